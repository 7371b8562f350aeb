#!/usr/bin/env python
"""Benchmark of the NeRF-DS render hot path on MI355X (BASELINE.json metric: rendered rays/s, 128 samples/ray,
full warp + NerfMLP).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic rays already resident in HBM: at N = 1 the
batch is BASELINE.json configs[1] - one 800x600 frame (480 000 rays) of the nerf_ds graph (SE(3) warp + hyper
sheet + mask + predicted-normal NerfMLP, 64 coarse + 64 fine -> 128 samples on the fine pass), rendered in
chunks of 65 536 rays (configs[2]) through the fused HIP kernel.  With N ranks every rank renders its own
480 000-ray block of an N-frame batch (weak scaling; rays are independent units) and each chunk ends with the
path's only exchange: one RCCL all-gather of the [65 536, 26] per-ray records, issued on a side stream so that it
overlaps the next chunk's kernel.

Prints ONE JSON line (rank 0).  `value` is the bf16 MFMA kernel (the arithmetic BASELINE.json's north_star names for
the roofline).  `roofline` is for the fused kernel: algorithmic FLOPs per launch (BASELINE.md section 2:
333.15 MFLOP/ray) over the mean launch duration measured with HIP events on the launch stream.  `cpu_baseline`
times the CPU oracle (torch fp32, all host cores) on a bounded sample of the same workload.  At N = 1 the same run
also times the other arithmetic modes on the same frame and measures every mode's composited-RGB error against the
CPU oracle on the baseline's sample (same rays, same injected uniforms): `parity_path` is the fastest mode that
meets north_star's 1e-4 (split bf16, three MFMAs per product), `other_paths` lists f16 and the mixed plan.

Other modes (not the driver's default line):
  --strong   BASELINE configs[2]: ONE 800x600 frame, every 65 536-ray chunk split over the N ranks in contiguous blocks
             (8 192 rays per rank at N = 8) through the drop-in top level, evaluation.render_image + make_model_fn:
             the all-gather of chunk i overlaps the kernel of chunk i + 1.  `scaling` = "strong".
  --train    BASELINE configs[3]: one training step (random-ray batch of 4096, MSE loss of both levels, backward,
             Adam) of the nerf_ds graph; metric "training rays/s"; roofline = MFMA on SURVEY 8d's algorithmic FLOPs (3 x the
             forward's); the HBM bytes of the layer-store design are reported beside it as `traffic` / `design_bytes_per_step`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, 'nerf-ds_amd'), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist

FLOP_PER_RAY = 333.15e6          # BASELINE.md section 2, nerf_ds graph, 192 field evaluations per ray
# dense MFMA peaks, MI355X_MICROARCH.md (bf16 = f16 rate; the split / mixed modes are priced against the same peak)
PEAK_TFLOPS = {'bf16': 2500.0, 'bf16x3': 2500.0, 'f32': 157.3, 'f16': 2500.0, 'mixed': 2500.0, 'bf16x3_fine': 2500.0, 'f16x3': 2500.0}
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
# where the committed PMC measurement of the headline kernel lives (separate rocprofv3 passes of this command, tools/prof_bench.sh)
def _latest(*names):
  for n in names:
    if os.path.exists(os.path.join(ROOT, n)):
      return n
  return names[0]


TRAFFIC_FILE = _latest('profiles/r6_bf16_hbm_traffic.json', 'profiles/r5_bf16_hbm_traffic.json', 'profiles/r4_bf16_hbm_traffic.json')
TRAIN_TRAFFIC_FILE = _latest('profiles/r6_train_hbm_traffic.json', 'profiles/r4_train_hbm_traffic.json')      # (round 5's file divided 9 steps' bytes by 4: not used)


def synth_rays(R, n_ids, seed, device):
  """Config-2 style synthetic frame: pinhole camera on a radius-1 sphere looking at the origin."""
  g = torch.Generator().manual_seed(seed)
  H, W = 600, 800
  idx = torch.arange(R) % (H * W)
  py, px = (idx // W).float() + 0.5, (idx % W).float() + 0.5
  focal = 0.5 * W / np.tan(0.5 * 0.6911)
  d_cam = torch.stack([(px - 0.5 * W) / focal, -(py - 0.5 * H) / focal, -torch.ones(R)], -1)
  d_cam = d_cam / d_cam.norm(dim=-1, keepdim=True)
  origins = torch.tensor([0.0, 0.0, 1.0]).expand(R, 3).contiguous()
  ids = torch.full((R, 1), int(torch.randint(0, n_ids, (1,), generator=g)), dtype=torch.int32)
  mask = (torch.rand(R, 1, generator=g) < 0.3).float()
  return dict(origins=origins.to(device), directions=d_cam.to(device), viewdirs=d_cam.to(device),
              metadata={'warp': ids.to(device)}, mask=mask.to(device))


def available_cores():
  """Cores this process may actually use: min(affinity mask, cgroup CPU quota) - os.cpu_count() over-reports in containers."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
      txt = open(path).read().split()
      if path.endswith('cpu.max'):
        if txt[0] != 'max':
          n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
      else:
        quota = int(txt[0])
        period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if quota > 0:
          n = min(n, max(1, quota // period))
    except (OSError, ValueError, IndexError):
      pass
  return max(1, n)


def cpu_baseline(cfg, params, budget_s=25.0, rays_full=None):
  """CPU oracle (torch fp32 restatement, vectorised over [R*S, K], all usable host cores) on a bounded ray sample.
  Returns (the cpu_baseline object, the sample: rays / uniforms / the oracle's composited rgb of both levels)."""
  from oracle import nerfds_oracle as O
  cores = min(available_cores(), 64)
  torch.set_num_threads(cores)
  model = O.NerfModel(cfg, params, torch.float32)
  keep = {}

  def run(R):
    src = synth_rays(R, cfg.num_warp_embeds, 1, 'cpu') if rays_full is None else \
        {k: (v[:R] if not isinstance(v, dict) else {kk: vv[:R] for kk, vv in v.items()}) for k, v in rays_full.items()}
    rays = {k: (v.cpu() if not isinstance(v, dict) else {kk: vv.cpu() for kk, vv in v.items()}) for k, v in src.items()}
    rng = np.random.default_rng(0)
    t, u = rng.random((R, cfg.num_coarse_samples)), rng.random((R, cfg.num_fine_samples))
    t0 = time.perf_counter()
    out = model.apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=cfg.predict_norm, compute_sigma_gradient=False)
    dt = time.perf_counter() - t0
    keep.update(rays=rays, t=t, u=u, rgb={lv: out[lv]['rgb'].numpy() for lv in out})
    return dt

  t_all = time.perf_counter()
  R = 16
  dt = run(R)                               # also warms the thread pool
  if rays_full is not None:                 # BASELINE configs[0]: the whole image, timed in full
    R = rays_full['origins'].shape[0]
    dt = run(R)
  while rays_full is None and time.perf_counter() - t_all + 4 * dt < budget_s and R < 16384:     # grow the sample while it fits the budget
    R *= 4
    dt = run(R)
  obj = {'value': R / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
         'sample': (f'{R} rays x ({cfg.num_coarse_samples} coarse + {cfg.num_coarse_samples + cfg.num_fine_samples} fine) samples of the same graph, torch-CPU fp32 oracle, '
                    f'{dt:.1f} s on {cores} threads, sigma-gradient off as on the GPU') if rays_full is None else
                   (f'the whole workload: {R} rays x {cfg.num_coarse_samples} samples, torch-CPU fp32 oracle, {dt:.1f} s on {cores} threads')}
  return obj, keep


TOLERANCE = 1e-4        # BASELINE.json north_star: composited RGB within 1e-4 rel of the reference on identical rays
PIXEL_FLOOR = 1e-2      # per-pixel relative error: |d rgb| / max(|rgb|, PIXEL_FLOOR) - dark pixels are not excused by the brightest one


def rgb_error(model, cfg, params, sample, precision):
  """Two statistics of the composited RGB against the oracle on the cpu_baseline sample (same rays, same injected sampling uniforms), worst of
  the two levels: (global) max |d rgb| / max |rgb| over rays and channels - the statistic of tests/test_gpu_parity.py - and (per pixel)
  max over rays and channels of |d rgb| / max(|rgb|, 1e-2)."""
  out = model.apply({'params': params}, sample['rays'], EXTRA, t_rand=sample['t'], u_rand=sample['u'],
                    use_predicted_norm=cfg.predict_norm, precision=precision)
  err = pix = 0.0
  for lv, ref in sample['rgb'].items():
    got = out[lv]['rgb'].cpu().numpy()
    d = np.abs(got - ref)
    err = max(err, float(d.max() / max(np.abs(ref).max(), 1e-6)))
    pix = max(pix, float((d / np.maximum(np.abs(ref), PIXEL_FLOOR)).max()))
  return err, pix


def rgb_error_by_level(model, cfg, params, sample, precision):
  """max |d rgb| / max |rgb| of the composited RGB against the oracle on the cpu_baseline sample, per level."""
  out = model.apply({'params': params}, sample['rays'], EXTRA, t_rand=sample['t'], u_rand=sample['u'],
                    use_predicted_norm=cfg.predict_norm, precision=precision)
  return {lv: float(np.abs(out[lv]['rgb'].cpu().numpy() - ref).max() / max(np.abs(ref).max(), 1e-6)) for lv, ref in sample['rgb'].items()}


def layer_dims(cfg):
  """(K, N) of every dense layer of the nerf_ds graph, per field evaluation (modules.py:57-83, 243-313): the traffic
  model of the training step's roofline."""
  def mlp(depth, width, in_dim, skip, heads):
    L = []
    for l in range(depth):
      L.append(((in_dim if l == 0 else width) + (in_dim if (l == skip and l > 0) else 0), width))
    return L + [(width, h) for h in heads]
  shared = mlp(8, 128, 44, 4, [1]) + mlp(6, 128, 33, 4, [3, 3]) + mlp(6, 64, 45, 4, [2])
  nerf = mlp(8, 256, 52, 4, [256, 4]) + [(256 + 24 + 256 + 24, 128), (128, 3)]
  return shared, nerf


def run_train(args, device, emit=True):
  """BASELINE configs[3]: forward + backward + Adam on a random-ray batch of 4096 (nerf_ds graph, 64 + 64 samples)."""
  from nerfds_amd import nerf_ds_config, init_params
  from nerfds_amd.training import Trainer
  R = args.train_rays
  cfg = nerf_ds_config(num_warp_embeds=64, near=0.3, far=1.7)
  params = init_params(cfg, 0, warp_head_scale=5e-2)
  rng = np.random.default_rng(2)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  f = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
  batch = dict(origins=f(rng.normal(size=(R, 3)) * 0.2), directions=f(d), viewdirs=f(d),
               metadata={'warp': torch.as_tensor(rng.integers(0, 64, (R, 1)), device=device)},
               mask=f(rng.random((R, 1)) < 0.3), rgb=f(rng.random((R, 3))))
  tr = Trainer(cfg, params, max_rays=R, device=device)
  losses = [tr.step(batch, EXTRA, 1e-3)['loss/total'] for _ in range(max(args.warmup, 1))]      # on-chip Philox jitter, new every step
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    losses.append(tr.step(batch, EXTRA, 1e-3)['loss/total'])
  torch.cuda.synchronize()
  dt = (time.perf_counter() - t0) / args.steps
  # Rows per ray.  NerfMLP: 64 coarse + 128 fine field evaluations.  Level-independent networks (mask / warp / hyper sheet): the reference
  # evaluates them 64 + 128 times too; since round 4 the plain step runs them - forwards and backwards - once per sample POSITION, 64 + 64
  # (nerfds_train.cpp run_merged; NERFDS_TRAIN_MERGED=0 restores one pass per level)
  merged = os.environ.get('NERFDS_TRAIN_MERGED', '1') != '0' and os.environ.get('NERFDS_TRAIN_FUSED_FWD', '1') != '0' and os.environ.get('NERFDS_TRAIN_FUSED_BWD', '1') != '0'
  S = 3 * 64
  M = R * S
  M_shared = R * (2 * 64 if merged else S)
  dims_shared, dims_nerf = layer_dims(cfg)
  # Traffic model of the step, per sample and dense layer (K inputs, N outputs), bytes:
  #   fused backward (default): forward writes Y as f16 + one ReLU bit per feature (2.125 N, hidden layers); the network's data-gradient
  #     chain writes g = dL/d(pre-activation) once (4 N); the weight gradient reads X (2 K as f16 from a hidden layer, 4 K from a raw
  #     input) and g (4 N).  dX never passes through HBM.
  #   layer-by-layer backward (NERFDS_TRAIN_FUSED_BWD=0, round 2): forward Y (4 N), weight gradient X + dY (4 K + 4 N), data gradient
  #     dY + Y + dX (8 N + 4 K); + X (4 K) in the forward when that is not fused either.
  fused_fwd = os.environ.get('NERFDS_TRAIN_FUSED_FWD', '1') != '0'
  fused_bwd = fused_fwd and os.environ.get('NERFDS_TRAIN_FUSED_BWD', '1') != '0'
  g16 = fused_bwd and os.environ.get('NERFDS_TRAIN_G16', '1') != '0'      # the chains hand g to the weight gradients as loss-scaled f16 (2 N) instead of fp32 (4 N)
  gb = 2 if g16 else 4
  if fused_bwd:
    per_row = lambda dims: sum((2.125 * N + gb * N if N > 6 else 0) + (gb if N > 6 else 4) * N + (2 * K if K in (64, 128, 256) else 4 * K) for K, N in dims)
    hbm_bytes = float(M) * per_row(dims_nerf) + float(M_shared) * per_row(dims_shared)
  else:
    hbm_bytes = 4.0 * M * sum((2 if fused_fwd else 3) * K + 4 * N for K, N in dims_shared + dims_nerf)
  traffic, traffic_source = None, None
  tpath = os.path.join(ROOT, TRAIN_TRAFFIC_FILE)
  if os.path.exists(tpath) and R == 4096 and fused_bwd:
    tj = json.load(open(tpath))
    traffic, traffic_source = tj['hbm_bytes_per_step'], f"{TRAIN_TRAFFIC_FILE} ({tj['measured_on']})"
  flop = 3 * FLOP_PER_RAY * R                    # fwd + bwd ~ 3 x forward, every network at 192 rows per ray (the algorithmic figure)
  # what the step executes: the level-independent networks at M_shared rows (128 per ray when merged), the NerfMLPs at 192 (SURVEY 8 layer table, MACs per sample)
  MAC_SHARED, MAC_NERF = 126080 + 91136 + 26368, 485376 + 65536 + 1024 + 72064
  flop_exec = 3 * 2.0 * (MAC_SHARED * float(M_shared) + MAC_NERF * float(M))
  # matrix-pipe floor of the EXECUTED work at the arithmetic the step runs: forward three bf16 MFMAs per product (split bf16), data gradient and
  # weight gradient ONE f16 MFMA per product with f16 g (3 + 1 + 1 of 9; NERFDS_TRAIN_BWD_F16=0: 3 + 3 + 1; fp32 g: 3 + 3 + 3)
  bwd_f16 = g16 and os.environ.get('NERFDS_TRAIN_BWD_F16', '0') == '1'
  floor_ms = ((5 / 9 if bwd_f16 else 7 / 9) if g16 else 1) * 3 * flop_exec / 2.5e15 * 1e3
  result = {
      'metric': f'training rays/sec (batch {R}, rgb-only objective: MSE of both levels + backward + Adam, full warp+NerfMLP)',
      'value': R / dt, 'unit': 'rays/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16x2 (forward and data-gradient chains: split bf16 operands, fp32 accumulate; activations stored as f16 + ReLU bits, the weight-gradient operand g as loss-scaled f16: weight gradients one f16 MFMA per product; data gradients and sums fp32)' if not bwd_f16 else 'bf16x2 forward, f16 backward (NERFDS_TRAIN_BWD_F16=1: data-gradient chains and weight gradients one f16 MFMA per product)',
      'data': 'synthetic',
      'config': {'workload': f"BASELINE configs[3]: training step, {R} random rays of 64 synthetic frames, 64 coarse + 64 fine samples, nerf_ds graph, "
                             'loss = MSE(fine) + MSE(coarse) ONLY (configs[3] as written; the full configs/nerf_ds.gin objective - norm loss, warp regulariser, '
                             'mask, back-facing - is the `full_objective` field / tools/objective_time.py), backward through every network, Adam; sampling jitter drawn on chip',
                 'rays_per_step': R, 'parallelism': 'single GPU', 'exchange': 'none (1 GPU)'},
      # SURVEY 8d prices the step by its ALGORITHMIC FLOPs (3 x 333.15 MFLOP per ray, ~30 MB of true input / parameter / gradient bytes): by that
      # definition the step is MFMA-bound and `frac` is the fraction of the 2.5 PFLOP/s dense peak.  The HBM figures describe the DESIGN (layer
      # activations stored for the backward), not the problem: `traffic` = measured bytes per step, `design_bytes_per_step` = this design's own model.
      'roofline': {'bound': 'mfma', 'achieved': flop / dt / 1e12, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': flop / dt / 2.5e15,
                   'traffic': traffic, 'traffic_source': traffic_source,
                   'kernel': ('whole step, level-independent networks once per sample position (three fused forward launches: coarse, new samples, fine NerfMLP; two NerfMLP '
                              'data-gradient chains + one chain per shared network over all positions; the hidden-layer weight gradients of an MLP in one launch)' if (fused_bwd and merged)
                              else 'whole step (per level: one fused forward launch, four fused data-gradient chains, one weight-gradient launch per layer segment)' if fused_bwd
                              else 'whole step (one fused forward launch per level + about 100 backward layer kernels; each is HBM-bound)' if fused_fwd
                              else 'whole step (about 150 layer kernels; each is HBM-bound)'),
                   'algorithmic_flop_per_step': flop, 'executed_flop_per_step': flop_exec, 'executed_tflops': flop_exec / dt / 1e12,
                   'design_bytes_per_step': hbm_bytes, 'design_hbm_gbps': hbm_bytes / dt / 1e9, 'design_hbm_frac_of_8000': hbm_bytes / dt / 8e12,
                   'traffic_model': (f"forward f16 Y + ReLU bits; chains write g once ({'f16' if g16 else 'fp32'}); weight gradient reads X (f16) + g" if fused_bwd else
                                     'fp32 activations: forward ' + ('Y (fused: X stays on chip)' if fused_fwd else 'X + Y') + ', weight gradient X + dY, data gradient dY + Y + dX'),
                   'mfma_floor_ms': floor_ms, 'ms_over_mfma_floor': dt * 1e3 / floor_ms},
      'loss_first': losses[0], 'loss_last': losses[-1],
  }
  # OPTION, measured beside the default: the data-gradient chains in one f16 MFMA per product (NERFDS_TRAIN_BWD_F16=1, read at every step; off by
  # default - it costs accuracy on small batches and loss-scale range, csrc/nerfds_train.cpp)
  if os.environ.get('NERFDS_TRAIN_BWD_F16') is None and g16 and not getattr(args, 'no_option_legs', False):
    try:
      os.environ['NERFDS_TRAIN_BWD_F16'] = '1'
      for _ in range(2):
        tr.step(batch, EXTRA, 1e-3)
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      nopt = max(3, min(args.steps, 10))
      for _ in range(nopt):
        tr.step(batch, EXTRA, 1e-3)
      torch.cuda.synchronize()
      dto = (time.perf_counter() - t1) / nopt
      result['option_f16_data_gradient_chains'] = {'env': 'NERFDS_TRAIN_BWD_F16=1', 'ms_per_step': dto * 1e3, 'value': R / dto, 'unit': 'rays/s', 'steps': nopt, 'warmup': 2,
                                                    'note': 'not the default: gradient error of the 16-ray golden case 2.2e-3 against 6.9e-4, narrower loss-scale window'}
    except (RuntimeError, FloatingPointError) as e:
      result['option_f16_data_gradient_chains'] = {'error': str(e)[:200]}
    finally:
      os.environ.pop('NERFDS_TRAIN_BWD_F16', None)
  # The objective real NeRF-DS training runs (configs/nerf_ds.gin: rgb + warp regulariser + back-facing + 3-D mask + the second-order norm loss,
  # whose tangent pass and its backward ride on fused chain kernels since round 5), on the same batch: the headline training number is
  # the rgb-only objective of BASELINE configs[3]; this field says what the shipped gin file's step costs next to it.
  if getattr(args, 'full_objective', True):
    full = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1, norm_loss_weight=0.001)
    try:
      for _ in range(2):
        tr.step(batch, EXTRA, 1e-3, objective=full)
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      nfull = max(3, min(args.steps, 10))
      for _ in range(nfull):
        tr.step(batch, EXTRA, 1e-3, objective=full)
      torch.cuda.synchronize()
      dtf = (time.perf_counter() - t1) / nfull
      result['full_objective'] = {'objective': 'configs/nerf_ds.gin: rgb + warp_reg + back_facing + predicted mask (sharp weights) + norm loss (second order)',
                                  'ms_per_step': dtf * 1e3, 'value': R / dtf, 'unit': 'rays/s', 'steps': nfull, 'warmup': 2,
                                  'ratio_to_rgb_only': dtf / dt}
    except (RuntimeError, FloatingPointError) as e:       # e.g. not enough device memory for the tangent workspace next to the render buffers
      result['full_objective'] = {'error': str(e)[:200]}
  if not args.no_cpu_baseline:
    from oracle import train_oracle as T
    cores = min(available_cores(), 64)
    torch.set_num_threads(cores)
    # bounded sample: slices of 256 rays of the same batch (autograd keeps ~4 MB per ray) until ~10 s of CPU work are on the clock
    Rc, done, dtc = 256, 0, 0.0
    while done < R and (dtc < 10.0 or done == 0):
      cb = {k: (v[done:done + Rc].cpu().numpy() if not isinstance(v, dict) else {'warp': v['warp'][done:done + Rc].cpu().numpy()}) for k, v in batch.items()}
      n = cb['origins'].shape[0]
      t, u = rng.random((n, 64)), rng.random((n, 64))
      t1 = time.perf_counter()
      T.loss_and_grads(cfg, params, cb, cb['rgb'], EXTRA, t, u, dtype=torch.float32)
      dtc += time.perf_counter() - t1
      done += n
    result['cpu_baseline'] = {'value': done / dtc, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
                              'sample': f'{done} rays (slices of {Rc}) x (64 + 128) samples, torch-CPU fp32 autograd through the oracle (loss + all gradients, no Adam), {dtc:.1f} s on {cores} threads'}
  else:
    result['cpu_baseline'] = None
  if emit:
    print(json.dumps(result), flush=True)
  return result


# ---- executed MFMA work (what the kernel issues) next to the algorithmic FLOPs (what the metric is defined on) ---------------
GRAPHS = {   # csrc/graphs.h: (depth, width, raw input features, has head) of the level-independent networks; trunk input; rgb condition
    'nerf_ds': dict(shared=[(8, 128, 44), (6, 128, 33), (6, 64, 45)], trunk_in=52, cond=48),
    'hypernerf': dict(shared=[(6, 128, 3 + 36 + 8), (6, 64, 45 - 1)], trunk_in=3 + 48 + 4, cond=3 + 24),
    'static': dict(shared=[], trunk_in=48, cond=24),
}


def stream_fragments(graph):
  """(shared, nerf) 1-KiB weight fragments (32 output rows x 16 k-slots) of one field evaluation of 32 samples: the walk of
  csrc/graphs.h shared_units / nerf_units at one unit per fragment.  One fragment = one MFMA in the bf16 / f16 kernels (three in
  split bf16), zero padding of ragged K and of the 32-row head tiles included, the activation-free bottleneck folded away."""
  ch = lambda f: -(-f // 16)
  def mlp(depth, width, in_feats, head, skip=4):
    n = 0
    for l in range(depth):
      n += (width // 32) * ((ch(in_feats) if l == 0 else width // 16) + (ch(in_feats) if (l == skip and l > 0) else 0))
    return n + (width // 16 if head else 0)
  g = GRAPHS[graph]
  shared = sum(mlp(d, w, i, True) for d, w, i in g['shared'])
  nerf = mlp(8, 256, g['trunk_in'], False) + 16 + 4 * (16 + ch(g['cond'])) + 8
  return shared, nerf


def executed_flop_per_ray(graph, nc, nf):
  """FLOPs of the MFMAs one ray issues (bf16 / f16 kernels): 32 768 per fragment and 32-sample tile.  The level-independent
  networks run once per sample POSITION (nc + nf positions), each level's NerfMLP on every sample of the level."""
  shared, nerf = stream_fragments(graph)
  t = lambda n: -(-n // 32)
  tiles = t(nc) * (shared + nerf) + ((t(nf) * shared + t(nc + nf) * nerf) if nf else 0)
  return 32768.0 * tiles


FLOP_PER_SAMPLE = {'nerf_ds': 1735168.0, 'hypernerf': 1420544.0, 'static': 1170688.0}     # SURVEY 8d (algorithmic, 2 FLOP / MAC)

# BASELINE configs[4] as SURVEY 8d resolves it: seven synthetic "scenes" = (seed, GLO rows, near, far)
SWEEP_SCENES = [(11, 163, 0.30, 1.70), (12, 881, 0.25, 1.60), (13, 424, 0.35, 1.90), (14, 741, 0.20, 1.50),
                (15, 309, 0.30, 2.00), (16, 511, 0.40, 1.80), (17, 256, 0.28, 1.75)]


def synth_rays_square(H, W, radius, device):
  """Config-1 camera: pinhole, focal = 0.5 W / tan(0.5 * 0.6911), on a radius-4 sphere looking at the origin."""
  idx = torch.arange(H * W)
  py, px = (idx // W).float() + 0.5, (idx % W).float() + 0.5
  focal = 0.5 * W / np.tan(0.5 * 0.6911)
  d = torch.stack([(px - 0.5 * W) / focal, -(py - 0.5 * H) / focal, -torch.ones(H * W)], -1)
  d = d / d.norm(dim=-1, keepdim=True)
  o = torch.tensor([0.0, 0.0, radius]).expand(H * W, 3).contiguous()
  return dict(origins=o.to(device), directions=d.to(device), viewdirs=d.to(device),
              metadata={'warp': torch.zeros((H * W, 1), dtype=torch.int32, device=device)}, mask=torch.zeros((H * W, 1), device=device))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--rays', type=int, default=None, help='rays per rank per step (default: one 800x600 frame; --graph static: the 64x64 image); with --strong: rays of the ONE frame')
  ap.add_argument('--chunk', type=int, default=65536)
  ap.add_argument('--precision', default='bf16', choices=['bf16', 'bf16x3', 'f32', 'f16', 'mixed', 'bf16x3_fine', 'f16x3'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-other-paths', action='store_true', help='skip the parity_path / other_paths legs (N = 1 only)')
  ap.add_argument('--graph', default='nerf_ds', choices=['nerf_ds', 'hypernerf', 'static'],
                  help="'hypernerf' = configs/base.gin graph (BASELINE configs[4] per SURVEY 8d; use with --samples 128); "
                       "'static' = BASELINE configs[0]: 64x64 image, 64 coarse samples, no warp / hyper / mask, CPU oracle timed in full")
  ap.add_argument('--samples', type=int, default=64, help='coarse = fine sample count (64 = the headline config; 128 = BASELINE configs[4])')
  ap.add_argument('--strong', action='store_true', help='BASELINE configs[2]: one frame, every chunk split over the ranks (render_image)')
  ap.add_argument('--sweep', action='store_true', help='BASELINE configs[4]: seven synthetic scenes, base.gin graph, 128 + 128 samples, one 800x600 frame each')
  ap.add_argument('--no-train-line', action='store_true', help='skip the train_step leg of the default run (N = 1 only)')
  ap.add_argument('--train', action='store_true', help='BASELINE configs[3]: the training step instead of the render')
  ap.add_argument('--train-rays', type=int, default=4096)
  ap.add_argument('--no-option-legs', action='store_true', help='--train: time the default step only (no NERFDS_TRAIN_BWD_F16 option leg): what the profiling scripts run, so that every launch they count belongs to a default step')
  ap.add_argument('--no-full-objective', dest='full_objective', action='store_false',
                  help='skip the full configs/nerf_ds.gin objective leg of the training line (profiles of the rgb step)')
  args = ap.parse_args()
  if args.sweep:
    args.graph, args.samples = 'hypernerf', 128
  if args.rays is None:
    args.rays = 64 * 64 if args.graph == 'static' else 480000

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
  # NERFDS_DIST_BACKEND=gloo (tests only): the N > 1 code path with every rank on whatever GPUs the box has, the exchange
  # staged through the host; the numbers of such a run mean nothing
  backend = os.environ.get('NERFDS_DIST_BACKEND', 'nccl')
  if backend != 'nccl':
    local_rank %= torch.cuda.device_count()
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if args.train:
    if world != 1:
      raise SystemExit('--train is the single-GPU step of BASELINE configs[3]')
    return run_train(args, device)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
      dist.init_process_group('nccl', device_id=device)      # "nccl" is RCCL on ROCm
    else:
      dist.init_process_group(backend)

  from nerfds_amd import nerf_ds_config, hypernerf_config, static_config, init_params
  from nerfds_amd.model import NerfModel
  from nerfds_amd.evaluation import TrainState, make_model_fn, render_image, all_gather_into
  from nerfds_amd import _native as N

  static = args.graph == 'static'
  nc, nf = args.samples, (0 if static else args.samples)
  if static:
    cfg = static_config(num_coarse_samples=nc)
  else:
    cfg = (nerf_ds_config if args.graph == 'nerf_ds' else hypernerf_config)(near=0.3, far=1.7, num_warp_embeds=256, num_coarse_samples=nc, num_fine_samples=nf)
  flop_per_ray = (nc + ((nc + nf) if nf else 0)) * FLOP_PER_SAMPLE[args.graph]        # nc coarse + (nc + nf) fine field evaluations per ray
  exec_per_ray = executed_flop_per_ray(args.graph, nc, nf)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)   # random-init weights
  model = NerfModel(cfg, device=device, precision=args.precision)
  model.load_params(params)
  variables = {'params': params}
  live = {'model': model, 'variables': variables}       # what the step functions render with (--sweep swaps it per scene)
  # resident in HBM before timing; --strong: every rank holds the rays of the ONE frame, weak: its own frame
  rays = synth_rays_square(64, 64, 4.0, device) if static else synth_rays(args.rays, cfg.num_warp_embeds, 100 + (0 if args.strong else rank), device)
  chunks = [(lo, min(lo + args.chunk, args.rays)) for lo in range(0, args.rays, args.chunk)]
  chunk_rays = [{k: (v[lo:hi] if not isinstance(v, dict) else {kk: vv[lo:hi] for kk, vv in v.items()})
                 for k, v in rays.items()} for lo, hi in chunks]
  compute = torch.cuda.current_stream(device)
  comm = torch.cuda.Stream(device) if world > 1 else None
  frame = torch.empty((args.rays, N.RAY_REC), dtype=torch.float32, device=device)               # this rank's records
  gathered = [torch.empty((world * (hi - lo), N.RAY_REC), dtype=torch.float32, device=device) for lo, hi in chunks[:2]] if world > 1 else None
  level = 'fine' if nf else 'coarse'

  def step_weak(seed, precision=None, out=None, out_coarse=None):
    """Every chunk: the fused kernel writes this rank's records straight into its frame buffer; with N ranks the
    all-gather of the chunk (the path's only exchange) runs on a side stream under the next chunk's kernel."""
    buf = frame if out is None else out
    for ci, (cr, (lo, hi)) in enumerate(zip(chunk_rays, chunks)):
      rec = {level: buf[lo:hi]}
      if out_coarse is not None:
        rec['coarse'] = out_coarse[lo:hi]
      live['model'].apply(live['variables'], cr, EXTRA, rngs={'coarse': seed, 'fine': seed + 500}, ray_offset=lo,
                  use_predicted_norm=cfg.predict_norm, return_points=False, mask_ratio=1, sharp_weights_std=0.1,
                  precision=precision, records_out=rec)
      if world > 1:
        slot = ci & 1
        ready = torch.cuda.Event()
        ready.record(compute)
        with torch.cuda.stream(comm):
          comm.wait_event(ready)
          all_gather_into(gathered[slot][:world * (hi - lo)], buf[lo:hi])
    if world > 1:
      compute.wait_stream(comm)

  state = TrainState.create(params, **EXTRA)
  model_fn = make_model_fn(model, precision=args.precision)
  frame_rays = {k: (v if not isinstance(v, dict) else dict(v)) for k, v in rays.items()}

  def step_strong(seed, precision=None):
    render_image(state, frame_rays, model_fn, device_count=world, rng=np.array([0, seed]), chunk=args.chunk, cfg=None, to_host=False)

  step = step_strong if args.strong else step_weak

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  def timed(steps, warmup, precision=None):
    for i in range(warmup):
      step(i, precision)
    sync()
    live['model'].kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for i in range(steps):
      step(warmup + i, precision)
    sync()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = live['model'].kernel_time_ms(reset=False)
    return elapsed, n_launch, kernel_ms

  launches_per_step = len(chunks)
  rays_per_launch = (args.rays / world if args.strong else args.rays) / launches_per_step
  kernel_name = 'nerfds::render_rays_kernel<%s, %%s>' % {'nerf_ds': 'GraphNerfDS', 'hypernerf': 'GraphHyperNeRF', 'static': 'GraphStatic'}[args.graph]

  def roofline_of(prec, n_launch, kernel_ms):
    """MFMA roofline of the fused kernel: ALGORITHMIC FLOPs per launch (fixed per ray) over the mean launch duration
    (HIP events on the launch stream).  `executed_*`: the MFMA work the kernel actually issues - zero padding in, bottleneck fold
    and the once-per-position evaluation of the level-independent networks out - for judging the matrix pipe itself."""
    launch_s = kernel_ms / max(n_launch, 1) * 1e-3
    ach = rays_per_launch * flop_per_ray / launch_s / 1e12 if n_launch else None
    mult = 3.0 if prec in ('bf16x3', 'f16x3') else 1.0       # split bf16 / split f16: three MFMAs per product (the mixed plan: only its warp field; not priced)
    exec_ray = exec_per_ray
    if prec == 'bf16x3_fine' and nf:              # split bf16 except the coarse level's NerfMLP: one f16 MFMA per product there
      shared_f, nerf_f = stream_fragments(args.graph)
      tl = lambda n: -(-n // 32)
      exec_ray, mult = 32768.0 * (3 * (tl(nc) + tl(nf)) * shared_f + tl(nc) * nerf_f + 3 * tl(nc + nf) * nerf_f), 1.0
    elif prec == 'bf16x3_fine':
      mult = 3.0
    exe = rays_per_launch * exec_ray * mult / launch_s / 1e12 if n_launch else None
    peak = PEAK_TFLOPS[prec]
    return {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': (ach / peak) if ach else None,
            'traffic': None, 'traffic_source': None, 'kernel': kernel_name % prec, 'avg_launch_ms': launch_s * 1e3, 'launches': n_launch,
            'algorithmic_flop_per_launch': rays_per_launch * flop_per_ray,
            'executed_mfma_flop_per_launch': (rays_per_launch * exec_ray * mult) if prec != 'mixed' else None,
            'executed_frac': (exe / (157.3 if prec == 'f32' else 2500.0)) if (exe and prec != 'mixed') else None}

  if args.sweep:
    return run_sweep(args, world, rank, device, live, cfg, chunks, timed, flop_per_ray, exec_per_ray, kernel_name)

  elapsed, n_launch, kernel_ms = timed(args.steps, args.warmup)
  if world > 1:
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    ms_per_step = elapsed * 1e3 / args.steps
    total_rays = args.rays if args.strong else args.rays * world
    value = total_rays / (elapsed / args.steps)
    roof = roofline_of(args.precision, n_launch, kernel_ms)
    # HBM bytes per launch cannot be read live (PMC passes are separate rocprofv3 runs of this same command): the
    # committed measurement of this round's kernel is quoted, with its source, when this run is the configuration it was taken on.
    tpath = os.path.join(ROOT, TRAFFIC_FILE)
    if (os.path.exists(tpath) and args.precision == 'bf16' and args.rays == 480000 and args.chunk == 65536 and args.samples == 64
        and args.graph == 'nerf_ds' and not args.strong):
      tj = json.load(open(tpath))
      roof['traffic'] = tj['hbm_bytes_per_launch']
      roof['traffic_source'] = f"{TRAFFIC_FILE} ({tj.get('measured_on', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command')})"
    if static:
      workload = ("BASELINE configs[0]: static Lego-style synthetic scene, 64x64 image (4096 rays), 64 samples/ray, coarse-only NerfMLP, "
                  'warp / hyper / mask / normal disabled, random-init weights')
      metric = 'rendered rays/sec (64 samples/ray, coarse-only NerfMLP, warp disabled)'
    else:
      nets = ('mask + predicted-normal NerfMLP (configs/nerf_ds.gin graph)' if args.graph == 'nerf_ds'
              else 'NerfMLP with posenc identity (configs/base.gin HyperNeRF graph)')
      what = ('ONE 800x600 frame (480000 rays), every 65536-ray chunk split over the ranks' if args.strong
              else '800x600 frame per GPU (480000 rays)')
      workload = (f"NeRF-DS 'bell'-shaped synthetic scene, {what}, {args.samples} coarse + {args.samples} fine "
                  f'samples ({2 * args.samples} on the fine pass, {3 * args.samples} field evaluations/ray), SE(3) warp + hyper-slice + '
                  f'{nets}, random-init weights')
      metric = 'rendered rays/sec (%d samples/ray, full warp+NerfMLP)' % (2 * args.samples)
    exchange = 'none (1 GPU)' if world == 1 else ('all-gather of the [chunk / N, 26] fp32 ray records of every chunk, on a side stream under the next chunk' if args.strong
                                                  else 'all-gather of [chunk, 26] fp32 ray records, on a side stream under the next chunk')
    result = {
        'metric': metric,
        'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': workload,
                   'rays_per_gpu_per_step': args.rays // world if args.strong else args.rays, 'chunk': args.chunk,
                   'parallelism': f'ray-shard x{world}' + (' of every chunk (evaluation.render_image)' if args.strong else ''),
                   'exchange': exchange},
        'roofline': roof,
        # both fractions of the headline kernel at the top level: ALGORITHMIC FLOPs (what the roofline object prices) and the MFMA FLOPs the
        # kernel actually executes (fewer: the shared networks run once per sample position, the linear bottleneck is folded) over the same duration
        'roofline_frac_algorithmic': roof.get('frac'),
        'roofline_frac_executed_mfma': (roof['executed_mfma_flop_per_launch'] / (roof['avg_launch_ms'] * 1e-3) / 1e12 / roof['peak']
                                        if roof.get('executed_mfma_flop_per_launch') and roof.get('avg_launch_ms') else None),
    }
    sample = None
    if world == 1 and not args.no_cpu_baseline:
      result['cpu_baseline'], sample = cpu_baseline(cfg, params, rays_full=(rays if static else None))
    else:
      result['cpu_baseline'] = None
    # north_star's contract on composited RGB, said at the top level of the line: `value` is the arithmetic named by --precision (bf16 by
    # default, the arithmetic the roofline target names) and carries its OWN verdict here; the number that holds the tolerance is
    # `parity_path` (split bf16) below, with its own meets_tolerance
    result['tolerance'] = TOLERANCE
    result['meets_tolerance'] = None            # unknown until an error is measured (no CPU baseline sample: N > 1 or --no-cpu-baseline)
    if sample:
      result['rgb_max_rel_err'], result['rgb_max_pixel_rel_err'] = rgb_error(model, cfg, params, sample, args.precision)
      result['meets_tolerance'] = bool(result['rgb_max_rel_err'] <= TOLERANCE)
      result['err_reference'] = ('CPU oracle (torch fp32) on the cpu_baseline sample: same rays, same injected uniforms; rgb_max_rel_err = max |d rgb| / max |rgb|, '
                                 'rgb_max_pixel_rel_err = max |d rgb| / max(|rgb|, 1e-2) per pixel and channel; worst of the levels')
    if world == 1 and not args.no_other_paths and not args.strong and not static:
      # The same frame in the other arithmetic modes, timed in this run.  The parity path - the arithmetic that meets north_star's
      # 1e-4 - gets the same treatment as the headline: >= 10 timed steps, its own roofline object, and its error measured twice:
      # against the CPU oracle on the baseline's sample and against the fp32-MFMA kernel (which matches the fp64 oracle to ~4e-6,
      # tests/test_gpu_parity.py) over EVERY ray of the frame.
      def full_frame_error(prec):
        ref_f, ref_c = torch.empty_like(frame), torch.empty_like(frame)
        got_f, got_c = torch.empty_like(frame), torch.empty_like(frame)
        step_weak(7, 'f32', ref_f, ref_c)
        step_weak(7, prec, got_f, got_c)
        torch.cuda.synchronize()
        err = pix = 0.0
        for g, r in ((got_f, ref_f), (got_c, ref_c)):
          d = (g[:, :3] - r[:, :3]).abs()
          err = max(err, float(d.max() / r[:, :3].abs().max()))
          pix = max(pix, float((d / r[:, :3].abs().clamp_min(PIXEL_FLOOR)).max()))
        return err, pix
      paths = {}
      for prec in ('f16x3', 'bf16x3', 'f16', 'mixed'):      # (the arithmetic that `value_at_tolerance` names is timed first, straight after the headline)
        if prec == args.precision:
          continue
        full = prec in ('bf16x3', 'f16x3')      # the parity-grade arithmetics get the headline's treatment
        steps = max(10, args.steps) if full else 3
        el, nl, kms = timed(steps, 2 if full else 1, prec)
        r = roofline_of(prec, nl, kms)
        e_glob, e_pix = rgb_error(model, cfg, params, sample, prec) if sample else (None, None)
        paths[prec] = {'precision': prec, 'value': args.rays / (el / steps), 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'steps': steps,
                       'warmup': 2 if full else 1, 'roofline': r, 'roofline_frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'],
                       'rgb_max_rel_err': e_glob, 'rgb_max_pixel_rel_err': e_pix}
      if 'mixed' in paths:
        plan = (C_int32 * 5)()
        N.load().nerfds_precision_plan(N.PREC['mixed'], plan)
        names = ('bf16', 'bf16x3', 'f32', 'f16')
        paths['mixed']['plan'] = dict(zip(('mask', 'warp', 'hyper', 'trunk', 'rgb'), (names[v] for v in plan)))
      pp = paths.pop('bf16x3', None)
      if pp is not None:
        pp['full_frame_rgb_max_rel_err'], pp['full_frame_rgb_max_pixel_rel_err'] = full_frame_error('bf16x3')
        pp['full_frame_reference'] = (f'the fp32-MFMA kernel (v_mfma_f32_32x32x2_f32, exact fp32 fma chains; itself within ~4e-6 of the fp64 oracle on the sample) on all '
                                      f'{args.rays} rays of the frame, both levels, same Philox sampling stream - NOT the CPU oracle, which sees the cpu_baseline sample only')
        errs = [e for e in (pp['rgb_max_rel_err'], pp['full_frame_rgb_max_rel_err']) if e is not None]
        pp['meets_1e-4'] = bool(errs) and max(errs) <= 1e-4
        pp['tolerance'], pp['meets_tolerance'] = TOLERANCE, pp['meets_1e-4']
        pix = [e for e in (pp['rgb_max_pixel_rel_err'], pp['full_frame_rgb_max_pixel_rel_err']) if e is not None]
        pp['meets_1e-4_per_pixel'] = bool(pix) and max(pix) <= 1e-4
        pp['note'] = ('split bf16 (hi + lo) operands, three MFMAs per product, fp32 accumulate: the fastest arithmetic that meets '
                      "north_star's 1e-4 on composited RGB (profiles/r4_precision_budget.md: no arithmetic below three MFMA-equivalents per product holds it at frame size)")
        pp['scene_dependence'] = ('holds 1e-4 on THIS frame; on other random-init scenes of the same graph a few dozen badly conditioned rays of 480 000 leave it (worst 1.2e-3) '
                                  'where the fp32-MFMA kernel and split f16 hold it: profiles/r6_parity_sweep.jsonl, DESIGN 11.7')
        result['parity_path'] = pp
      pq = paths.pop('f16x3', None)
      if pq is not None:
        # split f16 (hi + lo f16 operands, three MFMAs per product: NERFDS_PREC_F16X3, round 6): the arithmetic that holds 1e-4 on every frame of the scene sweep
        pq['full_frame_rgb_max_rel_err'], pq['full_frame_rgb_max_pixel_rel_err'] = full_frame_error('f16x3')
        pq['full_frame_reference'] = f'the fp32-MFMA kernel on all {args.rays} rays of the frame, both levels, same Philox sampling stream'
        errs = [e for e in (pq['rgb_max_rel_err'], pq['full_frame_rgb_max_rel_err']) if e is not None]
        pq['tolerance'], pq['meets_tolerance'] = TOLERANCE, bool(errs) and max(errs) <= TOLERANCE
        pq['note'] = ('split f16 (hi + lo, 11 + 11 significand bits) operands, three MFMAs per product, fp32 accumulate: fp32-MFMA-grade RGB at the MFMA count of split bf16; '
                      'holds 1e-4 on all 14 (graph, scene) frames of tools/parity_sweep.py (worst 5.7e-5, no ray of 6.7 M over 1e-4) and on 24 further nerf_ds scenes but for ONE ray of 23 M ray-levels (1.3e-4 on a coarse level; profiles/r6_parity_sweep_extra_24_scenes.jsonl); range of f16: an activation beyond 65504 is inf')
        result['parity_path_f16x3'] = pq
      # said at the TOP of the line: `value` is the arithmetic north_star's roofline clause names (bf16, outside the 1e-4 tolerance); the number that
      # satisfies the tolerance on BOTH levels - and not only on this frame (parity_path.scene_dependence) - is the split-f16 kernel's
      best = pq if (pq is not None and pq['meets_tolerance']) else (pp if (pp is not None and pp['meets_tolerance']) else None)
      result['value_at_tolerance'] = best['value'] if best else None
      result['roofline_frac_at_tolerance'] = best['roofline_frac'] if best else None
      result['precision_at_tolerance'] = best['precision'] if best else None
      # The same frame with the COARSE level's NerfMLP in one f16 MFMA per product (precision 'bf16x3_fine'): the fine level - the one render_fn returns,
      # evaluation.py:121-124 - sees of it only the weights its depths are drawn from and holds 1e-4; the coarse level's own RGB is f16-grade.  Its own
      # object, NOT the parity_path: `meets_tolerance_both_levels` is False by construction and says so.
      if args.precision != 'bf16x3_fine':
        steps = max(10, args.steps)
        el, nl, kms = timed(steps, 2, 'bf16x3_fine')
        r = roofline_of('bf16x3_fine', nl, kms)
        by_level = rgb_error_by_level(model, cfg, params, sample, 'bf16x3_fine') if sample else {}
        ref_f, ref_c, got_f, got_c = (torch.empty_like(frame) for _ in range(4))
        step_weak(7, 'f32', ref_f, ref_c)
        step_weak(7, 'bf16x3_fine', got_f, got_c)
        torch.cuda.synchronize()
        ff = {lv: float((g[:, :3] - rr[:, :3]).abs().max() / rr[:, :3].abs().max()) for lv, g, rr in (('fine', got_f, ref_f), ('coarse', got_c, ref_c))}
        worst_fine = max([e for e in (by_level.get('fine'), ff['fine']) if e is not None])
        worst_coarse = max([e for e in (by_level.get('coarse'), ff['coarse']) if e is not None])
        result['parity_path_fine_level'] = {
            'precision': 'bf16x3_fine', 'value': args.rays / (el / steps), 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'steps': steps, 'warmup': 2,
            'roofline': r, 'roofline_frac': r['frac'], 'avg_launch_ms': r['avg_launch_ms'],
            'fine_level_rgb_max_rel_err': by_level.get('fine'), 'coarse_level_rgb_max_rel_err': by_level.get('coarse'),
            'full_frame_fine_level_rgb_max_rel_err': ff['fine'], 'full_frame_coarse_level_rgb_max_rel_err': ff['coarse'],
            'tolerance': TOLERANCE, 'meets_tolerance_fine_level': bool(worst_fine <= TOLERANCE), 'meets_tolerance_both_levels': bool(max(worst_fine, worst_coarse) <= TOLERANCE),
            'note': ("split bf16 everywhere except the coarse level's NerfMLP (one f16 MFMA per product): render_fn / render_image return the FINE level only "
                     '(evaluation.py:121-124), and every array they return is within 1e-4; model.apply also returns the coarse level, whose composited RGB is f16-grade '
                     'in this mode.  The number that holds 1e-4 on BOTH levels is parity_path.')}
      result['other_paths'] = list(paths.values())
      if not args.no_train_line:
        # BASELINE configs[3] in the same run (the full line: --train): the 4096-ray training step, 10 timed steps after 3 warm-ups
        targs = argparse.Namespace(**vars(args))
        targs.steps, targs.warmup, targs.no_cpu_baseline = 10, 3, True
        tr = run_train(targs, device, emit=False)
        result['train_step'] = {k: tr[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'loss_first', 'loss_last', 'full_objective', 'option_f16_data_gradient_chains') if k in tr}
        result['train_step']['workload'] = tr['config']['workload']
        result['train_step']['roofline'] = {k: tr['roofline'][k] for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'algorithmic_flop_per_step',
                                                                         'executed_flop_per_step', 'design_bytes_per_step', 'design_hbm_gbps', 'mfma_floor_ms', 'ms_over_mfma_floor')}
        # the same step at the batch the reference itself trains at (configs/nerf_ds.gin:4 batch_size = 512; BASELINE configs[3] quotes 4096)
        targs.train_rays, targs.steps, targs.no_option_legs = 512, 20, True
        tr5 = run_train(targs, device, emit=False)
        result['train_step_reference_batch'] = {'rays': 512, 'ms_per_step': tr5['ms_per_step'], 'value': tr5['value'], 'unit': tr5['unit'], 'steps': 20, 'warmup': 3,
                                                'roofline_frac': tr5['roofline']['frac'], 'full_objective_ms_per_step': tr5.get('full_objective', {}).get('ms_per_step'),
                                                'note': 'configs/nerf_ds.gin:4 ships batch_size = 512: a chain of ~150 dependent launches, not a load (DESIGN 11.5)'}
    print(json.dumps(result), flush=True)

  if world > 1:
    dist.destroy_process_group()


def run_sweep(args, world, rank, device, live, cfg, chunks, timed, flop_per_ray, exec_per_ray, kernel_name):
  """BASELINE configs[4]: the seven-scene render sweep (render_pipeline.py:21-31 loops the scenes; the dataset is not here, so a
  scene is a (seed, GLO rows, near, far) tuple with its own random-init weights and its own context: the GLO table size is part of
  the model configuration).  Every rank renders its own 800x600 frame of each scene (weak scaling); one JSON line with the
  per-scene rates and the aggregate."""
  from nerfds_amd import init_params
  from nerfds_amd.model import NerfModel
  scenes = []
  tot_rays, tot_s, tot_kms, tot_launch = 0.0, 0.0, 0.0, 0
  for seed, n_ids, near, far in SWEEP_SCENES:
    scfg = cfg.replace(num_warp_embeds=n_ids, near=near, far=far)
    p = init_params(scfg, seed, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
    m = NerfModel(scfg, device=device, precision=args.precision)
    m.load_params(p)
    live['model'], live['variables'] = m, {'params': p}      # (the rays keep their frame; ids beyond a scene's table are clamped like a jnp gather)
    el, nl, kms = timed(args.steps, 1)
    if world > 1:
      t = torch.tensor([el], device=device, dtype=torch.float64)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      el = float(t.item())
    scenes.append({'seed': seed, 'glo_rows': n_ids, 'near': near, 'far': far, 'rays_per_s': args.rays * world / (el / args.steps),
                   'ms_per_frame': el * 1e3 / args.steps})
    tot_rays += args.rays * world * args.steps
    tot_s += el
    tot_kms += kms
    tot_launch += nl
  if rank == 0:
    launch_s = tot_kms / max(tot_launch, 1) * 1e-3
    rpl = args.rays / len(chunks)
    ach = rpl * flop_per_ray / launch_s / 1e12
    print(json.dumps({
        'metric': 'rendered rays/sec (256 samples/ray hierarchical, full warp+NerfMLP), 7-scene sweep', 'value': tot_rays / tot_s, 'unit': 'rays/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': 1, 'ms_per_step': tot_s * 1e3 / (args.steps * len(SWEEP_SCENES)), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[4]: seven synthetic dynamic-specular scenes (seed, GLO rows, near, far), one 800x600 frame each per GPU, '
                               'configs/base.gin HyperNeRF graph, 128 coarse + 128 fine samples (256 on the fine pass), random-init weights',
                   'rays_per_gpu_per_step': args.rays, 'chunk': args.chunk, 'parallelism': f'ray-shard x{world}', 'scenes': scenes},
        'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_TFLOPS[args.precision], 'unit': 'TFLOP/s', 'frac': ach / PEAK_TFLOPS[args.precision],
                     'traffic': None, 'kernel': kernel_name % args.precision, 'avg_launch_ms': launch_s * 1e3, 'launches': tot_launch,
                     'algorithmic_flop_per_launch': rpl * flop_per_ray, 'executed_mfma_flop_per_launch': rpl * exec_per_ray},
        'cpu_baseline': None}), flush=True)
  if world > 1:
    dist.destroy_process_group()


from ctypes import c_int32 as C_int32      # noqa: E402

if __name__ == '__main__':
  main()
