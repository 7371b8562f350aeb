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
             Adam) of the nerf_ds graph; metric "training rays/s"; roofline = HBM (every layer of the step is
             HBM-bound by construction).
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
PEAK_TFLOPS = {'bf16': 2500.0, 'bf16x3': 2500.0, 'f32': 157.3, 'f16': 2500.0, 'mixed': 2500.0}
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
# where the committed PMC measurement of the headline kernel lives (separate rocprofv3 passes of this command, tools/prof_bench.sh)
TRAFFIC_FILE = 'profiles/r2_bf16_hbm_traffic.json'
TRAIN_TRAFFIC_FILE = 'profiles/r2_train_hbm_traffic.json'


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


def cpu_baseline(cfg, params, budget_s=25.0):
  """CPU oracle (torch fp32 restatement, vectorised over [R*S, K], all usable host cores) on a bounded ray sample.
  Returns (the cpu_baseline object, the sample: rays / uniforms / the oracle's composited rgb of both levels)."""
  from oracle import nerfds_oracle as O
  cores = min(available_cores(), 64)
  torch.set_num_threads(cores)
  model = O.NerfModel(cfg, params, torch.float32)
  keep = {}

  def run(R):
    rays = {k: (v.cpu() if not isinstance(v, dict) else {kk: vv.cpu() for kk, vv in v.items()})
            for k, v in synth_rays(R, cfg.num_warp_embeds, 1, 'cpu').items()}
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
  while time.perf_counter() - t_all + 4 * dt < budget_s and R < 16384:     # grow the sample while it fits the budget
    R *= 4
    dt = run(R)
  obj = {'value': R / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
         'sample': f'{R} rays x ({cfg.num_coarse_samples} coarse + {cfg.num_coarse_samples + cfg.num_fine_samples} fine) samples of the same nerf_ds graph, torch-CPU fp32 oracle, '
                   f'{dt:.1f} s on {cores} threads, sigma-gradient off as on the GPU'}
  return obj, keep


def rgb_error(model, cfg, params, sample, precision):
  """Max over rays and channels of |rgb_gpu - rgb_oracle| / max |rgb_oracle| (the statistic of tests/test_gpu_parity.py),
  worst of the two levels, on the cpu_baseline sample: same rays, same injected sampling uniforms."""
  out = model.apply({'params': params}, sample['rays'], EXTRA, t_rand=sample['t'], u_rand=sample['u'],
                    use_predicted_norm=cfg.predict_norm, precision=precision)
  err = 0.0
  for lv, ref in sample['rgb'].items():
    got = out[lv]['rgb'].cpu().numpy()
    err = max(err, float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)))
  return err


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
  return shared + nerf


def run_train(args, device):
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
  S = 3 * 64                                     # field evaluations per ray: 64 coarse + 128 fine
  M = R * S
  dims = layer_dims(cfg)
  # traffic model of the step (fp32 activations in HBM): forward Y only (ONE fused launch per level keeps X on chip; layer by layer,
  # NERFDS_TRAIN_FUSED_FWD=0, it is X + Y); weight gradient X + dY; data gradient dY + Y (ReLU mask) + dX
  #   ->  N (+ K) + (K + N) + (K + 2N) floats per sample and layer, 4 bytes each
  fused_fwd = os.environ.get('NERFDS_TRAIN_FUSED_FWD', '1') != '0'
  hbm_bytes = 4.0 * M * sum((2 if fused_fwd else 3) * K + 4 * N for K, N in dims)
  traffic, traffic_source = None, None
  tpath = os.path.join(ROOT, TRAIN_TRAFFIC_FILE)
  if os.path.exists(tpath) and R == 4096 and fused_fwd:
    tj = json.load(open(tpath))
    traffic, traffic_source = tj['hbm_bytes_per_step'], f"{TRAIN_TRAFFIC_FILE} ({tj['measured_on']})"
  flop = 3 * FLOP_PER_RAY * R                    # fwd + bwd ~ 3 x forward
  result = {
      'metric': 'training rays/sec (batch 4096, MSE of both levels + backward + Adam, full warp+NerfMLP)',
      'value': R / dt, 'unit': 'rays/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt * 1e3,
      'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16x2 (split bf16 operands, fp32 accumulate, fp32 activations)',
      'data': 'synthetic',
      'config': {'workload': f"BASELINE configs[3]: training step, {R} random rays of 64 synthetic frames, 64 coarse + 64 fine samples, nerf_ds graph, "
                             'loss = MSE(fine) + MSE(coarse), backward through every network, Adam; sampling jitter drawn on chip',
                 'rays_per_step': R, 'parallelism': 'single GPU', 'exchange': 'none (1 GPU)'},
      'roofline': {'bound': 'hbm', 'achieved': hbm_bytes / dt / 1e9, 'peak': 8000.0, 'unit': 'GB/s', 'frac': hbm_bytes / dt / 8e12,
                   'traffic': traffic, 'traffic_source': traffic_source,
                   'kernel': 'whole step (one fused forward launch per level + about 100 backward layer kernels; each is HBM-bound)' if fused_fwd
                             else 'whole step (about 150 layer kernels; each is HBM-bound)',
                   'algorithmic_bytes_per_step': hbm_bytes,
                   'traffic_model': 'fp32 activations: forward ' + ('Y (fused: X stays on chip)' if fused_fwd else 'X + Y') + ', weight gradient X + dY, data gradient dY + Y + dX',
                   'algorithmic_tflops': flop / dt / 1e12, 'mfma_frac_of_2500': flop / dt / 2.5e15},
      'loss_first': losses[0], 'loss_last': losses[-1],
  }
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
  print(json.dumps(result), flush=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--rays', type=int, default=480000, help='rays per rank per step (800x600 frame); with --strong: rays of the ONE frame')
  ap.add_argument('--chunk', type=int, default=65536)
  ap.add_argument('--precision', default='bf16', choices=['bf16', 'bf16x3', 'f32', 'f16', 'mixed'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-other-paths', action='store_true', help='skip the parity_path / other_paths legs (N = 1 only)')
  ap.add_argument('--graph', default='nerf_ds', choices=['nerf_ds', 'hypernerf'], help="'hypernerf' = configs/base.gin graph (BASELINE config 5 per SURVEY 8d; use with --samples 128)")
  ap.add_argument('--samples', type=int, default=64, help='coarse = fine sample count (64 = the headline config; 128 = BASELINE config 5)')
  ap.add_argument('--strong', action='store_true', help='BASELINE configs[2]: one frame, every chunk split over the ranks (render_image)')
  ap.add_argument('--train', action='store_true', help='BASELINE configs[3]: the training step instead of the render')
  ap.add_argument('--train-rays', type=int, default=4096)
  args = ap.parse_args()

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

  from nerfds_amd import nerf_ds_config, hypernerf_config, init_params
  from nerfds_amd.model import NerfModel
  from nerfds_amd.evaluation import TrainState, make_model_fn, render_image, all_gather_into
  from nerfds_amd import _native as N

  make_cfg = nerf_ds_config if args.graph == 'nerf_ds' else hypernerf_config
  cfg = make_cfg(near=0.3, far=1.7, num_warp_embeds=256, num_coarse_samples=args.samples, num_fine_samples=args.samples)
  # 3 N field evaluations per ray x FLOP per sample (nerf_ds 1 735 168, base.gin graph 1 420 544)
  flop_per_ray = 3 * args.samples * (1735168.0 if args.graph == 'nerf_ds' else 1420544.0)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)   # random-init weights
  model = NerfModel(cfg, device=device, precision=args.precision)
  model.load_params(params)
  variables = {'params': params}
  # resident in HBM before timing; --strong: every rank holds the rays of the ONE frame, weak: its own frame
  rays = synth_rays(args.rays, cfg.num_warp_embeds, 100 + (0 if args.strong else rank), device)
  chunks = [(lo, min(lo + args.chunk, args.rays)) for lo in range(0, args.rays, args.chunk)]
  chunk_rays = [{k: (v[lo:hi] if not isinstance(v, dict) else {kk: vv[lo:hi] for kk, vv in v.items()})
                 for k, v in rays.items()} for lo, hi in chunks]
  compute = torch.cuda.current_stream(device)
  comm = torch.cuda.Stream(device) if world > 1 else None
  frame = torch.empty((args.rays, N.RAY_REC), dtype=torch.float32, device=device)               # this rank's records
  gathered = [torch.empty((world * (hi - lo), N.RAY_REC), dtype=torch.float32, device=device) for lo, hi in chunks[:2]] if world > 1 else None

  def step_weak(seed, precision=None):
    """Every chunk: the fused kernel writes this rank's records straight into its frame buffer; with N ranks the
    all-gather of the chunk (the path's only exchange) runs on a side stream under the next chunk's kernel."""
    for ci, (cr, (lo, hi)) in enumerate(zip(chunk_rays, chunks)):
      model.apply(variables, cr, EXTRA, rngs={'coarse': seed, 'fine': seed + 500}, ray_offset=lo,
                  use_predicted_norm=cfg.predict_norm, return_points=False, mask_ratio=1, sharp_weights_std=0.1,
                  precision=precision, records_out={'fine': frame[lo:hi]})
      if world > 1:
        slot = ci & 1
        ready = torch.cuda.Event()
        ready.record(compute)
        with torch.cuda.stream(comm):
          comm.wait_event(ready)
          all_gather_into(gathered[slot][:world * (hi - lo)], frame[lo:hi])
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
    model.kernel_time_ms(reset=True)
    t0 = time.perf_counter()
    for i in range(steps):
      step(warmup + i, precision)
    sync()
    elapsed = time.perf_counter() - t0
    n_launch, kernel_ms = model.kernel_time_ms(reset=False)
    return elapsed, n_launch, kernel_ms

  elapsed, n_launch, kernel_ms = timed(args.steps, args.warmup)
  if world > 1:
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    ms_per_step = elapsed * 1e3 / args.steps
    total_rays = args.rays if args.strong else args.rays * world
    value = total_rays / (elapsed / args.steps)
    launches_per_step = len(chunks)
    rays_per_launch = (args.rays / world if args.strong else args.rays) / launches_per_step
    avg_launch_s = (kernel_ms / max(n_launch, 1)) * 1e-3
    achieved = rays_per_launch * flop_per_ray / avg_launch_s / 1e12 if n_launch else None
    peak = PEAK_TFLOPS[args.precision]
    # HBM bytes per launch cannot be read live (PMC passes are separate rocprofv3 runs of this same command): the
    # committed measurement of this round's kernel is quoted, with its source, when this run is the configuration it was taken on.
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, TRAFFIC_FILE)
    if (os.path.exists(tpath) and args.precision == 'bf16' and args.rays == 480000 and args.chunk == 65536 and args.samples == 64
        and args.graph == 'nerf_ds' and not args.strong):
      tj = json.load(open(tpath))
      traffic, traffic_source = tj['hbm_bytes_per_launch'], f"{TRAFFIC_FILE} ({tj.get('measured_on', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command')})"
    nets = ('mask + predicted-normal NerfMLP (configs/nerf_ds.gin graph)' if args.graph == 'nerf_ds'
            else 'NerfMLP with posenc identity (configs/base.gin HyperNeRF graph)')
    what = ('ONE 800x600 frame (480000 rays), every 65536-ray chunk split over the ranks' if args.strong
            else '800x600 frame per GPU (480000 rays)')
    workload = (f"NeRF-DS 'bell'-shaped synthetic scene, {what}, {args.samples} coarse + {args.samples} fine "
                f'samples ({2 * args.samples} on the fine pass, {3 * args.samples} field evaluations/ray), SE(3) warp + hyper-slice + '
                f'{nets}, random-init weights')
    exchange = 'none (1 GPU)' if world == 1 else ('all-gather of the [chunk / N, 26] fp32 ray records of every chunk, on a side stream under the next chunk' if args.strong
                                                  else 'all-gather of [chunk, 26] fp32 ray records, on a side stream under the next chunk')
    result = {
        'metric': 'rendered rays/sec (%d samples/ray, full warp+NerfMLP)' % (2 * args.samples),
        'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'strong' if args.strong else 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': workload,
                   'rays_per_gpu_per_step': args.rays // world if args.strong else args.rays, 'chunk': args.chunk,
                   'parallelism': f'ray-shard x{world}' + (' of every chunk (evaluation.render_image)' if args.strong else ''),
                   'exchange': exchange},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': (achieved / peak) if achieved else None, 'traffic': traffic, 'traffic_source': traffic_source,
                     'kernel': 'nerfds::render_rays_kernel<%s, %s>' % ('GraphNerfDS' if args.graph == 'nerf_ds' else 'GraphHyperNeRF', args.precision),
                     'avg_launch_ms': avg_launch_s * 1e3, 'launches': n_launch,
                     'algorithmic_flop_per_launch': rays_per_launch * flop_per_ray},
    }
    sample = None
    if world == 1 and not args.no_cpu_baseline:
      result['cpu_baseline'], sample = cpu_baseline(cfg, params)
    else:
      result['cpu_baseline'] = None
    if world == 1 and not args.no_other_paths and not args.strong:
      # the same frame in the other arithmetic modes, timed in this run; error of every mode against the CPU oracle on the
      # baseline's sample (same rays, same uniforms) when the baseline ran
      paths = {}
      for prec in ('bf16x3', 'f16', 'mixed'):
        if prec == args.precision:
          continue
        steps = 2 if prec == 'bf16x3' else 3
        el, nl, kms = timed(steps, 1, prec)
        launch_s = kms / max(nl, 1) * 1e-3
        ach = rays_per_launch * flop_per_ray / launch_s / 1e12
        paths[prec] = {'precision': prec, 'value': args.rays / (el / steps), 'unit': 'rays/s', 'ms_per_step': el * 1e3 / steps, 'steps': steps,
                       'avg_launch_ms': launch_s * 1e3, 'roofline_frac': ach / PEAK_TFLOPS[prec],
                       'rgb_max_rel_err': rgb_error(model, cfg, params, sample, prec) if sample else None}
      plan = (C_int32 * 5)()
      N.load().nerfds_precision_plan(N.PREC['mixed'], plan)
      names = ('bf16', 'bf16x3', 'f32', 'f16')
      paths['mixed']['plan'] = dict(zip(('mask', 'warp', 'hyper', 'trunk', 'rgb'), (names[v] for v in plan)))
      if sample:
        result['rgb_max_rel_err'] = rgb_error(model, cfg, params, sample, args.precision)
      pp = paths.pop('bf16x3', None)
      if pp is not None:
        pp['meets_1e-4'] = (pp['rgb_max_rel_err'] is not None and pp['rgb_max_rel_err'] <= 1e-4)
        pp['note'] = ('split bf16 (hi + lo) operands, three MFMAs per product, fp32 accumulate: the fastest arithmetic that meets '
                      "north_star's 1e-4 on composited RGB (profiles/r2_precision_budget.md: no plan with a one-MFMA network does)")
        result['parity_path'] = pp
      result['other_paths'] = list(paths.values())
      result['err_reference'] = ('CPU oracle (torch fp32) on the cpu_baseline sample: same rays, same injected uniforms; statistic = max |d rgb| / max |rgb|, '
                                 'worst of coarse / fine') if sample else None
    print(json.dumps(result), flush=True)

  if world > 1:
    dist.destroy_process_group()


from ctypes import c_int32 as C_int32      # noqa: E402

if __name__ == '__main__':
  main()
