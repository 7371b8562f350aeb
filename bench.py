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
path's only exchange: one RCCL all-gather of the [65 536, 26] per-ray records.

Prints ONE JSON line (rank 0).  `roofline` is for the fused kernel: algorithmic FLOPs per launch (BASELINE.md
section 2: 333.15 MFLOP/ray) over the mean launch duration measured with HIP events on the launch stream.
`cpu_baseline` times the CPU oracle (torch fp32, all host cores) on a bounded sample of the same workload.
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
PEAK_TFLOPS = {'bf16': 2500.0, 'bf16x3': 2500.0, 'f32': 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)


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
  ids = torch.full((R, 1), int(torch.randint(0, n_ids, (1,), generator=g)), dtype=torch.int64)
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


def cpu_baseline(cfg, params, budget_s=15.0):
  """CPU oracle (torch fp32 restatement, vectorised over [R*S, K], all usable host cores) on a bounded ray sample."""
  from oracle import nerfds_oracle as O
  cores = min(available_cores(), 64)
  torch.set_num_threads(cores)
  model = O.NerfModel(cfg, params, torch.float32)

  def run(R):
    rays = {k: (v.cpu() if not isinstance(v, dict) else {kk: vv.cpu() for kk, vv in v.items()})
            for k, v in synth_rays(R, cfg.num_warp_embeds, 1, 'cpu').items()}
    rng = np.random.default_rng(0)
    t, u = rng.random((R, cfg.num_coarse_samples)), rng.random((R, cfg.num_fine_samples))
    t0 = time.perf_counter()
    model.apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=cfg.predict_norm, compute_sigma_gradient=False)
    return time.perf_counter() - t0

  t_all = time.perf_counter()
  R = 16
  dt = run(R)                               # also warms the thread pool
  while time.perf_counter() - t_all + 4 * dt < budget_s and R < 16384:     # grow the sample while it fits the budget
    R *= 4
    dt = run(R)
  return {'value': R / dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
          'sample': f'{R} rays x ({cfg.num_coarse_samples} coarse + {cfg.num_coarse_samples + cfg.num_fine_samples} fine) samples of the same nerf_ds graph, torch-CPU fp32 oracle, '
                    f'{dt:.1f} s on {cores} threads, sigma-gradient off as on the GPU'}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=10)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--rays', type=int, default=480000, help='rays per rank per step (800x600 frame)')
  ap.add_argument('--chunk', type=int, default=65536)
  ap.add_argument('--precision', default='bf16', choices=['bf16', 'bf16x3', 'f32'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--graph', default='nerf_ds', choices=['nerf_ds', 'hypernerf'], help="'hypernerf' = configs/base.gin graph (BASELINE config 5 per SURVEY 8d; use with --samples 128)")
  ap.add_argument('--samples', type=int, default=64, help='coarse = fine sample count (64 = the headline config; 128 = BASELINE config 5)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}')
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
  torch.cuda.set_device(local_rank)
  device = torch.device('cuda', local_rank)
  if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', device_id=device)      # "nccl" is RCCL on ROCm

  from nerfds_amd import nerf_ds_config, hypernerf_config, init_params
  from nerfds_amd.model import NerfModel
  from nerfds_amd.evaluation import all_gather_records

  make_cfg = nerf_ds_config if args.graph == 'nerf_ds' else hypernerf_config
  cfg = make_cfg(near=0.3, far=1.7, num_warp_embeds=256, num_coarse_samples=args.samples, num_fine_samples=args.samples)
  # 3 N field evaluations per ray x FLOP per sample (nerf_ds 1 735 168, base.gin graph 1 420 544)
  flop_per_ray = 3 * args.samples * (1735168.0 if args.graph == 'nerf_ds' else 1420544.0)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)   # random-init weights
  model = NerfModel(cfg, device=device, precision=args.precision)
  model.load_params(params)
  variables = {'params': params}
  rays = synth_rays(args.rays, cfg.num_warp_embeds, 100 + rank, device)       # resident in HBM before timing
  chunks = [(lo, min(lo + args.chunk, args.rays)) for lo in range(0, args.rays, args.chunk)]
  chunk_rays = [{k: (v[lo:hi] if not isinstance(v, dict) else {kk: vv[lo:hi] for kk, vv in v.items()})
                 for k, v in rays.items()} for lo, hi in chunks]

  def step(seed):
    out = None
    for ci, cr in enumerate(chunk_rays):
      model.apply(variables, cr, EXTRA, rngs={'coarse': seed * 1000 + ci, 'fine': seed * 1000 + ci + 500},
                  use_predicted_norm=cfg.predict_norm, return_points=False, mask_ratio=1, sharp_weights_std=0.1)
      out = all_gather_records(model.last_records['fine'])      # the path's only exchange (one collective per chunk)
    return out

  def sync():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  for i in range(args.warmup):
    step(i)
  sync()
  model.kernel_time_ms(reset=True)
  t0 = time.perf_counter()
  for i in range(args.steps):
    step(args.warmup + i)
  sync()
  elapsed = time.perf_counter() - t0
  n_launch, kernel_ms = model.kernel_time_ms(reset=False)
  if world > 1:
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    ms_per_step = elapsed * 1e3 / args.steps
    value = args.rays * world / (elapsed / args.steps)
    rays_per_launch = args.rays / len(chunks)
    avg_launch_s = (kernel_ms / max(n_launch, 1)) * 1e-3
    achieved = rays_per_launch * flop_per_ray / avg_launch_s / 1e12 if n_launch else None
    peak = PEAK_TFLOPS[args.precision]
    # HBM bytes per launch cannot be read live (PMC passes are separate rocprofv3 runs of this same command):
    # the committed measurement of the current kernel is used when present (profiles/, see tools/prof_bench.sh).
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', f'r1_{args.precision}_hbm_traffic.json')
    if os.path.exists(tpath) and args.rays == 480000 and args.chunk == 65536 and args.samples == 64 and args.graph == 'nerf_ds':
      traffic = json.load(open(tpath))['hbm_bytes_per_launch']
    nets = ('mask + predicted-normal NerfMLP (configs/nerf_ds.gin graph)' if args.graph == 'nerf_ds'
            else 'NerfMLP with posenc identity (configs/base.gin HyperNeRF graph)')
    workload = (f"NeRF-DS 'bell'-shaped synthetic scene, 800x600 frame per GPU (480000 rays), {args.samples} coarse + {args.samples} fine "
                f'samples ({2 * args.samples} on the fine pass, {3 * args.samples} field evaluations/ray), SE(3) warp + hyper-slice + '
                f'{nets}, random-init weights')
    result = {
        'metric': 'rendered rays/sec (%d samples/ray, full warp+NerfMLP)' % (2 * args.samples),
        'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {'workload': workload,
                   'rays_per_gpu_per_step': args.rays, 'chunk': args.chunk, 'parallelism': f'ray-shard x{world}',
                   'exchange': 'all-gather of [chunk, 26] fp32 ray records' if world > 1 else 'none (1 GPU)'},
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': (achieved / peak) if achieved else None, 'traffic': traffic,
                     'kernel': 'nerfds::render_rays_kernel<%s, %s>' % ('GraphNerfDS' if args.graph == 'nerf_ds' else 'GraphHyperNeRF', args.precision),
                     'avg_launch_ms': avg_launch_s * 1e3, 'launches': n_launch,
                     'algorithmic_flop_per_launch': rays_per_launch * flop_per_ray},
    }
    if world == 1 and not args.no_cpu_baseline:
      result['cpu_baseline'] = cpu_baseline(cfg, params)
    else:
      result['cpu_baseline'] = None
    print(json.dumps(result), flush=True)

  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
