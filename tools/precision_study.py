"""Development: error of each arithmetic mode against the fp32-MFMA kernel (which matches the fp64 oracle to ~4e-6) on
many rays, plus launch time at 65 536 rays.  One library per process: NERFDS_LIB selects a variant build.
usage (GPU box): python tools/precision_study.py [prec,prec,...] [rays]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params, _native as N
from nerfds_amd.model import NerfModel
import ctypes as C

precs = sys.argv[1].split(',') if len(sys.argv) > 1 else ['bf16', 'f16', 'mixed', 'bf16x3']
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device('cuda', 0)
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
plan = (C.c_int32 * 5)()
N.load().nerfds_precision_plan(4, plan)
names = ('bf16', 'bf16x3', 'f32', 'f16')
res = {'lib': os.path.basename(N.LIB_PATH), 'mixed_plan': [names[v] for v in plan]}

def rays_for(R, seed, spread, n_ids):
  rng = np.random.default_rng(seed)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  return dict(origins=torch.tensor(rng.normal(size=(R, 3)) * spread, dtype=torch.float32, device=dev),
              directions=torch.tensor(d, dtype=torch.float32, device=dev), viewdirs=torch.tensor(d, dtype=torch.float32, device=dev),
              metadata={'warp': torch.tensor(rng.integers(0, n_ids, (R, 1)), device=dev)},
              mask=torch.tensor((rng.random((R, 1)) < 0.3).astype(np.float32), device=dev)), rng

cases = {'trained': dict(seed=0, kw=dict(warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1), spread=0.1),
         'init': dict(seed=1, kw={}, spread=0.3)}
cfg = nerf_ds_config(num_warp_embeds=8)
m = NerfModel(cfg, device=dev)
for cname, c in cases.items():
  params = init_params(cfg, c['seed'], **c['kw'])
  rays, rng = rays_for(R, 2, c['spread'], 8)
  t = torch.tensor(rng.random((R, 64)), dtype=torch.float32, device=dev)
  u = torch.tensor(rng.random((R, 64)), dtype=torch.float32, device=dev)
  ref = m.apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, precision='f32')
  ref = {l: {k: v.clone() for k, v in ref[l].items()} for l in ref}
  for prec in precs:
    out = m.apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, precision=prec)
    e = {}
    for l in ('coarse', 'fine'):
      d = (out[l]['rgb'] - ref[l]['rgb']).abs()
      e[l] = dict(max=float(d.max() / ref[l]['rgb'].abs().max()), p999=float(torch.quantile(d.flatten(), 0.999)),
                  rms=float(d.pow(2).mean().sqrt()), depth=float((out[l]['depth'] - ref[l]['depth']).abs().max() / ref[l]['depth'].abs().max()))
    res.setdefault(cname, {})[prec] = e
    print(f"{res['lib']} {cname:8s} {prec:7s} rgb max-rel coarse {e['coarse']['max']:.2e} fine {e['fine']['max']:.2e}  (p99.9 {e['fine']['p999']:.1e} rms {e['fine']['rms']:.1e}) depth {e['fine']['depth']:.1e}", flush=True)

# timing at 65536 rays, on-chip jitter
RT = 65536
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rays, rng = rays_for(RT, 3, 0.2, 8)
for prec in precs:
  ms = []
  for it in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    m.apply({'params': params}, rays, EXTRA, rngs={'coarse': 1, 'fine': 2}, use_predicted_norm=True, precision=prec)
    torch.cuda.synchronize(); ms.append((time.time() - t0) * 1e3)
  best = min(ms[1:])
  res.setdefault('time_ms_65536', {})[prec] = best
  print(f"{res['lib']} time {prec:7s} {best:.2f} ms / 65536 rays = {RT / best / 1e3:.3f} Mrays/s  frac {RT * 333.15e6 / (best * 1e-3) / 2.5e15:.3f}", flush=True)
print('JSON ' + json.dumps(res))
