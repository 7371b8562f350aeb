"""GPU + CPU oracle: are the rays on which the split-bf16 kernel leaves 1e-4 (tools/parity_sweep.py: some random-init scenes of the nerf_ds graph) badly conditioned,
or is 16-bit operand arithmetic short?  Deterministic depths (use_stratified_sampling=False), one scene: full frame in the fp32-MFMA and the split-bf16 kernel, the 48 rays
with the largest difference, then BOTH kernels and the fp64 oracle on exactly those rays.
  python tools/parity_outliers.py --seed 15 --glo 309 --near 0.3 --far 2.0"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import bench
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel
from oracle import nerfds_oracle as O
ap = argparse.ArgumentParser()
ap.add_argument('--seed', type=int, default=15); ap.add_argument('--glo', type=int, default=309); ap.add_argument('--near', type=float, default=0.3); ap.add_argument('--far', type=float, default=2.0)
ap.add_argument('--rays', type=int, default=480000); ap.add_argument('--top', type=int, default=48)
a = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = nerf_ds_config(num_warp_embeds=a.glo, near=a.near, far=a.far, use_stratified_sampling=False)
params = init_params(cfg, a.seed, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rays = bench.synth_rays(a.rays, a.glo, a.seed, dev)
m = NerfModel(cfg, device=dev, precision='f32')
EX = dict(bench.EXTRA)
def render(prec, r):
  n = r['origins'].shape[0]
  f, c = torch.empty((n, 26), device=dev), torch.empty((n, 26), device=dev)
  for lo in range(0, n, 65536):
    hi = min(lo + 65536, n)
    cr = {k: (v[lo:hi] if not isinstance(v, dict) else {kk: vv[lo:hi] for kk, vv in v.items()}) for k, v in r.items()}
    m.apply({'params': params}, cr, EX, use_predicted_norm=True, precision=prec, records_out={'fine': f[lo:hi], 'coarse': c[lo:hi]})
  torch.cuda.synchronize()
  return {'fine': f[:, :3].clone(), 'coarse': c[:, :3].clone()}
full = {p: render(p, rays) for p in ('f32', 'bf16x3')}
out = {'scene': vars(a)}
for lv in ('fine', 'coarse'):
  d = (full['bf16x3'][lv] - full['f32'][lv]).abs().max(dim=1).values
  scale = float(full['f32'][lv].abs().max())
  out[f'{lv}_full_frame_max_rel'] = float(d.max()) / scale
  out[f'{lv}_rays_over_1e-4'] = int((d / scale > 1e-4).sum())
  idx = torch.topk(d, a.top).indices
  sub = {k: (v[idx] if not isinstance(v, dict) else {kk: vv[idx] for kk, vv in v.items()}) for k, v in rays.items()}
  sub_np = {k: (v.cpu().numpy() if not isinstance(v, dict) else {kk: vv.cpu().numpy() for kk, vv in v.items()}) for k, v in sub.items()}
  ref = O.NerfModel(cfg, params).apply(sub_np, EX, use_predicted_norm=True, compute_sigma_gradient=False)[lv]['rgb'].numpy()      # fp64
  for p in ('f32', 'bf16x3'):
    got = full[p][lv][idx].cpu().numpy()
    e = np.abs(got - ref).max(axis=1) / scale
    out[f'{lv}_{p}_vs_fp64_oracle_on_the_worst_{a.top}_rays'] = {'max': float(e.max()), 'median': float(np.median(e))}
  out[f'{lv}_bf16x3_vs_f32_on_the_worst_{a.top}_rays'] = {'max': float((d[idx] / scale).max()), 'median': float((d[idx] / scale).median())}
print(json.dumps(out))
