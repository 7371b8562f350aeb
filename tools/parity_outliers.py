"""GPU + CPU oracle: are the rays on which the split-bf16 kernel leaves 1e-4 (tools/parity_sweep.py: some random-init scenes of the nerf_ds graph) badly conditioned,
or is 16-bit operand arithmetic short?  Deterministic depths (use_stratified_sampling=False), one scene: full frame in the fp32-MFMA and the split-bf16 kernel, the 48 rays
with the largest difference, then the kernels and the fp64 oracle on exactly those rays.  --stratified: stratified depths from INJECTED uniforms (the oracle cannot draw the
on-chip Philox stream; injected t_rand / u_rand give kernel and oracle the same depths), --rank-by f16x3: the rays on which the split-f16 kernel is furthest from the fp32-MFMA one.
  python tools/parity_outliers.py --seed 15 --glo 309 --near 0.3 --far 2.0 [--stratified] [--rank-by f16x3]"""
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
ap.add_argument('--stratified', action='store_true'); ap.add_argument('--rank-by', default='bf16x3', choices=('bf16x3', 'f16x3'))
a = ap.parse_args()
dev = torch.device('cuda', 0)
cfg = nerf_ds_config(num_warp_embeds=a.glo, near=a.near, far=a.far, use_stratified_sampling=a.stratified)
params = init_params(cfg, a.seed, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rays = bench.synth_rays(a.rays, a.glo, a.seed, dev)
m = NerfModel(cfg, device=dev, precision='f32')
EX = dict(bench.EXTRA)
gen = torch.Generator(device='cpu').manual_seed(a.seed)
T = torch.rand((a.rays, cfg.num_coarse_samples), generator=gen).to(dev) if a.stratified else None
U = torch.rand((a.rays, cfg.num_fine_samples), generator=gen).to(dev) if a.stratified else None
PRECS = ('f32', 'bf16x3', 'f16x3')
def render(prec, r):
  n = r['origins'].shape[0]
  f, c = torch.empty((n, 26), device=dev), torch.empty((n, 26), device=dev)
  for lo in range(0, n, 65536):
    hi = min(lo + 65536, n)
    cr = {k: (v[lo:hi] if not isinstance(v, dict) else {kk: vv[lo:hi] for kk, vv in v.items()}) for k, v in r.items()}
    kw = dict(t_rand=T[lo:hi], u_rand=U[lo:hi]) if a.stratified else {}
    m.apply({'params': params}, cr, EX, use_predicted_norm=True, precision=prec, records_out={'fine': f[lo:hi], 'coarse': c[lo:hi]}, **kw)
  torch.cuda.synchronize()
  return {'fine': f[:, :3].clone(), 'coarse': c[:, :3].clone()}
full = {p: render(p, rays) for p in PRECS}
out = {'scene': vars(a)}
for lv in ('fine', 'coarse'):
  scale = float(full['f32'][lv].abs().max())
  for p in PRECS[1:]:
    dp = (full[p][lv] - full['f32'][lv]).abs().max(dim=1).values
    out[f'{lv}_{p}_full_frame_max_rel'] = float(dp.max()) / scale
    out[f'{lv}_{p}_rays_over_1e-4'] = int((dp / scale > 1e-4).sum())
  d = (full[a.rank_by][lv] - full['f32'][lv]).abs().max(dim=1).values
  idx = torch.topk(d, a.top).indices
  sub = {k: (v[idx] if not isinstance(v, dict) else {kk: vv[idx] for kk, vv in v.items()}) for k, v in rays.items()}
  sub_np = {k: (v.cpu().numpy() if not isinstance(v, dict) else {kk: vv.cpu().numpy() for kk, vv in v.items()}) for k, v in sub.items()}
  okw = dict(t_rand=T[idx].cpu().numpy(), u_rand=U[idx].cpu().numpy()) if a.stratified else {}
  ref = O.NerfModel(cfg, params).apply(sub_np, EX, use_predicted_norm=True, compute_sigma_gradient=False, **okw)[lv]['rgb'].numpy()      # fp64
  for p in PRECS:
    got = full[p][lv][idx].cpu().numpy()
    e = np.abs(got - ref).max(axis=1) / scale
    out[f'{lv}_{p}_vs_fp64_oracle_on_the_worst_{a.top}_rays'] = {'max': float(e.max()), 'median': float(np.median(e)), 'rays_over_1e-4': int((e > 1e-4).sum())}
  out[f'{lv}_{a.rank_by}_vs_f32_on_the_worst_{a.top}_rays'] = {'max': float((d[idx] / scale).max()), 'median': float((d[idx] / scale).median())}
print(json.dumps(out))
