"""Development: is a kernel deterministic run to run and equivariant under a permutation of the rays?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
cfg = nerf_ds_config(num_warp_embeds=16)
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
R = 4096
rng = np.random.default_rng(11)
d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
rays = dict(origins=rng.normal(size=(R, 3)) * 0.2, directions=d, viewdirs=d, metadata={'warp': rng.integers(0, 16, (R, 1))},
            mask=(rng.random((R, 1)) < 0.3).astype(np.float32))
m = NerfModel(cfg, device=torch.device('cuda', 0), precision=prec)
t, u = np.random.default_rng(0).random((R, 64)), np.random.default_rng(1).random((R, 64))
kw = dict(t_rand=t, u_rand=u, use_predicted_norm=True, return_samples=True)
a = m.apply({'params': params}, rays, EXTRA, **kw)
a = {l: {k: v.clone() for k, v in a[l].items()} for l in a}
b = m.apply({'params': params}, rays, EXTRA, **kw)
for l in ('coarse', 'fine'):
  for k in ('rgb', 'sigma', 'sample_rgb', 'warped_points', 'predicted_mask', 'z_vals'):
    dd = (a[l][k] - b[l][k]).abs()
    print(prec, 'run-to-run', l, k, 'max diff', float(dd.max()), 'n diff', int((dd > 0).sum()))
perm = torch.randperm(R, generator=torch.Generator().manual_seed(0)).numpy()
rp = {k: (v[perm] if k != 'metadata' else {'warp': v['warp'][perm]}) for k, v in rays.items()}
c = m.apply({'params': params}, rp, EXTRA, t_rand=t[perm], u_rand=u[perm], use_predicted_norm=True, return_samples=True)
for l in ('coarse', 'fine'):
  for k in ('rgb', 'sigma', 'sample_rgb', 'warped_points', 'predicted_mask'):
    dd = (a[l][k][perm] - c[l][k]).abs()
    bad = (dd.reshape(R, -1).max(1).values > 0).nonzero().flatten()
    print(prec, 'permuted', l, k, 'max diff', float(dd.max()), 'rays differing', len(bad), 'first', bad[:6].tolist(), 'sample idx', (dd.reshape(R, dd.shape[1], -1).max(2).values > 0).nonzero()[:4].tolist() if dd.dim() > 2 else '')
