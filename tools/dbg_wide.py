import sys
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.')
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel
from oracle import nerfds_oracle as O
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=128, num_fine_samples=128)
params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
R=7; rng = np.random.default_rng(12)
d = rng.normal(size=(R,3)); d/=np.linalg.norm(d,axis=-1,keepdims=True)
rays = dict(origins=rng.normal(size=(R,3))*0.2, directions=d, viewdirs=d, metadata={'warp': rng.integers(0,4,(R,1))}, mask=(rng.random((R,1))<0.3).astype(np.float32))
t,u = rng.random((R,128)), rng.random((R,128))
m = NerfModel(cfg, device=torch.device('cuda',0))
import copy
np.set_printoptions(precision=2, linewidth=250)
def variant(name):
  p = copy.deepcopy(params)
  for lv in ('nerf_mlps_coarse', 'nerf_mlps_fine'):
    k = p[lv]['rgb_mlp']['hidden_0']['kernel']
    if name == 'no_cond': k[256:] = 0
    if name == 'no_trunk': k[:256] = 0
    if name == 'no_vd': k[256:256+27] = 0
    if name == 'only_vd': k[:256] = 0; k[256+27:] = 0
    if name == 'only_rest': k[:256+27] = 0
    if name == 'no_bias': p[lv]['rgb_mlp']['hidden_0']['bias'][:] = 0; p[lv]['bottleneck']['bias'][:] = 0
  return p
for name in ('base', 'no_cond', 'no_trunk', 'no_vd', 'only_vd', 'only_rest', 'no_bias'):
  p = variant(name)
  ref = O.NerfModel(cfg, p).apply(rays, EX, t_rand=t, u_rand=u, use_predicted_norm=True, return_weights=True, return_points=True, compute_sigma_gradient=False)
  for prec in ('bf16x3',):
    out = m.apply({'params': p}, rays, EX, t_rand=t, u_rand=u, use_predicted_norm=True, return_samples=True, precision=prec)
    for level in ('coarse','fine'):
        for k in ('sample_rgb', 'sigma'):
            a = out[level][k].cpu().numpy().reshape(R, -1, out[level][k].shape[-1] if out[level][k].ndim==3 else 1); b = ref[level][k].numpy().reshape(a.shape)
            err = np.abs(a-b).max(-1)/max(np.abs(b).max(),1e-6)
            print(name, prec, level, k, 'max %.1e' % err.max(), 'n_bad', int((err > 1e-4).sum()))
