import sys
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.')
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
N = 128; prec = sys.argv[1]; outf = sys.argv[2]
cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=N, num_fine_samples=N)
params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
R=7; rng = np.random.default_rng(12)
d = rng.normal(size=(R,3)); d/=np.linalg.norm(d,axis=-1,keepdims=True)
rays = dict(origins=rng.normal(size=(R,3))*0.2, directions=d, viewdirs=d, metadata={'warp': rng.integers(0,4,(R,1))}, mask=(rng.random((R,1))<0.3).astype(np.float32))
t,u = rng.random((R,N)), rng.random((R,N))
m = NerfModel(cfg, device=torch.device('cuda',0))
res = {}
for rep in range(3):
    out = m.apply({'params': params}, rays, EX, t_rand=t, u_rand=u, use_predicted_norm=True, return_samples=True, precision=prec)
    for level in ('coarse','fine'):
        for k in ('predicted_norm', 'sample_rgb', 'sigma'):
            res[f'{rep}_{level}_{k}'] = out[level][k].cpu().numpy()
np.savez(outf, **res)
