import sys
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.')
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel
from oracle import nerfds_oracle as O
cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=8, num_fine_samples=8)
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rng = np.random.default_rng(1); R=12
d = rng.normal(size=(R,3)); d/=np.linalg.norm(d,axis=-1,keepdims=True)
rays = dict(origins=rng.normal(size=(R,3))*0.1, directions=d, viewdirs=d, metadata={'warp': rng.integers(0,4,(R,1))}, mask=np.zeros((R,1),np.float32))
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
t,u = rng.random((R,8)), rng.random((R,8))
ref = O.NerfModel(cfg, params).apply(rays, EX, t_rand=t, u_rand=u, use_predicted_norm=True, return_weights=True, return_points=True, compute_sigma_gradient=False)
m = NerfModel(cfg, device=torch.device('cuda',0))
for prec in ('f32','bf16x3','bf16'):
    out = m.apply({'params': params}, rays, EX, t_rand=t, u_rand=u, use_predicted_norm=True, return_samples=True, precision=prec)
    for level in ('coarse','fine'):
        for k in ('predicted_mask','warped_points','sigma','predicted_norm','sample_rgb','rgb'):
            a = out[level][k].cpu().numpy(); b = ref[level][k].numpy()
            print(prec, level, k, '%.2e' % (np.abs(a-b).max()/max(np.abs(b).max(),1e-6)))
