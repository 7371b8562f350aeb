"""GPU: device time per training step through the C ABI alone, return codes ignored and nothing read back - for MEASUREMENT builds whose results are wrong
on purpose (tools/variant.sh, NERFDS_LIB=...).  python tools/time_step_raw.py [--rays 4096] [--steps 30]"""
import argparse, ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd import _native as N
from nerfds_amd.training import Trainer
ap = argparse.ArgumentParser(); ap.add_argument('--rays', type=int, default=4096); ap.add_argument('--steps', type=int, default=30)
a, _ = ap.parse_known_args()
R = a.rays
dev = torch.device('cuda', 0)
cfg = nerf_ds_config(num_warp_embeds=64, near=0.3, far=1.7)
params = init_params(cfg, 0, warp_head_scale=5e-2)
rng = np.random.default_rng(2)
d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
f = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
o, dd, rgb = f(rng.normal(size=(R, 3)) * 0.2), f(d), f(rng.random((R, 3)))
wid = torch.as_tensor(rng.integers(0, 64, (R,)), device=dev).to(torch.int32).contiguous()
gm = f(rng.random((R,)) < 0.3)
tr = Trainer(cfg, params, max_rays=R, device=dev)
rays = N.Rays(num_rays=R, origins=o.data_ptr(), directions=dd.data_ptr(), viewdirs=dd.data_ptr(), warp_id=wid.data_ptr(), gt_mask=gm.data_ptr(), camera=None, first_pixel=0)
ex = N.Extra(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4., mask_ratio=1.0, near=0.3, far=1.7, use_stratified_sampling=1, use_linear_disparity=0)
s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
def call(i):
  rnd = N.Rand(t_rand=None, u_rand=None, seed=1000 + i, first_ray=0)
  tr._lib.nerfds_trainer_step(tr._h, C.byref(rays), rgb.data_ptr(), C.byref(ex), C.byref(rnd), None, C.c_float(0.0), 0, None, s)
for i in range(5): call(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps): call(i)
torch.cuda.synchronize()
print(f'{os.environ.get("NERFDS_LIB", "shipped")}: rays {R}: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step')
