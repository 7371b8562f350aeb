"""GPU: where the wall time of Trainer.step goes beyond the device timeline (BASELINE config 4 shape unless --rays): the Python marshalling, the C call's
launches, the per-step read-back (synchronisation).  python tools/step_overhead.py [--rays 4096] [--steps 30]"""
import argparse, ctypes as C, os, sys, time
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd import _native as N
from nerfds_amd.training import Trainer
ap = argparse.ArgumentParser(); ap.add_argument('--rays', type=int, default=4096); ap.add_argument('--steps', type=int, default=30)
a = ap.parse_args()
R = a.rays
dev = torch.device('cuda', 0)
cfg = nerf_ds_config(num_warp_embeds=64, near=0.3, far=1.7)
params = init_params(cfg, 0, warp_head_scale=5e-2)
rng = np.random.default_rng(2)
d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
f = lambda x: torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
batch = dict(origins=f(rng.normal(size=(R, 3)) * 0.2), directions=f(d), viewdirs=f(d), metadata={'warp': torch.as_tensor(rng.integers(0, 64, (R, 1)), device=dev)},
             mask=f(rng.random((R, 1)) < 0.3), rgb=f(rng.random((R, 3))))
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
tr = Trainer(cfg, params, max_rays=R, device=dev)
for _ in range(5):
  tr.step(batch, EX, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
  tr.step(batch, EX, 1e-3)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / a.steps
# the C call alone, structs built once, with the per-step read-back
wid = batch['metadata']['warp'].reshape(-1).to(torch.int32).contiguous()
gm = batch['mask'].reshape(-1).contiguous()
rays = N.Rays(num_rays=R, origins=batch['origins'].data_ptr(), directions=batch['directions'].data_ptr(), viewdirs=batch['viewdirs'].data_ptr(),
              warp_id=wid.data_ptr(), gt_mask=gm.data_ptr(), camera=None, first_pixel=0)
ex = N.Extra(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4., mask_ratio=1.0, near=0.3, far=1.7, use_stratified_sampling=1, use_linear_disparity=0)
loss = (C.c_float * 16)()
s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
lib = tr._lib
def call(i, host):
  rnd = N.Rand(t_rand=None, u_rand=None, seed=1000 + i, first_ray=0)
  rc = lib.nerfds_trainer_step(tr._h, C.byref(rays), batch['rgb'].data_ptr(), C.byref(ex), C.byref(rnd), None, C.c_float(1e-3), 0, loss if host else None, s)
  assert rc == 0, rc
for i in range(3): call(i, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps): call(i, True)
torch.cuda.synchronize()
c_sync = (time.perf_counter() - t0) / a.steps
t0 = time.perf_counter()
for i in range(a.steps): call(i, False)
t_enq = (time.perf_counter() - t0) / a.steps
torch.cuda.synchronize()
c_async = (time.perf_counter() - t0) / a.steps
print(f'rays {R}: Trainer.step {full * 1e3:.3f} ms | C call + per-step read-back {c_sync * 1e3:.3f} ms | C call, no read-back: {c_async * 1e3:.3f} ms per step (host enqueue alone {t_enq * 1e3:.3f} ms)')
