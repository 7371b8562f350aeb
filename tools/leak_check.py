"""GPU (development): create / step / destroy trainers and render contexts a dozen times; the free device memory must not drift (hipMalloc / hipFree balance)."""
import sys, os, gc, numpy as np, torch
sys.path.insert(0, 'nerf-ds_amd'); sys.path.insert(0, '.')
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.training import Trainer
from nerfds_amd.model import NerfModel
cfg = nerf_ds_config(num_warp_embeds=8, num_coarse_samples=16, num_fine_samples=16, near=0.3, far=1.7)
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rng = np.random.default_rng(0); R = 64
d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
batch = dict(origins=rng.normal(size=(R, 3)) * 0.2, directions=d, viewdirs=d, metadata={'warp': rng.integers(0, 8, (R, 1))}, mask=np.zeros((R, 1), np.float32), rgb=rng.random((R, 3)))
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
ob = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1, norm_loss_weight=0.01)
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
f0 = None
for it in range(12):
  tr = Trainer(cfg, params, max_rays=R)
  tr.step(batch, EX, 1e-3, objective=ob); tr.step(batch, EX, 1e-3)
  m = NerfModel(cfg, device=torch.device('cuda', 0)); m.apply({'params': params}, batch, EX, rngs={'coarse': 1, 'fine': 2}, use_predicted_norm=True, precision='bf16x3')
  del tr, m; gc.collect()
  f = free()
  if it == 1: f0 = f
  print(it, 'free MiB', f >> 20, flush=True)
print('drift after warm-up (MiB):', (f0 - f) / 2**20)
assert f0 - f < 64 * 2**20
