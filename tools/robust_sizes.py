"""Trainer at awkward sizes (ray counts that are not multiples of anything, 128+128 samples, coarse-only, the full objective): the two
GEMM modes must agree and stay finite.  Dev check for the GPU box: python tools/robust_sizes.py"""
import os, sys, numpy as np, torch
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'nerf-ds_amd')); sys.path.insert(0, _ROOT)
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.training import Trainer
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
OBJ = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1, norm_loss_weight=0.1)
dev = torch.device('cuda', 0)
for R, Nc, Nf, full in ((1000, 64, 64, False), (1000, 64, 64, True), (777, 128, 128, False), (4096, 128, 128, False), (4099, 64, 0, False), (31, 64, 64, True)):
  cfg = nerf_ds_config(num_warp_embeds=16, num_coarse_samples=Nc, num_fine_samples=Nf, near=0.3, far=1.7)
  params = init_params(cfg, 0, warp_head_scale=5e-2)
  rng = np.random.default_rng(2)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  f = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
  batch = dict(origins=f(rng.normal(size=(R, 3)) * 0.2), directions=f(d), viewdirs=f(d), metadata={'warp': torch.as_tensor(rng.integers(0, 16, (R, 1)), device=dev)},
               mask=f((rng.random((R, 1)) < 0.3)), rgb=f(rng.random((R, 3))))
  res = {}
  for mode in ('mfma', 'rocblas'):
    os.environ['NERFDS_TRAIN_GEMM'] = mode
    tr = Trainer(cfg, params, max_rays=R)
    st = tr.step(batch, EX, 0.0, grads_only=True, objective=OBJ if full else None)
    g = tr._download(1)
    res[mode] = (st['loss/total'], g)
    del tr; torch.cuda.empty_cache()
  a, b = res['mfma'], res['rocblas']
  rel = np.linalg.norm(a[1] - b[1]) / np.linalg.norm(b[1])
  print(R, Nc, Nf, full, 'loss', a[0], b[0], 'grad rel diff', rel, 'finite', np.isfinite(a[1]).all())
