"""GPU (development): the fine level of precision='bf16x3_fine' against the fp32-MFMA kernel on the shapes the frame test does not cover - the wide
work shape (128 + 128 samples) and the HyperNeRF graph.  usage: python tools/x3fine_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerfds_amd import nerf_ds_config, hypernerf_config, init_params
from nerfds_amd.model import NerfModel

for name, mk, n in (('nerf_ds 128+128', nerf_ds_config, 128), ('hypernerf 128+128', hypernerf_config, 128), ('hypernerf 64+64', hypernerf_config, 64)):
  cfg = mk(near=0.3, far=1.7, num_warp_embeds=16, num_coarse_samples=n, num_fine_samples=n)
  p = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  rng = np.random.default_rng(0); R = 8192
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  rays = dict(origins=rng.normal(size=(R, 3)) * 0.2, directions=d, viewdirs=d, metadata={'warp': rng.integers(0, 16, (R, 1))}, mask=np.zeros((R, 1)))
  ex = dict(nerf_alpha=8., warp_alpha=6., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
  m = NerfModel(cfg, device=torch.device('cuda', 0))
  o = {pr: m.apply({'params': p}, rays, ex, rngs={'coarse': 1, 'fine': 2}, use_predicted_norm=cfg.predict_norm, precision=pr) for pr in ('f32', 'bf16x3', 'bf16x3_fine')}
  for pr in ('bf16x3', 'bf16x3_fine'):
    print(name, pr, {lv: f"{float((o[pr][lv]['rgb'] - o['f32'][lv]['rgb']).abs().max() / o['f32'][lv]['rgb'].abs().max()):.2e}" for lv in ('fine', 'coarse')}, flush=True)
