#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_7; mkdir -p $O
( timeout 2400 python -m pytest tests -m gpu -q -s 2>&1 | grep -E "coarse|fine|passed|failed|FAILED|^E " | head -400 ) > $O/gputests_s.log 2>&1
python - > $O/sigma_grad_quantiles.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, 'nerf-ds_amd'); sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import test_render_image_gpu as T
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel
from oracle import nerfds_oracle as O
cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=12, num_fine_samples=12)
params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
f = T._frame_rays(5, 6, 4, 4)
rng = np.random.default_rng(1)
t, u = rng.random((30, 12)), rng.random((30, 12))
flat = {k: (v.reshape(30, -1) if k != 'metadata' else {'warp': v['warp'].reshape(30, 1)}) for k, v in f.items()}
ref = O.NerfModel(cfg, params).apply(flat, T.EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=True)
m = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32'); m.sigma_gradient_block = 16
out = m.apply({'params': params}, f, T.EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, use_sigma_gradient=True, precision='f32')
for level, S in (('coarse', 12), ('fine', 24)):
  got = out[level]['target_norm'].cpu().numpy()
  cos = (got.reshape(30, S, 3) * ref[level]['target_norm'].numpy()).sum(-1)
  e = np.sort((1 - cos).ravel())
  print(level, 'n', e.size, 'median', np.median(e), 'q90', np.quantile(e, .9), 'q97', np.quantile(e, .97), 'q99', np.quantile(e, .99), 'max', e.max())
PY
tail -5 $O/gputests_s.log; cat $O/sigma_grad_quantiles.txt | tail -4
