"""Development aid: checks the fused backward's first layers against numpy on the buffers of one step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_training as TT
from nerfds_amd.training import Trainer
R, Nc, Nf = 6, 8, 0
cfg, params, batch, t, u = TT._problem(R, Nc, Nf)
tr = Trainer(cfg, params, max_rays=R)
tr.step(batch, TT.EX, 0.0, t_rand=t, mask_ratio=1.0, grads_only=True)
M = R * Nc
def bits_to_mask(bits, W):      # [(row*2+h)*(W/32) + tile] u16 -> [M][W] bool
  b = bits.reshape(M, 2, W // 32)
  m = np.zeros((M, W), bool)
  for tl in range(W // 32):
    for h in range(2):
      for r in range(16):
        m[:, 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * h] = (b[:, h, tl] >> r) & 1
  return m
P = params['nerf_mlps_coarse']
rgb_h = tr.debug_read('rgb_h16', (M, 128), np.float16).astype(np.float32)
rgb_bits = tr.debug_read('rgb_bits', (M * 2 * 4,), np.uint16)
mk = bits_to_mask(rgb_bits, 128)
print('rgb mask == (h16 > 0):', (mk == (rgb_h > 0)).mean())
d_rgb = tr.debug_read('d_rgb_logit', (M, 3))
g_rgb = tr.debug_read('rgb_g', (M, 128))
Wr = np.asarray(P['rgb_mlp']['logit']['kernel'], np.float64)     # [128, 3]
want = (d_rgb.astype(np.float64) @ Wr.T) * (rgb_h > 0)
print('g_rgb rel err', np.abs(g_rgb - want).max() / np.abs(want).max(), ' |want|', np.abs(want).max(), '|got|', np.abs(g_rgb).max())
bad = np.abs(g_rgb - want) > 1e-3 * np.abs(want).max()
print('bad entries by row', bad.sum(1)[:16], 'by col', bad.sum(0)[:40])
h7 = tr.debug_read('trunk_h16_7', (M, 256), np.float16).astype(np.float32)
b7 = bits_to_mask(tr.debug_read('trunk_bits_7', (M * 2 * 8,), np.uint16), 256)
print('trunk7 mask == (h16 > 0):', (b7 == (h7 > 0)).mean())
g7 = tr.debug_read('trunk_g_7', (M, 256))
d_alpha = tr.debug_read('d_alpha', (M, 4))
K = np.asarray(P['rgb_mlp']['hidden_0']['kernel'], np.float64)      # rows [bott 256 | vd 24 | x 256 | nm 24]
Wb = np.asarray(P['bottleneck']['kernel'], np.float64)
F = Wb @ K[:256] + K[280:536]
Wa = np.asarray(P['alpha_mlp']['logit']['kernel'], np.float64)     # [256, 4]
want7 = (want @ F.T + d_alpha.astype(np.float64) @ Wa.T) * (h7 > 0)
print('g_7 rel err', np.abs(g7 - want7).max() / np.abs(want7).max())
unm = d_rgb.astype(np.float64) @ Wr.T
np.set_printoptions(precision=3, suppress=False, linewidth=200)
print('row0 got ', g_rgb[0, :16])
print('row0 want', want[0, :16])
print('row0 unm ', unm[0, :16])
# which unmasked column does each got column follow? (least squares over rows where got != 0)
for c in range(0, 40):
  nz = g_rgb[:, c] != 0
  if nz.sum() < 3:
    print(c, 'all zero'); continue
  best = min(range(128), key=lambda k: np.abs(unm[nz, k] - g_rgb[nz, c]).max())
  e = np.abs(unm[nz, best] - g_rgb[nz, c]).max() / np.abs(unm).max()
  print(c, '-> unmasked col', best, f'err {e:.1e}', ' nonzero rows', int(nz.sum()), ' mask rows', int(mk[:, c].sum()), ' same support', bool((nz == mk[:, c]).all()))
