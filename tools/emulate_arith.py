"""Development (CPU): what composited-RGB error would a given MFMA arithmetic give?  The oracle's nn.Dense is replaced by an
emulation of the operand formats (products exact, fp64 accumulation - the MFMA's fp32 accumulation is not the issue):
  f16     one f16 x f16 product
  bf16x3  split bf16: hi*hi + hi*lo + lo*hi
  f16mx8  f16 hi*hi + fp8-e4m3 cross terms  (w_hi8 * x_lo8 + w_lo8 * x_hi8, residuals pre-scaled by 2^SHIFT) - two
          MFMA-equivalents per product on gfx950 (v_mfma_scale_f32_32x32x64_f8f6f4 runs at twice the f16 rate)
  f16mx6  the same with fp6-e2m3-like cross terms (3 explicit significand bits), per-32-block power-of-two scale
usage: python tools/emulate_arith.py [rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from oracle import nerfds_oracle as O

R = int(sys.argv[1]) if len(sys.argv) > 1 else 96
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
f64 = torch.float64

def q(x, dt):
  return x.to(dt).to(f64)

def q_block(x, bits, axis):
  """power-of-two scale per 32-block along `axis`, then `bits` significand bits (round to nearest), range 2^-2..2^2 of the block max"""
  x = x.movedim(axis, -1)
  sh = x.shape
  pad = (-sh[-1]) % 32
  xp = torch.nn.functional.pad(x, (0, pad)).reshape(*sh[:-1], -1, 32)
  m = xp.abs().amax(-1, keepdim=True).clamp_min(1e-300)
  e = torch.floor(torch.log2(m))
  # element exponent clamped to [e - 3, e] (a 2-bit exponent field), significand `bits` bits
  ee = torch.floor(torch.log2(xp.abs().clamp_min(1e-300))).clamp(min=e - 3, max=e)
  step = torch.pow(2.0, ee - bits)
  y = torch.round(xp / step) * step
  return y.reshape(*sh[:-1], -1)[..., :sh[-1]].movedim(-1, axis)

MODE = 'exact'
SHIFT = 12
def dense(p, x):
  W, b = p['kernel'], p['bias']
  x = x.to(torch.float32).to(f64)          # layer inputs are fp32 values on the GPU
  if MODE == 'exact':
    return x @ W + b
  if MODE == 'f16':
    return q(x, torch.float16) @ q(W, torch.float16) + b
  if MODE == 'bf16x3':
    xh, wh = q(x, torch.bfloat16), q(W, torch.bfloat16)
    xl, wl = q(x - xh, torch.bfloat16), q(W - wh, torch.bfloat16)
    return xh @ wh + xl @ wh + xh @ wl + b
  xh, wh = q(x, torch.float16), q(W, torch.float16)
  s = 2.0 ** SHIFT
  if MODE == 'f16mx8':
    f8 = torch.float8_e4m3fn
    xl, wl = q((x - xh) * s, f8) / s, q((W - wh) * s, f8) / s
    return xh @ wh + xl @ q(W, f8) + q(x, f8) @ wl + b
  if MODE == 'f16mx6':
    xl, wl = q_block((x - xh), 3, -1), q_block((W - wh), 3, 0)
    return xh @ wh + xl @ q_block(W, 3, 0) + q_block(x, 3, -1) @ wl + b
  raise ValueError(MODE)
O.dense = dense

def case(name, seed, kw, spread):
  global MODE
  cfg = nerf_ds_config(num_warp_embeds=8)
  params = init_params(cfg, seed, **kw)
  rng = np.random.default_rng(2)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  rays = dict(origins=rng.normal(size=(R, 3)) * spread, directions=d, viewdirs=d, metadata={'warp': rng.integers(0, 8, (R, 1))},
              mask=(rng.random((R, 1)) < 0.3).astype(np.float32))
  t, u = rng.random((R, 64)), rng.random((R, 64))
  outs = {}
  for MODE in ('exact', 'f16', 'bf16x3', 'f16mx8', 'f16mx6'):
    o = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=False)
    outs[MODE] = {lv: o[lv]['rgb'].numpy() for lv in o}
  for m in ('f16', 'bf16x3', 'f16mx8', 'f16mx6'):
    e = {lv: np.abs(outs[m][lv] - outs['exact'][lv]).max() / np.abs(outs['exact'][lv]).max() for lv in ('coarse', 'fine')}
    print(f'{name:8s} {m:7s} rgb max-rel coarse {e["coarse"]:.2e} fine {e["fine"]:.2e}', flush=True)

case('trained', 0, dict(warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1), 0.1)
case('init', 1, {}, 0.3)
