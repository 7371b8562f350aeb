"""Measurement tool (GPU, development): which MFMA arithmetic keeps the composited RGB of the nerf_ds graph within north_star's 1e-4 AT FRAME SIZE?

Runs the torch restatement of the graph (oracle/nerfds_oracle.py - test infrastructure, used here as a measuring instrument, never by the
product) on the GPU in fp64 with every nn.Dense (modules.py:61-65, 74-78) replaced by an emulation of one candidate operand format: the
operands are rounded / split exactly as the candidate's MFMAs would see them, the products and sums are fp64 (an MFMA's fp32 accumulation is
not what separates the candidates).  All 480 000 rays of bench.py's frame (config 2: 800 x 600, 64 + 64 samples, trained-regime weights,
seeded jitter), error against the exact (fp64) run of the same rays.

  usage: python tools/precision_study_gpu.py [--rays 480000] [--chunk 8192] [--modes a,b,...] [--layers]

Cost model printed next to each row: bf16-MFMA-equivalents per product on gfx950 (MI355X_MICROARCH.md: i8 and block-scaled fp8 run at 2 x the
bf16 rate, fp6 / fp4 at 4 x).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np
import torch

from nerfds_amd import nerf_ds_config, init_params
from oracle import nerfds_oracle as O

EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
f64 = torch.float64


def q(x, dt):
  return x.to(dt).to(f64)


def split(x, dt):
  h = q(x, dt)
  return h, q(x - h, dt)


def mx8(x, axis):
  """OCP MX fp8 (e4m3, 448 max) with one power-of-two scale per 32-block along `axis` (the contraction axis)."""
  x = x.movedim(axis, -1)
  sh = x.shape
  pad = (-sh[-1]) % 32
  xp = torch.nn.functional.pad(x, (0, pad)).reshape(*sh[:-1], -1, 32)
  m = xp.abs().amax(-1, keepdim=True)
  e = torch.floor(torch.log2(m.clamp_min(1e-300))) - 8           # block max lands in [256, 512) -> just inside e4m3's range after the clamp
  s = torch.pow(2.0, e)
  y = (xp / s).clamp(-448, 448).to(torch.float8_e4m3fn).to(f64) * s
  return y.reshape(*sh[:-1], -1)[..., :sh[-1]].movedim(-1, axis)


def i8_slices(x, axis, bits=15, unsigned=False):
  """Symmetric fixed point with ONE scale per vector along the contraction axis (an integer MFMA cannot rescale inside the sum), cut into a
  high and a low signed 8-bit slice: x ~ s * (128 * hi + lo).  unsigned: the vector is known to be >= 0 (ReLU output): centre it first, one more
  bit (the centring constant times the weight column sums goes into the bias: free)."""
  m = x.abs().amax(axis, keepdim=True).clamp_min(1e-300)
  off = 0.5 * m if unsigned else torch.zeros_like(m)
  top = float(2 ** (bits - 1) - 1 - 64)                          # hi in [-127, 127], lo in [-64, 63]
  s = (m - off) / top
  qv = torch.round((x - off) / s)
  hi = torch.round(qv / 128.0)
  lo = qv - 128.0 * hi
  return s, hi, lo, off


MODE = 'exact'
PER_LAYER = {}          # id(kernel tensor) -> mode override
NAMES = {}


def emul(x, W, mode):
  if mode == 'exact':
    return x @ W
  if mode in ('f16', 'bf16'):
    dt = torch.float16 if mode == 'f16' else torch.bfloat16
    return q(x, dt) @ q(W, dt)
  if mode in ('bf16x3', 'f16x3'):
    dt = torch.float16 if mode == 'f16x3' else torch.bfloat16
    xh, xl = split(x, dt)
    wh, wl = split(W, dt)
    return xh @ wh + xl @ wh + xh @ wl
  if mode == 'f16w2':                                           # weights split exactly (22 bits), activations one f16: 2 MFMAs
    wh, wl = split(W, torch.float16)
    xh = q(x, torch.float16)
    return xh @ wh + xh @ wl
  if mode in ('f16mx8', 'bf16mx8'):                             # hi * hi in 16 bits + both cross terms on the block-scaled fp8 MFMA: 1 + 2 * 0.5
    dt = torch.float16 if mode == 'f16mx8' else torch.bfloat16
    xh, wh = q(x, dt), q(W, dt)
    return xh @ wh + mx8(x - xh, -1) @ mx8(W, 0) + mx8(x, -1) @ mx8(W - wh, 0)
  if mode in ('i8x2', 'i8x2u'):                                 # hh + hl + lh on the i8 MFMA: 3 * 0.5
    sx, xh, xl, xo = i8_slices(x, -1, unsigned=(mode == 'i8x2u') and bool((x >= 0).all()))
    sw, wh, wl, _ = i8_slices(W, 0)
    acc = (128.0 * 128.0) * (xh @ wh) + 128.0 * (xh @ wl + xl @ wh)      # the lo * lo term is dropped
    return acc * sx * sw + xo * W.sum(0, keepdim=True)
  raise ValueError(mode)


def dense(p, x):
  W, b = p['kernel'], p['bias']
  x = x.to(torch.float32).to(f64)                               # layer inputs are fp32 values on the GPU
  mode = PER_LAYER.get(id(W), MODE)
  return emul(x, W.to(torch.float32).to(f64), mode) + b


O.dense = dense
COST = {'exact': None, 'bf16': 1.0, 'f16': 1.0, 'bf16x3': 3.0, 'f16x3': 3.0, 'f16w2': 2.0, 'f16mx8': 2.0, 'bf16mx8': 2.0, 'i8x2': 1.5, 'i8x2u': 1.5}


def frame_rays(R, n_ids, seed):
  """bench.py synth_rays (config 2), same seed: one 800 x 600 frame."""
  g = torch.Generator(device='cpu').manual_seed(seed)
  H, W = 600, 800
  idx = torch.arange(R, device='cpu') % (H * W)
  py, px = (idx // W).float() + 0.5, (idx % W).float() + 0.5
  focal = 0.5 * W / np.tan(0.5 * 0.6911)
  d = torch.stack([(px - 0.5 * W) / focal, -(py - 0.5 * H) / focal, -torch.ones(R, device='cpu')], -1)
  d = d / d.norm(dim=-1, keepdim=True)
  ids = torch.full((R, 1), int(torch.randint(0, n_ids, (1,), generator=g, device='cpu')), dtype=torch.int32, device='cpu')
  mask = (torch.rand(R, 1, generator=g, device='cpu') < 0.3).float()
  return dict(origins=np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (R, 1)), directions=d.numpy(), viewdirs=d.numpy(),
              metadata={'warp': ids.numpy()}, mask=mask.numpy())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rays', type=int, default=480000)
  ap.add_argument('--chunk', type=int, default=8192)
  ap.add_argument('--modes', default='bf16x3,f16x3,f16mx8,bf16mx8,i8x2,i8x2u,f16w2,f16')
  ap.add_argument('--layers', action='store_true', help='layer granularity: every trunk layer / network alone in f16 (one MFMA), the rest split bf16')
  ap.add_argument('--levels', action='store_true', help='level granularity: the COARSE NerfMLP alone in one MFMA (f16 / bf16), everything else split bf16 - what the fine level (the one render_fn returns) sees of a cheap coarse pass')
  ap.add_argument('--out', default=None)
  ap.add_argument('--device', default='cuda')
  args = ap.parse_args()
  torch.set_default_device(args.device)
  cfg = nerf_ds_config(near=0.3, far=1.7, num_warp_embeds=256, num_coarse_samples=64, num_fine_samples=64)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  om = O.NerfModel(cfg, params)
  rays = frame_rays(args.rays, cfg.num_warp_embeds, 100)
  rng = np.random.default_rng(5)
  t_all, u_all = rng.random((args.rays, 64), dtype=np.float32), rng.random((args.rays, 64), dtype=np.float32)

  def walk(tree, path=''):
    for k, v in tree.items():
      if isinstance(v, dict):
        yield from walk(v, path + '/' + k)
      elif k == 'kernel':
        yield path, v
  for pth, ker in walk(om.params):
    NAMES[id(ker)] = pth

  def render(mode, per_layer=None):
    global MODE
    MODE = mode
    PER_LAYER.clear()
    PER_LAYER.update(per_layer or {})
    out = {'coarse': [], 'fine': []}
    for lo in range(0, args.rays, args.chunk):
      hi = min(lo + args.chunk, args.rays)
      sl = {k: (v[lo:hi] if not isinstance(v, dict) else {kk: vv[lo:hi] for kk, vv in v.items()}) for k, v in rays.items()}
      with torch.no_grad():
        o = om.apply(sl, EXTRA, t_rand=t_all[lo:hi], u_rand=u_all[lo:hi], use_predicted_norm=True, compute_sigma_gradient=False)
      for lv in out:
        out[lv].append(o[lv]['rgb'].to(torch.float64))
    return {lv: torch.cat(v, 0) for lv, v in out.items()}

  def errors(a, ref):
    res = {}
    for lv in ('coarse', 'fine'):
      d = (a[lv] - ref[lv]).abs()
      res[lv] = dict(max_rel_global=float(d.max() / ref[lv].abs().max()),
                     max_rel_pixel=float((d / ref[lv].abs().clamp_min(1e-2)).max()),
                     first_4096=float(d[:4096].max() / ref[lv][:4096].abs().max()))
    return res

  t0 = time.time()
  ref = render('exact')
  print(f'# exact run: {time.time() - t0:.1f} s for {args.rays} rays', flush=True)
  rows = []
  plans = [(m, None, COST[m]) for m in args.modes.split(',') if m]
  if args.layers:
    groups = {}
    for pth, ker in walk(om.params):
      net = pth.split('/')[1] if pth.startswith('/nerf_mlps') is False else '/'.join(pth.split('/')[1:4])
      groups.setdefault(pth, []).append(ker)
    # every dense layer of the fine / coarse trunk (both levels together), and every small network as a whole, alone in f16
    for l in range(8):
      sel = {id(k): 'f16' for pth, k in walk(om.params) if f'trunk_mlp/hidden_{l}' in pth}
      plans.append((f'x3, trunk hidden_{l} f16', sel, None))
    for name in ('rgb_mlp', 'alpha_mlp', 'bottleneck', 'warp_field', 'hyper_sheet_mlp', 'mask_mlp'):
      sel = {id(k): 'f16' for pth, k in walk(om.params) if name in pth}
      plans.append((f'x3, {name} f16', sel, None))
    sel = {id(k): 'f16' for pth, k in walk(om.params) if 'trunk_mlp/hidden_' in pth and not pth.endswith('hidden_0')}
    plans.append(('x3, trunk hidden_1..7 f16', sel, None))
  if args.levels:
    for m in ('f16', 'bf16', 'f16w2'):
      sel = {id(k): m for pth, k in walk(om.params) if pth.startswith('/nerf_mlps_coarse')}
      plans.append((f'x3, coarse NerfMLP {m}', sel, None))
  for name, sel, cost in plans:
    t0 = time.time()
    got = render('bf16x3' if sel is not None else name, sel)
    e = errors(got, ref)
    row = dict(plan=name, mfma_equiv_per_product=cost, seconds=round(time.time() - t0, 1), **{f'{lv}_{k}': v for lv, d in e.items() for k, v in d.items()})
    rows.append(row)
    print(json.dumps(row), flush=True)
  if args.out:
    json.dump(dict(rays=args.rays, rows=rows), open(args.out, 'w'), indent=1)


if __name__ == '__main__':
  main()
