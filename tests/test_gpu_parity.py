"""GPU parity tests: the fused HIP ray kernel, called through the C ABI (libnerfds_hip.so), against the CPU oracle
on identical rays, weights and injected sampling uniforms.

Tolerance (BASELINE.json north_star): composited RGB within 1e-4 relative of the reference semantics.  The
fp32-MFMA kernel (exact fp32 fma chains) and the split-bf16 kernel are held to that bar.  The one-MFMA-per-product
kernels cannot meet it (profiles/r2_precision_budget.md: >= 16 significand bits are needed on both operands of every
layer; profiles/r4_precision_budget.md: nor can any arithmetic below three MFMA-equivalents per product); their measured error is printed and
bounded at <= 2 x the worst value these tests measure: bf16 3e-2 (8 significand bits: 2.05e-2 measured), f16 2.4e-3 (1.24e-3), mixed (f16 with
the warp field in split bf16) 1.4e-3 (7.3e-4).  Statistics: global = max |d| / max |ref| over rays and channels; for the composited rgb of the
parity-grade modes also per pixel = max |d| / max(|ref|, 1e-2), bounded at 2e-4.
"""
import ctypes as C
import sys

import numpy as np
import pytest
import torch

from nerfds_amd import nerf_ds_config, static_config, init_params
from oracle import nerfds_oracle as O

pytestmark = pytest.mark.gpu

EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
# parity grade: north_star's 1e-4.  Throughput modes: <= 2 x the worst value these tests measure (round 4: bf16 2.05e-2, f16 1.24e-3, mixed 7.3e-4)
RTOL = {'f32': 1e-4, 'bf16x3': 1e-4, 'f16x3': 1e-4, 'bf16': 3e-2, 'f16': 2.4e-3, 'mixed': 1.4e-3}
PIXEL_FLOOR = 1e-2
FAST = ('bf16', 'f16', 'mixed')      # throughput arithmetic: composited maps only, bounded by RTOL


def _rays(R, n_ids, seed, spread=0.1):
  rng = np.random.default_rng(seed)
  d = rng.normal(size=(R, 3))
  d /= np.linalg.norm(d, axis=-1, keepdims=True)
  return dict(origins=rng.normal(size=(R, 3)) * spread, directions=d, viewdirs=d,
              metadata={'warp': rng.integers(0, n_ids, (R, 1))},
              mask=(rng.random((R, 1)) < 0.3).astype(np.float32)), rng


def _relerr(a, b):
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-6))


def _pixerr(a, b):
  """per-pixel relative error: max |a - b| / max(|b|, 1e-2) - a dark pixel is not excused by the brightest one (bench.py reports the same)"""
  return float((np.abs(a - b) / np.maximum(np.abs(b), PIXEL_FLOOR)).max())


def _model(cfg):
  from nerfds_amd.model import NerfModel
  return NerfModel(cfg, device=torch.device('cuda', 0))


def test_extension_is_loaded_and_mfma_layout():
  """The accumulator / operand maps the kernel relies on, checked with an asymmetric product."""
  from nerfds_amd import _native as N
  lib = N.load()
  rng = np.random.default_rng(0)
  a = rng.integers(-4, 5, (32, 16)).astype(np.float32)      # exactly representable in bf16
  b = rng.integers(-4, 5, (16, 32)).astype(np.float32)
  cb, cf = np.zeros((32, 32), np.float32), np.zeros((32, 32), np.float32)
  rc = lib.nerfds_debug_mfma(0, a.ctypes.data, b.ctypes.data, cb.ctypes.data, cf.ctypes.data)
  assert rc == 0
  assert np.array_equal(cb, a @ b) and np.array_equal(cf, a @ b)
  assert any('libnerfds_hip' in l for l in open('/proc/self/maps'))


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'bf16', 'f16', 'mixed', 'f16x3'])
def test_nerf_ds_graph_tiny(prec):
  cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=8, num_fine_samples=8)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 12
  rays, rng = _rays(R, 4, 1)
  t, u = rng.random((R, 8)), rng.random((R, 8))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, sharp_weights_std=0.1,
                                       return_weights=True, return_points=True, compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True,
                          sharp_weights_std=0.1, return_samples=True, precision=prec)
  tol = RTOL[prec]
  for level in ('coarse', 'fine'):
    r, g = ref[level], {k: v.cpu().numpy() for k, v in out[level].items()}
    if prec in FAST:         # throughput arithmetic: only the composited maps (measured error is printed)
      for k in ('rgb', 'depth', 'acc', 'ray_delta_x', 'ray_predicted_mask'):
        e = _relerr(g[k], r[k].numpy())
        print(f'{prec} {level} {k}: {e:.2e}', file=sys.stderr)
        assert e <= (tol if k == 'rgb' else 4 * tol), (level, k, e)
      continue
    assert np.allclose(g['z_vals'], r['z_vals'].numpy(), rtol=2e-6, atol=1e-6), level
    for k in ('sigma', 'predicted_mask', 'warped_points', 'predicted_norm', 'sample_rgb', 'weights', 'alpha',
              'accum_prod', 'back_facing', 'delta_x', 'sharp_weights'):
      e = _relerr(g[k], r[k].numpy())
      print(f'{prec} {level} per-sample {k}: {e:.2e}', file=sys.stderr)
      assert e < 20 * tol, (level, k, e)
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'ray_norm', 'ray_rotation_field', 'ray_translation_field',
              'ray_delta_x', 'ray_hyper_points', 'ray_predicted_mask', 'med_points', 'ray_hyper_c'):
      e = _relerr(g[k], r[k].numpy())
      print(f'{prec} {level} {k}: {e:.2e}', file=sys.stderr)
      lim = tol if k == 'rgb' else 10 * tol
      assert e <= lim, (level, k, e)
      if k == 'rgb':                 # the per-pixel form of the same bar
        ep = _pixerr(g[k], r[k].numpy())
        print(f'{prec} {level} rgb per-pixel: {ep:.2e}', file=sys.stderr)
        assert ep <= 2 * tol, (level, 'rgb per pixel', ep)


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'bf16', 'f16', 'mixed', 'f16x3'])
def test_nerf_ds_graph_full_samples_init_regime(prec):
  """64 + 64 samples (nerf_ds.gin), freshly initialised weights: theta ~ 1e-4 stresses exp_se3 (quirk 5)."""
  cfg = nerf_ds_config(num_warp_embeds=8)
  params = init_params(cfg, 1)
  R = 70                         # not a multiple of anything
  rays, rng = _rays(R, 8, 2, spread=0.3)
  t, u = rng.random((R, 64)), rng.random((R, 64))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, precision=prec)
  for level in ('coarse', 'fine'):
    e = _relerr(out[level]['rgb'].cpu().numpy(), ref[level]['rgb'].numpy())
    print(f'{prec} {level} rgb: {e:.2e}', file=sys.stderr)
    assert e <= RTOL[prec], (level, e)
    assert torch.isfinite(out[level]['ray_delta_x']).all()


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'bf16', 'f16', 'mixed', 'f16x3'])
def test_nerf_ds_graph_256_samples_per_ray(prec):
  """BASELINE.json configs[4] shape: 128 coarse + 128 fine (256 on the fine pass) - the WIDE kernel shape
  (2 rays per workgroup, twice the waves per ray)."""
  cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=128, num_fine_samples=128)
  params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 7
  rays, rng = _rays(R, 4, 12, spread=0.2)
  t, u = rng.random((R, 128)), rng.random((R, 128))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, return_samples=True,
                          precision=prec)
  assert out['fine']['z_vals'].shape == (R, 256) and torch.all(torch.diff(out['fine']['z_vals'], dim=-1) >= 0)
  for level in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'ray_delta_x'):
      e = _relerr(out[level][k].cpu().numpy(), ref[level][k].numpy())
      print(f'{prec} {level} {k}: {e:.2e}', file=sys.stderr)
      assert e <= (RTOL[prec] if k == 'rgb' else 10 * RTOL[prec]), (level, k, e)


def test_nerf_ds_trained_regime_and_deterministic_sampling():
  cfg = nerf_ds_config(num_warp_embeds=3, use_stratified_sampling=False)
  params = init_params(cfg, 2, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.2)
  R = 33
  rays, _ = _rays(R, 3, 3)
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, use_predicted_norm=True, compute_sigma_gradient=False)
  m = _model(cfg)
  for prec in ('f32', 'bf16x3'):
    out = m.apply({'params': params}, rays, EXTRA, use_predicted_norm=True, precision=prec)
    for level in ('coarse', 'fine'):
      for k in ('rgb', 'depth', 'acc', 'ray_predicted_mask', 'ray_delta_x'):
        e = _relerr(out[level][k].cpu().numpy(), ref[level][k].numpy())
        assert e <= (1e-4 if k == 'rgb' else 1e-3), (prec, level, k, e)


@pytest.mark.parametrize('graph', ['nerf_ds', 'hypernerf'])
def test_fine_level_parity_mode_on_both_two_level_graphs(graph):
  """precision='bf16x3_fine': the coarse level's NerfMLP in one f16 MFMA per product, everything else split bf16.  Fine level (what render_fn
  returns): every per-ray output at the parity bounds of the split-bf16 kernel; coarse level: the level-independent networks' outputs (split bf16
  here too: the fine level reuses them) at parity bounds, the NerfMLP's (rgb, depth, acc) at the f16 kernel's."""
  from nerfds_amd import hypernerf_config
  mk = nerf_ds_config if graph == 'nerf_ds' else hypernerf_config
  cfg = mk(num_warp_embeds=4, num_coarse_samples=16, num_fine_samples=16)
  params = init_params(cfg, 2, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 40
  rays, rng = _rays(R, 4, 3)
  t, u = rng.random((R, 16)), rng.random((R, 16))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=cfg.predict_norm, compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=cfg.predict_norm, precision='bf16x3_fine')
  for k in ('rgb', 'depth', 'acc', 'ray_delta_x', 'ray_hyper_points', 'med_depth'):
    e = _relerr(out['fine'][k].cpu().numpy(), ref['fine'][k].numpy())
    print(f'bf16x3_fine {graph} fine {k}: {e:.2e}', file=sys.stderr)
    assert e <= (1e-4 if k == 'rgb' else 1e-3), ('fine', k, e)
  assert _pixerr(out['fine']['rgb'].cpu().numpy(), ref['fine']['rgb'].numpy()) <= 2e-4
  for k in ('ray_delta_x', 'ray_hyper_points'):                  # warp field / hyper sheet at the coarse positions: split bf16 (weighted with f16-grade weights)
    assert _relerr(out['coarse'][k].cpu().numpy(), ref['coarse'][k].numpy()) <= 4 * RTOL['f16'], k
  for k in ('rgb', 'depth', 'acc'):
    e = _relerr(out['coarse'][k].cpu().numpy(), ref['coarse'][k].numpy())
    print(f'bf16x3_fine {graph} coarse {k}: {e:.2e}', file=sys.stderr)
    assert e <= (RTOL['f16'] if k == 'rgb' else 4 * RTOL['f16']), ('coarse', k, e)


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'f16', 'bf16', 'mixed', 'f16x3'])
def test_windows_partially_open(prec):
  """warp_alpha / nerf_alpha mid-schedule: fractional Hann windows on the top bands (model_utils.py:420-436)."""
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=16, num_fine_samples=16)
  params = init_params(cfg, 4, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  extra = dict(nerf_alpha=5.3, warp_alpha=2.6, hyper_alpha=0.4, hyper_sheet_alpha=3.7, norm_input_alpha=1.5)
  R = 9
  rays, rng = _rays(R, 2, 5)
  t, u = rng.random((R, 16)), rng.random((R, 16))
  ref = O.NerfModel(cfg, params).apply(rays, extra, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, extra, t_rand=t, u_rand=u, use_predicted_norm=True, precision=prec)
  for level in ('coarse', 'fine'):
    e = _relerr(out[level]['rgb'].cpu().numpy(), ref[level]['rgb'].numpy())
    print(f'windows {prec} {level} rgb: {e:.2e}', file=sys.stderr)
    assert e <= RTOL[prec], (level, e)


@pytest.mark.parametrize('graph', ['nerf_ds', 'static', 'hypernerf'])
@pytest.mark.parametrize('white,infinity', [(True, True), (False, False), (True, False)])
def test_white_background_and_no_sample_at_infinity(graph, white, infinity):
  """use_white_background (model_utils.py:144-145; BASELINE config 1 is a Lego-style, i.e. white-background, scene) and
  use_sample_at_infinity=False (last delta 1e-19 instead of 1e10, model_utils.py:124-126; acc then includes the last
  sample, :147-148) on all three compiled graphs, fp32-MFMA and split-bf16 kernels against the oracle."""
  from nerfds_amd import hypernerf_config
  kw = dict(use_white_background=white, use_sample_at_infinity=infinity)
  if graph == 'static':
    cfg = static_config(**kw)
  elif graph == 'hypernerf':
    cfg = hypernerf_config(num_warp_embeds=3, num_coarse_samples=16, num_fine_samples=16, **kw)
  else:
    cfg = nerf_ds_config(num_warp_embeds=3, num_coarse_samples=16, num_fine_samples=16, **kw)
  params = init_params(cfg, 7, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 21
  rays, rng = _rays(R, 3, 31, spread=0.2)
  nc, nf = cfg.num_coarse_samples, cfg.num_fine_samples
  t, u = rng.random((R, nc)), rng.random((R, max(nf, 1)))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u if nf else None, use_predicted_norm=cfg.predict_norm,
                                       compute_sigma_gradient=False)
  m = _model(cfg)
  for prec in ('f32', 'bf16x3'):
    out = m.apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u if nf else None, use_predicted_norm=cfg.predict_norm,
                  precision=prec)
    for level in ref:
      for k in ('rgb', 'depth', 'med_depth', 'acc'):
        e = _relerr(out[level][k].cpu().numpy(), ref[level][k].numpy())
        assert e <= (1e-4 if k == 'rgb' else 1e-3), (prec, level, k, e)
    if white:       # a ray that hits nothing is white, not black
      assert float(out[sorted(ref)[-1]]['rgb'].max()) <= 1.0 + 1e-5


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'f16x3'])
def test_mask_ratio_blends_gt_mask(prec):
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=8, num_fine_samples=8)
  params = init_params(cfg, 6, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 6
  rays, rng = _rays(R, 2, 7)
  rays['mask'] = np.ones((R, 1), np.float32)
  t, u = rng.random((R, 8)), rng.random((R, 8))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, mask_ratio=0.25,
                                       compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True,
                          mask_ratio=0.25, precision=prec)
  for k in ('rgb', 'ray_delta_x', 'ray_hyper_points'):
    assert _relerr(out['fine'][k].cpu().numpy(), ref['fine'][k].numpy()) <= 1e-4, k


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'bf16', 'f16', 'mixed', 'f16x3'])
def test_static_graph_config1(prec):
  """BASELINE.json configs[0]: 64 samples/ray, coarse only, warp disabled."""
  cfg = static_config()
  params = init_params(cfg, 0, bias_scale=0.05)
  R = 50
  rays, rng = _rays(R, 1, 8, spread=1.0)
  rays['origins'] = rays['origins'] * 0 + np.array([0.0, 0.0, -4.0]) + rng.normal(size=(R, 3)) * 0.01
  t = rng.random((R, 64))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, compute_sigma_gradient=False)['coarse']
  out = _model(cfg).apply({'params': params}, rays, EXTRA, t_rand=t, precision=prec)
  assert set(out) == {'coarse'}
  for k in ('rgb', 'depth', 'acc'):
    e = _relerr(out['coarse'][k].cpu().numpy(), ref[k].numpy())
    assert e <= (RTOL[prec] if k == 'rgb' else 10 * RTOL[prec]), (k, e)
  assert 'ray_rotation_field' not in out['coarse'] and out['coarse']['ray_hyper_points'].shape == (R, 0)


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['bf16x3', 'bf16x3_fine'])
def test_static_graph_two_levels(prec):
  """The static graph with a fine level (coarse + fine, 16 + 16).  'bf16x3_fine' has no kernel of its own on this graph: the call runs plain split
  bf16 - streams packed, sized AND launched at that one precision (round 5 packed the coarse level as f16 under a bf16x3 kernel) - so both levels
  hold north_star's 1e-4 and the two precisions agree bit for bit."""
  cfg = static_config(num_coarse_samples=16, num_fine_samples=16)
  params = init_params(cfg, 0, bias_scale=0.05)
  R = 40
  rays, rng = _rays(R, 1, 8, spread=1.0)
  rays['origins'] = rays['origins'] * 0 + np.array([0.0, 0.0, -4.0]) + rng.normal(size=(R, 3)) * 0.01
  t, u = rng.random((R, 16)), rng.random((R, 16))
  ref = O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u, compute_sigma_gradient=False)
  m = _model(cfg)
  out = m.apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, precision=prec)
  assert set(out) == {'coarse', 'fine'}
  for lv in ('coarse', 'fine'):
    e = _relerr(out[lv]['rgb'].cpu().numpy(), ref[lv]['rgb'].numpy())
    assert e <= RTOL['bf16x3'], (lv, e)
  base = m.apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, precision='bf16x3')
  for lv in ('coarse', 'fine'):
    assert np.array_equal(out[lv]['rgb'].cpu().numpy(), base[lv]['rgb'].cpu().numpy()), lv


@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'bf16', 'f16', 'mixed', 'f16x3'])
@pytest.mark.parametrize('Nc,Nf', [(16, 16), (128, 128)])
def test_hypernerf_base_gin_graph(prec, Nc, Nf):
  """configs/base.gin graph (BASELINE config 5 per SURVEY 8d): posenc identity, SE3 warp (6 bands), hyper sheet, no mask / normal;
  at its own 128 + 128 samples (wide kernel shape) and at a small count."""
  from nerfds_amd import hypernerf_config
  cfg = hypernerf_config(num_warp_embeds=5, num_coarse_samples=Nc, num_fine_samples=Nf)
  params = init_params(cfg, 2, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 9 if Nc > 64 else 37
  rays, rng = _rays(R, 5, 21, spread=0.2)
  t, u = rng.random((R, Nc)), rng.random((R, Nf))
  extra = dict(EXTRA, warp_alpha=6.0)
  ref = O.NerfModel(cfg, params).apply(rays, extra, t_rand=t, u_rand=u, return_weights=True, return_points=True, compute_sigma_gradient=False)
  out = _model(cfg).apply({'params': params}, rays, extra, t_rand=t, u_rand=u, precision=prec, return_samples=True)
  for level in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'ray_delta_x', 'ray_hyper_points', 'ray_rotation_field'):
      e = _relerr(out[level][k].cpu().numpy().reshape(ref[level][k].shape), ref[level][k].numpy())
      assert e <= (RTOL[prec] if k == 'rgb' else 10 * RTOL[prec]), (level, k, e)
    assert 'ray_predicted_mask' not in out[level] or float(out[level]['ray_predicted_mask'].abs().max()) == 0.0


def test_edge_cases_empty_single_and_leading_shape():
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=8, num_fine_samples=8)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  m = _model(cfg)
  rays, _ = _rays(0, 2, 0)
  out = m.apply({'params': params}, rays, EXTRA, use_predicted_norm=True)
  assert out['fine']['rgb'].shape == (0, 3)
  rays, rng = _rays(6, 2, 1)
  a = m.apply({'params': params}, rays, EXTRA, use_predicted_norm=True, t_rand=np.full((6, 8), .5), u_rand=np.full((6, 8), .5),
              precision='f32')['fine']['rgb'].cpu().numpy()
  hw = {k: (v.reshape(2, 3, -1) if k != 'metadata' else {'warp': v['warp'].reshape(2, 3, 1)}) for k, v in rays.items()}
  b = m.apply({'params': params}, hw, EXTRA, use_predicted_norm=True, t_rand=np.full((6, 8), .5), u_rand=np.full((6, 8), .5),
              precision='f32')['fine']['rgb'].cpu().numpy()
  assert b.shape == (2, 3, 3) and np.array_equal(a.reshape(2, 3, 3), b)          # deterministic, shape preserved
  big, last = dict(rays), dict(rays)                 # out-of-range GLO ids clamp to the last row, like a jnp gather
  big['metadata'], last['metadata'] = {'warp': np.full((6, 1), 7)}, {'warp': np.full((6, 1), 1)}
  kw = dict(use_predicted_norm=True, t_rand=np.full((6, 8), .5), u_rand=np.full((6, 8), .5), precision='f32')
  assert torch.equal(m.apply({'params': params}, big, EXTRA, **kw)['fine']['rgb'],
                     m.apply({'params': params}, last, EXTRA, **kw)['fine']['rgb'])
  with pytest.raises(ValueError):
    m.apply({'params': params}, rays, EXTRA, use_predicted_norm=False)


def test_philox_sampling_is_seeded_and_stratified():
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=16, num_fine_samples=16)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  m = _model(cfg)
  rays, _ = _rays(5, 2, 3)
  kw = dict(use_predicted_norm=True, return_samples=True, precision='f32')
  a = m.apply({'params': params}, rays, EXTRA, rngs={'coarse': 1, 'fine': 2}, **kw)
  b = m.apply({'params': params}, rays, EXTRA, rngs={'coarse': 1, 'fine': 2}, **kw)
  c = m.apply({'params': params}, rays, EXTRA, rngs={'coarse': 5, 'fine': 2}, **kw)
  za, zb, zc = (x['coarse']['z_vals'].cpu().numpy() for x in (a, b, c))
  assert np.array_equal(za, zb) and not np.array_equal(za, zc)
  edges = np.linspace(cfg.near, cfg.far, 16)
  mids = 0.5 * (edges[1:] + edges[:-1])
  lo, hi = np.r_[edges[0], mids], np.r_[mids, edges[-1]]
  assert np.all(za >= lo - 1e-6) and np.all(za <= hi + 1e-6)                     # one sample per stratum
  zf = a['fine']['z_vals'].cpu().numpy()
  assert zf.shape == (5, 32) and np.all(np.diff(zf, axis=-1) >= 0)              # sorted union


def test_frame_properties_at_config2_size():
  """Size-independent properties at the metric's shape (64+64 samples, thousands of rays, bf16 kernel)."""
  cfg = nerf_ds_config(num_warp_embeds=16)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 4096
  rays, _ = _rays(R, 16, 11, spread=0.2)
  m = _model(cfg)
  out = m.apply({'params': params}, rays, EXTRA, rngs={'coarse': 3, 'fine': 4}, use_predicted_norm=True, return_samples=True)
  f = out['fine']
  w = f['weights']
  assert torch.isfinite(m.last_records['fine']).all()
  assert torch.all(w >= 0) and torch.allclose(w.sum(-1), torch.ones(R, device=w.device), atol=1e-4)   # 1e10 tail closes the ray
  assert torch.allclose(f['acc'], w[:, :-1].sum(-1), atol=1e-5)
  assert torch.all((f['rgb'] >= 0) & (f['rgb'] <= 1 + 1e-5))
  assert torch.all(torch.diff(f['z_vals'], dim=-1) >= 0)
  assert torch.allclose(f['rgb'], (w[..., None] * f['sample_rgb']).sum(1), atol=1e-5)
  # permutation equivariance over rays (rays are independent units)
  perm = torch.randperm(R, generator=torch.Generator().manual_seed(0)).numpy()
  t, u = np.random.default_rng(0).random((R, 64)), np.random.default_rng(1).random((R, 64))
  a = m.apply({'params': params}, rays, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True)['fine']['rgb']
  rp = {k: (v[perm] if k != 'metadata' else {'warp': v['warp'][perm]}) for k, v in rays.items()}
  b = m.apply({'params': params}, rp, EXTRA, t_rand=t[perm], u_rand=u[perm], use_predicted_norm=True)['fine']['rgb']
  assert torch.equal(a[perm], b)


def test_render_frame_of_a_coarse_only_model():
  """frames.render_frame on the static coarse-only graph (NerfModel.last_records is keyed by the real level names)."""
  from nerfds_amd.camera import Camera
  from nerfds_amd.frames import render_frame
  import os
  cam = Camera.from_json(os.path.join(os.path.dirname(__file__), 'golden', 'reference_testdata_camera.json')).scale(0.05)
  cfg = static_config(near=0.5, far=3.0)
  params = init_params(cfg, 0, bias_scale=0.05)
  m = _model(cfg)
  rgb, dbg, rec = render_frame(m, {'params': params}, cam, 0, EXTRA, want_debug=False, precision='f32')
  H, W = cam.image_shape
  assert rgb.shape == (H, W, 3) and rec.shape == (H * W, 26) and torch.isfinite(rec).all()
  assert set(m.last_records) == {'coarse'}


def test_params_are_repacked_for_a_new_tree_and_tables_are_shape_checked():
  """apply() re-packs when it is handed another parameter tree object (temporaries included: the packed tree is kept alive,
  so a recycled address cannot alias it), reload_params=True covers in-place edits, and GLO tables / biases whose shapes
  disagree with the configuration are rejected before the library reads them."""
  import copy
  cfg = nerf_ds_config(num_warp_embeds=3, num_coarse_samples=8, num_fine_samples=8)
  p0 = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  m = _model(cfg)
  rays, rng = _rays(5, 3, 2)
  kw = dict(use_predicted_norm=True, t_rand=np.full((5, 8), .5), u_rand=np.full((5, 8), .5), precision='f32')
  a = m.apply({'params': p0}, rays, EXTRA, **kw)['fine']['rgb'].clone()
  outs = []
  for seed in (1, 2, 3):       # temporaries: each tree is freed after the call, CPython readily reuses the address
    outs.append(m.apply({'params': init_params(cfg, seed, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)}, rays, EXTRA, **kw)['fine']['rgb'].clone())
  assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2]) and not torch.equal(a, outs[0])
  p1 = copy.deepcopy(p0)
  b = m.apply({'params': p1}, rays, EXTRA, **kw)['fine']['rgb'].clone()
  assert torch.equal(a, b)
  p1['nerf_mlps_fine']['rgb_mlp']['logit']['bias'] = np.asarray(p1['nerf_mlps_fine']['rgb_mlp']['logit']['bias']) + 1.0
  c = m.apply({'params': p1}, rays, EXTRA, reload_params=True, **kw)['fine']['rgb']
  assert not torch.equal(b, c)
  bad = copy.deepcopy(p0)
  bad['warp_embed']['embed']['embedding'] = np.zeros((2, cfg.glo_num_dims), np.float32)       # fewer rows than num_warp_embeds
  with pytest.raises(ValueError, match='num_warp_embeds'):
    m.apply({'params': bad}, rays, EXTRA, **kw)
  bad = copy.deepcopy(p0)
  bad['mask_mlp']['MLP_0']['hidden_0']['bias'] = np.zeros((7,), np.float32)
  with pytest.raises(ValueError, match='bias'):
    m.apply({'params': bad}, rays, EXTRA, **kw)


@pytest.mark.parametrize('prec', ['bf16', 'f16', 'mixed', 'bf16x3', 'f32', 'f16x3'])
def test_kernels_are_deterministic_and_ray_order_independent(prec):
  """4096 rays at 64 + 64 samples, rendered twice and once with the rays permuted: bit-identical per ray in every arithmetic
  mode.  Guards the hand-placed synchronisation of the kernel (LDS-DMA ring protocol, asm epilogues whose MFMA -> VALU hazard
  hipcc does not pad): both kinds of mistake showed up as 1-2 % of the rays differing from run to run, far inside the
  error bounds of the oracle comparisons."""
  cfg = nerf_ds_config(num_warp_embeds=16)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 4096
  rays, _ = _rays(R, 16, 11, spread=0.2)
  m = _model(cfg)
  t, u = np.random.default_rng(0).random((R, 64)), np.random.default_rng(1).random((R, 64))
  kw = dict(t_rand=t, u_rand=u, use_predicted_norm=True, precision=prec)
  a = m.apply({'params': params}, rays, EXTRA, **kw)
  a = {lv: a[lv]['rgb'].clone() for lv in a}
  for _ in range(2):
    b = m.apply({'params': params}, rays, EXTRA, **kw)
    for lv in a:
      assert torch.equal(a[lv], b[lv]['rgb']), (prec, lv, int((a[lv] != b[lv]['rgb']).any(-1).sum()))
  perm = torch.randperm(R, generator=torch.Generator().manual_seed(0)).numpy()
  rp = {k: (v[perm] if k != 'metadata' else {'warp': v['warp'][perm]}) for k, v in rays.items()}
  c = m.apply({'params': params}, rp, EXTRA, t_rand=t[perm], u_rand=u[perm], use_predicted_norm=True, precision=prec)
  for lv in a:
    assert torch.equal(a[lv][perm], c[lv]['rgb']), (prec, lv)


# ---- rows B and Q of SURVEY 8a: pre-encoded / interpolated GLO metadata, render_opts ---------------------------------------------------------
def _tiny_case(seed=7, R=24, ns=16):
  cfg = nerf_ds_config(num_warp_embeds=6, num_coarse_samples=ns, num_fine_samples=ns)
  params = init_params(cfg, seed, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  rng = np.random.default_rng(seed)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  rays = dict(origins=rng.normal(size=(R, 3)) * 0.1, directions=d, viewdirs=d,
              metadata={'warp': rng.integers(0, 6, (R, 1))}, mask=np.zeros((R, 1)))
  return cfg, params, rng, rays, rng.random((R, ns)), rng.random((R, ns))


@pytest.mark.gpu
def test_encode_metadata_matches_the_oracle():
  """evaluation.encode_metadata / NerfModel._encode_embed (evaluation.py:29-50, models.py:271-294): one id channel and the 3-channel
  (left, right, progression) interpolation, out-of-range ids clamped like a jnp gather - bit for bit in fp32."""
  from nerfds_amd.evaluation import encode_metadata
  cfg, params, rng, rays, _, _ = _tiny_case()
  model = _model(cfg)
  om = O.NerfModel(cfg, params, dtype=torch.float32)
  ids1 = rng.integers(0, 9, (5, 7, 1))                      # some ids >= 6: clamped
  meta3 = np.concatenate([rng.integers(0, 6, (40, 1)), rng.integers(0, 8, (40, 1)), rng.random((40, 1))], -1).astype(np.float32)
  meta3[:3, 2] = [0.0, 1.0, 0.5]
  for meta in (ids1, meta3):
    got = encode_metadata(model, {'params': params}, {'warp': meta})
    want = O.encode_metadata(cfg, om.params, {'warp': meta})
    assert set(got) == {'encoded_warp', 'encoded_hyper'}
    for k in got:
      assert got[k].shape == tuple(meta.shape[:-1]) + (8,)
      np.testing.assert_array_equal(got[k].cpu().numpy(), want[k].numpy())
  m = model.encode_embed(ids1, 'mask').cpu().numpy()
  np.testing.assert_array_equal(m, O.encode_embed(torch.as_tensor(ids1), om.params['mask_embed']).numpy())


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'f16x3'])
def test_metadata_encoded_interpolated_embeddings(prec):
  """model.apply(metadata_encoded=True) on vectors interpolated between two GLO rows (the 3-channel metadata of models.py:271-294)
  against the oracle run the same way; with integer progression it reproduces the id path exactly."""
  from nerfds_amd.evaluation import encode_metadata
  cfg, params, rng, rays, t, u = _tiny_case(seed=8)
  R = t.shape[0]
  model = _model(cfg)
  meta3 = np.concatenate([rng.integers(0, 6, (R, 1)), rng.integers(0, 6, (R, 1)), rng.random((R, 1))], -1).astype(np.float32)
  enc = encode_metadata(model, {'params': params}, {'warp': meta3})
  enc_mask = model.encode_embed(meta3, 'mask')
  kw = dict(t_rand=t, u_rand=u, use_predicted_norm=True)
  rays_e = dict(rays, metadata=dict(enc, encoded_mask=enc_mask))
  got = model.apply({'params': params}, rays_e, EXTRA, metadata_encoded=True, precision=prec, **kw)
  om = O.NerfModel(cfg, params)
  oenc = O.encode_metadata(cfg, om.params, {'warp': meta3})
  oenc['encoded_mask'] = O.encode_embed(torch.as_tensor(meta3), om.params['mask_embed'])
  ref = O.to_numpy(om.apply(dict(rays, metadata=oenc), EXTRA, metadata_encoded=True, compute_sigma_gradient=False, **kw))
  for level in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'acc', 'ray_predicted_mask', 'ray_delta_x'):
      assert _relerr(got[level][k].cpu().numpy(), ref[level][k]) < 1e-4, (level, k)
  # the reference's own form: encoded warp vectors, the mask embedding still looked up from the ids (models.py:924-926)
  ids = rays['metadata']['warp']
  enc1 = encode_metadata(model, {'params': params}, {'warp': ids})
  a = model.apply({'params': params}, dict(rays, metadata=dict(enc1, warp=ids)), EXTRA, metadata_encoded=True, precision=prec, **kw)
  b = model.apply({'params': params}, rays, EXTRA, precision=prec, **kw)
  assert torch.equal(a['fine']['rgb'], b['fine']['rgb']) and torch.equal(a['coarse']['ray_delta_x'], b['coarse']['ray_delta_x'])


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'f16x3'])
def test_render_opts_filter_sigma(prec):
  """render_opts (filter_sigma, models.py:38-66, 1288): dust threshold and bounding box against the oracle; the per-sample 'sigma' stays
  unfiltered (models.py:1271) while alpha / weights see the filter.  FINE level only: NerfModel.__call__ forwards render_opts to the
  'fine' render_samples call (models.py:1545) and not to the 'coarse' one (models.py:1493-1517), so the coarse level - every output of it, and
  the fine depths drawn from its weights - is that of the render without options, bit for bit."""
  cfg, params, rng, rays, t, u = _tiny_case(seed=9)
  cfg = cfg.replace(use_mask_sharp_weights=False)
  model = _model(cfg)
  om = O.NerfModel(cfg, params)
  kw = dict(t_rand=t, u_rand=u, use_predicted_norm=True)
  plain = O.to_numpy(om.apply(rays, EXTRA, compute_sigma_gradient=False, return_weights=True, **kw))
  got_plain = model.apply({'params': params}, rays, EXTRA, precision=prec, return_weights=True, **kw)
  thr = float(np.median(plain['fine']['sigma']))
  for opts in ({'dust_threshold': thr}, {'bounding_box': (-0.3, 0.4, -0.5, 0.5, -0.2, 0.6)},
               {'dust_threshold': thr * 0.5, 'bounding_box': (-0.6, 0.6, -0.6, 0.6, -0.6, 0.6)}):
    ref = O.to_numpy(om.apply(rays, EXTRA, compute_sigma_gradient=False, return_weights=True, render_opts=opts, **kw))
    got = model.apply({'params': params}, rays, EXTRA, precision=prec, render_opts=opts, return_weights=True, **kw)
    assert _relerr(ref['fine']['rgb'], plain['fine']['rgb']) > 1e-3          # the options do something on this case
    for k in ref['coarse']:                                                   # ... and nothing at all to the coarse level
      assert np.array_equal(ref['coarse'][k], plain['coarse'][k]), k
    for k in got['coarse']:
      assert torch.equal(got['coarse'][k], got_plain['coarse'][k]), (opts, k)
    assert torch.equal(got['fine']['z_vals'], got_plain['fine']['z_vals'])    # the pdf of the resample is the unfiltered one
    for level in ('coarse', 'fine'):
      for k in ('rgb', 'depth', 'acc', 'ray_norm'):
        assert _relerr(got[level][k].cpu().numpy(), ref[level][k]) < 1e-4, (opts, level, k)
      assert _relerr(got[level]['sigma'].cpu().numpy(), ref[level]['sigma']) < 2e-4
      assert _relerr(got[level]['weights'].cpu().numpy(), ref[level]['weights']) < 2e-4
  with pytest.raises(ValueError):
    model.apply({'params': params}, rays, EXTRA, precision=prec, render_opts={'nonsense': 1}, **kw)
  # a single-level model never sees the options (only the 'coarse' call exists, models.py:1493-1517)
  cfg1 = static_config()
  p1 = init_params(cfg1, 3, bias_scale=0.1)
  R = t.shape[0]
  t1 = rng.random((R, cfg1.num_coarse_samples))
  m1 = _model(cfg1)
  a = m1.apply({'params': p1}, rays, EXTRA, precision=prec, t_rand=t1, use_predicted_norm=cfg1.predict_norm)
  b = m1.apply({'params': p1}, rays, EXTRA, precision=prec, t_rand=t1, use_predicted_norm=cfg1.predict_norm, render_opts={'dust_threshold': 1e9})
  assert torch.equal(a['coarse']['rgb'], b['coarse']['rgb']) and float(a['coarse']['acc'].max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize('prec', ['f32', 'bf16x3', 'f16x3'])
@pytest.mark.parametrize('cfg_inf,call_inf', [(True, False), (False, True), (True, None)])
def test_use_sample_at_infinity_override_reaches_the_fine_level_only(prec, cfg_inf, call_inf):
  """The use_sample_at_infinity kwarg of NerfModel.__call__ (models.py:1433, 1484-1485) goes to the 'fine' render_samples call (models.py:1544);
  the 'coarse' call keeps self.use_sample_at_infinity (models.py:1509).  Both levels against the oracle, plus sharp_weights, which come from
  cal_weights with its default sample_at_infinity=True whatever the level composites with (models.py:1239-1245, model_utils.py:162)."""
  cfg, params, rng, rays, t, u = _tiny_case(seed=12)
  cfg = cfg.replace(use_sample_at_infinity=cfg_inf)
  kw = dict(t_rand=t, u_rand=u, use_predicted_norm=True, return_weights=True, use_sample_at_infinity=call_inf)
  om = O.NerfModel(cfg, params)
  ref = O.to_numpy(om.apply(rays, EXTRA, compute_sigma_gradient=False, **kw))
  model = _model(cfg)
  got = model.apply({'params': params}, rays, EXTRA, precision=prec, **kw)
  for level in ('coarse', 'fine'):
    for k in ('rgb', 'depth', 'med_depth', 'acc', 'weights', 'sharp_weights'):
      e = _relerr(got[level][k].cpu().numpy(), ref[level][k])
      assert e <= (1e-4 if k == 'rgb' else 1e-3), (level, k, e)
  if call_inf is not None:
    base = model.apply({'params': params}, rays, EXTRA, precision=prec, **dict(kw, use_sample_at_infinity=None))
    for k in got['coarse']:                                                   # the override does not touch the coarse level
      assert torch.equal(got['coarse'][k], base['coarse'][k]), k
    assert not torch.equal(got['fine']['weights'][:, -1], base['fine']['weights'][:, -1])     # ... and does reach the fine one (the last sample's delta is 1e10 or 1e-19)


@pytest.mark.gpu
@pytest.mark.parametrize('stratified', [False, True])
def test_linear_disparity_sampling(stratified):
  """NerfModel.use_linear_disparity (model_utils.py:73-76): coarse depths linear in 1 / z, in the render kernel and in the trainer's forward."""
  cfg, params, rng, rays, t, u = _tiny_case(seed=10)
  cfg = cfg.replace(use_linear_disparity=True, near=0.3, far=1.7, use_stratified_sampling=stratified)
  kw = dict(t_rand=t, u_rand=u, use_predicted_norm=True)
  ref = O.to_numpy(O.NerfModel(cfg, params).apply(rays, EXTRA, compute_sigma_gradient=False, return_weights=True, **kw))
  got = _model(cfg).apply({'params': params}, rays, EXTRA, precision='f32', return_weights=True, **kw)
  lin = O.to_numpy(O.NerfModel(cfg.replace(use_linear_disparity=False), params).apply(rays, EXTRA, compute_sigma_gradient=False, **kw))
  assert _relerr(ref['fine']['rgb'], lin['fine']['rgb']) > 1e-3             # the option changes the picture
  for level in ('coarse', 'fine'):
    assert np.allclose(got[level]['z_vals'].cpu().numpy(), ref[level]['z_vals'], rtol=3e-6, atol=1e-6), level
    assert _relerr(got[level]['rgb'].cpu().numpy(), ref[level]['rgb']) < 1e-4
  from nerfds_amd.training import Trainer
  batch = dict(rays, rgb=rng.random((t.shape[0], 3)))
  tr = Trainer(cfg, params, max_rays=t.shape[0])
  stats = tr.step(batch, EXTRA, 0.0, t_rand=t, u_rand=u, mask_ratio=1.0, grads_only=True)
  want = float(((ref['fine']['rgb'] - batch['rgb']) ** 2).mean())
  assert abs(stats['loss/fine'] - want) < 2e-5 * max(want, 1e-3), (stats, want)
