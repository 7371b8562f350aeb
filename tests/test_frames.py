"""Frame output path: oracle known answers on CPU, byte-exact HIP-vs-oracle on the GPU."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))
sys.path.insert(0, ROOT)
from oracle import frame_oracle as FO      # noqa: E402


def _records(H, W, seed, near, far):
  rng = np.random.default_rng(seed)
  r = rng.normal(size=(H * W, FO.RAY_REC)).astype(np.float32)
  r[:, 0:3] = rng.random((H * W, 3)) * 1.2 - 0.1                      # rgb slightly out of [0, 1]: exercises the clip
  r[:, FO.F_MED_DEPTH] = rng.random(H * W) * (far - near) * 1.3 + near - 0.15 * (far - near)      # some depths outside [near, far]
  r[:7, FO.F_MED_DEPTH] = [near, far, 0.0, near - 1, far + 1, (near + far) / 2, near + (far - near) / 255]
  r[7, FO.F_NORM:FO.F_NORM + 3] = 0.0                                 # zero normal: eps clamp
  r[:, FO.F_MASK] = rng.random(H * W)
  r[:, FO.F_MED_POINTS:FO.F_MED_POINTS + 5] = rng.random((H * W, 5)) * 3.4 - 1.7
  return r


def test_oracle_known_answers():
  gray = FO.get_colormap('gray')
  # gray map, no inversion: colorize(x) = x inside [cmin, cmax], white above, black below (visualization.py:229-233)
  d = np.array([2.0, 6.0, 4.0, 1.0, 7.0], np.float32)
  c = FO.colorize(d, 2.0, 6.0, gray)
  np.testing.assert_allclose(c[:, 0], [0.0, 1.0, 0.5, 0.0, 1.0], atol=1e-6)
  ci = FO.colorize(d, 2.0, 6.0, gray, invert=True)
  np.testing.assert_allclose(ci[:, 0], [1.0, 0.0, 0.5, 1.0, 0.0], atol=1e-6)
  assert ci.dtype == np.float64
  np.testing.assert_array_equal(FO.image_to_uint8(np.array([-0.5, 0.0, 0.999, 1.0, 2.0, 0.5], np.float32)), [0, 0, 254, 255, 255, 127])
  with pytest.raises(ValueError):
    FO.image_to_uint8(np.zeros(3, np.int32))
  sb = FO.get_colormap('sinebow')
  np.testing.assert_allclose(sb[0], [1.0, 0.25, 0.25], atol=1e-12)     # sin^2(pi/2), sin^2(5 pi/6), sin^2(7 pi/6)
  n = FO.normalize_vector(np.array([[3.0, 0.0, 4.0], [0.0, 0.0, 0.0]], np.float32))
  np.testing.assert_allclose(n, [[0.6, 0.0, 0.8], [0.0, 0.0, 0.0]], atol=1e-7)


def test_oracle_mosaic_layout():
  H, W = 3, 4
  r = np.zeros((H * W, FO.RAY_REC), np.float32)
  r[:, 0:3] = [1.0, 0.0, 0.0]
  r[:, FO.F_MED_DEPTH] = 2.0                       # == near -> inverted gray map -> white
  r[:, FO.F_NORM:FO.F_NORM + 3] = [0.0, 0.0, 1.0]  # -> (0.5, 0.5, 1.0)
  r[:, FO.F_MASK] = 0.5
  r[:, FO.F_DELTA_X:FO.F_DELTA_X + 3] = [-0.05, 0.0, 0.2]
  r[:, FO.F_MED_POINTS:FO.F_MED_POINTS + 3] = [1.5, -1.5, 0.0]
  rgb, dbg = FO.frame_images(r, H, W, 2.0, 6.0, FO.get_colormap('gray'))
  assert rgb.shape == (H, W, 3) and dbg.shape == (2 * H, 3 * W, 3) and dbg.dtype == np.uint8
  np.testing.assert_array_equal(rgb[0, 0], [255, 0, 0])
  np.testing.assert_array_equal(dbg[0, 0], [255, 0, 0])               # rgb tile
  np.testing.assert_array_equal(dbg[0, W], [255, 255, 255])           # depth tile
  np.testing.assert_array_equal(dbg[0, 2 * W], [127, 127, 255])       # normal tile
  np.testing.assert_array_equal(dbg[H, 0], [127, 127, 127])           # mask tile
  np.testing.assert_array_equal(dbg[H, W], [127, 0, 255])             # |delta_x| * 10, clipped
  np.testing.assert_array_equal(dbg[H, 2 * W], [255, 0, 127])         # (p + 1.5) / 3


GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'frame_colorize_ref.npz')


def test_oracle_matches_reference_run_colorize():
  """The oracle's colourisation against outputs of the REFERENCE's own visualization.py, run in the build container by
  tests/golden/make_frame_golden.py (real 'magma' / 'turbo' tables, render.py:263's arguments and the function's other modes):
  bit for bit, float64."""
  g = np.load(GOLD)
  import matplotlib
  if matplotlib.__version__ == str(g['matplotlib_version']):       # the tables come out of matplotlib: same version, same bits
    for name in ('magma', 'sinebow'):
      np.testing.assert_array_equal(FO.get_colormap(name), g[f'table_{name}'])
  magma, turbo = g['table_magma'], g['table_turbo']
  got = FO.interpolate_colormap(g['interp_values'], magma)
  np.testing.assert_array_equal(got, g['interp_magma'])
  for ci in g['cases']:
    d = g[f'c{ci}_depth']
    near, far = (float(x) for x in g[f'c{ci}_near_far'])
    assert d.dtype == np.float32
    np.testing.assert_array_equal(FO.scale_values(d, near, far), g[f'c{ci}_scaled'])
    with np.errstate(invalid='ignore'):
      np.testing.assert_array_equal(FO.colorize(d, near, far, magma, invert=True), g[f'c{ci}_magma_inv'])
      np.testing.assert_array_equal(FO.colorize(d, near, far, magma), g[f'c{ci}_magma'])
      np.testing.assert_array_equal(FO.colorize(d, near, far, turbo, invert=True), g[f'c{ci}_turbo_inv'])
      np.testing.assert_array_equal(FO.colorize(d, near, far, magma, invert=True, clip=True), g[f'c{ci}_magma_inv_clip'])
      np.testing.assert_array_equal(FO.colorize(d, colormap=magma), g[f'c{ci}_magma_auto'])
    assert g[f'c{ci}_magma_inv'].dtype == np.float64


@pytest.mark.gpu
def test_hip_depth_tile_matches_reference_run_colorize():
  """nerfds_frame_images on the fixture's depth maps with the real magma table: the depth tile of the debug mosaic equals
  image_to_uint8 of the REFERENCE-run colorize output byte for byte (render.py:263-268)."""
  import torch
  from nerfds_amd.frames import frame_images, get_colormap
  g = np.load(GOLD)
  np.testing.assert_array_equal(get_colormap('magma'), g['table_magma'])      # the product's own table lookup (same matplotlib in the image)
  for ci in g['cases']:
    d = g[f'c{ci}_depth']
    H, W = d.shape
    near, far = (float(x) for x in g[f'c{ci}_near_far'])
    r = _records(H, W, 11 + int(ci), near, far)
    r[:, FO.F_MED_DEPTH] = d.reshape(-1)
    want = FO.image_to_uint8(g[f'c{ci}_magma_inv'])
    for table in ('magma', g['table_magma']):
      _, dbg = frame_images(torch.from_numpy(r).cuda(), H, W, near, far, colormap=table)
      got = dbg.cpu().numpy()[:H, W:2 * W]
      diff = got != want
      assert int(diff.sum()) == 0, f'case {ci}: {int(diff.sum())} differing bytes, first at {np.argwhere(diff)[0]}'


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,cmap', [(5, 7, 'sinebow'), (64, 48, 'gray'), (600, 800, 'random')])
def test_hip_frame_images_are_byte_exact(H, W, cmap):
  import torch
  from nerfds_amd.frames import frame_images
  near, far = 0.3, 1.7
  r = _records(H, W, 3, near, far)
  table = np.random.default_rng(9).random((256, 3)) if cmap == 'random' else FO.get_colormap(cmap)
  want_rgb, want_dbg = FO.frame_images(r, H, W, near, far, table)
  rgb, dbg = frame_images(torch.from_numpy(r).cuda(), H, W, near, far, colormap=table)
  assert int((rgb.cpu().numpy() != want_rgb).sum()) == 0
  diff = dbg.cpu().numpy() != want_dbg
  assert int(diff.sum()) == 0, f'{int(diff.sum())} differing bytes, first at {np.argwhere(diff)[0]}'
  only_rgb, none = frame_images(torch.from_numpy(r).cuda(), H, W, near, far, want_debug=False)
  assert none is None and int((only_rgb.cpu().numpy() != want_rgb).sum()) == 0


@pytest.mark.gpu
def test_frame_images_from_a_render_dict():
  import torch
  from nerfds_amd import _native as N
  from nerfds_amd.frames import frame_images, raw_result
  H, W = 6, 5
  r = _records(H, W, 4, 0.3, 1.7)
  rec = torch.from_numpy(r).cuda()
  render = {k: rec[:, o:o + n].reshape(H, W, n) for k, (o, n) in N.RAY_FIELDS.items()}
  a = frame_images(rec, H, W, 0.3, 1.7, colormap='sinebow')
  b = frame_images(render, H, W, 0.3, 1.7, colormap='sinebow')
  assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
  raw = raw_result(render)
  assert set(raw) == {'rgb', 'med_depth', 'ray_norm', 'ray_delta_x', 'med_points', 'ray_predicted_mask', 'ray_rotation_field'}


@pytest.mark.gpu
def test_render_frame_composes_camera_render_and_frame_output():
  """camera -> rays -> fused render -> uint8 frames in one call equals the three steps done separately."""
  import json
  import torch
  from nerfds_amd import init_params, nerf_ds_config
  from nerfds_amd.camera import Camera, camera_to_rays
  from nerfds_amd.frames import frame_images, render_frame
  from nerfds_amd.model import NerfModel
  cam = Camera.from_json(os.path.join(ROOT, 'tests', 'golden', 'reference_testdata_camera.json')).scale(0.05)
  H, W = cam.image_shape
  cfg = nerf_ds_config(num_warp_embeds=3, num_coarse_samples=8, num_fine_samples=8, use_stratified_sampling=False)
  params = init_params(cfg, 1, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  model = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32')
  EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
  rgb, dbg, rec = render_frame(model, {'params': params}, cam, 2, EX, chunk=97, colormap='sinebow')
  assert rgb.shape == (H, W, 3) and dbg.shape == (2 * H, 3 * W, 3) and rgb.dtype == torch.uint8
  rays = camera_to_rays(cam, torch.device('cuda', 0))
  rd = dict(origins=rays['origins'], directions=rays['directions'], viewdirs=rays['directions'],
            metadata={'warp': torch.full((H, W, 1), 2, dtype=torch.int32)})
  model.apply({'params': params}, rd, EX, use_predicted_norm=True)
  rec2 = model.last_records['fine']
  assert float((rec - rec2).abs().max()) < 1e-5
  rgb2, dbg2 = frame_images(rec2, H, W, cfg.near, cfg.far, colormap='sinebow')
  assert int((rgb.int() - rgb2.int()).abs().max()) <= 1 and int((dbg.int() - dbg2.int()).abs().max()) <= 1
