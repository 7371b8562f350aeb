"""Camera -> ray generation.  CPU: the oracle's pins (round trips on the reference's own
testdata/camera.json fixture, closed form without distortion).  GPU: HIP kernel and fused render path vs the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import camera_oracle as CO

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_testdata_camera.json')


def test_fixture_is_the_reference_testdata_and_legacy_key_is_honoured():
  c = CO.Camera.from_json(FIXTURE)
  assert tuple(c.image_size) == (2448, 3264) and c.tangential_distortion[0] == pytest.approx(0.001109850269091041)
  assert np.allclose(c.orientation @ c.orientation.T, np.eye(3), atol=1e-12)


def test_undistort_round_trip_and_unit_rays():
  c = CO.Camera.from_json(FIXTURE)
  rng = np.random.default_rng(0)
  px = np.stack([rng.uniform(0, 2448, 500), rng.uniform(0, 3264, 500)], -1)
  local = c.pixel_to_local_rays(px)
  x, y = local[:, 0] / local[:, 2], local[:, 1] / local[:, 2]
  back = np.stack(c.project_local(x, y), -1)
  assert np.abs(back - px).max() < 1e-7                    # Newton x10 converged: project(undistort(p)) == p
  d = c.pixels_to_rays(px)
  assert np.allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-14)
  assert np.allclose(d, (c.orientation.T @ local.T).T, atol=1e-14)


def test_distortion_free_closed_form_and_camera_to_rays_shapes():
  R = np.eye(3)
  c = CO.Camera(R, [1.0, 2.0, 3.0], 100.0, [4.0, 3.0], [8, 6])
  px = c.get_pixel_centers()
  assert px.shape == (6, 8, 2) and px[0, 0].tolist() == [0.5, 0.5] and px[5, 7].tolist() == [7.5, 5.5]
  d = c.pixels_to_rays(px)
  ref = np.stack([(px[..., 0] - 4) / 100, (px[..., 1] - 3) / 100, np.ones((6, 8))], -1)
  assert np.allclose(d, ref / np.linalg.norm(ref, axis=-1, keepdims=True), atol=1e-15)
  out = CO.camera_to_rays(c)
  assert out['origins'].shape == (6, 8, 3) and np.all(out['origins'] == np.float32([1, 2, 3])) and out['directions'].dtype == np.float32


@pytest.mark.gpu
def test_hip_camera_to_rays_matches_oracle_on_reference_fixture():
  from nerfds_amd.camera import Camera, camera_to_rays
  cam = Camera.from_json(FIXTURE)
  out = camera_to_rays(cam, torch.device('cuda', 0))
  ref = CO.Camera.from_json(FIXTURE)
  H, W = cam.image_shape
  assert out['directions'].shape == (H, W, 3) and out['pixels'].shape == (H, W, 2)
  sub = (slice(0, H, 97), slice(0, W, 89))
  d_ref = ref.pixels_to_rays(ref.get_pixel_centers()[sub])
  assert np.abs(out['directions'][sub].cpu().numpy() - d_ref).max() < 5e-6           # fp32 kernel vs fp64 oracle
  assert torch.equal(out['origins'][0, 0].cpu(), torch.tensor(cam.position))
  assert np.array_equal(out['pixels'][sub].cpu().numpy(), ref.get_pixel_centers()[sub].astype(np.float32))
  n = torch.linalg.norm(out['directions'], dim=-1)                                     # property at full size (8 M rays)
  assert float((n - 1).abs().max()) < 1e-6
  px = torch.tensor([[10.25, 20.5], [2000.0, 3000.0]], device='cuda')
  assert np.abs(cam.pixels_to_rays(px).cpu().numpy() - ref.pixels_to_rays(px.cpu().numpy().astype(np.float64))).max() < 5e-6
  with pytest.raises(ValueError):
    cam.pixels_to_rays(px.double())


@pytest.mark.gpu
def test_fused_camera_render_equals_render_of_generated_rays():
  from nerfds_amd import nerf_ds_config, init_params
  from nerfds_amd.camera import Camera, camera_to_rays
  from nerfds_amd.model import NerfModel
  cam = Camera.from_json(FIXTURE)
  cam.image_size = np.array([24, 16], np.uint32)              # a 24x16 image with the same distortion model
  cam.principal_point = np.float32([12.0, 8.0])
  cam.focal_length = np.float32(30.0)
  cam.position = np.float32([0.0, 0.0, 1.0])
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=16, num_fine_samples=16)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  dev = torch.device('cuda', 0)
  m = NerfModel(cfg, device=dev)
  H, W = cam.image_shape
  extra = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
  meta = {'warp': torch.ones((H, W, 1), dtype=torch.int64, device=dev)}
  t, u = np.full((H * W, 16), 0.5), np.full((H * W, 16), 0.25)
  kw = dict(use_predicted_norm=True, t_rand=t, u_rand=u, precision='f32')
  rays = camera_to_rays(cam, dev)
  a = m.apply({'params': params}, dict(origins=rays['origins'], directions=rays['directions'], viewdirs=rays['directions'],
                                       metadata=meta, mask=None), extra, **kw)['fine']
  b = m.apply({'params': params}, dict(camera=cam, metadata=meta, mask=None), extra, **kw)['fine']
  assert b['rgb'].shape == (H, W, 3)
  for k in ('rgb', 'depth', 'ray_delta_x', 'med_points'):
    assert torch.equal(a[k], b[k]), k                          # same arithmetic -> bit-identical
