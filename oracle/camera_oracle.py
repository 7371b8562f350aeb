"""CPU ORACLE for camera -> ray generation.  TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's pinhole + radial/tangential camera (paths under /root/reference/hypernerf/):
  _compute_residual_and_jacobian   camera.py:24-72
  _radial_and_tangential_undistort camera.py:75-106   (Newton, 10 iterations, eps = 1e-9)
  Camera.from_json                 camera.py:140-161  (legacy "tangential" key)
  Camera.pixel_to_local_rays       camera.py:226-243
  Camera.pixels_to_rays            camera.py:245-270
  Camera.get_pixel_centers         camera.py:364-368
  camera_to_rays                   datasets/core.py:51-76

Pinning: the reference's only fixture for this code is hypernerf/testdata/camera.json (copied as DATA to
tests/golden/reference_testdata_camera.json); its test file (camera_test.py of upstream HyperNeRF) is absent, so the
pins are mathematical: project(undistort(p)) == p round trips, and the closed form for a distortion-free camera.
"""
import json

import numpy as np


def compute_residual_and_jacobian(x, y, xd, yd, k1=0.0, k2=0.0, k3=0.0, p1=0.0, p2=0.0):
  """camera.py:24-72."""
  r = x * x + y * y
  d = 1.0 + r * (k1 + r * (k2 + k3 * r))
  fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd
  fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd
  d_r = (k1 + r * (2.0 * k2 + 3.0 * k3 * r))
  d_x = 2.0 * x * d_r
  d_y = 2.0 * y * d_r
  fx_x = d + d_x * x + 2.0 * p1 * y + 6.0 * p2 * x
  fx_y = d_y * x + 2.0 * p1 * x + 2.0 * p2 * y
  fy_x = d_x * y + 2.0 * p2 * y + 2.0 * p1 * x
  fy_y = d + d_y * y + 2.0 * p2 * x + 6.0 * p1 * y
  return fx, fy, fx_x, fx_y, fy_x, fy_y


def radial_and_tangential_undistort(xd, yd, k1=0, k2=0, k3=0, p1=0, p2=0, eps=1e-9, max_iterations=10):
  """camera.py:75-106."""
  x, y = xd.copy(), yd.copy()
  for _ in range(max_iterations):
    fx, fy, fx_x, fx_y, fy_x, fy_y = compute_residual_and_jacobian(x, y, xd, yd, k1, k2, k3, p1, p2)
    denominator = fy_x * fx_y - fx_x * fy_y
    x_numerator = fx * fy_y - fy * fx_y
    y_numerator = fy * fx_x - fx * fy_x
    ok = np.abs(denominator) > eps
    safe = np.where(ok, denominator, 1.0)
    x = x + np.where(ok, x_numerator / safe, 0.0)
    y = y + np.where(ok, y_numerator / safe, 0.0)
  return x, y


class Camera:
  def __init__(self, orientation, position, focal_length, principal_point, image_size, skew=0.0, pixel_aspect_ratio=1.0,
               radial_distortion=None, tangential_distortion=None, dtype=np.float64):
    self.dtype = dtype
    self.orientation = np.array(orientation, dtype)
    self.position = np.array(position, dtype)
    self.focal_length = dtype(focal_length)
    self.principal_point = np.array(principal_point, dtype)
    self.skew = dtype(skew)
    self.pixel_aspect_ratio = dtype(pixel_aspect_ratio)
    self.radial_distortion = np.array([0, 0, 0] if radial_distortion is None else radial_distortion, dtype)
    self.tangential_distortion = np.array([0, 0] if tangential_distortion is None else tangential_distortion, dtype)
    self.image_size = np.array(image_size, np.uint32)

  @classmethod
  def from_json(cls, path, dtype=np.float64):
    """camera.py:140-161."""
    j = json.load(open(path))
    if 'tangential' in j:
      j['tangential_distortion'] = j['tangential']
    return cls(j['orientation'], j['position'], j['focal_length'], j['principal_point'], j['image_size'], j['skew'],
               j['pixel_aspect_ratio'], j['radial_distortion'], j['tangential_distortion'], dtype=dtype)

  def pixel_to_local_rays(self, pixels):
    """camera.py:226-243."""
    y = (pixels[..., 1] - self.principal_point[1]) / (self.focal_length * self.pixel_aspect_ratio)
    x = (pixels[..., 0] - self.principal_point[0] - y * self.skew) / self.focal_length
    if np.any(self.radial_distortion != 0) or np.any(self.tangential_distortion != 0):
      x, y = radial_and_tangential_undistort(x, y, *self.radial_distortion, *self.tangential_distortion)
    dirs = np.stack([x, y, np.ones_like(x)], axis=-1)
    return dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)

  def pixels_to_rays(self, pixels):
    """camera.py:245-270."""
    batch_shape = pixels.shape[:-1]
    local = self.pixel_to_local_rays(pixels.reshape(-1, 2))
    rays = (self.orientation.T @ local[..., None])[..., 0]
    rays = rays / np.linalg.norm(rays, axis=-1, keepdims=True)
    return rays.reshape(*batch_shape, 3)

  def get_pixel_centers(self):
    """camera.py:364-368."""
    xx, yy = np.meshgrid(np.arange(self.image_size[0], dtype=self.dtype), np.arange(self.image_size[1], dtype=self.dtype))
    return np.stack([xx, yy], axis=-1) + 0.5

  def project_local(self, x, y):
    """Forward distortion + intrinsics (camera.py:272-311 `project`, camera frame part) - used for round-trip pins."""
    k1, k2, k3 = self.radial_distortion
    p1, p2 = self.tangential_distortion
    r2 = x * x + y * y
    dist = 1.0 + r2 * (k1 + r2 * (k2 + k3 * r2))
    xd = x * dist + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * dist + 2 * p2 * x * y + p1 * (r2 + 2 * y * y)
    return (self.focal_length * xd + self.skew * yd + self.principal_point[0],
            self.focal_length * self.pixel_aspect_ratio * yd + self.principal_point[1])


def camera_to_rays(camera):
  """datasets/core.py:51-76."""
  H, W = int(camera.image_size[1]), int(camera.image_size[0])
  pixels = camera.get_pixel_centers()
  return {'origins': np.tile(camera.position[None, None, :], (H, W, 1)).astype(np.float32),
          'directions': camera.pixels_to_rays(pixels).astype(np.float32),
          'pixels': pixels.astype(np.float32)}
