"""Camera -> rays on the GPU: the host-side mirror of ``hypernerf.camera.Camera`` (from_json, image_shape,
pixels_to_rays, get_pixel_centers: camera.py:140-161, 214-216, 245-270, 364-368) and of
``datasets.camera_to_rays`` (datasets/core.py:51-76).  The arithmetic runs in the HIP kernel
(csrc/camera_kernel.hip); this file only marshals parameters.  render.py:201 calls ``camera_to_rays(camera)``
once per frame on the host in numpy; here the 36 B/ray never cross PCIe."""
from __future__ import annotations

import ctypes as C
import json
from typing import Optional

import numpy as np
import torch

from . import _native as N


class Camera:
  def __init__(self, orientation, position, focal_length, principal_point, image_size, skew=0.0, pixel_aspect_ratio=1.0,
               radial_distortion=None, tangential_distortion=None):
    self.orientation = np.array(orientation, np.float32).reshape(3, 3)
    self.position = np.array(position, np.float32).reshape(3)
    self.focal_length = np.float32(focal_length)
    self.principal_point = np.array(principal_point, np.float32).reshape(2)
    self.skew = np.float32(skew)
    self.pixel_aspect_ratio = np.float32(pixel_aspect_ratio)
    self.radial_distortion = np.array([0, 0, 0] if radial_distortion is None else radial_distortion, np.float32)
    self.tangential_distortion = np.array([0, 0] if tangential_distortion is None else tangential_distortion, np.float32)
    self.image_size = np.array(image_size, np.uint32)        # (width, height)

  @classmethod
  def from_json(cls, path):
    with open(path) as fp:
      j = json.load(fp)
    if 'tangential' in j:                                     # legacy key (camera.py:147-149)
      j['tangential_distortion'] = j['tangential']
    return cls(j['orientation'], j['position'], j['focal_length'], j['principal_point'], j['image_size'], j['skew'],
               j['pixel_aspect_ratio'], j['radial_distortion'], j['tangential_distortion'])

  @property
  def image_shape(self):
    return int(self.image_size[1]), int(self.image_size[0])

  def scale(self, scale: float) -> 'Camera':
    """camera.py:370-387 (render.py:183 renders at ``image_scale``): intrinsics and image size scaled, pose and distortion kept."""
    if scale <= 0:
      raise ValueError('scale needs to be positive.')
    w, h = (int(round(float(v) * scale)) for v in self.image_size)
    return Camera(self.orientation, self.position, self.focal_length * scale, self.principal_point * scale, (w, h), self.skew,
                  self.pixel_aspect_ratio, self.radial_distortion, self.tangential_distortion)

  def _struct(self) -> N.CameraStruct:
    s = N.CameraStruct()
    s.orientation[:] = self.orientation.reshape(-1).tolist()
    s.position[:] = self.position.tolist()
    s.focal_length = float(self.focal_length)
    s.principal_point[:] = self.principal_point.tolist()
    s.skew, s.pixel_aspect_ratio = float(self.skew), float(self.pixel_aspect_ratio)
    s.radial_distortion[:] = self.radial_distortion.tolist()
    s.tangential_distortion[:] = self.tangential_distortion.tolist()
    s.image_width, s.image_height = int(self.image_size[0]), int(self.image_size[1])
    return s

  def pixels_to_rays(self, pixels: torch.Tensor) -> torch.Tensor:
    """camera.py:245-270 for device pixels [..., 2] -> unit world-frame directions [..., 3]."""
    if pixels.shape[-1] != 2:
      raise ValueError('The last dimension of pixels must be 2.')          # camera.py:254-255
    if pixels.dtype != torch.float32:
      raise ValueError(f'pixels dtype ({pixels.dtype!r}) must match camera dtype (torch.float32)')   # camera.py:256-258
    if not pixels.is_cuda:
      raise RuntimeError('pixels must live on the GPU: there is no CPU path')
    flat = pixels.reshape(-1, 2).contiguous()
    out = torch.empty((flat.shape[0], 3), device=pixels.device, dtype=torch.float32)
    _call(self, pixels.device, 0, flat.shape[0], flat, None, out, None)
    return out.reshape(*pixels.shape[:-1], 3)


def _call(camera: Camera, device, first_pixel, n, pixels, origins, directions, pixels_out):
  lib = N.load()
  ptr = lambda t: (t.data_ptr() if t is not None else None)
  s = camera._struct()
  rc = lib.nerfds_camera_to_rays(device.index or 0, C.byref(s), first_pixel, n, ptr(pixels), ptr(origins), ptr(directions),
                                 ptr(pixels_out), C.c_void_p(torch.cuda.current_stream(device).cuda_stream))
  if rc != 0:
    raise RuntimeError(f'nerfds_camera_to_rays failed ({rc}): {N.last_error(None)}')


def camera_to_rays(camera: Camera, device: Optional[torch.device] = None):
  """datasets/core.py:51-76: {'origins', 'directions', 'pixels'} of shape [H, W, .], generated on the GPU."""
  if not torch.cuda.is_available():
    raise RuntimeError('camera_to_rays needs an MI355X; there is no CPU path')
  device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
  H, W = camera.image_shape
  o = torch.empty((H, W, 3), device=device, dtype=torch.float32)
  d = torch.empty((H, W, 3), device=device, dtype=torch.float32)
  p = torch.empty((H, W, 2), device=device, dtype=torch.float32)
  _call(camera, device, 0, H * W, None, o, d, p)
  return {'origins': o, 'directions': d, 'pixels': p}
