"""Frame output path that follows the gather (render.py:220-277): ray records of one frame -> the uint8 rgb frame
and the 2 x 3 debug mosaic the reference writes into its videos, on the GPU (csrc/frame_kernel.hip), byte-exact with
the numpy code of render.py:231-268 / visualization.py:186-235 / image_utils.py:124-131; plus the raw-result dict of
render.py:222-229.  There is no CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _native as N

# keys render.py:192-193 keeps in the raw results it np.save's
RELEVANT_KEYS = ('rgb', 'med_depth', 'ray_norm', 'ray_delta_x', 'med_points', 'ray_predicted_mask', 'ray_rotation_field')


def get_colormap(name: str = 'magma', num_bins: int = 256) -> np.ndarray:
  """[256, 3] float64 table (visualization.get_colormap, visualization.py:173-183).  'sinebow' and 'gray' are analytic;
  anything else is looked up in matplotlib, as the reference does (cm.get_cmap(name) resampled to 256 bins)."""
  if name == 'sinebow':
    h = np.linspace(0, 1, num_bins)
    f = lambda x: np.sin(np.pi * x) ** 2
    return np.stack([f(3 / 6 - h), f(5 / 6 - h), f(7 / 6 - h)], -1)
  if name == 'gray':
    g = np.linspace(0, 1, num_bins)
    return np.stack([g, g, g], -1)
  try:
    from matplotlib import cm
  except ImportError as e:
    raise RuntimeError(f"colormap {name!r} needs matplotlib (not installed): pass a [256, 3] table or use 'sinebow' / 'gray'") from e
  return np.asarray(cm.get_cmap(name)(np.linspace(0, 1, num_bins))[:, :3], np.float64)


def records_from_render(render: Dict[str, torch.Tensor]) -> torch.Tensor:
  """Rebuilds the [H * W, 26] record tensor from a render_image result dict (the inverse of evaluation.unpack)."""
  any_v = render['rgb']
  H, W = any_v.shape[:2]
  rec = torch.zeros((H * W, N.RAY_REC), dtype=torch.float32, device=any_v.device)
  for k, (o, n) in N.RAY_FIELDS.items():
    if k in render:
      rec[:, o:o + n] = render[k].reshape(H * W, n).to(torch.float32)
  return rec


def frame_images(records: torch.Tensor, height: int, width: int, near: float, far: float, colormap='magma',
                 want_debug: bool = True, stream: Optional[torch.cuda.Stream] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
  """records: CUDA float32 [H * W, 26] (or a render_image dict) -> (rgb uint8 [H, W, 3], debug uint8 [2H, 3W, 3])."""
  if isinstance(records, dict):
    records = records_from_render(records)
  if not records.is_cuda:
    raise RuntimeError('frame_images runs on the GPU: records must be a CUDA tensor (there is no CPU path)')
  lib = N.load()
  rec = records.reshape(-1, N.RAY_REC).contiguous().to(torch.float32)
  if rec.shape[0] != height * width:
    raise ValueError(f'{rec.shape[0]} records for a {height} x {width} frame')
  dev = rec.device
  rgb = torch.empty((height, width, 3), dtype=torch.uint8, device=dev)
  dbg = torch.empty((2 * height, 3 * width, 3), dtype=torch.uint8, device=dev) if want_debug else None
  lut = None
  if want_debug:
    table = get_colormap(colormap) if isinstance(colormap, str) else np.asarray(colormap, np.float64)
    if table.shape != (256, 3):
      raise ValueError('colormap must be [256, 3]')
    lut = torch.from_numpy(np.ascontiguousarray(table)).to(dev)
  s = stream if stream is not None else torch.cuda.current_stream(dev)
  rc = lib.nerfds_frame_images(dev.index or 0, rec.data_ptr(), height, width, float(near), float(far),
                               lut.data_ptr() if lut is not None else None, rgb.data_ptr(),
                               dbg.data_ptr() if dbg is not None else None, C.c_void_p(s.cuda_stream))
  if rc != 0:
    raise RuntimeError(f'nerfds_frame_images failed ({rc}): {N.last_error(None)}')
  return rgb, dbg


def render_frame(model, variables, camera, warp_id: int, extra_params, *, gt_mask=None, chunk: int = 65536, colormap='magma',
                 precision: Optional[str] = None, seed: int = 0, want_debug: bool = True):
  """One iteration of render.py's frame loop (render.py:196-268) without leaving the GPU: pixel centres -> rays
  (camera.py:245-270) -> the fused ray kernel, chunk by chunk -> ray records -> uint8 frames.  Nothing per-ray crosses PCIe:
  the camera goes in as 27 scalars, the frames come out as bytes.

  Returns (rgb_u8 [H, W, 3], debug_u8 [2H, 3W, 3] or None, records [H * W, 26]).  ``gt_mask``: optional [H, W] / [H * W] mask
  (``batch['mask']``, render.py:215-216)."""
  cfg = model.cfg
  H, W = camera.image_shape
  n = H * W
  dev = model.device
  records = torch.empty((n, N.RAY_REC), dtype=torch.float32, device=dev)
  mask = None
  if gt_mask is not None:
    mask = (gt_mask if isinstance(gt_mask, torch.Tensor) else torch.as_tensor(np.asarray(gt_mask))).to(dev, torch.float32).reshape(-1)
  level = 'fine' if cfg.num_fine_samples > 0 else 'coarse'      # NerfModel.last_records is keyed by the real level names
  for ci, first in enumerate(range(0, n, chunk)):
    cnt = min(chunk, n - first)
    rays = {'camera': camera, 'pixel_range': (first, cnt),
            'metadata': {'warp': torch.full((cnt, 1), int(warp_id), dtype=torch.int32, device=dev)}}
    if mask is not None:
      rays['mask'] = mask[first:first + cnt]
    model.apply(variables, rays, extra_params, rngs={'coarse': seed * 1000 + ci, 'fine': seed * 1000 + ci + 500},
                use_predicted_norm=cfg.predict_norm, precision=precision)
    records[first:first + cnt] = model.last_records[level]
  rgb, dbg = frame_images(records, H, W, cfg.near, cfg.far, colormap=colormap, want_debug=want_debug)
  return rgb, dbg, records


def raw_result(render: Dict[str, torch.Tensor]) -> Dict[str, np.ndarray]:
  """The per-frame dict render.py:222-229 appends to ``raw_result_list`` (host numpy copies of the relevant keys)."""
  return {k: render[k].detach().cpu().numpy() for k in RELEVANT_KEYS if k in render}
