"""A caller-defined loss on the rendered rays: ``torch.autograd`` over the C ABI's ``nerfds_trainer_forward`` / ``nerfds_render_rays_bwd``.

The reference differentiates an arbitrary ``_loss_fn`` through ``model.apply`` with ``jax.value_and_grad`` (hypernerf/training.py:441-494);
``Trainer.step`` bakes the reference's own loss menu into the fused step.  This module is the loss-agnostic route: the parameters are ONE flat
fp32 leaf tensor (a view of the trainer's device vector, ``requires_grad``), ``DifferentiableRenderer.__call__`` returns ``rgb`` / ``depth`` /
``acc`` of both levels as tensors of the autograd graph, and ``loss.backward()`` leaves d loss / d parameters in ``params.grad`` - any torch loss,
any torch optimizer (in-place updates of ``params`` ARE updates of the library's parameter vector; its weight streams are re-packed from it at
the start of every pass).  The backward re-runs the forward (include/nerfds.h: the workspace holds one level's activations), so a step costs
the forward twice; ``Trainer.step`` stays the fast path for the reference's objective.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _native as N
from .training import Trainer


class _LevelOut(C.Structure):
  _fields_ = [('rgb', C.c_void_p), ('depth', C.c_void_p), ('acc', C.c_void_p)]


class _LevelCot(C.Structure):
  _fields_ = [('d_rgb', C.c_void_p), ('d_depth', C.c_void_p), ('d_acc', C.c_void_p)]


def _bind(lib):
  if getattr(lib, '_autograd_bound', False):
    return lib
  lib.nerfds_trainer_forward.argtypes = [C.c_void_p, C.POINTER(N.Rays), C.POINTER(N.Extra), C.POINTER(N.Rand), C.POINTER(_LevelOut), C.POINTER(_LevelOut),
                                         C.c_void_p]
  lib.nerfds_render_rays_bwd.argtypes = [C.c_void_p, C.POINTER(N.Rays), C.POINTER(N.Extra), C.POINTER(N.Rand), C.POINTER(_LevelCot), C.POINTER(_LevelCot),
                                         C.c_void_p]
  lib._autograd_bound = True
  return lib


class _RenderRays(torch.autograd.Function):
  """(params) -> (rgb_fine, depth_fine, acc_fine, rgb_coarse, depth_coarse, acc_coarse); everything else rides in ``call`` (not differentiated)."""

  @staticmethod
  def forward(ctx, params: torch.Tensor, renderer: 'DifferentiableRenderer', call: Dict[str, Any]):
    ctx.renderer, ctx.call = renderer, call
    ctx.version = params._version          # torch's own in-place counter of the leaf
    return renderer._forward(call)

  @staticmethod
  def backward(ctx, *cots):
    r = ctx.renderer
    if r.params._version != ctx.version:
      raise RuntimeError('the parameters were modified in place between this forward and its backward (the backward re-runs the forward on the CURRENT parameters)')
    return r._backward(ctx.call, cots), None, None


class DifferentiableRenderer:
  """``render = DifferentiableRenderer(cfg, params, max_rays)``; ``out = render(rays_dict, extra_params, seed=...)``;
  ``loss = f(out['fine']['rgb'], out['fine']['acc'], ...)``; ``loss.backward()``; ``render.params.grad`` is the flat gradient
  (``render.trainer.leaves`` names the slices by their Flax paths; ``render.grads_tree()`` returns the tree)."""

  def __init__(self, cfg, params: Optional[Dict[str, Any]] = None, max_rays: int = 4096, device: Optional[torch.device] = None):
    self.trainer = Trainer(cfg, params, max_rays=max_rays, device=device)
    self.cfg, self.device = cfg, self.trainer.device
    self._lib = _bind(self.trainer._lib)
    self.params = self.trainer.params_tensor().requires_grad_(True)      # a LEAF aliasing the library's parameter vector
    self.auto_loss_scale = True
    self.loss_scale_adjust = 0      # log2 offset of the backward's loss scale (set from max |cotangent| when auto_loss_scale; this renderer's, not the trainer's)

  # -- marshalling ---------------------------------------------------------------------------------------
  def _pack(self, rays_dict, extra_params, t_rand, u_rand, mask_ratio, near, far, seed, ray_offset):
    dev, cfg = self.device, self.cfg
    f32 = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).detach().to(dev, torch.float32).contiguous()
    origins = f32(rays_dict['origins']).reshape(-1, 3)
    R = origins.shape[0]
    if R > self.trainer.max_rays:
      raise ValueError(f'{R} rays for a renderer built with max_rays={self.trainer.max_rays}')
    directions = f32(rays_dict['directions']).reshape(-1, 3)
    viewdirs = f32(rays_dict['viewdirs']).reshape(-1, 3) if rays_dict.get('viewdirs') is not None else directions
    wid = rays_dict['metadata']['warp']
    wid = (wid if isinstance(wid, torch.Tensor) else torch.as_tensor(np.asarray(wid).astype(np.int64))).to(dev).reshape(-1).to(torch.int32).contiguous()
    gt_mask = f32(rays_dict['mask']).reshape(-1) if rays_dict.get('mask') is not None else None
    keep = [origins, directions, viewdirs, wid, gt_mask]
    rays = N.Rays(num_rays=R, origins=origins.data_ptr(), directions=directions.data_ptr(), viewdirs=viewdirs.data_ptr(), warp_id=wid.data_ptr(),
                  gt_mask=gt_mask.data_ptr() if gt_mask is not None else None, camera=None, first_pixel=0)
    g = lambda k, d=0.0: float(extra_params[k]) if extra_params.get(k) is not None else d
    ex = N.Extra(nerf_alpha=g('nerf_alpha'), warp_alpha=g('warp_alpha'), hyper_alpha=g('hyper_alpha'), hyper_sheet_alpha=g('hyper_sheet_alpha'),
                 norm_input_alpha=g('norm_input_alpha'), mask_ratio=float(mask_ratio), near=float(cfg.near if near is None else near),
                 far=float(cfg.far if far is None else far), use_stratified_sampling=int(cfg.use_stratified_sampling),
                 use_linear_disparity=int(cfg.use_linear_disparity))
    rnd = N.Rand(t_rand=None, u_rand=None, seed=int(seed) & 0xFFFFFFFFFFFFFFFF, first_ray=int(ray_offset))
    if t_rand is not None:
      t = f32(t_rand).reshape(R, cfg.num_coarse_samples); keep.append(t); rnd.t_rand = t.data_ptr()
    if u_rand is not None and cfg.num_fine_samples > 0:
      u = f32(u_rand).reshape(R, cfg.num_fine_samples); keep.append(u); rnd.u_rand = u.data_ptr()
    return dict(R=R, rays=rays, ex=ex, rnd=rnd, keep=keep)

  def __call__(self, rays_dict: Dict[str, Any], extra_params: Dict[str, Any], *, t_rand=None, u_rand=None, mask_ratio: float = 1.0,
               near: Optional[float] = None, far: Optional[float] = None, seed: int = 0, ray_offset: int = 0) -> Dict[str, Dict[str, torch.Tensor]]:
    """NerfModel.apply on the current ``params`` (models.py:1419-1565): {'coarse': {rgb [R,3], depth [R], acc [R]}, 'fine': {...}} as differentiable
    tensors.  Sampling jitter: injected ``t_rand`` / ``u_rand``, else the on-chip Philox stream keyed by ``seed`` (pass a new one every step)."""
    call = self._pack(rays_dict, extra_params, t_rand, u_rand, mask_ratio, near, far, seed, ray_offset)
    outs = _RenderRays.apply(self.params, self, call)
    two = self.cfg.num_fine_samples > 0
    ret = {'coarse': dict(zip(('rgb', 'depth', 'acc'), outs[3:6]))}
    if two:
      ret['fine'] = dict(zip(('rgb', 'depth', 'acc'), outs[0:3]))
    return ret

  def _stream(self):
    return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def _forward(self, call):
    R, dev = call['R'], self.device
    two = self.cfg.num_fine_samples > 0
    mk = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    fine = (mk(R, 3), mk(R), mk(R)) if two else (mk(0, 3), mk(0), mk(0))
    coarse = (mk(R, 3), mk(R), mk(R))
    lf = _LevelOut(*(x.data_ptr() for x in fine)) if two else None
    lc = _LevelOut(*(x.data_ptr() for x in coarse))
    rc = self._lib.nerfds_trainer_forward(self.trainer._h, C.byref(call['rays']), C.byref(call['ex']), C.byref(call['rnd']),
                                          C.byref(lf) if lf is not None else None, C.byref(lc), self._stream())
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_forward failed ({rc}): {(self._lib.nerfds_trainer_last_error(self.trainer._h) or b"").decode()}')
    return (*fine, *coarse)

  def _backward(self, call, cots):
    R, tr = call['R'], self.trainer
    two = self.cfg.num_fine_samples > 0
    c = [None if (x is None or x.numel() == 0) else x.detach().to(self.device, torch.float32).contiguous() for x in cots]
    # The backward's stored f16 g is scaled for head gradients of a mean squared error's size, at most 2 / (3 R) per unit of colour error
    # (csrc/nerfds_train.cpp g_scale): the caller's cotangents are brought to that size by a power of two.  The adjustment is THIS renderer's
    # (self.loss_scale_adjust) - the trainer's own policy (Trainer.step's overflow ladder) is restored after the call, so a caller who mixes
    # render.trainer.step() with autograd passes does not inherit a scale chosen for someone else's cotangents.
    adjust = int(getattr(self, 'loss_scale_adjust', 0))
    if self.auto_loss_scale:
      gmax = max([float(x.abs().max()) for x in c if x is not None] + [0.0])
      if gmax > 0.0 and np.isfinite(gmax):
        adjust = int(np.clip(np.floor(np.log2((2.0 / (3.0 * R)) / gmax)), -40, 16))
    ptr = lambda x: x.data_ptr() if x is not None else None
    df = _LevelCot(ptr(c[0]), ptr(c[1]), ptr(c[2])) if two else None
    dc = _LevelCot(ptr(c[3]), ptr(c[4]), ptr(c[5]))
    retries = 0
    try:
      while True:
        self._lib.nerfds_trainer_set_loss_scale_adjust(tr._h, adjust)
        rc = self._lib.nerfds_render_rays_bwd(tr._h, C.byref(call['rays']), C.byref(call['ex']), C.byref(call['rnd']),
                                              C.byref(df) if df is not None else None, C.byref(dc), self._stream())
        if rc != 0:
          raise RuntimeError(f'nerfds_render_rays_bwd failed ({rc}): {(self._lib.nerfds_trainer_last_error(tr._h) or b"").decode()}')
        grad = tr.grads_tensor().clone()
        if bool(torch.isfinite(grad).all()) or retries >= tr.max_overflow_retries or adjust <= -38:
          self.loss_scale_adjust = adjust
          return grad
        retries += 1                    # the scaled f16 g left f16's range: a quarter of the scale, same samples (as Trainer.step)
        adjust -= 2
    finally:
      tr._set_numerics()                # the trainer's own policy back in place

  # -- conveniences ----------------------------------------------------------------------------------------
  def grads_tree(self) -> Dict[str, Any]:
    """``params.grad`` as the Flax-named tree (numpy)."""
    if self.params.grad is None:
      raise RuntimeError('no gradient yet: call backward() on a loss first')
    return self.trainer._tree(self.params.grad.detach().cpu().numpy())

  def params_tree(self) -> Dict[str, Any]:
    return self.trainer.get_params()
