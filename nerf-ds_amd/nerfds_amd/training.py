"""Host-side mirror of the reference's training step for BASELINE config 4 (hypernerf/training.py:198-511 reduced to the
first-order MSE objective): ``Trainer`` wraps the C-ABI ``nerfds_trainer_*`` (csrc/nerfds_train.cpp),
``train_step`` keeps the call shape ``train_step(model, rng_key, state, batch, scalar_params)``.  No CPU fallback."""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _native as N
from .config import NerfModelConfig
from .model import _cfg_struct

GRADS_ONLY = 1
SIGMA_GRAD = 2


class Objective(C.Structure):
  _fields_ = [('warp_reg_loss_weight', C.c_float), ('warp_reg_loss_alpha', C.c_float), ('warp_reg_loss_scale', C.c_float),
              ('back_facing_reg_weight', C.c_float), ('predicted_mask_loss_weight', C.c_float), ('sharp_weights_std', C.c_float),
              ('use_mask_sharp_weights', C.c_int32), ('norm_loss_weight', C.c_float), ('hyper_reg_loss_weight', C.c_float),
              ('background_loss_weight', C.c_float), ('background_loss_alpha', C.c_float), ('background_loss_scale', C.c_float),
              ('background_points', C.c_void_p), ('background_ids', C.c_void_p), ('num_background_points', C.c_int64),
              ('elastic_loss_weight', C.c_float), ('elastic_reduce_by_weight', C.c_int32), ('mask_occlusion_reg_loss_weight', C.c_float)]


class Numerics(C.Structure):
  """nerfds_train_numerics (include/nerfds.h): the numeric policy of the trainer's f16 storage."""
  _fields_ = [('loss_scale_log2_adjust', C.c_int32), ('tangent_scale_log2_adjust', C.c_int32), ('chain_arith', C.c_int32), ('fp32_step', C.c_int32), ('diagnose', C.c_int32)]


# nerfds_trainer_overflow_sources bits (include/nerfds.h NERFDS_OVF_*)
OVF_ACTIVATION, OVF_PRIMAL_G, OVF_TANGENT, OVF_COTANGENT, OVF_FP32, OVF_FP32_SECOND_ORDER, OVF_FP32_BACKWARD = 1, 2, 4, 8, 16, 32, 64
_OVF_NAMES = ((OVF_ACTIVATION, 'f16 activation'), (OVF_PRIMAL_G, 'loss-scaled f16 g of the primal chains'), (OVF_TANGENT, 'stored f16 tangents'),
              (OVF_COTANGENT, 'stored f16 cotangents of the tangent pass'), (OVF_FP32, 'fp32 value of the forward pass (loss / head output / warp)'),
              (OVF_FP32_SECOND_ORDER, 'fp32 tangent / cotangent of the second-order terms'), (OVF_FP32_BACKWARD, 'fp32 cotangent of the backward pass'))


def overflow_names(mask: int) -> str:
  return ', '.join(n for b, n in _OVF_NAMES if mask & b) or 'unattributed (an array a later level overwrote)'


class _DevVec:
  """Zero-copy torch view of a library-owned fp32 device vector (through __cuda_array_interface__)."""

  def __init__(self, ptr: int, n: int):
    self.__cuda_array_interface__ = {'shape': (n,), 'typestr': '<f4', 'data': (int(ptr), False), 'version': 2}


def allreduce_mean_(t: torch.Tensor, force: bool = False) -> torch.Tensor:
  """jax.lax.pmean(grad, 'batch') (training.py:502): in-place mean over the ranks of the default process group.
  ``force``: issue the collective also in a one-rank group (Trainer.step(data_parallel=True); a test executes RCCL that way)."""
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force):
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t /= dist.get_world_size()
  return t


def _bind(lib):
  if getattr(lib, '_trainer_bound', False):
    return lib
  lib.nerfds_trainer_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(N.ModelCfg), C.c_int64]
  lib.nerfds_trainer_destroy.argtypes = [C.c_void_p]
  lib.nerfds_trainer_param_count.argtypes = [C.c_void_p]
  lib.nerfds_trainer_param_count.restype = C.c_int64
  lib.nerfds_trainer_num_leaves.argtypes = [C.c_void_p]
  lib.nerfds_trainer_leaf.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
  lib.nerfds_trainer_params.argtypes = [C.c_void_p]
  lib.nerfds_trainer_params.restype = C.c_void_p
  lib.nerfds_trainer_grads.argtypes = [C.c_void_p]
  lib.nerfds_trainer_grads.restype = C.c_void_p
  lib.nerfds_trainer_reset_optimizer.argtypes = [C.c_void_p]
  lib.nerfds_trainer_set_step.argtypes = [C.c_void_p, C.c_int64]
  lib.nerfds_trainer_get_step.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
  lib.nerfds_trainer_set_loss_scale_adjust.argtypes = [C.c_void_p, C.c_int32]
  lib.nerfds_trainer_set_numerics.argtypes = [C.c_void_p, C.POINTER(Numerics)]
  lib.nerfds_trainer_get_numerics.argtypes = [C.c_void_p, C.POINTER(Numerics)]
  lib.nerfds_trainer_overflow_sources.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
  lib.nerfds_trainer_nonfinite.argtypes = [C.c_void_p]
  lib.nerfds_trainer_download.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
  lib.nerfds_trainer_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
  lib.nerfds_trainer_step.argtypes = [C.c_void_p, C.POINTER(N.Rays), C.c_void_p, C.POINTER(N.Extra), C.POINTER(N.Rand), C.POINTER(Objective),
                                      C.c_float, C.c_uint32, C.POINTER(C.c_float), C.c_void_p]
  lib.nerfds_trainer_apply.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
  lib.nerfds_trainer_clip_gradients.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p]
  lib.nerfds_trainer_target_norm.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
  lib.nerfds_trainer_last_error.argtypes = [C.c_void_p]
  lib.nerfds_trainer_debug_read.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_longlong]
  lib.nerfds_trainer_debug_read.restype = C.c_longlong
  lib.nerfds_trainer_last_error.restype = C.c_char_p
  lib._trainer_bound = True
  return lib


class Trainer:
  """Owns the flat fp32 parameter / gradient / Adam vectors on one MI355X."""

  def __init__(self, cfg: NerfModelConfig, params: Optional[Dict[str, Any]] = None, max_rays: int = 4096,
               device: Optional[torch.device] = None, warp_embeds=None):
    """``warp_embeds``: the list of warp ids present in the dataset (model.warp_embeds = embeddings_dict['warp'], models.py:248-250): what the
    background regulariser draws its random ids from (training.py:163).  None = range(num_warp_embeds), which is the same thing when the
    training ids are contiguous from 0 - with sparse ids it would regularise GLO rows no ray ever trains."""
    cfg.validate()
    self.cfg = cfg
    self.warp_embeds = None if warp_embeds is None else torch.as_tensor(np.asarray(list(warp_embeds)).astype(np.int64)).reshape(-1)
    if self.warp_embeds is not None and (self.warp_embeds.numel() == 0 or int(self.warp_embeds.min()) < 0 or int(self.warp_embeds.max()) >= cfg.num_warp_embeds):
      raise ValueError('warp_embeds must be a non-empty list of ids in [0, num_warp_embeds)')
    self._lib = _bind(N.load())
    if not torch.cuda.is_available():
      raise RuntimeError('Trainer needs an MI355X (torch.cuda is not available); there is no CPU path')
    self.device = N.resolve_device(device)
    self._cstruct = _cfg_struct(cfg)
    h = C.c_void_p()
    rc = self._lib.nerfds_trainer_create(C.byref(h), self.device.index, C.byref(self._cstruct), max_rays)
    if rc != 0:
      msg = (self._lib.nerfds_trainer_last_error(None) or b'').decode()
      raise (NotImplementedError if rc == -95 else RuntimeError)(f'nerfds_trainer_create failed ({rc}): {msg}')
    self._h = h
    self.max_rays = max_rays
    # Numeric policy of the f16 storage (include/nerfds.h nerfds_train_numerics), moved by the overflow ladder of step():
    #   loss_scale_adjust     log2 offset of the primal chains' stored g, lowered by 2 when THAT array overflowed
    #   tangent_scale_adjust  log2 offset of the second-order terms' stored tangents / cotangents, lowered by 2 when THOSE overflowed
    #   split_chains          every chain in split bf16 (fp32's exponent range between the layers) instead of one f16 MFMA per product
    #   fp32_step             fp32 activations and g, layer by layer: the arithmetic range of the reference's step (the ladder's last rung)
    # each relaxes one notch towards the default after `loss_scale_growth_interval` clean steps.
    self.loss_scale_adjust = 0
    self.tangent_scale_adjust = 0
    self.split_chains = False
    self.fp32_step = False
    self.loss_scale_growth_interval = 1000
    self.max_overflow_retries = 12
    self._clean_steps = 0
    self.record_leaves = False     # diagnosis: overflow_events also lists the gradient leaves that were non-finite (downloads the gradient per event)
    self.overflow_events = []      # (step call index, source mask, action) of every skipped-and-repeated attempt, newest last (bounded)
    self._calls = 0
    self.num_params = int(self._lib.nerfds_trainer_param_count(h))
    self.leaves = []
    name = C.create_string_buffer(256)
    off, rows, cols = C.c_int64(), C.c_int32(), C.c_int32()
    for i in range(self._lib.nerfds_trainer_num_leaves(h)):
      self._lib.nerfds_trainer_leaf(h, i, name, 256, C.byref(off), C.byref(rows), C.byref(cols))
      self.leaves.append((name.value.decode(), off.value, rows.value, cols.value))
    if params is not None:
      self.set_params(params)

  def __del__(self):
    h = getattr(self, '_h', None)
    if h:
      self._lib.nerfds_trainer_destroy(h)
      self._h = None

  # -- flat vector <-> Flax-named tree -----------------------------------------------------------------------
  def _download(self, which: int) -> np.ndarray:
    out = np.empty(self.num_params, np.float32)
    rc = self._lib.nerfds_trainer_download(self._h, which, out.ctypes.data)
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_download failed ({rc})')
    return out

  def _tree(self, flat: np.ndarray) -> Dict[str, Any]:
    tree: Dict[str, Any] = {}
    for name, off, rows, cols in self.leaves:
      parts = name.split('/')
      node = tree
      for p in parts[:-1]:
        node = node.setdefault(p, {})
      a = flat[off:off + rows * cols]
      node[parts[-1]] = a.reshape(cols).copy() if parts[-1] == 'bias' else a.reshape(rows, cols).copy()
    return tree

  def set_params(self, params: Dict[str, Any]) -> None:
    flat = np.zeros(self.num_params, np.float32)
    for name, off, rows, cols in self.leaves:
      node = params
      for p in name.split('/'):
        node = node[p]
      a = np.asarray(node, np.float32)
      if a.size != rows * cols:
        raise ValueError(f'{name}: expected {rows * cols} values, got shape {a.shape}')
      flat[off:off + rows * cols] = a.ravel()
    rc = self._lib.nerfds_trainer_upload(self._h, 0, flat.ctypes.data)
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_upload failed ({rc})')
    self._lib.nerfds_trainer_reset_optimizer(self._h)

  def grads_tensor(self) -> torch.Tensor:
    """The flat gradient vector as a torch CUDA tensor aliasing the library's memory (for the RCCL all-reduce)."""
    return torch.as_tensor(_DevVec(self._lib.nerfds_trainer_grads(self._h), self.num_params), device=self.device)

  def params_tensor(self) -> torch.Tensor:
    return torch.as_tensor(_DevVec(self._lib.nerfds_trainer_params(self._h), self.num_params), device=self.device)

  def apply_gradients(self, learning_rate: float, stream: Optional[torch.cuda.Stream] = None) -> None:
    """optimizer.apply_gradient (training.py:508) on the gradient vector as it stands."""
    s = stream if stream is not None else torch.cuda.current_stream(self.device)
    rc = self._lib.nerfds_trainer_apply(self._h, float(learning_rate), C.c_void_p(s.cuda_stream))
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_apply failed ({rc})')

  def target_norm(self, level: str = 'fine') -> np.ndarray:
    """out[level]['target_norm'] (models.py:1328) of the last step run with sigma_gradient=True: [R, S, 3]."""
    lv = 1 if level == 'fine' else 0
    S = self.cfg.num_coarse_samples + (self.cfg.num_fine_samples if lv else 0)
    out = np.empty((self._last_rays, S, 3), np.float32)
    rc = self._lib.nerfds_trainer_target_norm(self._h, lv, self._last_rays, out.ctypes.data)
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_target_norm failed ({rc}): {(self._lib.nerfds_trainer_last_error(self._h) or b"").decode()}')
    return out

  def debug_read(self, name: str, shape, dtype=np.float32) -> np.ndarray:
    """Development / tests: host copy of an internal buffer of the last step (include/nerfds.h nerfds_trainer_debug_read)."""
    out = np.empty(shape, dtype)
    rc = self._lib.nerfds_trainer_debug_read(self._h, name.encode(), out.ctypes.data, out.nbytes)
    if rc < 0:
      raise RuntimeError(f'nerfds_trainer_debug_read failed ({rc}): {(self._lib.nerfds_trainer_last_error(self._h) or b"").decode()}')
    return out

  def _set_numerics(self, diagnose: bool = False) -> None:
    n = Numerics(int(self.loss_scale_adjust), int(self.tangent_scale_adjust), 1 if self.split_chains else 0, 1 if self.fp32_step else 0, 1 if diagnose else 0)
    rc = self._lib.nerfds_trainer_set_numerics(self._h, C.byref(n))
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_set_numerics failed ({rc}): {(self._lib.nerfds_trainer_last_error(self._h) or b"").decode()}')

  def overflow_sources(self) -> int:
    """NERFDS_OVF_* mask of the stored arrays of the last step that hold an inf / NaN (nerfds_trainer_overflow_sources; synchronises, scans the workspace)."""
    m = C.c_uint32(0)
    rc = self._lib.nerfds_trainer_overflow_sources(self._h, C.byref(m))
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_overflow_sources failed ({rc}): {(self._lib.nerfds_trainer_last_error(self._h) or b"").decode()}')
    return int(m.value)

  def _escalate(self, src: int, unattributed_turn: int) -> str:
    """One rung of the overflow ladder: changes the policy so that the source `src` of the failed attempt cannot recur; returns what it did."""
    # Causality runs activation -> tangents -> cotangents -> primal g (the second-order chains' outputs feed the primal backward; a non-finite value
    # spreads to everything behind it): the EARLIEST source set is the one to remove, what lies behind it is its consequence.
    acts = []
    second = src & (OVF_TANGENT | OVF_COTANGENT)
    if src & (OVF_ACTIVATION | OVF_FP32) or (not second and src & (OVF_FP32_SECOND_ORDER | OVF_FP32_BACKWARD)):
      # an activation beyond f16's range, or an fp32 value of the step itself that is inf / NaN with no f16 array in front of it: no power of two helps
      self.fp32_step = True
      return 'fp32 step'
    if second:
      if not self.split_chains:
        self.split_chains = True
        acts.append('split-bf16 chains')
      if self.tangent_scale_adjust > -16:
        self.tangent_scale_adjust -= 2
        acts.append(f'tangent scale 2^{self.tangent_scale_adjust}')
      elif not acts:
        self.fp32_step = True
        acts.append('fp32 step')
    elif src & OVF_PRIMAL_G:
      if self.loss_scale_adjust > -24:
        self.loss_scale_adjust -= 2
        acts.append(f'loss scale 2^{self.loss_scale_adjust}')
      else:
        self.fp32_step = True
        acts.append('fp32 step')
    if not acts:
      # unattributed (the coarse NerfMLP's arrays are overwritten by the fine level's): second-order side first, then the primal scale, alternating;
      # after two rounds of both, the fp32 step
      if unattributed_turn >= 4:
        self.fp32_step = True
        acts.append('fp32 step')
      elif unattributed_turn % 2 == 0:
        self.split_chains = True
        self.tangent_scale_adjust = max(self.tangent_scale_adjust - 2, -16)
        acts.append(f'split-bf16 chains, tangent scale 2^{self.tangent_scale_adjust}')
      else:
        self.loss_scale_adjust = max(self.loss_scale_adjust - 2, -24)
        acts.append(f'loss scale 2^{self.loss_scale_adjust}')
    return ' + '.join(acts)

  def _relax(self) -> None:
    """After `loss_scale_growth_interval` clean steps: one notch back towards the default policy (fp32 step first, then the scales, then the chains)."""
    if self.fp32_step:
      self.fp32_step = False
    elif self.loss_scale_adjust < 0 or self.tangent_scale_adjust < 0:
      self.loss_scale_adjust = min(self.loss_scale_adjust + 1, 0) if self.loss_scale_adjust < 0 else self.loss_scale_adjust
      self.tangent_scale_adjust = min(self.tangent_scale_adjust + 1, 0)
    elif self.split_chains:
      self.split_chains = False

  def nonfinite_leaves(self):
    """Names of the gradient leaves that hold an inf / NaN after the last attempt, with their counts (diagnosis: downloads the gradient vector)."""
    g = self._download(1)
    out = []
    for name, off, rows, cols in self.leaves:
      n = int((~np.isfinite(g[off:off + rows * cols])).sum())
      if n:
        out.append((name, n, rows * cols))
    return out

  def nonfinite(self) -> bool:
    """True if the last Adam update was skipped because the gradient vector held an inf / NaN (nerfds_trainer_nonfinite; synchronises)."""
    rc = self._lib.nerfds_trainer_nonfinite(self._h)
    if rc < 0:
      raise RuntimeError(f'nerfds_trainer_nonfinite failed ({rc})')
    return bool(rc)

  @property
  def optimizer_step(self) -> int:
    """OptimizerState.step: the number of Adam updates actually applied (an update skipped for a non-finite gradient does not count)."""
    v = C.c_int64()
    rc = self._lib.nerfds_trainer_get_step(self._h, C.byref(v))
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_get_step failed ({rc})')
    return int(v.value)

  def get_params(self) -> Dict[str, Any]:
    return self._tree(self._download(0))

  def get_grads(self) -> Dict[str, Any]:
    return self._tree(self._download(1))

  def get_opt_state(self):
    """(grad_ema tree, grad_sq_ema tree): the Adam moments, shaped like the parameter tree (flax 0.3.4 _AdamParamState per leaf) - what
    checkpoint.save_checkpoint(..., opt_state=) writes so that a run can be resumed (training.py:59-66)."""
    return self._tree(self._download(2)), self._tree(self._download(3))

  def set_opt_state(self, grad_ema: Dict[str, Any], grad_sq_ema: Dict[str, Any], step: int) -> None:
    """Restores the Adam moments and the optimizer step count (checkpoint.restore_optimizer_state)."""
    for which, tree in ((2, grad_ema), (3, grad_sq_ema)):
      flat = np.empty(self.num_params, np.float32)
      for name, off, rows, cols in self.leaves:
        node = tree
        for part in name.split('/'):
          node = node[part]
        flat[off:off + rows * cols] = np.asarray(node, np.float32).reshape(-1)
      rc = self._lib.nerfds_trainer_upload(self._h, which, flat.ctypes.data)
      if rc != 0:
        raise RuntimeError(f'nerfds_trainer_upload failed ({rc})')
    rc = self._lib.nerfds_trainer_set_step(self._h, int(step))
    if rc != 0:
      raise RuntimeError(f'nerfds_trainer_set_step failed ({rc})')

  # -- one step ------------------------------------------------------------------------------------------
  def step(self, batch: Dict[str, Any], extra_params: Dict[str, Any], learning_rate: float = 0.0, *, t_rand=None, u_rand=None,
           mask_ratio: float = 1.0, near: Optional[float] = None, far: Optional[float] = None, grads_only: bool = False,
           sigma_gradient: bool = False, objective: Optional[Dict[str, float]] = None, grad_max_val: float = 0.0,
           grad_max_norm: float = 0.0, seed: Optional[int] = None, ray_offset: Optional[int] = None,
           stream: Optional[torch.cuda.Stream] = None, data_parallel: Optional[bool] = None) -> Dict[str, float]:
    """One optimisation step.  Sampling jitter (cfg.use_stratified_sampling; the reference always draws it,
    model_utils.py:84,217): injected ``t_rand`` / ``u_rand``, else the on-chip Philox stream keyed by ``seed``; with
    ``seed=None`` the trainer's own step counter is used, so that successive steps never sample the same depths.
    ``data_parallel``: None = all-reduce the gradient vector when a process group with more than one rank is initialised
    (training.py:502); False = never (a local pass such as NerfModel's target_norm); True = always (also at world size 1).
    ``ray_offset``: Philox counter of the first ray; None = rank * max_rays, so that the ranks of a data-parallel step draw
    independent jitter as the reference's per-device keys do (training.py:228 under pmap)."""
    dev = self.device
    f32 = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))).to(dev, torch.float32).contiguous()
    origins = f32(batch['origins']).reshape(-1, 3)
    R = origins.shape[0]
    directions = f32(batch['directions']).reshape(-1, 3)
    viewdirs = f32(batch['viewdirs']).reshape(-1, 3) if batch.get('viewdirs') is not None else directions
    target = f32(batch['rgb']).reshape(-1, 3)[:, :3].contiguous()
    wid = batch['metadata']['warp']
    wid = (wid if isinstance(wid, torch.Tensor) else torch.as_tensor(np.asarray(wid).astype(np.int64))).to(dev).reshape(-1).to(torch.int32).contiguous()
    gt_mask = f32(batch['mask']).reshape(-1) if batch.get('mask') is not None else None
    keep = [origins, directions, viewdirs, target, wid, gt_mask]
    rays = N.Rays(num_rays=R, origins=origins.data_ptr(), directions=directions.data_ptr(), viewdirs=viewdirs.data_ptr(),
                  warp_id=wid.data_ptr(), gt_mask=gt_mask.data_ptr() if gt_mask is not None else None, camera=None, first_pixel=0)
    g = lambda k, d=0.0: float(extra_params[k]) if extra_params.get(k) is not None else d
    ex = N.Extra(nerf_alpha=g('nerf_alpha'), warp_alpha=g('warp_alpha'), hyper_alpha=g('hyper_alpha'), hyper_sheet_alpha=g('hyper_sheet_alpha'),
                 norm_input_alpha=g('norm_input_alpha'), mask_ratio=float(mask_ratio), near=float(self.cfg.near if near is None else near),
                 far=float(self.cfg.far if far is None else far), use_stratified_sampling=int(self.cfg.use_stratified_sampling),
                 use_linear_disparity=int(self.cfg.use_linear_disparity))
    if seed is None:
      self._auto_seed = getattr(self, '_auto_seed', 0) + 1
      seed = (0x5DEECE66D * self._auto_seed + 0xB) & 0xFFFFFFFFFFFFFFFF
    if ray_offset is None:
      import torch.distributed as dist
      ray_offset = dist.get_rank() * self.max_rays if (dist.is_available() and dist.is_initialized()) else 0
    rnd = N.Rand(t_rand=None, u_rand=None, seed=int(seed) & 0xFFFFFFFFFFFFFFFF, first_ray=int(ray_offset))
    if t_rand is not None:
      t = f32(t_rand).reshape(R, self.cfg.num_coarse_samples); keep.append(t); rnd.t_rand = t.data_ptr()
    if u_rand is not None and self.cfg.num_fine_samples > 0:
      u = f32(u_rand).reshape(R, self.cfg.num_fine_samples); keep.append(u); rnd.u_rand = u.data_ptr()
    loss = (C.c_float * 16)()
    ob = None
    if objective:       # scalar_params / SpecularConfig names (training.py:36-56)
      ob = Objective(warp_reg_loss_weight=objective.get('warp_reg_loss_weight', 0.0), warp_reg_loss_alpha=objective.get('warp_reg_loss_alpha', -2.0),
                     warp_reg_loss_scale=objective.get('warp_reg_loss_scale', 0.001), back_facing_reg_weight=objective.get('back_facing_reg_weight', 0.0),
                     predicted_mask_loss_weight=objective.get('predicted_mask_loss_weight', 0.0), sharp_weights_std=objective.get('sharp_weights_std', 1.0),
                     use_mask_sharp_weights=int(self.cfg.use_mask_sharp_weights), norm_loss_weight=objective.get('norm_loss_weight', 0.0),
                     hyper_reg_loss_weight=objective.get('hyper_reg_loss_weight', 0.0))
      ob.mask_occlusion_reg_loss_weight = float(objective.get('mask_occlusion_reg_loss_weight', 0.0))      # training.py:409-417
      if objective.get('elastic_loss_weight', 0.0):         # training.py:112-156, 274-295
        if objective.get('elastic_loss_type', 'log_svals') != 'log_svals':
          raise NotImplementedError("elastic_loss_type: only 'log_svals' (the reference's default) is built")
        method = objective.get('elastic_reduce_method', 'median')
        if method not in ('median', 'weight'):
          raise ValueError(f'elastic_reduce_method {method!r}')
        ob.elastic_loss_weight, ob.elastic_reduce_by_weight = float(objective['elastic_loss_weight']), int(method == 'weight')
      if objective.get('background_loss_weight', 0.0):      # training.py:159-183, 468-479: batch['background_points'] (+ noise, random warp ids)
        if batch.get('background_points') is None:
          raise ValueError("the background loss needs batch['background_points']")
        bp = f32(batch['background_points']).reshape(-1, 3)
        gen = torch.Generator(device='cpu').manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        bid = batch.get('background_ids')        # the reference draws them: random.choice(key, model.warp_embeds, ...); injectable for tests
        if bid is None:     # random.choice(key, model.warp_embeds, ...): uniform over the ids PRESENT in the dataset
          if self.warp_embeds is not None:
            bid = self.warp_embeds[torch.randint(0, self.warp_embeds.numel(), (bp.shape[0],), generator=gen)]
          else:
            bid = torch.randint(0, self.cfg.num_warp_embeds, (bp.shape[0],), generator=gen)
        bid = (bid if isinstance(bid, torch.Tensor) else torch.as_tensor(np.asarray(bid).astype(np.int64))).to(dev).reshape(-1).to(torch.int32).contiguous()
        std = float(objective.get('background_noise_std', 0.0))
        if std != 0.0:
          bp = bp + std * torch.randn(bp.shape, generator=gen).to(dev)
        bp = bp.contiguous()
        keep += [bp, bid]
        ob.background_loss_weight = float(objective['background_loss_weight'])
        ob.background_loss_alpha = float(objective.get('background_loss_alpha', -2.0))
        ob.background_loss_scale = float(objective.get('background_loss_scale', 0.001))
        ob.background_points, ob.background_ids, ob.num_background_points = bp.data_ptr(), bid.data_ptr(), int(bp.shape[0])
    s = stream if stream is not None else torch.cuda.current_stream(dev)
    import torch.distributed as dist
    grouped = dist.is_available() and dist.is_initialized()
    if data_parallel is None:
      data_parallel = grouped and dist.get_world_size() > 1
    elif data_parallel and not grouped:
      raise RuntimeError('data_parallel=True needs an initialised torch.distributed process group')
    clip = grad_max_val > 0.0 or grad_max_norm > 0.0
    deferred = bool(data_parallel or clip)          # Adam runs in nerfds_trainer_apply, after the all-reduce / the clip
    flags = (GRADS_ONLY if (grads_only or deferred) else 0) | (SIGMA_GRAD if sigma_gradient else 0)
    self._last_rays = R
    retries = unattributed = 0
    self._calls += 1

    def attempt(diagnose: bool = False) -> bool:
      """forward + backward (+ all-reduce, clip, Adam) under the current policy; True = the gradient was non-finite and the update was skipped"""
      self._set_numerics(diagnose)
      rc = self._lib.nerfds_trainer_step(self._h, C.byref(rays), target.data_ptr(), C.byref(ex), C.byref(rnd), C.byref(ob) if ob is not None else None,
                                         float(learning_rate), flags, loss, C.c_void_p(s.cuda_stream))
      overflow = rc == -34      # NERFDS_ENONFINITE: the update of this step was skipped (include/nerfds.h)
      if rc != 0 and not overflow:
        raise RuntimeError(f'nerfds_trainer_step failed ({rc}): {(self._lib.nerfds_trainer_last_error(self._h) or b"").decode()}')
      if not overflow and deferred:
        if data_parallel:       # one rank per GPU, each with its own rays: ONE all-reduce of the 6 MB gradient vector (training.py:502)
          with torch.cuda.stream(s):
            allreduce_mean_(self.grads_tensor(), force=True)
        if clip:                # utils.clip_gradients after the pmean (training.py:502-504)
          rc = self._lib.nerfds_trainer_clip_gradients(self._h, float(grad_max_val), float(grad_max_norm), C.c_void_p(s.cuda_stream))
          if rc != 0:
            raise RuntimeError(f'nerfds_trainer_clip_gradients failed ({rc})')
        if not grads_only:
          self.apply_gradients(learning_rate, s)
          overflow = self.nonfinite()      # nerfds_trainer_apply cannot report it (asynchronous): read the device flag back
      return overflow

    def sources() -> int:
      raw = self.overflow_sources()
      if data_parallel:         # every rank must climb the same rung: OR of the masks (as a MAX over one int per bit - RCCL has no bitwise OR)
        bits = torch.tensor([1 if raw & (1 << b) else 0 for b in range(30)], dtype=torch.int32, device=dev)
        dist.all_reduce(bits, op=dist.ReduceOp.MAX)
        raw = sum(int(v) << b for b, v in enumerate(bits.tolist()))
      return raw

    while True:
      # One attempt = forward + backward (+ all-reduce, clip, Adam).  A non-finite gradient skips the update as a whole on the device (parameters,
      # moments and step count untouched).  The reference's fp32 step cannot overflow (training.py:494-508); this one stores f16, so the attempt is
      # DIAGNOSED (which stored array holds the inf: overflow_sources; the coarse NerfMLP's arrays, which the fine level overwrites, by repeating the
      # attempt once with numerics.diagnose) and repeated - same seed, same samples - under a policy without that source (_escalate: the primal loss
      # scale for the primal g, split-bf16 chains / the tangent scale for the second-order terms, the fp32 step for an activation), ending in the
      # fp32 step; only what is non-finite THERE - as it would be in the reference - is raised.  The gradient all-reduce spreads an inf / NaN to
      # every rank and the source masks are OR-ed over the ranks, so all ranks take the same decision.
      if not attempt():
        break
      self._clean_steps = 0
      raw = sources()
      if raw & 127 == 0 and not self.fp32_step and self.cfg.num_fine_samples > 0:
        if not attempt(diagnose=True):      # (the sums of a step are atomics: a borderline overflow need not recur - then this attempt's update stands)
          self.overflow_events.append((self._calls, 0, 'did not recur', 0))
          break
        raw = sources()
      src = raw & 127           # the class bits; raw >> 8: which fp32 array (diagnosis, kept in overflow_events)
      policy = (f'loss_scale_adjust = {self.loss_scale_adjust}, tangent_scale_adjust = {self.tangent_scale_adjust}, split_chains = {self.split_chains}, '
                f'fp32_step = {self.fp32_step}')
      if self.fp32_step or retries >= self.max_overflow_retries:
        # (fp32 step: no f16 storage took part - the gradient of THIS batch at THESE parameters is inf / NaN in fp32 arithmetic, as in the reference)
        raise FloatingPointError(f'non-finite gradient: the Adam update of this step was skipped; source: {overflow_names(src)} (fp32 detail bits {raw >> 8:#x}) [after {retries} '
                                 f'repeated attempts; {policy}]')
      action = self._escalate(src, unattributed)
      unattributed += 1 if src == 0 else 0
      retries += 1
      self.overflow_events.append((self._calls, src, action, raw >> 8) + ((self.nonfinite_leaves(),) if self.record_leaves else ()))
      del self.overflow_events[:-256]
    if not grads_only:
      self._clean_steps += 1
      if self._clean_steps >= self.loss_scale_growth_interval:
        self._relax()
        self._clean_steps = 0
    del keep
    fine, coarse = float(loss[0]), float(loss[1])
    two = self.cfg.num_fine_samples > 0
    stats = {'loss/fine': fine, 'loss/coarse': coarse}
    aux = 0.0
    for k, name in enumerate(('warp_reg', 'back_facing', 'predicted_mask', 'norm')):
      stats[f'loss/{name}/fine'], stats[f'loss/{name}/coarse'] = float(loss[2 + k]), float(loss[6 + k])
      aux += (float(loss[2 + k]) if two else 0.0) + float(loss[6 + k])
    stats['loss/hyper_reg/fine'], stats['loss/hyper_reg/coarse'] = float(loss[10]), float(loss[11])      # training.py:312-321
    aux += (float(loss[10]) if two else 0.0) + float(loss[11])
    stats['loss/background'] = float(loss[12])                                                           # training.py:468-479
    stats['loss/elastic'] = float(loss[13])                                                              # training.py:274-295 (coarse level)
    aux += float(loss[12]) + float(loss[13])
    stats['loss/mask_occlusion_reg/fine'], stats['loss/mask_occlusion_reg/coarse'] = float(loss[14]), float(loss[15])      # training.py:409-417
    aux += (float(loss[14]) if two else 0.0) + float(loss[15])
    stats['loss/total'] = (fine + coarse if two else coarse) + aux
    return stats


def train_step(trainer: Trainer, rng_key, state, batch, scalar_params, **static_flags):
  """``training.train_step`` look-alike (training.py:198-216, 511): returns (state, stats, rng_key, None).  ``state`` carries
  ``extra_params`` (evaluation.TrainState); ``scalar_params`` needs ``learning_rate`` and optionally ``mask_ratio``."""
  lr = float(getattr(scalar_params, 'learning_rate', scalar_params['learning_rate'] if isinstance(scalar_params, dict) else 0.0))
  mr = getattr(scalar_params, 'mask_ratio', scalar_params.get('mask_ratio', 1.0) if isinstance(scalar_params, dict) else 1.0)
  # training.py:228 splits rng_key into (rng_key, fine_key, coarse_key, reg_key) every step: here the key is an integer, the
  # sampling seed of this step is derived from it and the advanced key is returned.
  key = int(np.asarray(rng_key).ravel()[-1]) if rng_key is not None else 0
  key = (key * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
  stats = trainer.step(batch, state.extra_params, lr, mask_ratio=float(mr), t_rand=static_flags.get('t_rand'), u_rand=static_flags.get('u_rand'),
                       seed=key)
  return state, stats, key, None
