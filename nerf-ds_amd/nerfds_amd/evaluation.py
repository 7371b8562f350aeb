"""Chunked, ray-sharded frame rendering: the drop-in for hypernerf/evaluation.py:53-149 (``render_image``)
and for the pmapped ``_model_fn`` of render.py:139-163.

Reference scheme: one host process, ``jax.pmap`` over D local devices, rays of a chunk reshaped to
``[D, R_c / D, ...]`` (utils.shard, utils.py:295-299), an in-pmap ``all_gather`` of the WHOLE two-level output
dict (render.py:155), device->host copy of the fine level, ``unshard`` + drop padding (utils.py:307-312).

Here: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI), each rank renders its
contiguous block of the chunk with the fused HIP kernel, and the only exchange is ONE all-gather of the
``[R_c / D, 26]`` per-ray record tensor (104 B/ray) of the returned level - the path has no other collective.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N


@dataclass
class TrainState:
  """model_utils.TrainState (model_utils.py:28-52) reduced to what the render path reads."""
  optimizer: Any = None
  nerf_alpha: Optional[float] = None
  warp_alpha: Optional[float] = None
  hyper_alpha: Optional[float] = None
  hyper_sheet_alpha: Optional[float] = None
  norm_loss_weight: Optional[float] = None
  norm_input_alpha: Optional[float] = None
  norm_voxel_lr: Optional[float] = None
  norm_voxel_ratio: Optional[float] = None

  @property
  def extra_params(self):
    return {'nerf_alpha': self.nerf_alpha, 'warp_alpha': self.warp_alpha, 'hyper_alpha': self.hyper_alpha,
            'hyper_sheet_alpha': self.hyper_sheet_alpha, 'norm_loss_weight': self.norm_loss_weight,
            'norm_input_alpha': self.norm_input_alpha, 'norm_voxel_lr': self.norm_voxel_lr,
            'norm_voxel_ratio': self.norm_voxel_ratio}

  @classmethod
  def create(cls, params, **alphas):
    return cls(optimizer=SimpleNamespace(target={'model': params}), **alphas)


def encode_metadata(model, params, metadata) -> Dict[str, torch.Tensor]:
  """evaluation.encode_metadata (evaluation.py:29-50): metadata ids - ``[..., 1]``, or ``[..., 3]`` = (left id, right id,
  progression), NerfModel._encode_embed models.py:271-294 - to per-ray GLO vectors, on the GPU (csrc/embed_kernel.hip).  The result goes
  back into ``rays_dict['metadata']`` for ``model.apply(..., metadata_encoded=True)``.  Of the reference's three entries the built graphs
  have ``encoded_warp`` and ``encoded_hyper`` (hyper_use_warp_embed: the same table and the same metadata, models.py:296-319);
  ``use_nerf_embed`` graphs are not built."""
  params = params['params'] if 'params' in params else params
  if params is not getattr(model, '_params_ref', None):
    model.load_params(params)
  enc = {}
  if model.cfg.use_warp:
    enc['encoded_warp'] = model.encode_embed(metadata['warp'], 'warp')
    if model.cfg.has_hyper:
      enc['encoded_hyper'] = enc['encoded_warp']
  return enc


def _world():
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank(), dist.get_world_size()
  return 0, 1


def shard_bounds(num_chunk_rays: int, device_count: int, proc_id: int):
  """Padding (edge replication, evaluation.py:99-107) and this rank's [lo, hi) block of the padded chunk."""
  remainder = num_chunk_rays % device_count
  padding = (device_count - remainder) if remainder else 0
  per = (num_chunk_rays + padding) // device_count
  return padding, proc_id * per, (proc_id + 1) * per


def pad_edge(x: torch.Tensor, padding: int) -> torch.Tensor:
  """jnp.pad(x, ((0, padding), (0, 0)), mode='edge') (evaluation.py:103-105)."""
  if padding == 0:
    return x
  return torch.cat([x, x[-1:].expand(padding, *x.shape[1:])], dim=0)


def tree_map(fn, tree):
  if isinstance(tree, dict):
    return {k: tree_map(fn, v) for k, v in tree.items()}
  return fn(tree)


def all_gather_into(out: torch.Tensor, rec: torch.Tensor) -> None:
  """``out[D * R_local, 26]`` <- every rank's ``rec[R_local, 26]`` in rank order (render.py:155 ``all_gather``), one collective,
  issued on the current stream.  RCCL ("nccl") gathers straight into ``out``; any other backend (gloo: the CPU tests, and the
  two-processes-on-one-GPU tests of the N > 1 code paths) goes through the list API on host copies."""
  world = dist.get_world_size()
  if dist.get_backend() == 'nccl':
    dist.all_gather_into_tensor(out, rec)
    return
  host = rec.contiguous().cpu()
  parts = [torch.empty_like(host) for _ in range(world)]
  dist.all_gather(parts, host)
  out.copy_(torch.cat(parts, dim=0))


def all_gather_records(rec: torch.Tensor) -> torch.Tensor:
  """[R_local, 26] on every rank -> [D * R_local, 26] on every rank (render.py:155 ``all_gather``), one collective."""
  rank, world = _world()
  if world == 1:
    return rec
  rec = rec.contiguous()
  out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
  all_gather_into(out, rec)
  return out


def records_to_dict(rec: torch.Tensor, cfg) -> Dict[str, torch.Tensor]:
  """Splits a [R, 26] record tensor into the reference's per-ray keys (models.py:1312-1415)."""
  o = {}
  for k, (a, n) in N.RAY_FIELDS.items():
    v = rec[:, a:a + n]
    o[k] = v[:, 0] if k in ('depth', 'med_depth', 'acc') else v
  o['med_points'] = o['med_points'][:, None, :3 + cfg.num_hyper_dims]
  o['ray_hyper_points'] = o['ray_hyper_points'][:, :cfg.num_hyper_dims]
  o['ray_hyper_c'] = torch.zeros_like(o['ray_hyper_points'])
  if not cfg.use_warp:
    del o['ray_rotation_field'], o['ray_translation_field']
  if not cfg.use_predicted_mask:
    del o['ray_predicted_mask']
  if not cfg.predict_norm:
    del o['ray_norm']
  return o


def make_model_fn(model, use_predicted_norm: Optional[bool] = None, sharp_weights_std: float = 0.1,
                  precision: Optional[str] = None, render_fn: Optional[Callable] = None,
                  gather_levels=('fine',)) -> Callable:
  """The ``_model_fn`` of render.py:139-155: render the local shard, all-gather the per-ray records.

  ``render_fn(params, rays, extra_params, key) -> (rec_fine [R,26], rec_coarse or None)`` may be injected
  (the gloo CPU tests use a deterministic stand-in so the sharding/gather logic runs without a GPU).
  ``gather_levels``: which levels of a two-level model are exchanged between the ranks (render_image returns 'fine'
  unless ``default_ret_key`` says otherwise; pass ``('coarse',)`` or both for that case).

  The returned callable has the reference's signature.  It also carries ``render_local`` (the same without the
  exchange, optionally writing the records straight into caller-owned buffers) which ``render_image`` uses to
  overlap the exchange of chunk i with the rendering of chunk i + 1.
  """
  cfg = model.cfg if model is not None else None
  upn = (cfg.predict_norm if use_predicted_norm is None else use_predicted_norm) if cfg is not None else False

  def _default_render(params, rays_dict, extra_params, keys, records_out=None):
    model.apply({'params': params}, rays_dict, extra_params, rngs={'coarse': keys[0], 'fine': keys[1]},
                use_predicted_norm=upn, return_points=False, return_nv_details=False, mask_ratio=1,
                sharp_weights_std=sharp_weights_std, precision=precision, records_out=records_out)
    return model.last_records.get('fine'), model.last_records['coarse']

  render = render_fn or _default_render

  def render_local(key_0, key_1, key_2, params, rays_dict, extra_params, records_out=None):
    """{'fine': rec, 'coarse': rec} (single-level model: {'coarse': rec}) of this rank's rays, no exchange."""
    if render_fn is None:
      rec_fine, rec_coarse = render(params, rays_dict, extra_params, (key_0, key_1, key_2), records_out)
    else:
      rec_fine, rec_coarse = render(params, rays_dict, extra_params, (key_0, key_1, key_2))
    if rec_fine is None or rec_coarse is None:        # single-level model: its only record is the 'coarse' level
      return {'coarse': rec_coarse if rec_coarse is not None else rec_fine}
    return {'fine': rec_fine, 'coarse': rec_coarse}

  def model_fn(key_0, key_1, key_2, params, rays_dict, extra_params):
    out = render_local(key_0, key_1, key_2, params, rays_dict, extra_params)
    if 'fine' not in out:
      return {'coarse': all_gather_records(out['coarse']), '_gathered': ('coarse',)}
    # The reference all-gathers both levels and evaluation.py:121-126 then drops 'coarse'.  Here only the levels in
    # `gather_levels` cross xGMI (default: the one render_image returns); the others stay the local shard and
    # render_image refuses to return them.
    out['_gathered'] = tuple(gather_levels)
    for lv in gather_levels:
      out[lv] = all_gather_records(out[lv])
    return out

  model_fn.render_local = render_local
  model_fn.gather_levels = tuple(gather_levels)
  model_fn.on_device = render_fn is None
  model_fn.device = model.device if model is not None else None
  model_fn.levels = (('coarse', 'fine') if cfg.num_fine_samples > 0 else ('coarse',)) if cfg is not None else ('coarse', 'fine')
  return model_fn


def render_image(state, rays_dict, model_fn, device_count, rng, chunk=8192, default_ret_key=None, cfg=None,
                 to_host: bool = True):
  """evaluation.render_image (evaluation.py:53-149).

  ``rays_dict`` leaves have leading shape [H, W]; returns a dict of [H, W, ...] maps of the fine level
  (coarse when there is no fine level).  ``device_count`` is the number of ranks the chunk is sharded over
  and must equal the torch.distributed world size (1 without a process group).

  With a ``make_model_fn`` callable on the GPU the frame is assembled in ONE device buffer: rays and ids are converted
  once, every chunk's local shard is rendered on the compute stream (world 1: straight into the frame buffer), the
  all-gather of chunk i runs on a side stream while chunk i + 1 renders, and the host copy - if asked for - happens
  once at the end (the reference copies every chunk to the host, evaluation.py:126).
  """
  rank, world = _world()
  if device_count != world:
    raise ValueError(f'device_count={device_count} but the process group has {world} ranks')
  as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a))
  rays_dict = tree_map(as_t, rays_dict)
  batch_shape = tuple(rays_dict['origins'].shape[:-1])
  num_rays = int(np.prod(batch_shape))
  rays_dict = tree_map(lambda x: x.reshape(num_rays, -1), rays_dict)
  seed = int(np.asarray(rng).ravel()[-1]) if rng is not None else 0
  key_0, key_1, key_2 = (seed * 4 + 1) * world + rank, (seed * 4 + 2) * world + rank, (seed * 4 + 3) * world + rank
  params = state.optimizer.target['model']
  num_batches = int(math.ceil(num_rays / chunk))
  fast = getattr(model_fn, 'on_device', False) and hasattr(model_fn, 'render_local')
  if fast:
    rec = _render_image_device(state, rays_dict, model_fn, params, (key_0, key_1, key_2), num_rays, num_batches, chunk,
                               device_count, rank, default_ret_key)
    if to_host:
      rec = rec.cpu()
  else:
    ret_chunks = []
    for batch_idx in range(num_batches):
      ray_idx = batch_idx * chunk
      chunk_rays = tree_map(lambda x: x[ray_idx:ray_idx + chunk], rays_dict)
      num_chunk_rays = chunk_rays['origins'].shape[0]
      padding, lo, hi = shard_bounds(num_chunk_rays, device_count, rank)
      chunk_rays = tree_map(lambda x: pad_edge(x, padding)[lo:hi], chunk_rays)         # evaluation.py:99-118
      model_out = model_fn(key_0, key_1, key_2, params, chunk_rays, state.extra_params)   # evaluation.py:119
      ret_key = default_ret_key or ('fine' if 'fine' in model_out else 'coarse')          # evaluation.py:121-124
      if world > 1 and ret_key not in model_out.get('_gathered', (ret_key,)):
        raise ValueError(f"level '{ret_key}' was not all-gathered by this model_fn: build it with make_model_fn(..., gather_levels=('{ret_key}',))")
      rec = model_out[ret_key]
      if padding:
        rec = rec[:-padding]                                                               # utils.unshard (utils.py:307-312)
      ret_chunks.append(rec.cpu() if to_host else rec)                                     # evaluation.py:126
    rec = torch.cat(ret_chunks, dim=0)
  if cfg is None:
    return {'records': rec.reshape(*batch_shape, rec.shape[-1])}
  out = records_to_dict(rec, cfg)
  return {k: v.reshape(*batch_shape, *v.shape[1:]) for k, v in out.items()}             # evaluation.py:143-147


def _render_image_device(state, rays_dict, model_fn, params, keys, num_rays, num_batches, chunk, device_count, rank,
                         default_ret_key):
  """The chunk loop of render_image on the GPU: [num_rays, 26] records of the returned level, on the device."""
  dev = model_fn.device
  world = device_count
  ret_key = default_ret_key or model_fn.levels[-1]                                        # evaluation.py:121-124
  if ret_key not in model_fn.levels:
    raise KeyError(ret_key)
  if world > 1 and len(model_fn.levels) > 1 and ret_key not in model_fn.gather_levels:
    raise ValueError(f"level '{ret_key}' was not all-gathered by this model_fn: build it with make_model_fn(..., gather_levels=('{ret_key}',))")

  # one conversion for the whole frame: the per-chunk slices below are contiguous views, so NerfModel.apply launches no
  # conversion kernels per chunk
  def to_dev(x):
    return x.to(dev, torch.int32 if not x.dtype.is_floating_point else torch.float32).contiguous()
  rays_dict = tree_map(to_dev, rays_dict)
  frame = torch.empty((num_rays, N.RAY_REC), dtype=torch.float32, device=dev)
  compute = torch.cuda.current_stream(dev)
  # NERFDS_FORCE_COLLECTIVES=1 (tests): a one-rank process group takes the sharded path below - staging buffers, side stream and the
  # all-gather itself - so that the RCCL branch of all_gather_into executes on a one-GPU box
  forced = os.environ.get('NERFDS_FORCE_COLLECTIVES') == '1' and dist.is_available() and dist.is_initialized()
  if world == 1 and not forced:
    for batch_idx in range(num_batches):
      ray_idx = batch_idx * chunk
      chunk_rays = tree_map(lambda x: x[ray_idx:ray_idx + chunk], rays_dict)
      n = chunk_rays['origins'].shape[0]
      model_fn.render_local(*keys, params, chunk_rays, state.extra_params, records_out={ret_key: frame[ray_idx:ray_idx + n]})
    return frame
  # world > 1: the local shard of chunk i is rendered into one of two staging buffers on the compute stream; its exchange
  # (ONE all-gather of [R_c / D, 26], render.py:155) runs on a side stream while chunk i + 1 renders into the other buffer
  comm = torch.cuda.Stream(dev)
  per_max = (min(chunk, num_rays) + world - 1) // world
  staging = [torch.empty((per_max, N.RAY_REC), dtype=torch.float32, device=dev) for _ in range(2)]
  # everything the loop needs is allocated once: two staging buffers, two (ready, done) event pairs, one buffer for the gather of a padded
  # chunk (only the frame's last chunk can be ragged) - the loop itself allocates nothing, whatever the number of chunks or ranks
  ready_ev = [torch.cuda.Event() for _ in range(2)]
  done_ev = [torch.cuda.Event() for _ in range(2)]
  padded = None
  done = [None, None]
  for batch_idx in range(num_batches):
    ray_idx = batch_idx * chunk
    chunk_rays = tree_map(lambda x: x[ray_idx:ray_idx + chunk], rays_dict)
    n = chunk_rays['origins'].shape[0]
    padding, lo, hi = shard_bounds(n, device_count, rank)
    if padding:
      chunk_rays = tree_map(lambda x: pad_edge(x, padding), chunk_rays)                 # evaluation.py:99-107
    local = tree_map(lambda x: x[lo:hi], chunk_rays)
    slot, per = batch_idx & 1, hi - lo
    if done[slot] is not None:
      compute.wait_event(done[slot])                         # the exchange of chunk i - 2 has read this buffer
    rec = staging[slot][:per]
    model_fn.render_local(*keys, params, local, state.extra_params, records_out={ret_key: rec})
    ready = ready_ev[slot]
    ready.record(compute)
    with torch.cuda.stream(comm):
      comm.wait_event(ready)
      if padding == 0:
        all_gather_into(frame[ray_idx:ray_idx + n], rec)
      else:
        if padded is None or padded.shape[0] < world * per:
          padded = torch.empty((world * per_max, N.RAY_REC), dtype=torch.float32, device=dev)
        full = padded[:world * per]
        all_gather_into(full, rec)
        frame[ray_idx:ray_idx + n].copy_(full[:n])                                       # utils.unshard: drop the padding
      done[slot] = done_ev[slot]
      done[slot].record(comm)
  compute.wait_stream(comm)
  return frame


render_image_on_rays = render_image   # the name BASELINE.json's north_star uses for the same surface
