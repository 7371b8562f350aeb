"""Reads the reference's checkpoints (``checkpoint_<step>`` files written by ``training.save_checkpoint``,
training.py:59-66, restored at render.py:128 / eval.py / train.py through ``flax.training.checkpoints``) into the
numpy parameter tree + ``extra_params`` that ``NerfModel.apply`` / ``TrainState`` take.

The file format belongs to a third-party dependency that is not under /root/reference: ``flax==0.3.4``
(requirements.txt:6), ``flax.serialization.to_bytes``: ONE msgpack document of the state dict, where

* every ndarray is ``ExtType(1, packb((shape, dtype.name, arr.tobytes('C'))))``, numpy scalars ``ExtType(3, same)``,
  python complex ``ExtType(2, packb((re, im)))``;
* arrays above 2**30 bytes are split into ``{'__msgpack_chunked_array__': True, 'shape': {'0': d0, ...}, 'chunks': {'0': ...}}``;
* dataclasses serialise field by field, lists/tuples as dicts keyed by the decimal index.

The state dict of ``model_utils.TrainState`` (model_utils.py:24-52) is
``{'optimizer': {'target': {'model': <params>}, 'state': {'step': ..., 'param_states': ...}}, 'nerf_alpha': ..., ...}``.
The optimizer is ``flax.optim.Adam`` (train.py:297-301): ``state`` = ``OptimizerState(step, param_states)`` with one
``_AdamParamState(grad_ema, grad_sq_ema)`` per parameter leaf (flax/optim/adam.py of 0.3.4), i.e. ``param_states`` mirrors ``target``
with ``{'grad_ema': ..., 'grad_sq_ema': ...}`` at every leaf.  ``save_checkpoint`` writes the COMPLETE state dict - every TrainState field,
``None`` schedule scalars as nil, the Adam moments (zeros when none are given) - which is what ``from_state_dict`` needs to restore into a
``TrainState`` target (training.py:59-66 / train.py:335-338 resume).
PARITY UNPINNED for the byte format: no checkpoint file ships with the reference and flax cannot be imported here, so
reader and writer are pinned only by the published format above (hand-assembled byte vectors in tests/test_checkpoint.py) and by
round trips.
"""
from __future__ import annotations

import os
import re
from typing import Any, Dict, Optional, Tuple

import msgpack
import numpy as np

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3
_CHUNK_KEY = '__msgpack_chunked_array__'
MAX_CHUNK_SIZE = 2 ** 30

EXTRA_PARAM_KEYS = ('nerf_alpha', 'warp_alpha', 'hyper_alpha', 'hyper_sheet_alpha', 'norm_loss_weight',
                    'norm_input_alpha', 'norm_voxel_lr', 'norm_voxel_ratio')      # model_utils.py:44-52


def _dtype_from_name(name: str):
  if name == 'bfloat16':
    raise ValueError('bfloat16 leaves are not produced by the reference (all parameters are float32)')
  return np.dtype(name)


def _ndarray_from_bytes(data: bytes) -> np.ndarray:
  shape, dtype_name, buf = msgpack.unpackb(data, raw=True)
  arr = np.frombuffer(buf, dtype=_dtype_from_name(dtype_name.decode())).reshape(tuple(shape), order='C')
  return arr.copy()            # own the memory (frombuffer views are read-only)


def _ext_unpack(code: int, data: bytes):
  if code == _EXT_NDARRAY:
    return _ndarray_from_bytes(data)
  if code == _EXT_NPSCALAR:
    return _ndarray_from_bytes(data)[()]
  if code == _EXT_COMPLEX:
    re_, im_ = msgpack.unpackb(data)
    return complex(re_, im_)
  return msgpack.ExtType(code, data)


def _unchunk(tree):
  if isinstance(tree, dict):
    if tree.get(_CHUNK_KEY):
      chunks = tree['chunks']
      flat = np.concatenate([np.asarray(chunks[str(i)]).ravel() for i in range(len(chunks))])
      shape = tree['shape']
      shape = tuple(shape[str(i)] for i in range(len(shape))) if isinstance(shape, dict) else tuple(shape)
      return flat.reshape(shape)
    return {k: _unchunk(v) for k, v in tree.items()}
  return tree


def msgpack_restore(encoded: bytes) -> Dict[str, Any]:
  """``flax.serialization.msgpack_restore``: bytes -> nested dict of numpy arrays / scalars."""
  return _unchunk(msgpack.unpackb(encoded, ext_hook=_ext_unpack, raw=False, strict_map_key=False))


def _ndarray_to_bytes(arr: np.ndarray) -> bytes:
  arr = np.asarray(arr)
  if arr.dtype.hasobject:
    raise ValueError('object arrays cannot be serialised')
  return msgpack.packb((arr.shape, arr.dtype.name, arr.tobytes('C')), use_bin_type=True)


def _ext_pack(x):
  if isinstance(x, np.ndarray):
    return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(x))
  if isinstance(x, np.generic):
    return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_to_bytes(np.asarray(x)))
  if isinstance(x, complex):
    return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((x.real, x.imag)))
  return x


def _chunk(tree):
  if isinstance(tree, dict):
    return {k: _chunk(v) for k, v in tree.items()}
  if isinstance(tree, (list, tuple)):
    return {str(i): _chunk(v) for i, v in enumerate(tree)}
  if isinstance(tree, np.ndarray) and tree.nbytes > MAX_CHUNK_SIZE:
    per = max(1, MAX_CHUNK_SIZE // tree.dtype.itemsize)
    flat = tree.ravel()
    return {_CHUNK_KEY: True, 'shape': {str(i): int(d) for i, d in enumerate(tree.shape)},
            'chunks': {str(i): flat[o:o + per] for i, o in enumerate(range(0, flat.size, per))}}
  return tree


def msgpack_serialize(tree: Dict[str, Any]) -> bytes:
  """``flax.serialization.msgpack_serialize`` for nested dicts of numpy arrays / python scalars."""
  return msgpack.packb(_chunk(tree), default=_ext_pack, strict_types=True, use_bin_type=True)


# ---- checkpoint directory handling (flax.training.checkpoints) ------------------------------------------------

def _natural_key(name: str):
  return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', name)]


def latest_checkpoint(ckpt_dir: str, prefix: str = 'checkpoint_') -> Optional[str]:
  """The file ``restore_checkpoint(ckpt_dir, ...)`` would pick: highest step in natural order, tmp files ignored."""
  if os.path.isfile(ckpt_dir):
    return ckpt_dir
  if not os.path.isdir(ckpt_dir):
    return None
  names = [n for n in os.listdir(ckpt_dir) if n.startswith(prefix) and not n.endswith('tmp')]
  if not names:
    return None
  return os.path.join(ckpt_dir, sorted(names, key=_natural_key)[-1])


def _to_float_tree(tree):
  if isinstance(tree, dict):
    return {k: _to_float_tree(v) for k, v in tree.items()}
  return np.ascontiguousarray(np.asarray(tree), dtype=np.float32)


def restore_checkpoint(path: str) -> Tuple[Dict[str, Any], Dict[str, float], int]:
  """Returns (params['model'] tree as float32 numpy, extra_params, step) of the newest checkpoint under ``path``.

  Mirrors what render.py:117-130 keeps of the restored state: ``state.optimizer.target['model']`` and
  ``state.extra_params`` (unset / None schedule values are dropped, as ``TrainState.create`` expects)."""
  f = latest_checkpoint(path)
  if f is None:
    raise FileNotFoundError(f'no checkpoint_* file under {path!r}')
  with open(f, 'rb') as fh:
    state = msgpack_restore(fh.read())
  try:
    target = state['optimizer']['target']
  except (KeyError, TypeError) as e:
    raise ValueError(f'{f}: not a TrainState checkpoint (no optimizer/target)') from e
  params = target['model'] if 'model' in target else target
  extra = {}
  for k in EXTRA_PARAM_KEYS:
    v = state.get(k)
    if v is not None:
      extra[k] = float(np.asarray(v))
  step = int(np.asarray(state['optimizer'].get('state', {}).get('step', 0)))
  return _to_float_tree(params), extra, step


def restore_optimizer_state(path: str) -> Optional[Tuple[Dict[str, Any], Dict[str, Any], int]]:
  """(grad_ema tree, grad_sq_ema tree, step) of the Adam state in the newest checkpoint under ``path`` (trees shaped like
  params['model']), or None if the file carries no ``param_states`` (a parameters-only file)."""
  f = latest_checkpoint(path)
  if f is None:
    raise FileNotFoundError(f'no checkpoint_* file under {path!r}')
  with open(f, 'rb') as fh:
    state = msgpack_restore(fh.read())
  st = state.get('optimizer', {}).get('state', {})
  ps = st.get('param_states')
  if ps is None:
    return None
  ps = ps['model'] if 'model' in ps else ps

  def pick(tree, key):
    if isinstance(tree, dict) and set(tree) == {'grad_ema', 'grad_sq_ema'} and not isinstance(tree['grad_ema'], dict):
      return np.ascontiguousarray(np.asarray(tree[key]), dtype=np.float32)
    return {k: pick(v, key) for k, v in tree.items()}
  return pick(ps, 'grad_ema'), pick(ps, 'grad_sq_ema'), int(np.asarray(st.get('step', 0)))


def _param_states(params, grad_ema, grad_sq_ema):
  if isinstance(params, dict):
    return {k: _param_states(v, None if grad_ema is None else grad_ema[k], None if grad_sq_ema is None else grad_sq_ema[k]) for k, v in params.items()}
  p = np.asarray(params)
  z = lambda a: np.zeros(p.shape, np.float32) if a is None else np.ascontiguousarray(np.asarray(a, np.float32).reshape(p.shape))
  return {'grad_ema': z(grad_ema), 'grad_sq_ema': z(grad_sq_ema)}


def save_checkpoint(ckpt_dir: str, params_model: Dict[str, Any], extra_params: Dict[str, float], step: int,
                    opt_state: Optional[Tuple[Dict[str, Any], Dict[str, Any]]] = None) -> str:
  """``training.save_checkpoint`` (training.py:59-66): writes ``checkpoint_<step>`` holding the complete TrainState state dict of flax
  0.3.4 - parameters, Adam ``step`` and ``param_states`` (``opt_state`` = (grad_ema tree, grad_sq_ema tree), e.g. ``Trainer.get_opt_state()``;
  zeros, a freshly created optimizer, when None), and all eight schedule scalars (nil when unset)."""
  os.makedirs(ckpt_dir, exist_ok=True)
  ema, sq = opt_state if opt_state is not None else (None, None)
  state = {'optimizer': {'target': {'model': params_model},
                         'state': {'step': np.asarray(step, np.int32), 'param_states': {'model': _param_states(params_model, ema, sq)}}}}
  for k in EXTRA_PARAM_KEYS:
    state[k] = np.float32(extra_params[k]) if extra_params.get(k) is not None else None
  path = os.path.join(ckpt_dir, f'checkpoint_{int(step)}')
  tmp = path + 'tmp'
  with open(tmp, 'wb') as fh:
    fh.write(msgpack_serialize(state))
  os.replace(tmp, path)
  return path
