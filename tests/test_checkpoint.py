"""flax-msgpack checkpoint reader: hand-assembled byte vectors of the published flax 0.3.4 format
and round trips.  (No reference checkpoint exists in /root/reference: the byte format is 'parity unpinned'.)"""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'nerf-ds_amd'))
from nerfds_amd import checkpoint as ck            # noqa: E402
from nerfds_amd import init_params, nerf_ds_config  # noqa: E402
from nerfds_amd.params import tree_leaves           # noqa: E402


def _ext_ndarray(shape, dtype_name, payload, code=1):
  inner = bytes([0x93, 0x90 | len(shape)]) + bytes(shape) + bytes([0xa0 | len(dtype_name)]) + dtype_name.encode() + \
      bytes([0xc4, len(payload)]) + payload
  return bytes([0xc7, len(inner), code]) + inner


def test_hand_assembled_document():
  w = np.array([[1.5, -2.0, 3.25], [0.0, 7.0, -0.125]], np.float32)
  doc = bytes([0x82]) + b'\xa6kernel' + _ext_ndarray((2, 3), 'float32', w.tobytes()) + \
      b'\xa4step' + _ext_ndarray((), 'int32', struct.pack('<i', 250000), code=3)
  out = ck.msgpack_restore(doc)
  assert out['kernel'].dtype == np.float32 and out['kernel'].shape == (2, 3)
  np.testing.assert_array_equal(out['kernel'], w)
  assert out['step'] == 250000 and isinstance(out['step'], np.integer)
  out['kernel'][0, 0] = 9.0        # restored leaves are writable, owned arrays


def test_writer_emits_the_same_bytes_as_the_hand_assembled_vector():
  w = np.arange(4, dtype=np.float32)
  assert ck.msgpack_serialize({'k': w}) == bytes([0x81]) + b'\xa1k' + _ext_ndarray((4,), 'float32', w.tobytes())


def test_chunked_leaf(monkeypatch):
  monkeypatch.setattr(ck, 'MAX_CHUNK_SIZE', 64)
  a = np.arange(100, dtype=np.float32).reshape(10, 10)
  out = ck.msgpack_restore(ck.msgpack_serialize({'a': a, 'l': [np.float32(1), np.float32(2)]}))
  np.testing.assert_array_equal(out['a'], a)
  assert set(out['l']) == {'0', '1'}          # lists serialise as dicts keyed by index, like flax


def test_round_trip_of_the_nerf_ds_tree(tmp_path):
  cfg = nerf_ds_config(num_warp_embeds=3)
  params = init_params(cfg, 5, bias_scale=0.1)
  extra = dict(nerf_alpha=8.0, warp_alpha=4.0, hyper_alpha=1.0, hyper_sheet_alpha=6.0, norm_input_alpha=4.0, norm_loss_weight=None)
  for step in (9, 10, 2):
    ck.save_checkpoint(str(tmp_path), params, extra, step)
  (tmp_path / 'checkpoint_11tmp').write_bytes(b'garbage')      # an interrupted save is ignored
  assert os.path.basename(ck.latest_checkpoint(str(tmp_path))) == 'checkpoint_10'      # natural, not lexicographic, order
  got, got_extra, step = ck.restore_checkpoint(str(tmp_path))
  assert step == 10
  assert got_extra == {k: v for k, v in extra.items() if v is not None}
  a, b = tree_leaves(params), tree_leaves(got)
  assert [n for n, _ in a] == [n for n, _ in b]
  for (n, x), (_, y) in zip(a, b):
    assert y.dtype == np.float32 and y.flags['C_CONTIGUOUS']
    np.testing.assert_array_equal(np.asarray(x, np.float32), y, err_msg=n)


def test_errors(tmp_path):
  with pytest.raises(FileNotFoundError):
    ck.restore_checkpoint(str(tmp_path))
  p = tmp_path / 'checkpoint_1'
  p.write_bytes(ck.msgpack_serialize({'not': {'a': np.zeros(1, np.float32)}}))
  with pytest.raises(ValueError):
    ck.restore_checkpoint(str(p))


def test_complete_train_state_with_adam_moments(tmp_path):
  """The writer emits every field flax 0.3.4's from_state_dict(TrainState) needs (training.py:59-66): target, OptimizerState.step,
  one {grad_ema, grad_sq_ema} per parameter leaf, all eight schedule scalars (nil when unset)."""
  cfg = nerf_ds_config(num_warp_embeds=3)
  params = init_params(cfg, 5, bias_scale=0.1)
  rng = np.random.default_rng(0)
  ema = {k: v for k, v in params.items()}
  def rand_like(tree):
    return {k: rand_like(v) for k, v in tree.items()} if isinstance(tree, dict) else rng.normal(size=np.asarray(tree).shape).astype(np.float32)
  ema, sq = rand_like(params), rand_like(params)
  path = ck.save_checkpoint(str(tmp_path), params, dict(nerf_alpha=8.0, warp_alpha=4.0), 77, opt_state=(ema, sq))
  raw = ck.msgpack_restore(open(path, 'rb').read())
  assert set(raw) == {'optimizer'} | set(ck.EXTRA_PARAM_KEYS) and raw['hyper_alpha'] is None and float(raw['nerf_alpha']) == 8.0
  assert set(raw['optimizer']) == {'target', 'state'} and set(raw['optimizer']['state']) == {'step', 'param_states'}
  assert raw['optimizer']['state']['step'].shape == () and raw['optimizer']['state']['step'].dtype == np.int32
  ps = raw['optimizer']['state']['param_states']['model']
  names = [n for n, _ in tree_leaves(params)]
  assert [n for n, _ in tree_leaves(ps)] == [f'{n}/{m}' for n in names for m in ('grad_ema', 'grad_sq_ema')]
  got_ema, got_sq, step = ck.restore_optimizer_state(str(tmp_path))
  assert step == 77
  for (n, a), (_, b) in zip(tree_leaves(ema), tree_leaves(got_ema)):
    np.testing.assert_array_equal(a, b, err_msg=n)
  for (n, a), (_, b) in zip(tree_leaves(sq), tree_leaves(got_sq)):
    np.testing.assert_array_equal(a, b, err_msg=n)
  # a fresh optimizer: zeros of every leaf's shape
  ck.save_checkpoint(str(tmp_path), params, {}, 78)
  z, _, _ = ck.restore_optimizer_state(str(tmp_path))
  assert all(not np.any(v) and v.shape == np.asarray(p).shape for (_, v), (_, p) in zip(tree_leaves(z), tree_leaves(params)))
