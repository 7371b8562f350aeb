"""World-size-2 CPU (gloo) tests of the ray-sharding / all-gather logic of evaluation.render_image.

The fused kernel needs a GPU, so a deterministic stand-in for the per-rank render (a function of the rays only) is
injected; what is under test is the host logic the reference implements in evaluation.py:53-149 + utils.shard /
unshard: chunking, edge padding to a multiple of the device count, contiguous per-rank blocks, one all-gather per
chunk, dropping the padding, and the [H, W, ...] reshape.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nerfds_amd import nerf_ds_config
from nerfds_amd.evaluation import TrainState, make_model_fn, render_image, shard_bounds, pad_edge


def _fake_render(params, rays_dict, extra_params, keys):
  """[R, 26] 'records' that depend only on the ray: column k = sum(origin) * (k + 1) + direction_x + warp id."""
  o, d = rays_dict['origins'].double(), rays_dict['directions'].double()
  ids = rays_dict['metadata']['warp'].double().reshape(-1, 1)
  base = o.sum(-1, keepdim=True) * torch.arange(1, 27).double() + d[:, :1] + ids
  rec_fine = (base * params['scale']).float()
  return rec_fine, (rec_fine * 0.5)


def _rays(H, W):
  rng = np.random.default_rng(0)
  return dict(origins=rng.normal(size=(H, W, 3)).astype(np.float32), directions=rng.normal(size=(H, W, 3)).astype(np.float32),
              metadata={'warp': rng.integers(0, 5, (H, W, 1))}, mask=np.zeros((H, W, 1), np.float32))


def _expected(H, W):
  r = _rays(H, W)
  flat = {k: (torch.as_tensor(v).reshape(H * W, -1) if k != 'metadata' else {'warp': torch.as_tensor(v['warp']).reshape(H * W, 1)})
          for k, v in r.items()}
  return _fake_render({'scale': 2.0}, flat, None, None)[0].reshape(H, W, 26)


def _worker(rank, world, port, H, W, chunk, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    cfg = nerf_ds_config()
    state = TrainState.create({'scale': 2.0}, nerf_alpha=8.0, warp_alpha=4.0)
    model_fn = make_model_fn(None, render_fn=_fake_render)
    out = render_image(state, _rays(H, W), model_fn, device_count=world, rng=np.array([0, 7]), chunk=chunk, cfg=cfg)
    q.put((rank, {k: v.numpy() for k, v in out.items()}))
  finally:
    dist.destroy_process_group()


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


@pytest.mark.parametrize('H,W,chunk', [(5, 7, 16), (4, 4, 5), (3, 3, 100)])
def test_render_image_two_ranks_matches_single_process(H, W, chunk):
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, H, W, chunk, q)) for r in range(2)]
  for p in procs:
    p.start()
  results = dict(q.get(timeout=120) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  exp = _expected(H, W).numpy()
  for rank in (0, 1):                     # every rank ends up with the whole frame (all-gather, render.py:155)
    out = results[rank]
    assert out['rgb'].shape == (H, W, 3) and out['med_points'].shape == (H, W, 1, 5) and out['depth'].shape == (H, W)
    assert np.array_equal(out['rgb'], exp[..., 0:3])
    assert np.array_equal(out['depth'], exp[..., 3])
    assert np.array_equal(out['ray_predicted_mask'], exp[..., 20:21])
    assert np.array_equal(out['med_points'][..., 0, :], exp[..., 21:26])
    assert np.all(out['ray_hyper_c'] == 0)


def test_single_process_render_image_and_helpers():
  cfg = nerf_ds_config()
  state = TrainState.create({'scale': 2.0})
  out = render_image(state, _rays(3, 5), make_model_fn(None, render_fn=_fake_render), device_count=1, rng=None, chunk=4, cfg=cfg)
  assert np.array_equal(out['rgb'].numpy(), _expected(3, 5).numpy()[..., :3])
  with pytest.raises(ValueError):
    render_image(state, _rays(3, 5), make_model_fn(None, render_fn=_fake_render), device_count=2, rng=None, chunk=4, cfg=cfg)
  # evaluation.py:99-118: padding to a multiple of the device count, contiguous blocks per rank
  assert shard_bounds(10, 4, 0) == (2, 0, 3) and shard_bounds(10, 4, 3) == (2, 9, 12) and shard_bounds(8, 4, 1) == (0, 2, 4)
  x = torch.arange(6.).reshape(3, 2)
  assert torch.equal(pad_edge(x, 2), torch.cat([x, x[-1:], x[-1:]]))            # mode='edge'
  assert state.extra_params['nerf_alpha'] is None and 'norm_input_alpha' in state.extra_params


def _grad_worker(rank, world, port, q):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from nerfds_amd.training import allreduce_mean_
    g = torch.arange(10, dtype=torch.float32) * (rank + 1)          # this rank's "gradient vector"
    allreduce_mean_(g)
    q.put((rank, g.numpy()))
  finally:
    dist.destroy_process_group()


def test_gradient_mean_over_two_ranks():
  """training.py:502 jax.lax.pmean(grad): the data-parallel training step all-reduces the flat gradient vector."""
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = dict(q.get(timeout=120) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
  want = np.arange(10, dtype=np.float32) * 1.5
  np.testing.assert_allclose(got[0], want)
  np.testing.assert_allclose(got[1], want)
  from nerfds_amd.training import allreduce_mean_
  t = torch.ones(3)
  assert allreduce_mean_(t) is t and float(t.sum()) == 3.0           # no process group: identity
