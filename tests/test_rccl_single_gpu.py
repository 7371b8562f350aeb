"""RCCL on the one GPU a test box has: a world-size-1 "nccl" process group runs the two collectives of the hot path -
the all-gather of per-ray records (render.py:155 / evaluation.py:91-129 in the reference; evaluation.all_gather_into here) under
render_image's sharded chunk loop, and the gradient all-reduce of the training step (training.py:502; training.allreduce_mean_).
One subprocess, because a process group and RCCL's communicator are process-wide state."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.evaluation import TrainState, make_model_fn, render_image
from nerfds_amd.model import NerfModel
from nerfds_amd.training import Trainer
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)           # "nccl" is RCCL on ROCm
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=16, num_fine_samples=16, use_stratified_sampling=False)
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rng = np.random.default_rng(0)
H, W = 23, 41
d = rng.normal(size=(H, W, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
rays = dict(origins=(rng.normal(size=(H, W, 3)) * 0.1).astype(np.float32), directions=d.astype(np.float32), viewdirs=d.astype(np.float32),
            metadata={'warp': rng.integers(0, 4, (H, W, 1))}, mask=(rng.random((H, W, 1)) < 0.3).astype(np.float32))
model = NerfModel(cfg, device='cuda', precision='f32')                          # unindexed device: resolves to cuda:0
state = TrainState.create(params, **EXTRA)
fn = make_model_fn(model, precision='f32')
plain = render_image(state, rays, fn, device_count=1, rng=np.array([0, 3]), chunk=300, cfg=cfg)
os.environ['NERFDS_FORCE_COLLECTIVES'] = '1'
gathered = render_image(state, rays, fn, device_count=1, rng=np.array([0, 3]), chunk=300, cfg=cfg)      # all_gather_into_tensor per chunk
res = {'render_equal': all(bool(torch.equal(plain[k], gathered[k])) for k in ('rgb', 'depth', 'med_points'))}
R = 64
flat = {k: (np.asarray(v).reshape(H * W, -1)[:R] if k != 'metadata' else {'warp': np.asarray(v['warp']).reshape(H * W, 1)[:R]}) for k, v in rays.items()}
flat['rgb'] = rng.random((R, 3)).astype(np.float32)
tr = Trainer(cfg, params, max_rays=R, device='cuda')
a = tr.step(flat, EXTRA, 0.0, grads_only=True, seed=7, data_parallel=False)
ga = tr.grads_tensor().clone()
b = tr.step(flat, EXTRA, 0.0, grads_only=True, seed=7, data_parallel=True)      # all_reduce(SUM) / 1 through RCCL
gb = tr.grads_tensor().clone()
torch.cuda.synchronize()
res['loss_equal'] = abs(a['loss/total'] - b['loss/total']) <= 1e-6 * abs(a['loss/total'])
res['grad_rel_diff'] = float((ga - gb).norm() / ga.norm())                     # float atomics: the two steps differ in the last bits
res['backend'] = dist.get_backend()
res['rccl_loaded'] = any('librccl' in ln for ln in open('/proc/self/maps'))
dist.destroy_process_group()
print('RESULT ' + json.dumps(res))
'''


def _free_port():
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    return s.getsockname()[1]


def test_rccl_all_gather_and_all_reduce_at_world_size_one():
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY='0')
  env.pop('NERFDS_FORCE_COLLECTIVES', None)
  r = subprocess.run([sys.executable, '-c', f'ROOT = {ROOT!r}\n' + WORKER], env=env, capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-3000:]
  line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')]
  assert line, r.stdout[-2000:]
  res = json.loads(line[0][7:])
  assert res['backend'] == 'nccl' and res['rccl_loaded']
  assert res['render_equal']                  # the frame that went through the all-gather equals the frame that did not
  assert res['loss_equal'] and res['grad_rel_diff'] < 1e-5
