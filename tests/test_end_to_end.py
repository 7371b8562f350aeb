"""Whole-stack integration on the GPU: teacher frame (camera -> rays -> fused render) -> training steps on random-ray batches of that
frame (trainer: full nerf_ds.gin objective) -> checkpoint written / re-read in the reference's flax-msgpack format -> student rendered
by the fused kernel -> uint8 frames.  Checks that the pieces compose, not image quality."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))

EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)


@pytest.mark.gpu
def test_train_checkpoint_render_round_trip(tmp_path):
  import torch
  from nerfds_amd import checkpoint as ck
  from nerfds_amd import init_params, nerf_ds_config
  from nerfds_amd.camera import Camera, camera_to_rays
  from nerfds_amd.evaluation import TrainState
  from nerfds_amd.frames import render_frame
  from nerfds_amd.model import NerfModel
  from nerfds_amd.sched import build
  from nerfds_amd.training import Trainer, train_step
  dev = torch.device('cuda', 0)
  cam = Camera.from_json(os.path.join(ROOT, 'tests', 'golden', 'reference_testdata_camera.json')).scale(0.02)     # 65 x 49 pixels
  H, W = cam.image_shape
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=16, num_fine_samples=16, use_stratified_sampling=False)
  teacher = init_params(cfg, 11, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  student = init_params(cfg, 12, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  model = NerfModel(cfg, device=dev, precision='f32')
  _, _, rec_t = render_frame(model, {'params': teacher}, cam, 1, EX, want_debug=False)
  target = rec_t[:, 0:3].contiguous()                                   # the teacher's rgb is the training target
  rays = camera_to_rays(cam, dev)
  o, d = rays['origins'].reshape(-1, 3), rays['directions'].reshape(-1, 3)

  def mse_of(params):
    _, _, rec = render_frame(model, {'params': params}, cam, 1, EX, want_debug=False)
    return float(((rec[:, 0:3] - target) ** 2).mean())

  before = mse_of(student)
  B = 1024
  tr = Trainer(cfg, student, max_rays=B, device=dev)
  lr = build({'type': 'exponential', 'initial_value': 2e-3, 'final_value': 2e-4, 'num_steps': 60})
  state = TrainState.create(student, **EX)
  obj = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, norm_loss_weight=0.01)
  gen = torch.Generator(device='cpu').manual_seed(0)
  first = last = None
  for step in range(60):
    idx = torch.randint(0, H * W, (B,), generator=gen).to(dev)
    batch = dict(origins=o[idx], directions=d[idx], viewdirs=d[idx], metadata={'warp': torch.ones((B, 1), dtype=torch.int32)},
                 mask=torch.zeros((B, 1)), rgb=target[idx])
    stats = tr.step(batch, state.extra_params, lr(step), objective=obj, grad_max_norm=10.0)
    first = stats['loss/total'] if first is None else first
    last = stats['loss/total']
  assert last < 0.6 * first, (first, last)
  # train_step look-alike keeps the reference's call shape
  state, stats, _, _ = train_step(tr, None, state, batch, {'learning_rate': 1e-4})
  assert set(stats) >= {'loss/fine', 'loss/coarse', 'loss/total'}
  # checkpoint round trip in the reference's format, then render through the fused kernel
  trained = tr.get_params()
  path = ck.save_checkpoint(str(tmp_path), trained, EX, 61)
  restored, extra, step = ck.restore_checkpoint(str(tmp_path))
  assert step == 61 and extra == {k: float(v) for k, v in EX.items()} and os.path.basename(path) == 'checkpoint_61'
  after = mse_of(restored)
  assert after < 0.7 * before, (before, after)
  rgb, dbg, _ = render_frame(model, {'params': restored}, cam, 1, extra, colormap='sinebow')
  assert rgb.shape == (H, W, 3) and dbg.shape == (2 * H, 3 * W, 3) and rgb.dtype == torch.uint8
