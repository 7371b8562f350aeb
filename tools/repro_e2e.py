"""Reproduces tests/test_end_to_end.py's training loop N times (different student seeds / batch seeds) and logs, per step, the loss and
the trainer's loss_scale_adjust, plus every overflow event - the round-5 driver failure (FloatingPointError after 8 retries) hunted down.
  python tools/repro_e2e.py [--runs 20] [--steps 60] [--rays 1024]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))

EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--runs', type=int, default=20)
  ap.add_argument('--steps', type=int, default=60)
  ap.add_argument('--rays', type=int, default=1024)
  ap.add_argument('--lr', type=float, default=2e-3)
  ap.add_argument('--nc', type=int, default=16)
  ap.add_argument('--nf', type=int, default=16)
  a = ap.parse_args()
  import torch
  from nerfds_amd import init_params, nerf_ds_config
  from nerfds_amd.camera import Camera, camera_to_rays
  from nerfds_amd.frames import render_frame
  from nerfds_amd.model import NerfModel
  from nerfds_amd.sched import build
  from nerfds_amd.training import Trainer
  dev = torch.device('cuda', 0)
  cam = Camera.from_json(os.path.join(ROOT, 'tests', 'golden', 'reference_testdata_camera.json')).scale(0.02)
  H, W = cam.image_shape
  cfg = nerf_ds_config(num_warp_embeds=2, num_coarse_samples=a.nc, num_fine_samples=a.nf, use_stratified_sampling=False)
  teacher = init_params(cfg, 11, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  model = NerfModel(cfg, device=dev, precision='f32')
  _, _, rec_t = render_frame(model, {'params': teacher}, cam, 1, EX, want_debug=False)
  target = rec_t[:, 0:3].contiguous()
  rays = camera_to_rays(cam, dev)
  o, d = rays['origins'].reshape(-1, 3), rays['directions'].reshape(-1, 3)
  obj = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, norm_loss_weight=0.01)
  B = a.rays
  fails = 0
  for run in range(a.runs):
    student = init_params(cfg, 12 + (run // 2), warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
    tr = Trainer(cfg, student, max_rays=B, device=dev)
    tr.record_leaves = True
    lr = build({'type': 'exponential', 'initial_value': a.lr, 'final_value': a.lr / 10, 'num_steps': a.steps})
    gen = torch.Generator(device='cpu').manual_seed(run % 2)
    log = []
    err = None
    for step in range(a.steps):
      idx = torch.randint(0, H * W, (B,), generator=gen).to(dev)
      batch = dict(origins=o[idx], directions=d[idx], viewdirs=d[idx], metadata={'warp': torch.ones((B, 1), dtype=torch.int32)},
                   mask=torch.zeros((B, 1)), rgb=target[idx])
      try:
        stats = tr.step(batch, EX, lr(step), objective=obj, grad_max_norm=10.0)
      except FloatingPointError as e:
        err = f'step {step}: {e}'
        break
      log.append((step, round(stats['loss/total'], 6)))
    events = list(tr.overflow_events)
    print(json.dumps({'run': run, 'first': log[0][1] if log else None, 'last': log[-1][1] if log else None, 'final_policy': [tr.loss_scale_adjust, tr.tangent_scale_adjust, tr.split_chains, tr.fp32_step],
                      'n_events': len(events), 'events': events[:3], 'error': err}), flush=True)
    fails += err is not None
    del tr
  print(f'{fails} of {a.runs} runs raised')
  return 1 if fails else 0


if __name__ == '__main__':
  sys.exit(main())
