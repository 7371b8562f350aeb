"""Teacher / student run of the trainer at BASELINE config-4 sizes (4096 random rays per step of a teacher frame, 64+64 samples,
nerf_ds graph, rgb loss + the regularisers of configs/nerf_ds.gin): prints one JSON line with the loss every 25 steps and the
held-out frame MSE before / after."""
import json, os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))
import torch
from nerfds_amd import init_params, nerf_ds_config
from nerfds_amd.camera import Camera, camera_to_rays
from nerfds_amd.frames import render_frame
from nerfds_amd.model import NerfModel
from nerfds_amd.sched import build
from nerfds_amd.training import Trainer

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
dev = torch.device('cuda', 0)
cam = Camera.from_json(os.path.join(ROOT, 'tests', 'golden', 'reference_testdata_camera.json')).scale(0.1)
H, W = cam.image_shape
cfg = nerf_ds_config(num_warp_embeds=2, use_stratified_sampling=True)
teacher = init_params(cfg, 11, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
model = NerfModel(cfg, device=dev, precision='f32')
_, _, rec_t = render_frame(model, {'params': teacher}, cam, 1, EX, want_debug=False)
target = rec_t[:, 0:3].contiguous()
rays = camera_to_rays(cam, dev)
o, d = rays['origins'].reshape(-1, 3), rays['directions'].reshape(-1, 3)
B = 4096
lr = build({'type': 'exponential', 'initial_value': 1e-3, 'final_value': 1e-4, 'num_steps': STEPS})
obj = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1)
out = {'config': f'{W}x{H} teacher frame, {B} rays/step, 64+64 samples, {STEPS} steps, lr 1e-3 -> 1e-4, grad clip 10', 'modes': {}}
for mode in ('mfma',):
  student = init_params(cfg, 12, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  mse = lambda p: float(((render_frame(model, {'params': p}, cam, 1, EX, want_debug=False)[2][:, 0:3] - target) ** 2).mean())
  before = mse(student)
  tr = Trainer(cfg, student, max_rays=B, device=dev)
  gen = torch.Generator(device='cpu').manual_seed(0)
  curve = []
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for step in range(STEPS):
    idx = torch.randint(0, H * W, (B,), generator=gen).to(dev)
    batch = dict(origins=o[idx], directions=d[idx], viewdirs=d[idx], metadata={'warp': torch.ones((B, 1), dtype=torch.int32)},
                 mask=torch.zeros((B, 1)), rgb=target[idx])
    t_rand, u_rand = torch.rand((B, 64), generator=gen).to(dev), torch.rand((B, 64), generator=gen).to(dev)     # stratified samples
    stats = tr.step(batch, EX, lr(step), t_rand=t_rand, u_rand=u_rand, objective=obj, grad_max_norm=10.0)
    if step % 25 == 0 or step == STEPS - 1: curve.append((step, round(stats['loss/total'], 6)))
  torch.cuda.synchronize(); dt = time.perf_counter() - t0
  out['modes'][mode] = {'frame_mse_before': before, 'frame_mse_after': mse(tr.get_params()), 'loss': curve, 'seconds': round(dt, 2)}
print(json.dumps(out))
