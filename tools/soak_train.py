"""GPU: a few hundred 4096-ray steps under the full configs/nerf_ds.gin objective on one fixed synthetic batch set (development evidence, not the bench):
the loss terms at the start and the end, the loss-scale adjustments and skipped updates along the way.  usage: python tools/soak_train.py [steps]"""
import sys, time, numpy as np, torch
sys.path.insert(0, 'nerf-ds_amd'); sys.path.insert(0, '.')
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.training import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
R = 4096
cfg = nerf_ds_config(num_warp_embeds=64, num_coarse_samples=64, num_fine_samples=64, near=0.3, far=1.7)
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rng = np.random.default_rng(0)
def make_batch():
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  o = rng.normal(size=(R, 3)) * 0.2
  # a smooth synthetic target: colour as a function of the ray, so there is something to fit
  rgb = 0.5 + 0.5 * np.sin(3.0 * d + o)
  b = dict(origins=o, directions=d, viewdirs=d, metadata={'warp': rng.integers(0, 64, (R, 1))}, mask=(rng.random((R, 1)) < 0.3).astype(np.float32), rgb=rgb)
  return {k: (torch.as_tensor(np.asarray(v), dtype=torch.float32).cuda() if k != 'metadata' else {'warp': torch.as_tensor(v['warp']).cuda()}) for k, v in b.items()}
batches = [make_batch() for _ in range(8)]
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
ob = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1, norm_loss_weight=0.001)
tr = Trainer(cfg, params, max_rays=R)
log, adj = [], []
t0 = time.time()
for i in range(steps):
  s = tr.step(batches[i % len(batches)], EX, 1e-3, objective=ob, seed=i)
  adj.append(tr.loss_scale_adjust)
  if i < 8 or i >= steps - 8 or i % 50 == 0:
    log.append((i, {k: float(v) for k, v in s.items() if k.startswith('loss/')}))
torch.cuda.synchronize()
dt = time.time() - t0
for i, l in log:
  print(f'step {i:4d}  total {l["loss/total"]:.5f}  rgb fine {l.get("loss/fine", float("nan")):.5f} coarse {l.get("loss/coarse", float("nan")):.5f}  norm fine {l.get("loss/norm/fine", 0):.6f}  mask fine {l.get("loss/mask/fine", 0):.6f}')
first = np.mean([l['loss/total'] for i, l in log if i < 8]); last = np.mean([l['loss/total'] for i, l in log if i >= steps - 8])
print(f'{steps} steps in {dt:.1f} s ({dt / steps * 1e3:.1f} ms per step incl. the per-step read-back of the loss terms); mean total loss of the first 8 steps {first:.5f}, of the last 8 {last:.5f}; '
      f'optimizer steps applied {tr.optimizer_step}; loss-scale adjustment min {min(adj)} max {max(adj)} final {adj[-1]}; overflow events (call, source mask, action, fp32 detail) {tr.overflow_events}; '
      f'final policy: tangent_scale_adjust {tr.tangent_scale_adjust} split_chains {tr.split_chains} fp32_step {tr.fp32_step}')
assert np.isfinite(last) and last < first and tr.optimizer_step == steps, (first, last, tr.optimizer_step)
