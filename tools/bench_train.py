"""BASELINE config 4: full training step (forward + backward + Adam) on a random-ray batch of 4096, nerf_ds graph,
64 coarse + 64 fine samples.  Prints one JSON line (step time, rays/s, algorithmic fwd+bwd TFLOP/s)."""
import json, os, sys, time
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(_ROOT, 'nerf-ds_amd')); sys.path.insert(0, _ROOT)
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.training import Trainer

R = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
full = len(sys.argv) > 3 and sys.argv[3] == 'full'      # the whole configs/nerf_ds.gin objective (incl. the second-order norm loss)
OBJ = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1,
           norm_loss_weight=0.1) if full else None
cfg = nerf_ds_config(num_warp_embeds=64, near=0.3, far=1.7)
params = init_params(cfg, 0, warp_head_scale=5e-2)
rng = np.random.default_rng(2)
d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
dev = torch.device('cuda', 0)
f = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
batch = dict(origins=f(rng.normal(size=(R, 3)) * 0.2), directions=f(d), viewdirs=f(d), metadata={'warp': torch.as_tensor(rng.integers(0, 64, (R, 1)), device=dev)},
             mask=f((rng.random((R, 1)) < 0.3)), rgb=f(rng.random((R, 3))))
t_rand, u_rand = f(rng.random((R, 64))), f(rng.random((R, 64)))
EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
tr = Trainer(cfg, params, max_rays=R)
losses = []
for _ in range(3):
  losses.append(tr.step(batch, EX, 1e-3, t_rand=t_rand, u_rand=u_rand, objective=OBJ)['loss/total'])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
  losses.append(tr.step(batch, EX, 1e-3, t_rand=t_rand, u_rand=u_rand, objective=OBJ)['loss/total'])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flop = 3 * 333.15e6 * R          # fwd + bwd ~ 3 x forward
print(json.dumps({'config': 'BASELINE config 4: train step, %d rays, 64+64 samples, nerf_ds graph, %s%s' % (R, 'hand-written MFMA layers (split-bf16 operands, fp32 accumulation) + HIP kernels', ', full nerf_ds.gin objective incl. second-order norm loss' if full else ', rgb loss only'),
                  'ms_per_step': dt * 1e3, 'rays_per_s': R / dt, 'algorithmic_tflops': flop / dt / 1e12,
                  'loss_first': losses[0], 'loss_last': losses[-1], 'params': tr.num_params}))
