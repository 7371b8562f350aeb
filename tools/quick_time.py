"""Ad-hoc timing of the fused kernel (development aid, not the benchmark)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
from nerfds_amd import nerf_ds_config, init_params
from nerfds_amd.model import NerfModel

R = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
precs = sys.argv[2].split(',') if len(sys.argv) > 2 else ['bf16']
cfg = nerf_ds_config(num_warp_embeds=16)
params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
rng = np.random.default_rng(0)
d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
dev = torch.device('cuda', 0)
rays = dict(origins=torch.tensor(rng.normal(size=(R, 3)) * 0.2, dtype=torch.float32, device=dev),
            directions=torch.tensor(d, dtype=torch.float32, device=dev),
            metadata={'warp': torch.tensor(rng.integers(0, 16, (R, 1)), device=dev)},
            mask=torch.zeros((R, 1), device=dev))
rays['viewdirs'] = rays['directions']
extra = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
m = NerfModel(cfg, device=dev)
for prec in precs:
  for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    m.apply({'params': params}, rays, extra, rngs={'coarse': 1, 'fine': 2}, use_predicted_norm=True, precision=prec)
    torch.cuda.synchronize(); dt = time.time() - t0
    flops = R * 333.15e6
    print(f'{prec}: R={R} {dt*1e3:.2f} ms  {R/dt/1e6:.3f} Mrays/s  {flops/dt/1e12:.1f} TFLOP/s (algorithmic)', flush=True)
