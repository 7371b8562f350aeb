"""Development: trainer gradients (fused / layer-by-layer forward) against the fp64 oracle at a given size."""
import os, sys, numpy as np
sys.path.insert(0, 'nerf-ds_amd'); sys.path.insert(0, '.')
from nerfds_amd.training import Trainer
from tests.test_training import tree_leaves, EX, _problem
from oracle import train_oracle as T
def run(R, nc, nf, ratio=1.0):
  cfg, params, batch, t, u = _problem(R, nc, nf)
  L, G, out = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u if nf else None, mask_ratio=ratio)
  want = dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  for mode in ('1', '0'):
    os.environ['NERFDS_TRAIN_FUSED_FWD'] = mode
    tr = Trainer(cfg, params, max_rays=R)
    st = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u if nf else None, mask_ratio=ratio, grads_only=True)
    got = dict(tree_leaves(tr.get_grads()))
    errs = []
    for name, w in want.items():
      g = got[name].reshape(w.shape)
      l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
      errs.append((float(l2), name))
    errs.sort()
    print(R, nc, nf, 'fused' if mode == '1' else 'layer', 'loss', st['loss/coarse'], L['coarse'], 'worst l2', errs[-3:], flush=True)
    del tr
for a in [(33, 24, 0), (33, 12, 0), (200, 24, 0)]:
  run(*a)
