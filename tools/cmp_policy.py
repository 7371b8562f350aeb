"""GPU: every rung of the overflow ladder (default / split-bf16 chains / fp32 step) against the fp64 oracle on the rgb-only step - worst gradient leaves."""
import os, sys, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))
from tests.test_training import _problem, EX, OBJECTIVE, tree_leaves
from nerfds_amd.training import Trainer
from oracle import train_oracle as T
for (R, Nc, Nf) in [(24, 16, 16), (64, 16, 16)]:
  cfg, params, batch, t, u = _problem(R, Nc, Nf)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u)
  want = dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  for pol in ('default', 'split_chains', 'fp32_step'):
    tr = Trainer(cfg, params, max_rays=R)
    if pol != 'default': setattr(tr, pol, True)
    tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
    got = dict(tree_leaves(tr.get_grads()))
    errs = sorted(((float(np.linalg.norm(got[k].reshape(w.shape) - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))), k) for k, w in want.items()), reverse=True)
    print(R, pol, ['%.2e %s' % e for e in errs[:4]])
