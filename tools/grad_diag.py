"""Development aid: per-leaf gradient error of the trainer against the fp64 autograd oracle (tests/test_training.py's problem)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_training as TT
from nerfds_amd.training import Trainer
from oracle import train_oracle as T
R, Nc, Nf = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (6, 8, 8)
cfg, params, batch, t, u = TT._problem(R, Nc, Nf)
L, G, out = T.loss_and_grads(cfg, params, batch, batch['rgb'], TT.EX, t, u if Nf else None, mask_ratio=1.0)
tr = Trainer(cfg, params, max_rays=R)
stats = tr.step(batch, TT.EX, 0.0, t_rand=t, u_rand=u if Nf else None, mask_ratio=1.0, grads_only=True)
print('loss', stats['loss/coarse'], L['coarse'], stats['loss/fine'], L.get('fine'))
got = dict(TT.tree_leaves(tr.get_grads())); want = dict(TT.tree_leaves(G))
gmax = max(np.abs(v).max() for v in want.values())
for name, w in want.items():
  g = got[name].reshape(w.shape)
  l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
  print(f'{name:55s} l2 {l2:.2e}  |want| {np.linalg.norm(w):.2e} |got| {np.linalg.norm(g):.2e}' + ('   <<<' if l2 > 4e-3 else ''))
