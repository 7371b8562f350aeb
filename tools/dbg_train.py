import sys
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_training import _problem, EX
from nerfds_amd.training import Trainer
from nerfds_amd.params import tree_leaves
from oracle import train_oracle as T
R,Nc,Nf = 64,16,16
cfg, params, batch, t, u = _problem(R,Nc,Nf)
L,G,out = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, mask_ratio=0.7)
import torch
L32,G32,_ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, mask_ratio=0.7, dtype=torch.float32)
w32 = dict(tree_leaves(G32))
tr = Trainer(cfg, params, max_rays=R)
stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, mask_ratio=0.7)
print(stats, L)
got = dict(tree_leaves(tr.get_grads())); want = dict(tree_leaves(G))
gmax = max(np.abs(v).max() for v in want.values())
for name,w in want.items():
    g = got[name].reshape(w.shape)
    e32 = np.abs(w32[name]-w).max()/max(np.abs(w).max(),1e-3*gmax)
    l2 = np.linalg.norm(g-w)/max(np.linalg.norm(w),1e-12)
    print('%-46s want %.1e maxerr %.1e  l2 %.1e | oracle f32-vs-f64 maxerr %.1e' % (name, np.abs(w).max(), np.abs(g-w).max()/max(np.abs(w).max(),1e-3*gmax), l2, e32))
