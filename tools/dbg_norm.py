import sys
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from test_training import _problem, EX, OBJECTIVE
from nerfds_amd.training import Trainer
from nerfds_amd.params import tree_leaves
from oracle import train_oracle as T
Nc, Nf = int(sys.argv[1]), int(sys.argv[2])
cfg, params, batch, t, u = _problem(24, Nc, Nf)
u = u if Nf else None
ob = dict(norm_loss_weight=0.05)
L,G,_ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
L32,G32,_ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob, dtype=torch.float32)
L0,G0,_ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u)
tr = Trainer(cfg, params, max_rays=24)
stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
print({k:v for k,v in stats.items() if 'norm' in k}, {k:v for k,v in L.items() if 'norm' in k})
got, want, w32, base = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G)), dict(tree_leaves(G32)), dict(tree_leaves(G0))
gmax = max(np.abs(v).max() for v in want.values())
for name, w in want.items():
    g = got[name].reshape(w.shape); fl = 1e-3*gmax*np.sqrt(w.size)
    l2 = np.linalg.norm(g-w)/max(np.linalg.norm(w), fl); n32 = np.linalg.norm(w32[name]-w)/max(np.linalg.norm(w), fl)
    mv = np.linalg.norm(w-base[name])/max(np.linalg.norm(w),1e-12)
    if 'trunk_mlp/hidden_0' in name or 'trunk_mlp/hidden_1/' in name or 'branches_w/logit/kernel' in name: print('%-46s l2 %.1e  oracle-f32 %.1e  norm-loss share %.2f' % (name, l2, n32, mv))
