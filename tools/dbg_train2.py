import sys, copy
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from test_training import _problem, EX
from nerfds_amd.training import Trainer
from nerfds_amd.params import tree_leaves
from oracle import train_oracle as T
R,Nc,Nf = 6,8,0
cfg, params, batch, t, u = _problem(R,Nc,Nf)
L,G,out = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, None)
tr = Trainer(cfg, params, max_rays=R)
tr.step(batch, EX, 0.0, t_rand=t, grads_only=True)
got = dict(tree_leaves(tr.get_grads())); want = dict(tree_leaves(G))
for name in ('hyper_sheet_mlp/MLP_0/logit/bias', 'hyper_sheet_mlp/MLP_0/logit/kernel'):
    print(name, 'want', want[name].ravel()[:4], 'got', got[name].ravel()[:4])
# FD through the HIP forward and through the oracle forward
for eng in ('hip', 'oracle'):
  fd = []
  for j in range(2):
    vals = []
    for s in (1, -1):
      p = copy.deepcopy(params); b = np.array(p['hyper_sheet_mlp']['MLP_0']['logit']['bias'], np.float64); b[j] += s*1e-3; p['hyper_sheet_mlp']['MLP_0']['logit']['bias'] = b
      if eng == 'hip':
        tr.set_params(p); vals.append(tr.step(batch, EX, 0.0, t_rand=t, grads_only=True)['loss/coarse'])
      else:
        vals.append(T.loss_and_grads(cfg, p, batch, batch['rgb'], EX, t, None)[0]['total'])
    fd.append((vals[0]-vals[1])/2e-3)
  print(eng, 'FD', fd)
