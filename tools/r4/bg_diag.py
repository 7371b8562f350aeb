import sys, numpy as np, torch
sys.path.insert(0,'nerf-ds_amd'); sys.path.insert(0,'.')
import tests.test_training as TT
from oracle import train_oracle as T
from nerfds_amd.training import Trainer
cfg, params, batch, t, u = TT._problem(24, 8, 8)
rng = np.random.default_rng(11); B=101
batch['background_points'] = rng.uniform(-1.0, 1.0, (B, 3)).astype(np.float32)
batch['background_ids'] = rng.integers(0, cfg.num_warp_embeds, (B,))
for ob in (None, dict(background_loss_weight=1.0)):
  L,G,_=T.loss_and_grads(cfg, params, batch, batch['rgb'], TT.EX, t, u, objective=ob)
  tr = Trainer(cfg, params, max_rays=24)
  st = tr.step(batch, TT.EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  got, want = dict(TT.tree_leaves(tr.get_grads())), dict(TT.tree_leaves(G))
  gmax=max(np.abs(v).max() for v in want.values())
  e={n: float(np.linalg.norm(got[n].reshape(w.shape)-w)/max(np.linalg.norm(w),1e-3*gmax*np.sqrt(w.size))) for n,w in want.items()}
  print('objective', ob, 'gmax', gmax)
  for k,v in sorted(e.items(), key=lambda kv:-kv[1])[:6]: print('  ', k, f'{v:.2e}', 'norm', f'{np.linalg.norm(want[k]):.3e}')
