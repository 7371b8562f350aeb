"""TEST INFRASTRUCTURE - CPU oracle of the first-order training step of BASELINE config 4:
``loss = mean((rgb_fine - gt)^2) + mean((rgb_coarse - gt)^2)`` (training.py:265-274 with utils.l2_loss utils.py:492-493,
training.py:459-466, 481) and its gradient w.r.t. every parameter (``jax.value_and_grad(_loss_fn)``, training.py:494),
obtained with torch autograd through the forward oracle (nerfds_oracle.NerfModel), which carries the reference's
stop_gradients (fine z samples model_utils.py:241; the normal fed to the rgb branch models.py:1132-1133).
The sigma-gradient (row M) is not evaluated: with an MSE-only loss ``target_norm`` has no consumer.
Also the optimiser the reference uses: flax.optim.Adam (flax==0.3.4, not under /root/reference; published update rule
restated: beta1 0.9, beta2 0.999, eps 1e-8, weight_decay 0, bias-corrected) - training.py:508, train.py:297-301.
PARITY UNPINNED, like the forward oracle.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline import this.
"""
import numpy as np
import torch

from . import nerfds_oracle as O


def _leaves(tree, prefix=''):
  for k, v in tree.items():
    if isinstance(v, dict):
      yield from _leaves(v, prefix + k + '/')
    else:
      yield prefix + k, v


def loss_and_grads(cfg, params, rays_dict, target_rgb, extra_params, t_rand, u_rand, dtype=torch.float64,
                   use_predicted_norm=True, mask_ratio=1.0):
  """Returns (loss dict {'fine','coarse','total'}, grads tree shaped like ``params``, model outputs)."""
  model = O.NerfModel(cfg, params, dtype=dtype)
  leaves = list(_leaves(model.params))
  for _, v in leaves:
    v.requires_grad_(True)
  out = model.apply(rays_dict, extra_params, t_rand=t_rand, u_rand=u_rand, use_predicted_norm=use_predicted_norm,
                    mask_ratio=mask_ratio, return_weights=True, return_points=True, compute_sigma_gradient=False)
  gt = torch.as_tensor(np.asarray(target_rgb)).to(dtype)
  losses = {level: ((out[level]['rgb'][..., :3] - gt) ** 2).mean() for level in out}      # training.py:265-274
  total = sum(losses.values())                                                             # training.py:481
  grads = torch.autograd.grad(total, [v for _, v in leaves], allow_unused=True)
  flat = {n: (g if g is not None else torch.zeros_like(v)).detach().cpu().numpy() for (n, v), g in zip(leaves, grads)}
  tree = {}
  for n, g in flat.items():
    node = tree
    parts = n.split('/')
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = g
  losses = {k: float(v) for k, v in losses.items()}
  losses['total'] = float(total)
  return losses, tree, {lvl: {k: v.detach() for k, v in o.items() if torch.is_tensor(v)} for lvl, o in out.items()}


def adam_step(param, grad, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
  """flax.optim.Adam.apply_param_gradient (flax 0.3.4): step is the 0-based count BEFORE this update."""
  m = (1.0 - b1) * grad + b1 * m
  v = (1.0 - b2) * grad * grad + b2 * v
  t = step + 1.0
  m_hat = m / (1.0 - b1 ** t)
  v_hat = v / (1.0 - b2 ** t)
  return param - lr * m_hat / (np.sqrt(v_hat) + eps), m, v
