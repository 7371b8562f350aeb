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


def general_loss_with_squared_residual(x_sq, alpha, scale):
  """utils.py:208-263 (Barron's general robust loss), for the finite alpha values not equal to 0 or 2 plus those two."""
  eps = float(np.finfo(np.float32).eps)
  scale = max(eps, scale)
  loss_two = 0.5 * x_sq / scale ** 2
  if alpha == 2:
    return scale * loss_two
  if alpha == 0:
    return scale * torch.log1p(torch.clamp(loss_two, max=3e37))
  a = (1.0 if alpha >= 0 else -1.0) * max(eps, abs(alpha))
  b = max(eps, abs(alpha - 2))
  return scale * (b / a) * ((loss_two / (0.5 * b) + 1) ** (0.5 * alpha) - 1)


def auxiliary_losses(cfg, out, rays_dict, objective, dtype, level='coarse'):
  """The first-order auxiliary terms of _compute_loss_and_stats for one level (training.py:297-310, 334-339, 386-408)."""
  terms = {}
  weights = out['weights'].detach()                                            # lax.stop_gradient(model_out['weights'])
  if objective.get('warp_reg_loss_weight', 0.0):
    idx = O.compute_depth_index(weights)
    warp_mag = ((out['points'] - out['warped_points'][..., :3]) ** 2).sum(-1)
    resid = torch.take_along_dim(warp_mag, idx[..., None], dim=-1)
    terms['warp_reg'] = objective['warp_reg_loss_weight'] * general_loss_with_squared_residual(
        resid, objective.get('warp_reg_loss_alpha', -2.0), objective.get('warp_reg_loss_scale', 0.001)).mean()
  if objective.get('back_facing_reg_weight', 0.0):
    terms['back_facing'] = objective['back_facing_reg_weight'] * (weights * out['back_facing']).mean()
  if objective.get('predicted_mask_loss_weight', 0.0):
    pm = out['predicted_mask'].squeeze(-1)
    gt = torch.as_tensor(np.asarray(rays_dict['mask'])).to(dtype).reshape(-1)
    w = out['sharp_weights'].detach() if cfg.use_mask_sharp_weights else weights
    terms['predicted_mask'] = objective['predicted_mask_loss_weight'] * ((gt - (w * pm).sum(-1)) ** 2).mean()
  if objective.get('mask_occlusion_reg_loss_weight', 0.0):                     # training.py:409-417
    low = torch.clamp(0.01 - weights, min=0.0)
    terms['mask_occlusion_reg'] = objective['mask_occlusion_reg_loss_weight'] * (low * out['predicted_mask'].squeeze(-1).abs()).sum(-1).mean()
  if objective.get('hyper_reg_loss_weight', 0.0) and level == 'coarse':        # training.py:312-321; the COARSE level only (training.py:461-466: the fine
    #                                                                             level is evaluated with use_hyper_reg_loss at its default, False)
    resid = (out['warped_points'][..., 3:] ** 2).sum(-1)
    terms['hyper_reg'] = objective['hyper_reg_loss_weight'] * (weights * general_loss_with_squared_residual(resid, 0.0, 0.05)).sum(1).mean()
  if objective.get('norm_loss_weight', 0.0):                                   # training.py:323-332 (second order: target_norm is NOT detached)
    diff = out['predicted_norm'] - out['target_norm']
    terms['norm'] = objective['norm_loss_weight'] * (weights * torch.sqrt((diff ** 2).sum(-1))).mean()
  return terms


def loss_and_grads(cfg, params, rays_dict, target_rgb, extra_params, t_rand, u_rand, dtype=torch.float64,
                   use_predicted_norm=True, mask_ratio=1.0, objective=None):
  """Returns (loss dict {'fine','coarse','total'}, grads tree shaped like ``params``, model outputs)."""
  model = O.NerfModel(cfg, params, dtype=dtype)
  leaves = list(_leaves(model.params))
  for _, v in leaves:
    v.requires_grad_(True)
  out = model.apply(rays_dict, extra_params, t_rand=t_rand, u_rand=u_rand, use_predicted_norm=use_predicted_norm,
                    mask_ratio=mask_ratio, return_weights=True, return_points=True,
                    compute_sigma_gradient='differentiable' if (objective or {}).get('norm_loss_weight', 0.0) else False,
                    sharp_weights_std=(objective or {}).get('sharp_weights_std', 1.0))
  gt = torch.as_tensor(np.asarray(target_rgb)).to(dtype)
  losses = {level: ((out[level]['rgb'][..., :3] - gt) ** 2).mean() for level in out}      # training.py:265-274
  aux = {level: auxiliary_losses(cfg, out[level], rays_dict, objective, dtype, level) for level in out} if objective else {}
  el = None
  if objective and objective.get('elastic_loss_weight', 0.0):                                # training.py:112-156, 274-295: coarse level only (461-466)
    o = out['coarse']
    pts = o['points'].detach().clone().requires_grad_(True)
    warped, _ = O.se3_field_warp(cfg, model.params['warp_field'], pts, o['warp_metadata'], extra_params['warp_alpha'])
    # samples are independent: row c of every sample's Jacobian d warp / d point (warping.py:276-278: metadata - mask included - held fixed)
    J = torch.stack([torch.autograd.grad(warped[..., c].sum(), pts, create_graph=True)[0] for c in range(3)], dim=-2)      # [R, S, 3 (out), 3 (in)]
    w = o['weights'].detach()
    by_weight = objective.get('elastic_reduce_method', 'median') == 'weight'
    if not by_weight:
      J = torch.take_along_dim(J, O.compute_depth_index(w)[..., None, None, None], dim=-3)    # the median-depth sample's Jacobian
    sq = (torch.log(torch.clamp(torch.linalg.svdvals(J), min=1e-6)) ** 2).sum(-1)            # 'log_svals'
    l = general_loss_with_squared_residual(sq, -2.0, 0.03)
    if by_weight:
      l = w * l
    el = objective['elastic_loss_weight'] * l.sum(-1).mean()
  bg = None
  if objective and objective.get('background_loss_weight', 0.0):                             # training.py:159-183, 468-479 (ids / noise injected)
    pts = torch.as_tensor(np.asarray(rays_dict['background_points'])).to(dtype).reshape(-1, 3)
    ids = torch.as_tensor(np.asarray(rays_dict['background_ids']).astype(np.int64)).reshape(-1)
    embed = model.params['warp_embed']['embed']['embedding'][ids]                             # models.py:767
    if cfg.use_mask_in_warp:
      embed = torch.cat([embed, torch.zeros_like(embed[..., :1])], dim=-1)                    # "assume background has 0 mask"
    warped, _ = O.se3_field_warp(cfg, model.params['warp_field'], pts, embed, extra_params['warp_alpha'])
    bg = objective['background_loss_weight'] * general_loss_with_squared_residual(
        ((warped[..., :3] - pts) ** 2).sum(-1), objective.get('background_loss_alpha', -2.0), objective.get('background_loss_scale', 0.001)).mean()
  total = sum(losses.values()) + sum(v for a in aux.values() for v in a.values()) + (bg if bg is not None else 0.0) + (el if el is not None else 0.0)      # training.py:481
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
  for level, a in aux.items():
    for k, v in a.items():
      losses[f'{k}/{level}'] = float(v)
  if bg is not None:
    losses['background'] = float(bg)
  if el is not None:
    losses['elastic'] = float(el)
  losses['total'] = float(total)
  return losses, tree, {lvl: {k: v.detach() for k, v in o.items() if torch.is_tensor(v)} for lvl, o in out.items()}


def custom_loss_and_grads(cfg, params, rays_dict, extra_params, t_rand, u_rand, loss_fn, dtype=torch.float64, use_predicted_norm=True, mask_ratio=1.0):
  """``jax.value_and_grad(_loss_fn)`` for an ARBITRARY loss of the model outputs (training.py:441-494): ``loss_fn(out)`` gets the
  {'coarse': {...}, 'fine': {...}} dict of torch tensors (rgb, depth, acc, weights, ...) and returns a scalar.  Returns (loss, grads tree, out)."""
  model = O.NerfModel(cfg, params, dtype=dtype)
  leaves = list(_leaves(model.params))
  for _, v in leaves:
    v.requires_grad_(True)
  out = model.apply(rays_dict, extra_params, t_rand=t_rand, u_rand=u_rand, use_predicted_norm=use_predicted_norm, mask_ratio=mask_ratio,
                    return_weights=True, return_points=True, compute_sigma_gradient=False)
  total = loss_fn(out)
  grads = torch.autograd.grad(total, [v for _, v in leaves], allow_unused=True)
  tree = {}
  for (n, v), g in zip(leaves, grads):
    node = tree
    parts = n.split('/')
    for p in parts[:-1]:
      node = node.setdefault(p, {})
    node[parts[-1]] = (g if g is not None else torch.zeros_like(v)).detach().cpu().numpy()
  return float(total), tree, {lvl: {k: v.detach() for k, v in o.items() if torch.is_tensor(v)} for lvl, o in out.items()}


def adam_step(param, grad, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
  """flax.optim.Adam.apply_param_gradient (flax 0.3.4): step is the 0-based count BEFORE this update."""
  m = (1.0 - b1) * grad + b1 * m
  v = (1.0 - b2) * grad * grad + b2 * v
  t = step + 1.0
  m_hat = m / (1.0 - b1 ** t)
  v_hat = v / (1.0 - b2 ** t)
  return param - lr * m_hat / (np.sqrt(v_hat) + eps), m, v
