"""CPU ORACLE for the NeRF-DS volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain torch-on-CPU (float64 for golden vectors, float32 for the timed CPU
baseline), the algorithm of the reference's render path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
(``nerf-ds_amd/``) never does.

PARITY UNPINNED.  The reference (JAX 0.3.15 / Flax 0.3.4, requirements.txt:2,6,7) can neither be
imported nor compiled in the authoring container (no jax/flax/gin; pure Python, nothing to compile)
and it ships no tests, golden vectors or fixtures for this path.  The oracle is
therefore pinned only by known-answer tests derived from the mathematics
(tests/test_oracle_kat.py: posenc closed forms, exp_se3 == scipy expm, constant-density
compositing, linear inverse-CDF, MLP skip semantics vs an independent torch.nn stack) and by
fp64-vs-fp32 self-consistency, not by outputs of the reference itself.

Each function cites the reference lines it follows (paths relative to /root/reference/).
Randomness: the reference draws ``t_rand`` / ``u`` from JAX threefry streams
(model_utils.py:84,217) which cannot be reproduced without JAX; here they are inputs.

Deliberate decisions on reference quirks (SURVEY.md section 8a "quirks"):
  1. ``sharpen_weights`` row-gather quirk (model_utils.py:181-182) is reproduced literally.
  2. ``use_warp=False`` graphs: identity warp, no rotation/translation fields (reference would raise).
  3. coarse-only graphs skip the ``del out['fine'][...]`` (reference would raise KeyError).
  4. ``1 - alpha + 1e-10`` kept.  5. ``theta = |w|`` has no epsilon - kept.
"""
from __future__ import annotations

import math
from typing import Any, Dict, Optional

import numpy as np
import torch

# ----------------------------------------------------------------------------------------------
# model_utils.py
# ----------------------------------------------------------------------------------------------


def posenc_window(min_deg, max_deg, alpha, dtype):
  """model_utils.py:420-436."""
  bands = torch.arange(min_deg, max_deg, dtype=dtype)
  x = torch.clip(torch.as_tensor(alpha, dtype=dtype) - bands, 0.0, 1.0)
  return 0.5 * (1 + torch.cos(math.pi * x + math.pi))


def posenc(x, min_deg, max_deg, use_identity=False, alpha=None):
  """model_utils.py:398-417.  Layout of the output is [freq][sin, cos][channel], flattened."""
  batch_shape = x.shape[:-1]
  scales = 2.0 ** torch.arange(min_deg, max_deg, dtype=x.dtype)
  xb = x[..., None, :] * scales[:, None]                                    # (*, F, C)
  four_feat = torch.sin(torch.stack([xb, xb + 0.5 * math.pi], dim=-2))      # (*, F, 2, C)
  if alpha is not None:
    window = posenc_window(min_deg, max_deg, alpha, x.dtype)
    four_feat = window[..., None, None] * four_feat
  four_feat = four_feat.reshape((*batch_shape, -1))
  if use_identity:
    return torch.cat([x, four_feat], dim=-1)
  return four_feat


def normalize_vector(vector):
  """model_utils.py:438-442 (eps is float32 machine epsilon whatever the dtype)."""
  eps = float(np.finfo(np.float32).eps)
  return vector / torch.sqrt(torch.clamp_min(torch.sum(vector ** 2, dim=-1, keepdim=True), eps))


def sample_along_rays(t_rand, origins, directions, num_coarse_samples, near, far,
                      use_stratified_sampling, use_linear_disparity=False):
  """model_utils.py:55-92; ``t_rand`` [B, Nc] replaces ``random.uniform(key, ...)`` (line 84)."""
  batch_size = origins.shape[0]
  dtype = origins.dtype
  t_vals = torch.linspace(0., 1., num_coarse_samples, dtype=dtype)
  if not use_linear_disparity:
    z_vals = near * (1. - t_vals) + far * t_vals
  else:
    z_vals = 1. / (1. / near * (1. - t_vals) + 1. / far * t_vals)
  if use_stratified_sampling:
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    z_vals = lower + (upper - lower) * t_rand.to(dtype)
  else:
    z_vals = z_vals[None, :].expand(batch_size, num_coarse_samples)
  return z_vals, origins[..., None, :] + z_vals[..., :, None] * directions[..., None, :]


def _dists_alpha_accum(sigma, z_vals, dirs, sample_at_infinity, eps, scale=1.0):
  """Shared body of volumetric_rendering (model_utils.py:123-135) and cal_weights (163-175)."""
  last_sample_z = 1e10 if sample_at_infinity else 1e-19
  dists = torch.cat([
      z_vals[..., 1:] - z_vals[..., :-1],
      torch.full_like(z_vals[..., :1], last_sample_z)], -1)
  dists = dists * torch.linalg.norm(dirs[..., None, :], dim=-1)
  alpha = 1.0 - torch.exp(-scale * sigma * dists)
  accum_prod = torch.cat([
      torch.ones_like(alpha[..., :1]),
      torch.cumprod(1.0 - alpha[..., :-1] + eps, dim=-1)], dim=-1)
  return alpha, accum_prod


def cal_weights(sigma, z_vals, dirs, sample_at_infinity=True, eps=1e-10, scale=1):
  """model_utils.py:162-177."""
  alpha, accum_prod = _dists_alpha_accum(sigma, z_vals, dirs, sample_at_infinity, eps, scale)
  return alpha * accum_prod


def sharpen_weights(weights, z_vals, std=0.01):
  """model_utils.py:180-190, including the row-gather quirk on line 182:
  ``z_vals[max_weights_idx]`` indexes ROWS (rays) of z_vals with the per-ray argmax."""
  max_weights_idx = torch.argmax(weights, dim=1)
  # [R, S]: row gather, NOT take_along_axis.  jnp gathers clamp out-of-bounds indices (the sample
  # index can exceed the number of rays), hence the clamp.
  max_weights_idx = torch.clamp(max_weights_idx, max=z_vals.shape[0] - 1)
  max_weights_z_val = z_vals[max_weights_idx]
  gaussian_filter = torch.exp(-0.5 * ((z_vals - max_weights_z_val) / std) ** 2) / (std * math.sqrt(2 * math.pi))
  sharp_weights = weights * gaussian_filter
  return sharp_weights / torch.sum(sharp_weights, dim=1)[..., None]


def compute_opaqueness_mask(weights, depth_threshold=0.5):
  """model_utils.py:272-293."""
  cumulative_contribution = torch.cumsum(weights, dim=-1)
  opaqueness = cumulative_contribution >= depth_threshold
  false_padding = torch.zeros_like(opaqueness[..., :1])
  padded_opaqueness = torch.cat([false_padding, opaqueness[..., :-1]], dim=-1)
  return torch.logical_xor(opaqueness, padded_opaqueness).to(weights.dtype)


def compute_depth_index(weights, depth_threshold=0.5):
  """model_utils.py:296-299 (argmax of an all-zero mask is 0)."""
  return torch.argmax(compute_opaqueness_mask(weights, depth_threshold), dim=-1)


def compute_depth_map(weights, z_vals, depth_threshold=0.5):
  """model_utils.py:302-317."""
  return torch.sum(compute_opaqueness_mask(weights, depth_threshold) * z_vals, dim=-1)


def volumetric_rendering(rgb, sigma, z_vals, dirs, use_white_background, sample_at_infinity=True,
                         eps=1e-10, use_sharp_weights=False, sharp_weights_std=1.0):
  """model_utils.py:95-159."""
  alpha, accum_prod = _dists_alpha_accum(sigma, z_vals, dirs, sample_at_infinity, eps)
  weights = alpha * accum_prod
  if use_sharp_weights:
    weights = sharpen_weights(weights, z_vals, std=sharp_weights_std)
  rgb = (weights[..., None] * rgb).sum(dim=-2)
  exp_depth = (weights * z_vals).sum(dim=-1)
  med_depth = compute_depth_map(weights, z_vals)
  acc = weights.sum(dim=-1)
  if use_white_background:
    rgb = rgb + (1. - acc[..., None])
  if sample_at_infinity:
    acc = weights[..., :-1].sum(dim=-1)
  return {'rgb': rgb, 'depth': exp_depth, 'med_depth': med_depth, 'acc': acc,
          'weights': weights, 'alpha': alpha, 'accum_prod': accum_prod}


def piecewise_constant_pdf(u, bins, weights, num_coarse_samples, use_stratified_sampling):
  """model_utils.py:193-241; ``u`` [B, N] replaces ``random.uniform`` (line 217)."""
  eps = 1e-5
  weights = weights + eps
  pdf = weights / weights.sum(dim=-1, keepdim=True)
  cdf = torch.cumsum(pdf, dim=-1)
  cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
  if use_stratified_sampling:
    u = u.to(cdf.dtype)
  else:
    u = torch.linspace(0., 1., num_coarse_samples, dtype=cdf.dtype)
    u = u.expand(*cdf.shape[:-1], num_coarse_samples)
  mask = (u[..., None, :] >= cdf[..., :, None])

  def minmax(x):
    x0 = torch.max(torch.where(mask, x[..., None], x[..., :1, None]), dim=-2).values
    x1 = torch.min(torch.where(~mask, x[..., None], x[..., -1:, None]), dim=-2).values
    x0 = torch.minimum(x0, x[..., -2:-1])
    x1 = torch.maximum(x1, x[..., 1:2])
    return x0, x1

  bins_g0, bins_g1 = minmax(bins)
  cdf_g0, cdf_g1 = minmax(cdf)
  denom = cdf_g1 - cdf_g0
  denom = torch.where(denom < eps, torch.ones_like(denom), denom)
  t = (u - cdf_g0) / denom
  return (bins_g0 + t * (bins_g1 - bins_g0)).detach()


def sample_pdf(u, bins, weights, origins, directions, z_vals, num_coarse_samples,
               use_stratified_sampling):
  """model_utils.py:244-269."""
  z_samples = piecewise_constant_pdf(u, bins, weights, num_coarse_samples, use_stratified_sampling)
  z_vals = torch.sort(torch.cat([z_vals, z_samples], dim=-1), dim=-1).values
  return z_vals, origins[..., None, :] + z_vals[..., None] * directions[..., None, :]


# ----------------------------------------------------------------------------------------------
# rigid_body.py
# ----------------------------------------------------------------------------------------------


def skew(w):
  """rigid_body.py:26-41, batched over leading dims."""
  z = torch.zeros_like(w[..., 0])
  return torch.stack([
      torch.stack([z, -w[..., 2], w[..., 1]], -1),
      torch.stack([w[..., 2], z, -w[..., 0]], -1),
      torch.stack([-w[..., 1], w[..., 0], z], -1)], -2)


def exp_so3(w, theta):
  """rigid_body.py:59-74 (Rodrigues)."""
  W = skew(w)
  eye = torch.eye(3, dtype=w.dtype)
  th = theta[..., None, None]
  return eye + torch.sin(th) * W + (1.0 - torch.cos(th)) * (W @ W)


def exp_se3(S, theta, rotation_only=False, inverse=False):
  """rigid_body.py:77-101.  Returns (R [...,3,3], p [...,3]) instead of the 4x4 block matrix."""
  w, v = S[..., :3], S[..., 3:]
  W = skew(w)
  R = exp_so3(w, theta)
  eye = torch.eye(3, dtype=S.dtype)
  th = theta[..., None, None]
  G = th * eye + (1.0 - torch.cos(th)) * W + (th - torch.sin(th)) * (W @ W)
  p = (G @ v[..., None])[..., 0]
  if rotation_only:
    p = p * 0
  if inverse:
    p = -(R.transpose(-1, -2) @ p[..., None])[..., 0]
    R = R.transpose(-1, -2)
  return R, p


# ----------------------------------------------------------------------------------------------
# modules.py
# ----------------------------------------------------------------------------------------------


def dense(p, x):
  """flax nn.Dense: y = x @ kernel + bias."""
  return x @ p['kernel'] + p['bias']


def mlp(p, x, depth, skips=(), hidden_activation=torch.relu, output_channels=0,
        output_activation=None):
  """modules.py:57-83: the RAW input is re-concatenated before layer i for i in skips."""
  inputs = x
  for i in range(depth):
    if i in skips:
      x = torch.cat([x, inputs], dim=-1)
    x = hidden_activation(dense(p[f'hidden_{i}'], x))
  if output_channels > 0:
    x = dense(p['logit'], x)
    if output_activation is not None:
      x = output_activation(x)
  return x


def glo_embed(p, ids):
  """modules.py:336-348."""
  if ids.shape[-1] == 1:
    ids = ids[..., 0]
  return p['embed']['embedding'][ids.long()]


def encode_embed(embed, p):
  """NerfModel._encode_embed (models.py:271-294): one id channel -> the GLO row; three channels (left id, right id, progression)
  -> (1 - progression) * row(left) + progression * row(right).  Ids are cast like ``astype(jnp.uint32)`` and clamped like a jnp gather."""
  table = p['embed']['embedding']
  rows = table.shape[0]
  row = lambda ids: table[ids.to(torch.int64).clamp(0, rows - 1)]
  if embed.shape[-1] == 3:
    left, right, progression = embed[..., 0], embed[..., 1], embed[..., 2:3].to(table.dtype)
    return (1.0 - progression) * row(left) + progression * row(right)
  return row(embed[..., 0])


def encode_metadata(cfg, params, metadata):
  """evaluation.encode_metadata (evaluation.py:29-50) for the built graphs: encode_warp_embed (models.py:321-322) and
  encode_hyper_embed (models.py:296-319: hyper_use_warp_embed -> the warp table and the warp metadata)."""
  enc = {}
  if cfg.use_warp:
    enc['encoded_warp'] = encode_embed(torch.as_tensor(np.asarray(metadata['warp'])), params['warp_embed'])
  if cfg.has_hyper:
    enc['encoded_hyper'] = encode_embed(torch.as_tensor(np.asarray(metadata['warp'])), params['warp_embed'])
  return enc


def filter_sigma(points, sigma, render_opts):
  """models.py:38-66."""
  if render_opts is None:
    return sigma
  if 'dust_threshold' in render_opts:
    dust_thres = render_opts.get('dust_threshold', 0.0)
    sigma = (sigma >= dust_thres).to(sigma.dtype) * sigma
  if 'bounding_box' in render_opts:
    xmin, xmax, ymin, ymax, zmin, zmax = render_opts['bounding_box']
    render_mask = ((points[..., 0] >= xmin) & (points[..., 0] <= xmax) & (points[..., 1] >= ymin) & (points[..., 1] <= ymax)
                   & (points[..., 2] >= zmin) & (points[..., 2] <= zmax))
    sigma = render_mask.to(sigma.dtype) * sigma
  return sigma


def hyper_sheet_mlp(cfg, p, points, embed, alpha):
  """modules.py:367-392."""
  feat = posenc(points, cfg.hyper_sheet_min_deg, cfg.hyper_sheet_max_deg, alpha=alpha)
  inputs = torch.cat([feat, embed], dim=-1)
  s = cfg.hyper_sheet_mlp
  return mlp(p['MLP_0'], inputs, s.depth, s.skips, output_channels=cfg.hyper_sheet_output_channels)


def mask_mlp(cfg, p, points, embed, alpha):
  """modules.py:409-434 (+ ``MaskMLP.output_activation = @jax.nn.relu``, nerf_ds.gin:118)."""
  feat = posenc(points, cfg.mask_min_deg, cfg.mask_max_deg, alpha=alpha)
  inputs = torch.cat([feat, embed], dim=-1) if cfg.use_mask_embed else feat
  s = cfg.mask_mlp
  out = mlp(p['MLP_0'], inputs, s.depth, s.skips, output_channels=1)
  return torch.relu(out) if cfg.mask_output_relu else out


# ----------------------------------------------------------------------------------------------
# warping.py
# ----------------------------------------------------------------------------------------------


def se3_field_transform(cfg, p, points, metadata_embed, warp_alpha):
  """warping.py:209-222: trunk -> (w, v) -> unit screw axis + theta."""
  feat = posenc(points, cfg.warp_min_deg, cfg.warp_max_deg,
                use_identity=cfg.warp_use_posenc_identity, alpha=warp_alpha)
  inputs = torch.cat([feat, metadata_embed], dim=-1)
  s = cfg.warp_trunk
  trunk_output = mlp(p['trunk'], inputs, s.depth, s.skips)
  w = dense(p['branches_w']['logit'], trunk_output)
  v = dense(p['branches_v']['logit'], trunk_output)
  theta = torch.linalg.norm(w, dim=-1)
  w = w / theta[..., None]
  v = v / theta[..., None]
  return torch.cat([w, v], dim=-1), theta


def se3_field_warp(cfg, p, points, metadata_embed, warp_alpha, vector=None, inverse=False,
                   with_translation=False):
  """warping.py:200-237."""
  screw_axis, theta = se3_field_transform(cfg, p, points, metadata_embed, warp_alpha)
  rotation_only = vector is not None and not with_translation
  R, t = exp_se3(screw_axis, theta, rotation_only=rotation_only, inverse=inverse)
  src = points if vector is None else vector
  warped = (R @ src[..., None])[..., 0] + t          # from_homogenous(T @ to_homogenous(src)), w == 1
  return warped, screw_axis


# ----------------------------------------------------------------------------------------------
# models.py
# ----------------------------------------------------------------------------------------------


def to_torch(tree, dtype=torch.float64):
  if isinstance(tree, dict):
    return {k: to_torch(v, dtype) for k, v in tree.items()}
  return torch.as_tensor(np.asarray(tree)).to(dtype)


class NerfModel:
  """Restatement of ``NerfModel`` (models.py:71-1565) for the configurations of ``NerfModelConfig``."""

  def __init__(self, cfg, params, dtype=torch.float64):
    cfg.validate()
    self.cfg = cfg
    self.dtype = dtype
    self.params = to_torch(params, dtype)

  # -- models.py:710-764 ------------------------------------------------------------------------
  def map_points(self, points, warp_embed, hyper_embed, extra_params, mask, use_warp=True):
    cfg, P = self.cfg, self.params
    if cfg.use_mask_in_warp and warp_embed is not None:
      warp_embed = torch.cat([warp_embed, mask], dim=-1)
    if cfg.use_mask_in_hyper and hyper_embed is not None:
      hyper_embed = torch.cat([hyper_embed, mask], dim=-1)
    if cfg.use_warp and use_warp:                                     # models.py:609-630
      spatial_points, _ = se3_field_warp(cfg, P['warp_field'], points, warp_embed, extra_params['warp_alpha'])
    else:
      spatial_points = points
    hyper_points = None
    if cfg.use_hyper and cfg.hyper_slice_method == 'bendy_sheet':     # models.py:642-670
      hyper_points = hyper_sheet_mlp(cfg, P['hyper_sheet_mlp'], points, hyper_embed,
                                     extra_params['hyper_sheet_alpha'])
    if hyper_points is not None and cfg.use_hyper_for_sigma:
      warped_points = torch.cat([spatial_points, hyper_points], dim=-1)
    else:
      warped_points = spatial_points
    return warped_points

  # -- models.py:581-607 ------------------------------------------------------------------------
  def map_vectors(self, points, vectors, warp_embed, extra_params, mask, inverse=False,
                  with_translation=False):
    cfg, P = self.cfg, self.params
    if not cfg.use_warp:
      return vectors
    if cfg.use_mask_in_warp:
      warp_embed = torch.cat([warp_embed, mask], dim=-1)
    out, _ = se3_field_warp(cfg, P['warp_field'], points, warp_embed, extra_params['warp_alpha'],
                            vector=vectors, inverse=inverse, with_translation=with_translation)
    return out

  # -- models.py:493-523, 393-429 ---------------------------------------------------------------
  def pre_process_query(self, points, viewdirs, extra_params):
    cfg = self.cfg
    rgb_condition = None
    if cfg.use_viewdirs:
      rgb_condition = posenc(viewdirs, cfg.viewdir_min_deg, cfg.viewdir_max_deg,
                             use_identity=cfg.use_posenc_identity)
    points_feat = posenc(points[..., :3], cfg.spatial_point_min_deg, cfg.spatial_point_max_deg,
                         use_identity=cfg.use_posenc_identity, alpha=extra_params['nerf_alpha'])
    if points.shape[-1] > 3:
      hyper_feats = posenc(points[..., 3:], cfg.hyper_point_min_deg, cfg.hyper_point_max_deg,
                           use_identity=False, alpha=extra_params['hyper_alpha'])
      points_feat = torch.cat([points_feat, hyper_feats], dim=-1)
    return points_feat, rgb_condition

  def _sigma_of_points(self, level, points, warp_embed, hyper_embed, viewdirs, mask, extra_params, use_warp):
    """cal_single_pt_sigma (models.py:1035-1063), batched."""
    cfg = self.cfg
    nerf = self.params[f'nerf_mlps_{level}']
    warped_points = self.map_points(points, warp_embed, hyper_embed, extra_params, mask, use_warp)
    points_feat, rgb_condition = self.pre_process_query(warped_points, viewdirs, extra_params)
    flat = points_feat.reshape(-1, points_feat.shape[-1])
    trunk_output = mlp(nerf['trunk_mlp'], flat, cfg.nerf_trunk_depth, tuple(cfg.nerf_skips))   # modules.py:252
    if rgb_condition is not None:
      bottleneck = dense(nerf['bottleneck'], trunk_output)                                        # modules.py:255
    else:
      bottleneck = trunk_output
    alpha_out = dense(nerf['alpha_mlp']['logit'], trunk_output)                                   # modules.py:273-274
    sigma = alpha_out[..., :cfg.alpha_channels]
    norm = alpha_out[..., cfg.alpha_channels:cfg.alpha_channels + 3] if cfg.predict_norm else None
    return sigma, norm, warped_points, trunk_output, bottleneck, rgb_condition

  # -- models.py:867-1417 -----------------------------------------------------------------------
  def render_samples(self, level, points, z_vals, directions, viewdirs, metadata, extra_params,
                     gt_mask, use_warp=True, use_sample_at_infinity=False, use_predicted_norm=False,
                     mask_ratio=1, sharp_weights_std=1.0, compute_sigma_gradient=True, metadata_encoded=False, render_opts=None):
    cfg, P = self.cfg, self.params
    out = {'points': points}
    R, S = points.shape[:2]
    batch_shape = points.shape[:-1]

    warp_embed = hyper_embed = mask_embed = None
    if use_warp and cfg.use_warp:                                    # models.py:897-904
      warp_embed = metadata['encoded_warp'].to(self.dtype) if metadata_encoded else glo_embed(P['warp_embed'], metadata['warp'])
    if cfg.has_hyper:                                                # models.py:907-916
      hyper_embed = metadata['encoded_hyper'].to(self.dtype) if metadata_encoded else warp_embed
    if cfg.use_predicted_mask:                                       # models.py:924-928 (ids even when metadata_encoded)
      # 'encoded_mask' has no reference counterpart: 3-channel metadata has no integer id for the mask table (include/nerfds.h)
      mask_embed = metadata['encoded_mask'].to(self.dtype) if (metadata_encoded and 'encoded_mask' in metadata) \
          else glo_embed(P['mask_embed'], metadata['warp'])

    def bcast(e):
      return None if e is None else e[:, None, :].expand(*batch_shape, e.shape[-1])
    warp_embed, hyper_embed, mask_embed = bcast(warp_embed), bcast(hyper_embed), bcast(mask_embed)
    gt_mask_b = bcast(gt_mask.to(self.dtype)) if gt_mask is not None else None

    if cfg.use_predicted_mask:                                       # models.py:955-975
      predicted_mask = mask_mlp(cfg, P['mask_mlp'], points, mask_embed, extra_params['warp_alpha'])
      out['predicted_mask'] = predicted_mask
      mask = predicted_mask * mask_ratio + gt_mask_b * (1 - mask_ratio)
    else:
      predicted_mask = None
      mask = gt_mask_b
    if warp_embed is not None:      # what the warp field sees next to the point (map_points): test infrastructure for the warp Jacobian (training.py:279)
      out['warp_metadata'] = torch.cat([warp_embed, mask], dim=-1) if cfg.use_mask_in_warp else warp_embed

    # value_and_grad(cal_single_pt_sigma) (models.py:1065-1077); mask enters as a constant input.
    if compute_sigma_gradient == 'differentiable':
      # The reference keeps the whole value_and_grad result in the graph (no stop_gradient on target_norm): the
      # norm loss is second order in the warp / hyper / trunk weights.  Used by oracle/train_oracle.py only.
      pts = points.clone().requires_grad_(True)
      with torch.enable_grad():
        sigma, norm, warped_points, trunk_output, bottleneck, rgb_condition = self._sigma_of_points(
            level, pts, warp_embed, hyper_embed, viewdirs, mask, extra_params, use_warp)
        grad, = torch.autograd.grad(sigma.sum(), pts, create_graph=True)
      sigma_gradient = normalize_vector(-grad)
    elif compute_sigma_gradient:
      pts = points.detach().clone().requires_grad_(True)
      mask_c = mask.detach() if mask is not None else None
      with torch.enable_grad():
        sigma, norm, warped_points, trunk_output, bottleneck, rgb_condition = self._sigma_of_points(
            level, pts, warp_embed, hyper_embed, viewdirs, mask_c, extra_params, use_warp)
        grad, = torch.autograd.grad(sigma.sum(), pts)
      sigma_gradient = normalize_vector(-grad.detach())
      sigma, warped_points = sigma.detach(), warped_points.detach()
      trunk_output, bottleneck = trunk_output.detach(), bottleneck.detach()
      norm = norm.detach() if norm is not None else None
    else:
      sigma, norm, warped_points, trunk_output, bottleneck, rgb_condition = self._sigma_of_points(
          level, points, warp_embed, hyper_embed, viewdirs, mask, extra_params, use_warp)
      sigma_gradient = None

    if norm is not None:
      norm = norm.reshape(R, S, 3)

    norm_input = None
    if use_predicted_norm and cfg.predict_norm:                      # models.py:1113-1133
      normalized_norm = normalize_vector(norm)
      norm_input = self.map_vectors(points, normalized_norm, warp_embed, extra_params, mask, inverse=True)
      norm_input = norm_input.detach()                               # stop_norm_gradient=True (models.py:179, 1132-1133)
    norm_input_feat = None
    if norm_input is not None:                                       # models.py:1137-1150
      norm_input = normalize_vector(norm_input)
      if cfg.norm_input_posenc:
        norm_input_feat = posenc(norm_input, cfg.norm_input_min_deg, cfg.norm_input_max_deg,
                                 use_identity=cfg.use_posenc_identity, alpha=extra_params['norm_input_alpha'])
      else:
        norm_input_feat = norm_input

    extra_rgb_condition = trunk_output if cfg.use_x_in_rgb_condition else None   # models.py:1201-1213

    # sharp weights (models.py:1236-1246)
    sigma_raw = sigma.reshape(R, S)
    sigmoid_sigma = torch.nn.functional.softplus(filter_sigma(points, sigma_raw, render_opts))   # the filter on the RAW density here (models.py:1236-1237)
    weights_sg = cal_weights(sigmoid_sigma, z_vals, directions)
    if cfg.use_mask_sharp_weights:
      out['sharp_weights'] = sharpen_weights(weights_sg, z_vals, std=sharp_weights_std)

    # query_rgb (modules.py:288-313): [bottleneck, rgb_condition, extra_rgb_condition, norm]
    rgb_input = trunk_output
    if rgb_condition is not None:
      cond = rgb_condition[:, None, :].expand(R, S, rgb_condition.shape[-1]).reshape(R * S, -1)
      rgb_input = torch.cat([bottleneck, cond], dim=-1)
    if extra_rgb_condition is not None:
      rgb_input = torch.cat([rgb_input, extra_rgb_condition], dim=-1)
    if norm_input_feat is not None:
      rgb_input = torch.cat([rgb_input, norm_input_feat.reshape(R * S, -1)], dim=-1)
    nerf = P[f'nerf_mlps_{level}']
    rgb = mlp(nerf['rgb_mlp'], rgb_input, cfg.nerf_rgb_branch_depth, (), output_channels=cfg.rgb_channels)

    # post_process_query (models.py:567-579); noise_std is None.
    rgb = torch.sigmoid(rgb.reshape(R, S, cfg.rgb_channels))
    sigma = torch.nn.functional.softplus(sigma_raw)
    out['sigma'] = sigma

    if cfg.predict_norm and sigma_gradient is not None:              # models.py:1273-1277
      sigma_gradient_r = normalize_vector(self.map_vectors(points, sigma_gradient, warp_embed, extra_params, mask))

    rotation_field = translation_field = None
    if cfg.use_warp:                                                 # models.py:1291-1305
      ref = normalize_vector(torch.ones_like(points))
      rotation_field = normalize_vector(self.map_vectors(points, ref, warp_embed, extra_params, mask)[..., :3])
      translation_field = self.map_vectors(points, torch.zeros_like(points), warp_embed, extra_params, mask,
                                           with_translation=True)[..., :3]

    warped_points = warped_points.reshape(R, S, warped_points.shape[-1])
    out['warped_points'] = warped_points
    sigma = filter_sigma(points, sigma, render_opts)                 # models.py:1288 (out['sigma'] above stays unfiltered)
    out.update(volumetric_rendering(rgb, sigma, z_vals, directions,
                                    use_white_background=cfg.use_white_background,
                                    sample_at_infinity=use_sample_at_infinity))
    out['sample_rgb'] = rgb       # oracle-only extra (per-sample colour), for debugging the HIP path
    out['z_vals'] = z_vals        # oracle-only extra

    if cfg.predict_norm:                                             # models.py:1324-1344
      out['predicted_norm'] = norm
      if sigma_gradient is not None:
        out['target_norm'] = sigma_gradient_r.reshape(R, S, 3)
      back_facing = (norm * viewdirs[:, None, :]).sum(-1)
      out['back_facing'] = torch.relu(back_facing) ** 2

    weights = out['weights']                                         # models.py:1346-1415
    if norm is not None:
      out['ray_norm'] = (weights[..., None] * norm).sum(dim=-2)
    elif sigma_gradient is not None:
      out['ray_norm'] = (weights[..., None] * sigma_gradient).sum(dim=-2)
    if rotation_field is not None:
      out['ray_rotation_field'] = (weights[..., None] * rotation_field).sum(dim=-2)
      out['ray_translation_field'] = (weights[..., None] * translation_field).sum(dim=-2)
    delta_x = warped_points[..., :3] - points
    out['delta_x'] = delta_x
    out['ray_delta_x'] = (weights[..., None] * delta_x).sum(dim=-2)
    hyper_points = warped_points[..., 3:]
    out['ray_hyper_points'] = (weights[..., None] * hyper_points).sum(dim=-2)
    out['ray_hyper_c'] = torch.zeros_like(out['ray_hyper_points'])
    if cfg.use_predicted_mask:
      out['ray_predicted_mask'] = (weights[..., None] * predicted_mask).sum(dim=-2)
    depth_indices = compute_depth_index(weights)
    out['med_points'] = torch.take_along_dim(
        warped_points, depth_indices[..., None, None].expand(R, 1, warped_points.shape[-1]), dim=-2)
    return out

  # -- models.py:1419-1565 ----------------------------------------------------------------------
  def apply(self, rays_dict: Dict[str, Any], extra_params: Dict[str, Any], *, t_rand=None, u_rand=None,
            use_warp=True, return_points=False, return_weights=False, near=None, far=None,
            use_sample_at_infinity=None, use_predicted_norm=False, mask_ratio=1, sharp_weights_std=1.0,
            compute_sigma_gradient=True, metadata_encoded=False, render_opts=None):
    cfg, dt = self.cfg, self.dtype
    as_t = lambda a: torch.as_tensor(np.asarray(a)).to(dt)
    origins, directions = as_t(rays_dict['origins']), as_t(rays_dict['directions'])
    viewdirs = as_t(rays_dict['viewdirs']) if 'viewdirs' in rays_dict else directions
    metadata = {k: (torch.as_tensor(np.asarray(v)).to(dt) if k.startswith('encoded_') else torch.as_tensor(np.asarray(v).astype(np.int64)))
                for k, v in rays_dict.get('metadata', {}).items()}
    mask = as_t(rays_dict['mask']) if rays_dict.get('mask') is not None else None
    use_warp = cfg.use_warp and use_warp
    near = cfg.near if near is None else near
    far = cfg.far if far is None else far
    if use_sample_at_infinity is None:
      use_sample_at_infinity = cfg.use_sample_at_infinity
    if t_rand is not None:
      t_rand = as_t(t_rand)
    if u_rand is not None:
      u_rand = as_t(u_rand)
    common = dict(use_warp=use_warp, use_predicted_norm=use_predicted_norm, mask_ratio=mask_ratio,
                  sharp_weights_std=sharp_weights_std, compute_sigma_gradient=compute_sigma_gradient,
                  metadata_encoded=metadata_encoded)
    # Per-level kwargs of the two render_samples calls (models.py:1493-1517 vs 1528-1552; DESIGN.md section 2 has the whole table): every
    # kwarg is forwarded identically to both levels EXCEPT
    #   use_sample_at_infinity  coarse = self.use_sample_at_infinity (:1509), fine = the per-call override (:1484-1485, :1544)
    #   render_opts             coarse = not passed, i.e. None (:884),        fine = render_opts (:1545)
    #   coarse_depth            coarse = None (:1502), fine = coarse_ret['depth'] (:1537) - read only under use_coarse_depth_for_mask
    #                           (:956-958, default False :210, no gin file sets it; not built: config.py rejects it)

    z_vals, points = sample_along_rays(t_rand, origins, directions, cfg.num_coarse_samples, near, far,
                                       cfg.use_stratified_sampling, cfg.use_linear_disparity)
    coarse_ret = self.render_samples('coarse', points, z_vals, directions, viewdirs, metadata, extra_params,
                                     mask, use_sample_at_infinity=cfg.use_sample_at_infinity, **common)
    out = {'coarse': coarse_ret}
    if cfg.num_fine_samples > 0:
      z_vals_mid = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
      z_vals, points = sample_pdf(u_rand, z_vals_mid, coarse_ret['weights'][..., 1:-1], origins, directions,
                                  z_vals, cfg.num_fine_samples, cfg.use_stratified_sampling)
      out['fine'] = self.render_samples('fine', points, z_vals, directions, viewdirs, metadata, extra_params,
                                        mask, use_sample_at_infinity=use_sample_at_infinity, render_opts=render_opts, **common)
    for level in out:
      if not return_weights:
        del out[level]['weights']
      if not return_points:
        del out[level]['points']
        del out[level]['warped_points']
    return out


def to_numpy(tree):
  if isinstance(tree, dict):
    return {k: to_numpy(v) for k, v in tree.items()}
  return tree.detach().cpu().numpy()
