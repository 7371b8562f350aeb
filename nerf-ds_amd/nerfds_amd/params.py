"""Parameter tree of the NeRF-DS model, in the reference's Flax naming.

The tree mirrors what ``NerfModel.init`` produces (models.py:324-391, names per Flax
conventions, SURVEY.md section 5 "Checkpoint / resume"): ``nn.Dense`` kernels are ``[in, out]``
with ``y = x @ kernel + bias``; ``nn.Embed`` tables are ``[num_embeddings, features]``.

    warp_embed/embed/embedding            [N, 8]
    mask_embed/embed/embedding            [N, 8]
    mask_mlp/MLP_0/hidden_{i}|logit       MaskMLP            (modules.py:409-434)
    warp_field/trunk/hidden_{i}           SE3Field trunk     (warping.py:166-172)
    warp_field/branches_w|branches_v/logit                   (warping.py:174-193)
    hyper_sheet_mlp/MLP_0/hidden_{i}|logit HyperSheetMLP     (modules.py:367-392)
    nerf_mlps_{coarse,fine}/trunk_mlp/hidden_{i}, bottleneck,
                            alpha_mlp/logit, rgb_mlp/hidden_0|logit  (modules.py:122-152)

Leaves are numpy float32 arrays; the host layer uploads / packs them.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .config import NerfModelConfig, MLPSpec


def mlp_layer_dims(in_dim: int, spec: MLPSpec, out_dim: int = 0) -> List[Tuple[str, int, int]]:
  """(name, fan_in, fan_out) of every Dense in a reference ``MLP`` (modules.py:57-83)."""
  dims = []
  width_in = in_dim
  for i in range(spec.depth):
    k = width_in + (in_dim if i in spec.skips else 0)
    dims.append((f'hidden_{i}', k, spec.width))
    width_in = spec.width
  if out_dim > 0:
    dims.append(('logit', width_in, out_dim))
  return dims


def nerf_mlp_layer_dims(cfg: NerfModelConfig) -> Dict[str, List[Tuple[str, int, int]]]:
  trunk = MLPSpec(cfg.nerf_trunk_depth, cfg.nerf_trunk_width, tuple(cfg.nerf_skips))
  rgb = MLPSpec(cfg.nerf_rgb_branch_depth, cfg.nerf_rgb_branch_width, ())
  return {
      'trunk_mlp': mlp_layer_dims(cfg.trunk_in_dim, trunk),
      'bottleneck': [('bottleneck', cfg.nerf_trunk_width, cfg.nerf_trunk_width)],
      'alpha_mlp': [('logit', cfg.nerf_trunk_width, cfg.alpha_out_dim)],
      'rgb_mlp': mlp_layer_dims(cfg.rgb_in_dim, rgb, cfg.rgb_channels),
  }


def levels(cfg: NerfModelConfig) -> List[str]:
  return ['coarse', 'fine'] if cfg.num_fine_samples > 0 else ['coarse']


def _glorot(rng, fan_in, fan_out):
  a = np.sqrt(6.0 / (fan_in + fan_out))
  return rng.uniform(-a, a, size=(fan_in, fan_out)).astype(np.float32)


def _dense(rng, fan_in, fan_out, kernel_init='glorot', scale=1.0, bias_scale=0.0):
  if kernel_init == 'glorot':
    k = _glorot(rng, fan_in, fan_out)
  elif kernel_init == 'uniform':        # jax.nn.initializers.uniform(scale): U[0, scale)
    k = rng.uniform(0.0, scale, size=(fan_in, fan_out)).astype(np.float32)
  elif kernel_init == 'normal':         # jax.nn.initializers.normal(stddev)
    k = (rng.standard_normal((fan_in, fan_out)) * scale).astype(np.float32)
  else:
    raise ValueError(kernel_init)
  if bias_scale > 0:
    b = rng.uniform(-bias_scale, bias_scale, size=(fan_out,)).astype(np.float32)
  else:
    b = np.zeros((fan_out,), np.float32)
  return {'kernel': k, 'bias': b}


def init_params(cfg: NerfModelConfig, seed: int = 0, *, warp_head_scale: float = 1e-4,
                small_head_scale: float = 1e-5, bias_scale: float = 0.0) -> Dict:
  """Random-initialised parameter tree with the reference's initialisers.

  ``warp_head_scale`` is the U[0, s) scale of the SE3 w/v heads (1e-4 at init, warping.py:156-157);
  ``small_head_scale`` the N(0, s) std of the hyper-sheet / mask output layers (1e-5, modules.py:362,404).
  A "trained-like" regime for tests and the benchmark raises these and ``bias_scale`` so that every
  branch of the graph carries signal.
  """
  cfg.validate()
  rng = np.random.default_rng(seed)
  p: Dict = {}
  if cfg.use_warp:
    p['warp_embed'] = {'embed': {'embedding': rng.uniform(0, 0.05, (cfg.num_warp_embeds, cfg.glo_num_dims)).astype(np.float32)}}
    trunk = {name: _dense(rng, i, o, bias_scale=bias_scale) for name, i, o in mlp_layer_dims(cfg.warp_in_dim, cfg.warp_trunk)}
    w = cfg.warp_trunk.width
    p['warp_field'] = {
        'trunk': trunk,
        'branches_w': {'logit': _dense(rng, w, 3, 'uniform', warp_head_scale, bias_scale * warp_head_scale)},
        'branches_v': {'logit': _dense(rng, w, 3, 'uniform', warp_head_scale, bias_scale * warp_head_scale)},
    }
  if cfg.use_predicted_mask:
    p['mask_embed'] = {'embed': {'embedding': rng.uniform(0, 0.05, (cfg.num_warp_embeds, cfg.glo_num_dims)).astype(np.float32)}}
    layers = {}
    for name, i, o in mlp_layer_dims(cfg.mask_in_dim, cfg.mask_mlp, 1):
      layers[name] = (_dense(rng, i, o, 'normal', small_head_scale, bias_scale * small_head_scale)
                      if name == 'logit' else _dense(rng, i, o, bias_scale=bias_scale))
    p['mask_mlp'] = {'MLP_0': layers}
  if cfg.has_hyper:
    layers = {}
    for name, i, o in mlp_layer_dims(cfg.hyper_in_dim, cfg.hyper_sheet_mlp, cfg.hyper_sheet_output_channels):
      layers[name] = (_dense(rng, i, o, 'normal', small_head_scale, bias_scale * small_head_scale)
                      if name == 'logit' else _dense(rng, i, o, bias_scale=bias_scale))
    p['hyper_sheet_mlp'] = {'MLP_0': layers}
  for level in levels(cfg):
    sub = {}
    for group, dims in nerf_mlp_layer_dims(cfg).items():
      if group == 'bottleneck':
        name, i, o = dims[0]
        sub['bottleneck'] = _dense(rng, i, o, bias_scale=bias_scale)
      else:
        sub[group] = {name: _dense(rng, i, o, bias_scale=bias_scale) for name, i, o in dims}
    p[f'nerf_mlps_{level}'] = sub
  return p


def tree_map(fn, tree):
  if isinstance(tree, dict):
    return {k: tree_map(fn, v) for k, v in tree.items()}
  return fn(tree)


def tree_leaves(tree, prefix=''):
  if isinstance(tree, dict):
    for k, v in tree.items():
      yield from tree_leaves(v, f'{prefix}/{k}' if prefix else k)
  else:
    yield prefix, tree


def param_count(tree) -> int:
  return sum(int(np.prod(v.shape)) for _, v in tree_leaves(tree))
