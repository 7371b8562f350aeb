"""Generates the self-golden fixtures (ORACLE-generated, not reference-generated: the reference cannot be run
here and ships no vectors - SURVEY.md section 4, section 8c).

    python tests/golden/make_golden.py

Each .npz holds the seeded inputs (rays, injected sampling uniforms, extra_params, the init_params arguments
and a checksum of the generated weights) and the fp64 oracle's per-ray outputs.  Weights are regenerated from the
seed (numpy PCG64, stable across numpy versions); the checksum guards against drift.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, 'nerf-ds_amd'), ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)

from nerfds_amd import nerf_ds_config, static_config, hypernerf_config, init_params       # noqa: E402
from nerfds_amd.params import tree_leaves                                # noqa: E402
from oracle import nerfds_oracle as O                                    # noqa: E402

EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)
KEYS = ('rgb', 'depth', 'med_depth', 'acc', 'ray_norm', 'ray_rotation_field', 'ray_translation_field', 'ray_delta_x',
        'ray_hyper_points', 'ray_predicted_mask', 'med_points')

CASES = {
    'static_tiny': dict(graph='static', cfg_kw=dict(num_coarse_samples=8), R=16, init_kw=dict(bias_scale=0.05), seed=0),
    'nerfds_tiny_init': dict(graph='nerf_ds', cfg_kw=dict(num_warp_embeds=4, num_coarse_samples=8, num_fine_samples=8), R=8,
                             init_kw=dict(), seed=1),
    'nerfds_tiny_trained': dict(graph='nerf_ds', cfg_kw=dict(num_warp_embeds=4, num_coarse_samples=8, num_fine_samples=8), R=8,
                                init_kw=dict(warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1), seed=2),
    'hypernerf_tiny': dict(graph='hypernerf', cfg_kw=dict(num_warp_embeds=3, num_coarse_samples=8, num_fine_samples=8), R=8,
                           init_kw=dict(warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1), seed=3),
}
GRAPHS = {'static': static_config, 'nerf_ds': nerf_ds_config, 'hypernerf': hypernerf_config}
# training-step fixture (oracle/train_oracle.py): loss and a digest of every gradient leaf (sum, sum of squares, first 4 values)
TRAIN_CASE = 'nerfds_tiny_trained'


def build_case(name):
  c = CASES[name]
  cfg = GRAPHS[c['graph']](**c['cfg_kw'])
  params = init_params(cfg, c['seed'], **c['init_kw'])
  rng = np.random.default_rng(1000 + c['seed'])
  R = c['R']
  d = rng.normal(size=(R, 3))
  d /= np.linalg.norm(d, axis=-1, keepdims=True)
  rays = dict(origins=rng.normal(size=(R, 3)) * 0.1, directions=d, viewdirs=d,
              metadata={'warp': rng.integers(0, cfg.num_warp_embeds, (R, 1))},
              mask=(rng.random((R, 1)) < 0.3).astype(np.float64))
  t = rng.random((R, cfg.num_coarse_samples))
  u = rng.random((R, max(cfg.num_fine_samples, 1)))
  return cfg, params, rays, t, u


def weights_checksum(params):
  return float(sum(np.asarray(v, np.float64).sum() for _, v in tree_leaves(params)))


def run_oracle(cfg, params, rays, t, u):
  return O.NerfModel(cfg, params).apply(rays, EXTRA, t_rand=t, u_rand=u if cfg.num_fine_samples else None,
                                        use_predicted_norm=cfg.predict_norm, compute_sigma_gradient=False)


def main():
  for name in CASES:
    cfg, params, rays, t, u = build_case(name)
    out = run_oracle(cfg, params, rays, t, u)
    blob = dict(origins=rays['origins'], directions=rays['directions'], warp_id=rays['metadata']['warp'], mask=rays['mask'],
                t_rand=t, u_rand=u, weights_checksum=weights_checksum(params))
    for level, o in out.items():
      for k in KEYS:
        if k in o:
          blob[f'{level}/{k}'] = o[k].numpy()
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **blob)
    print(name, 'saved', {k: v.shape for k, v in blob.items() if hasattr(v, 'shape') and '/' in k and 'rgb' in k})


def train_case():
  cfg, params, rays, t, u = build_case(TRAIN_CASE)
  target = np.random.default_rng(77).random((rays['origins'].shape[0], 3))
  return cfg, params, rays, t, u, target


def run_train_oracle():
  from oracle import train_oracle as T
  cfg, params, rays, t, u, target = train_case()
  losses, grads, _ = T.loss_and_grads(cfg, params, rays, target, EXTRA, t, u)
  return losses, dict(tree_leaves(grads))


def grad_digest(name, g):
  """Seeded subsample (<= 256 entries) + L2 norm + max-abs of a gradient leaf: keeps the fixture small (the tree has 1.5 M entries)."""
  flat = np.asarray(g, np.float64).ravel()
  idx = np.random.default_rng(abs(hash_name(name)) % (2 ** 32)).choice(flat.size, size=min(256, flat.size), replace=False)
  idx.sort()
  return idx, flat[idx], float(np.linalg.norm(flat)), float(np.abs(flat).max())


def hash_name(name):
  h = 0
  for ch in name:
    h = (h * 131 + ord(ch)) % (2 ** 61 - 1)
  return h


def main_train():
  losses, grads = run_train_oracle()
  blob = {'loss/' + k: v for k, v in losses.items()}
  for name, g in grads.items():
    idx, vals, norm, amax = grad_digest(name, g)
    blob['idx/' + name], blob['val/' + name], blob['norm/' + name], blob['max/' + name] = idx, vals, norm, amax
  np.savez_compressed(os.path.join(HERE, 'train_' + TRAIN_CASE + '.npz'), **blob)
  print('train fixture saved', losses)


if __name__ == '__main__':
  main()
  main_train()
