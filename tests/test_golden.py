"""Golden-vector tests.  The fixtures under tests/golden/ are ORACLE-generated (tests/golden/make_golden.py): the
reference ships none.  CPU: the oracle still reproduces them (pins it against accidental edits).  GPU: the HIP path,
through the C ABI, matches them without the oracle being run at all."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_golden as G      # noqa: E402


def _load(name):
  z = np.load(os.path.join(HERE, 'golden', name + '.npz'))
  cfg, params, rays, t, u = G.build_case(name)
  assert np.array_equal(z['origins'], rays['origins']) and np.array_equal(z['t_rand'], t)          # seeded inputs are stable
  assert abs(float(z['weights_checksum']) - G.weights_checksum(params)) < 1e-9
  return z, cfg, params, rays, t, u


@pytest.mark.parametrize('name', list(G.CASES))
def test_oracle_reproduces_golden(name):
  z, cfg, params, rays, t, u = _load(name)
  out = G.run_oracle(cfg, params, rays, t, u)
  n = 0
  for level, o in out.items():
    for k in G.KEYS:
      if k in o:
        assert np.allclose(o[k].numpy(), z[f'{level}/{k}'], rtol=1e-12, atol=1e-14), (level, k)
        n += 1
  assert n >= 6


@pytest.mark.gpu
@pytest.mark.parametrize('name', list(G.CASES))
@pytest.mark.parametrize('prec,tol', [('f32', 1e-4), ('bf16x3', 1e-4), ('f16x3', 1e-4)])
def test_hip_matches_golden(name, prec, tol):
  from nerfds_amd.model import NerfModel
  z, cfg, params, rays, t, u = _load(name)
  out = NerfModel(cfg, device=torch.device('cuda', 0)).apply({'params': params}, rays, G.EXTRA, t_rand=t,
                                                             u_rand=u if cfg.num_fine_samples else None,
                                                             use_predicted_norm=cfg.predict_norm, precision=prec)
  for level, o in out.items():
    for k in G.KEYS:
      if f'{level}/{k}' not in z.files or k not in o:
        continue
      ref, got = z[f'{level}/{k}'], o[k].cpu().numpy()
      assert ref.shape == got.shape, (level, k)
      if ref.size == 0:          # e.g. ray_hyper_points of the static graph is [R, 0]
        continue
      err = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)
      assert err <= (tol if k == 'rgb' else 10 * tol), (level, k, err)        # 1e-4 rel on composited RGB (north_star)


def test_train_oracle_reproduces_golden():
  z = np.load(os.path.join(HERE, 'golden', 'train_' + G.TRAIN_CASE + '.npz'))
  losses, grads = G.run_train_oracle()
  for k, v in losses.items():
    assert abs(v - float(z['loss/' + k])) < 1e-12
  for name, g in grads.items():
    idx, vals, norm, amax = G.grad_digest(name, g)
    assert np.array_equal(idx, z['idx/' + name]) and np.allclose(vals, z['val/' + name], rtol=1e-10, atol=1e-14), name
    assert abs(norm - float(z['norm/' + name])) <= 1e-10 * max(norm, 1e-30)


# fp32 g arrays (NERFDS_TRAIN_G16=0) / 16-bit g arrays (the default): measured 3.3e-4 / 6.1e-5 and - with round 3's bf16 g - 3.4e-3 / 7.5e-4
# round 4 (f16 g, merged step): 3.6e-4 / 1.1e-4 and 6.9e-4 / 1.1e-4; the bounds are 2 x those (the float atomics of the sums reorder from run to run)
TOLS = {False: (8e-4, 2.5e-4), True: (1.4e-3, 2.5e-4)}


@pytest.mark.gpu
@pytest.mark.parametrize('g16', [False, True])
def test_hip_training_step_matches_golden(g16, monkeypatch):
  """The HIP training step against COMMITTED gradient digests (the oracle is not run): per leaf a seeded subsample of
  256 entries, the L2 norm and the max-abs value of the fp64 autograd gradient.  The trainer's layers are hand-written
  split-bf16 MFMA kernels (tests/test_training.py).  Measured in round 3 with the fused backward (activations as f16 + ReLU bits,
  data gradient chained in registers): worst max-abs 3.3e-4 of the leaf maximum, worst norm 6.1e-5 - bounds MAX_TOL / NORM_TOL are
  a few times that (the float atomics of the weight-gradient sums reorder from run to run).  Round 2's layer-by-layer backward
  needed 1e-1 / 3e-2 here; what it lost is what a round trip of every dX through fp32 HBM arrays and a second rounding to split
  bf16 costs on the ill-conditioned posenc backward of this trained-regime case.
  g16=True is the shipped default: the chains hand g to the weight-gradient kernels in 16 bits (round 3: bf16; round 4: loss-scaled f16, which
  only tightens what follows).  On this 16-ray case the 8-bit rounding of bf16 showed in the cancelling column sums (bias gradients of the warp field: 3.4e-3 / 7.5e-4, ten times the
  fp32-g figure, still 30x inside round 2's bounds); it averages down with the row count (the 19 200-row gradient test of
  tests/test_training.py holds its fp32-era bounds in this mode).  g16=False keeps the chain arithmetic pinned at the tight bounds."""
  from nerfds_amd.training import Trainer
  monkeypatch.setenv('NERFDS_TRAIN_G16', '1' if g16 else '0')
  MAX_TOL, NORM_TOL = TOLS[g16]
  gemm = 'mfma'
  from nerfds_amd.params import tree_leaves
  z = np.load(os.path.join(HERE, 'golden', 'train_' + G.TRAIN_CASE + '.npz'))
  cfg, params, rays, t, u, target = G.train_case()
  tr = Trainer(cfg, params, max_rays=rays['origins'].shape[0])
  stats = tr.step(dict(rays, rgb=target), G.EXTRA, 0.0, t_rand=t, u_rand=u, grads_only=True)
  assert abs(stats['loss/fine'] - float(z['loss/fine'])) < 2e-5 and abs(stats['loss/coarse'] - float(z['loss/coarse'])) < 2e-5
  got = dict(tree_leaves(tr.get_grads()))
  gmax = max(float(z[k]) for k in z.files if k.startswith('max/'))
  worst = {'max': (0.0, ''), 'norm': (0.0, '')}
  for name, g in got.items():
    idx, want, norm, amax = z['idx/' + name], z['val/' + name], float(z['norm/' + name]), float(z['max/' + name])
    flat = np.asarray(g, np.float64).ravel()
    scale = max(amax, 1e-3 * gmax)
    e_max = float(np.abs(flat[idx] - want).max() / scale)
    e_norm = float(abs(np.linalg.norm(flat) - norm) / max(norm, 1e-3 * gmax * np.sqrt(flat.size)))
    worst['max'] = max(worst['max'], (e_max, name))
    worst['norm'] = max(worst['norm'], (e_norm, name))
  import sys
  print(f"golden training digests: worst max-abs {worst['max'][0]:.2e} ({worst['max'][1]}), worst norm {worst['norm'][0]:.2e} ({worst['norm'][1]})", file=sys.stderr)
  assert worst['max'][0] < MAX_TOL and worst['norm'][0] < NORM_TOL, worst
