"""Training step of BASELINE config 4: HIP trainer vs the autograd oracle (oracle/train_oracle.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))
sys.path.insert(0, ROOT)
from nerfds_amd import init_params, nerf_ds_config      # noqa: E402
from nerfds_amd.params import tree_leaves                # noqa: E402

EX = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)


def _problem(R, Nc, Nf, seed=1, n_ids=4):
  cfg = nerf_ds_config(num_warp_embeds=n_ids, num_coarse_samples=Nc, num_fine_samples=Nf, near=0.3, far=1.7)
  params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  rng = np.random.default_rng(seed)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  batch = dict(origins=rng.normal(size=(R, 3)) * 0.2, directions=d, viewdirs=d, metadata={'warp': rng.integers(0, n_ids, (R, 1))},
               mask=(rng.random((R, 1)) < 0.3).astype(np.float32), rgb=rng.random((R, 3)))
  return cfg, params, batch, rng.random((R, Nc)), rng.random((R, max(Nf, 1)))


def test_adam_oracle_known_answer():
  from oracle import train_oracle as T
  # first step of Adam moves every coordinate by lr * sign(g) (bias-corrected m / sqrt(v) = g / |g|)
  p, m, v = T.adam_step(np.array([1.0, -2.0]), np.array([0.5, -3.0]), np.zeros(2), np.zeros(2), 0, 1e-2)
  np.testing.assert_allclose(p, [1.0 - 1e-2, -2.0 + 1e-2], rtol=1e-6)


def test_oracle_gradient_matches_finite_differences_where_no_stop_gradient_applies():
  import copy
  from oracle import train_oracle as T
  cfg, params, batch, t, _ = _problem(5, 8, 0)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, None)
  assert set(L) == {'coarse', 'total'}
  for path, idx in [(('nerf_mlps_coarse', 'rgb_mlp', 'hidden_0', 'kernel'), (300, 7)), (('nerf_mlps_coarse', 'bottleneck', 'kernel'), (2, 3))]:
    vals = []
    for s in (1, -1):
      p = copy.deepcopy(params)
      node = p
      for k in path[:-1]:
        node = node[k]
      a = np.array(node[path[-1]], np.float64); a[idx] += s * 1e-6; node[path[-1]] = a
      vals.append(T.loss_and_grads(cfg, p, batch, batch['rgb'], EX, t, None)[0]['total'])
    g = G
    for k in path:
      g = g[k]
    np.testing.assert_allclose(g[idx], (vals[0] - vals[1]) / 2e-6, rtol=1e-4)


# Every dense layer of the trainer runs on the hand-written MFMA kernels (train_gemm.hip): split-bf16 operands, fp32
# accumulation; there is no library GEMM in the product (a shape the kernels do not cover is NERFDS_ENOTSUP).  Operands carry 16
# mantissa bits (hi + lo) - finer than the reference's own matmuls at jnp's default precision (one bf16 pass on TPU, TF32 on
# NVIDIA GPUs) - except in the forward of the warp field, whose output feeds the 2^7-frequency posenc of the template: there the
# three-way split (24 bits) is used (with 16 bits the warp-field gradients were 1 % off the fp64 oracle).  The rgb-loss gradients
# meet the 4e-3 bound that fp32 GEMMs meet; with the auxiliary and the second-order norm losses, whose gradients reach the warp
# field through the ill-conditioned posenc backward (tests/test_golden.py has the argument), the 16-bit trunk GEMMs leave ~9e-3
# on warp-field leaves: bound 1.5e-2.
L2_TOL = {'mfma': 4e-3}
# measured in round 3 (printed by the tests): rgb loss 2.9e-3 worst leaf at <= 64 rays and 4.0e-3 at 19 200 rows; first-order auxiliary
# losses 2.5e-3; second-order norm loss 6.4e-3 (5.8e-3 at 19 200 rows).  Bounds are <= 2 x those.
L2_TOL_2ND = {'mfma': 1.2e-2}
L2_TOL_AUX = 5e-3        # first-order auxiliary losses (test_auxiliary_losses_match_the_oracle)


@pytest.fixture(params=['mfma'])
def gemm(request):
  return request.param


@pytest.mark.gpu
@pytest.mark.parametrize('R,Nc,Nf,ratio', [(6, 8, 8, 1.0), (64, 16, 16, 0.7), (33, 12, 0, 1.0)])
def test_hip_gradients_match_autograd_oracle(R, Nc, Nf, ratio, gemm):
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(R, Nc, Nf)
  L, G, out = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u if Nf else None, mask_ratio=ratio)
  tr = Trainer(cfg, params, max_rays=R)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u if Nf else None, mask_ratio=ratio, grads_only=True)
  assert abs(stats['loss/coarse'] - L['coarse']) < 2e-5 * max(1.0, L['coarse'])
  if Nf:
    assert abs(stats['loss/fine'] - L['fine']) < 2e-5 * max(1.0, L['fine'])
  got = dict(tree_leaves(tr.get_grads()))
  want = dict(tree_leaves(G))
  assert set(got) == set(want)
  # Tolerance: the HIP path is fp32 (as the reference trains), the oracle fp64.  At these widths / 2^7 posenc frequencies
  # / theta ~ 5e-2 the oracle's OWN fp32-vs-fp64 difference is 3e-4 .. 3e-3 per leaf (max-abs) - measured below and used
  # as the yardstick: per leaf, relative L2 error < 4e-3 and max-abs error < max(1e-2, 6 x the oracle's fp32 noise).
  import torch
  _, G32, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u if Nf else None, mask_ratio=ratio, dtype=torch.float32)
  w32 = dict(tree_leaves(G32))
  gmax = max(np.abs(v).max() for v in want.values())
  worst_l2, worst_max = (0.0, ''), (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    scale = max(np.abs(w).max(), 1e-3 * gmax)        # per-leaf scale, floored so all-zero leaves (stop-gradient) compare absolutely
    err = np.abs(g - w).max() / scale
    noise = np.abs(w32[name] - w).max() / scale
    l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
    # a ReLU whose pre-activation is ~0 can land on the other side with 16-bit operands; at 96 .. 2048 samples one flipped
    # unit is visible in the max-abs error of a leaf (not in its L2 error), so max-abs is bounded at 2.5e-2
    max_tol = 2.5e-2          # measured 1.1e-2
    worst_l2, worst_max = max(worst_l2, (float(l2), name)), max(worst_max, (float(err), name))
    assert l2 < L2_TOL[gemm] and err < max_tol, f'{name}: l2 {l2:.2e}, max {err:.2e} (oracle fp32 noise {noise:.2e})'
  print(f'gradients vs oracle ({R} rays, {Nc}+{Nf}): worst l2 {worst_l2[0]:.2e} ({worst_l2[1]}), worst max-abs {worst_max[0]:.2e} ({worst_max[1]})', file=sys.stderr)
  # the normal channels of the alpha head receive no gradient (stop_gradient, models.py:1132-1133)
  for lv in (['coarse', 'fine'] if Nf else ['coarse']):
    assert np.abs(got[f'nerf_mlps_{lv}/alpha_mlp/logit/kernel'][:, 1:]).max() == 0.0


@pytest.mark.gpu
def test_f16_data_gradient_chains_option(monkeypatch):
  """NERFDS_TRAIN_BWD_F16=1 (off by default; read at every step): the primal data-gradient chains in one f16 MFMA per product.  Same oracle, same
  leaves; measured 2.97e-3 at this size (2.90e-3 with the default split-bf16 chains) - the option is bounded like the default here, and its cost on
  smaller batches and on the loss-scale window is stated where it is defined (csrc/nerfds_train.cpp)."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  monkeypatch.setenv('NERFDS_TRAIN_BWD_F16', '1')
  cfg, params, batch, t, u = _problem(64, 16, 16)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u)
  tr = Trainer(cfg, params, max_rays=64)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
  assert abs(stats['loss/total'] - L['total']) < 2e-5 * max(1.0, abs(L['total']))
  got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    worst = max(worst, (float(np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))), name))
  print(f'f16 data-gradient chains (64 rays, 16+16): worst l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)
  assert worst[0] < 4e-3, worst


@pytest.mark.gpu
@pytest.mark.parametrize('R,Nc,Nf,reduction', [(24, 8, 8, 'mean'), (40, 16, 16, 'sum'), (17, 12, 0, 'mean')])
def test_caller_defined_loss_through_autograd(R, Nc, Nf, reduction):
  """The loss-agnostic backward (include/nerfds.h nerfds_trainer_forward / nerfds_render_rays_bwd behind nerfds_amd.autograd): a loss the fused step does
  not know - L1 on the colours of both levels, a squared pull on the fine level's acc, a weight on the coarse depth - written in torch on the
  renderer's outputs, differentiated with loss.backward(), against torch autograd through the fp64 oracle (jax.value_and_grad of an arbitrary _loss_fn,
  training.py:441-494): forward values, the loss and EVERY parameter leaf.  'sum' reduction: cotangents R times larger than a mean's - the loss scale
  of the backward's f16 g follows the cotangents."""
  import torch
  from nerfds_amd.autograd import DifferentiableRenderer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(R, Nc, Nf, seed=11)
  top = 'fine' if Nf else 'coarse'
  red = (lambda x: x.mean()) if reduction == 'mean' else (lambda x: x.sum())

  def loss_fn(out, gt):
    l = red((out[top]['rgb'][..., :3] - gt).abs()) + 0.3 * red((out[top]['acc'] - 0.7) ** 2) + 0.1 * red(out['coarse']['depth'])
    if Nf:
      l = l + 0.5 * red((out['coarse']['rgb'][..., :3] - gt).abs())
    return l
  want_loss, G, ref = T.custom_loss_and_grads(cfg, params, batch, EX, t, u if Nf else None, lambda o: loss_fn(o, torch.as_tensor(batch['rgb'], dtype=torch.float64)))
  render = DifferentiableRenderer(cfg, params, max_rays=R)
  out = render(batch, EX, t_rand=t, u_rand=u if Nf else None)
  assert set(out) == set(ref)
  for lv in out:
    for k in ('rgb', 'depth', 'acc'):
      e = np.abs(out[lv][k].detach().cpu().numpy() - ref[lv][k].numpy()).max() / max(np.abs(ref[lv][k].numpy()).max(), 1e-6)
      assert e < 2e-4, (lv, k, e)
  loss = loss_fn(out, torch.as_tensor(batch['rgb'], dtype=torch.float32, device=render.device))
  assert abs(float(loss) - want_loss) < 2e-4 * max(1.0, abs(want_loss))
  loss.backward()
  assert render.params.grad is not None and bool(torch.isfinite(render.params.grad).all())
  got, want = dict(tree_leaves(render.grads_tree())), dict(tree_leaves(G))
  assert set(got) == set(want)
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
    err = np.abs(g - w).max() / max(np.abs(w).max(), 1e-3 * gmax)
    worst = max(worst, (float(l2), name))
    # L1 cotangents are +-1 / N wherever the error changes sign: a sample whose colour error is ~0 can take the other sign at 16-bit operands, which a
    # squared error does not see (its cotangent is ~0 there); and the depth / acc terms reach the warp field through the ill-conditioned posenc
    # backward like the auxiliary losses do (L2_TOL_2ND's argument): 1.5e-2 where the rgb-MSE test has 4e-3 (measured: 9.7e-3 on a warp-field bias
    # at 17 rays x 12 samples, the worst leaf of the three cases)
    assert l2 < 1.5e-2 and err < 4e-2, f'{name}: l2 {l2:.2e}, max {err:.2e}'
  print(f'caller-defined loss ({R} rays, {Nc}+{Nf}, {reduction}): worst l2 {worst[0]:.2e} ({worst[1]}), loss scale adjust {render.trainer.loss_scale_adjust}', file=sys.stderr)
  # any torch optimizer on the leaf: the in-place update IS the library's parameter vector, and the next pass sees it
  opt = torch.optim.SGD([render.params], lr=1e-2 if reduction == 'mean' else 1e-2 / R)
  first = float(loss)
  for _ in range(5):
    opt.step(); opt.zero_grad()
    out = render(batch, EX, t_rand=t, u_rand=u if Nf else None)
    loss = loss_fn(out, torch.as_tensor(batch['rgb'], dtype=torch.float32, device=render.device))
    loss.backward()
  assert float(loss) < first, (first, float(loss))


@pytest.mark.gpu
def test_adam_update_and_loss_decrease():
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(64, 16, 16)
  tr = Trainer(cfg, params, max_rays=64)
  s0 = tr.step(batch, EX, 1e-3, t_rand=t, u_rand=u)
  g = dict(tree_leaves(tr.get_grads()))
  p1 = dict(tree_leaves(tr.get_params()))
  p0 = dict(tree_leaves(params))
  for name in ('nerf_mlps_fine/rgb_mlp/logit/kernel', 'warp_field/trunk/hidden_2/kernel', 'mask_embed/embed/embedding'):
    want, _, _ = T.adam_step(np.asarray(p0[name], np.float64), g[name].astype(np.float64), 0.0, 0.0, 0, 1e-3)
    np.testing.assert_allclose(p1[name], want.reshape(p1[name].shape), rtol=0, atol=2e-6)
  losses = [s0['loss/total']] + [tr.step(batch, EX, 1e-3, t_rand=t, u_rand=u)['loss/total'] for _ in range(20)]
  assert losses[-1] < 0.8 * losses[0], losses


@pytest.mark.gpu
def test_trainer_forward_agrees_with_the_fused_render_kernel():
  import torch
  from nerfds_amd.model import NerfModel
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(48, 16, 16)
  tr = Trainer(cfg, params, max_rays=48)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
  m = NerfModel(cfg, device=torch.device('cuda', 0))
  out = m.apply({'params': params}, batch, EX, t_rand=t, u_rand=u, use_predicted_norm=True, precision='f32')
  gt = torch.as_tensor(batch['rgb'], dtype=torch.float32, device='cuda')
  for lv in ('fine', 'coarse'):
    mse = float(((out[lv]['rgb'] - gt) ** 2).mean())
    assert abs(mse - stats[f'loss/{lv}']) < 1e-5, (lv, mse, stats)


@pytest.mark.gpu
def test_training_step_at_config4_size():
  """BASELINE configs[3] at its own size: 4096 random rays x (64 + 64) samples, nerf_ds graph, jitter drawn on chip.  No oracle at
  786 432 field evaluations; the size-independent properties: the trainer's two losses equal the fused render kernel's MSE on the
  same rays and the same Philox seed (fp32-MFMA kernel: 1e-5), every gradient leaf is finite and the gradient is not zero, the same
  seed gives the same loss again (to the order of its atomic summation), another seed another one, and a few Adam steps on the fixed batch bring the loss down."""
  import torch
  from nerfds_amd.model import NerfModel
  import nerfds_amd.model as M
  from nerfds_amd.training import Trainer
  R = 4096
  cfg = nerf_ds_config(num_warp_embeds=64, near=0.3, far=1.7)          # 64 + 64 samples, stratified
  assert cfg.num_coarse_samples == 64 and cfg.num_fine_samples == 64 and cfg.use_stratified_sampling
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  rng = np.random.default_rng(2)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  batch = dict(origins=(rng.normal(size=(R, 3)) * 0.2).astype(np.float32), directions=d.astype(np.float32), viewdirs=d.astype(np.float32),
               metadata={'warp': rng.integers(0, 64, (R, 1))}, mask=(rng.random((R, 1)) < 0.3).astype(np.float32),
               rgb=rng.random((R, 3)).astype(np.float32))
  tr = Trainer(cfg, params, max_rays=R)
  a = tr.step(batch, EX, 0.0, grads_only=True, seed=77)
  g = dict(tree_leaves(tr.get_grads()))
  assert all(np.isfinite(v).all() for v in g.values()) and sum(float(np.abs(v).sum()) for v in g.values()) > 0
  b = tr.step(batch, EX, 0.0, grads_only=True, seed=77)
  c = tr.step(batch, EX, 0.0, grads_only=True, seed=78)
  # (the per-ray losses are summed with float atomics: equal up to the summation order)
  assert abs(a['loss/fine'] - b['loss/fine']) < 1e-6 and abs(a['loss/coarse'] - b['loss/coarse']) < 1e-6 and abs(a['loss/fine'] - c['loss/fine']) > 1e-5
  m = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32')
  old = M._seed_from_rngs
  M._seed_from_rngs = lambda rngs: 77
  try:
    out = m.apply({'params': params}, batch, EX, use_predicted_norm=True, precision='f32')
  finally:
    M._seed_from_rngs = old
  gt = torch.as_tensor(batch['rgb'], device='cuda')
  for lv in ('fine', 'coarse'):
    mse = float(((out[lv]['rgb'] - gt) ** 2).mean())
    assert abs(mse - a[f'loss/{lv}']) < 1e-5 * max(1.0, mse), (lv, mse, a)
  first = tr.step(batch, EX, 1e-3, seed=1)['loss/total']
  for i in range(8):
    last = tr.step(batch, EX, 1e-3, seed=2 + i)['loss/total']
  assert last < first, (first, last)


@pytest.mark.gpu
def test_training_step_at_reference_batch():
  """The shape the reference actually trains at - configs/nerf_ds.gin:4 ships batch_size = 512 (BASELINE config 4 quotes 4096), 64 + 64 samples - under
  the objective that file selects (warp regulariser, back-facing regulariser, 3-D mask supervision on sharpened weights, the second-order norm loss:
  nerf_ds.gin:58-64, 82-87, 105-126), init_lr 1e-3, gradient clipping off as shipped: 200 steps on 8 fixed batches.  The reference's fp32 step never
  skips an update (training.py:494-508): here every one of the 200 updates must be applied on the FIRST attempt - no overflow event, the numeric
  policy still at its default - and the loss must come down."""
  import torch
  from nerfds_amd.training import Trainer
  R, steps = 512, 200
  cfg = nerf_ds_config(num_warp_embeds=64, num_coarse_samples=64, num_fine_samples=64, near=0.3, far=1.7)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  rng = np.random.default_rng(0)

  def make_batch():
    d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = rng.normal(size=(R, 3)) * 0.2
    b = dict(origins=o, directions=d, viewdirs=d, mask=(rng.random((R, 1)) < 0.3).astype(np.float32), rgb=0.5 + 0.5 * np.sin(3.0 * d + o))   # a smooth target
    b = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32).cuda() for k, v in b.items()}
    b['metadata'] = {'warp': torch.as_tensor(rng.integers(0, 64, (R, 1))).cuda()}
    return b
  batches = [make_batch() for _ in range(8)]
  ob = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1, norm_loss_weight=0.001)
  tr = Trainer(cfg, params, max_rays=R)
  tot = []
  for i in range(steps):
    tot.append(tr.step(batches[i % 8], EX, 1e-3, objective=ob, seed=i)['loss/total'])
  first, last = float(np.mean(tot[:8])), float(np.mean(tot[-8:]))
  print(f'reference batch (512 rays, 64 + 64, nerf_ds.gin objective): total loss {first:.5f} -> {last:.5f} over {steps} steps; overflow events {tr.overflow_events}', file=sys.stderr)
  assert tr.optimizer_step == steps and not tr.overflow_events, tr.overflow_events
  assert (tr.loss_scale_adjust, tr.tangent_scale_adjust, tr.split_chains, tr.fp32_step) == (0, 0, False, False)
  assert np.isfinite(last) and last < 0.8 * first, (first, last)


@pytest.mark.gpu
@pytest.mark.parametrize('R,nc,nf', [(1001, 48, 16), (70, 128, 128), (301, 24, 0)])
def test_fused_forward_equals_the_layer_by_layer_forward(monkeypatch, R, nc, nf):
  """The step's forward is ONE launch per level (train_fwd_kernel.hip: the render kernel's field evaluation writing
  every activation); NERFDS_TRAIN_FUSED_FWD=0 runs the same forward as ~50 layer kernels.  Same parameters, rays and jitter: the two
  losses agree to 1e-5, every gradient leaf to 2.5e-2 of its largest entry and the median leaf to 2e-3 (both forwards round operands
  to split bf16 in a different order, and a ReLU unit within that rounding of zero switches its whole gradient path: the same
  bounds as against the fp64 oracle above), with the ragged sizes (48 + 16 samples, 1000 rays: tail lanes and a tail workgroup) that the kernel has to clamp."""
  from nerfds_amd.training import Trainer
  cfg = nerf_ds_config(num_warp_embeds=16, near=0.3, far=1.7, num_coarse_samples=nc, num_fine_samples=nf)
  params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  rng = np.random.default_rng(5)
  d = rng.normal(size=(R, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  batch = dict(origins=(rng.normal(size=(R, 3)) * 0.2).astype(np.float32), directions=d.astype(np.float32), viewdirs=d.astype(np.float32),
               metadata={'warp': rng.integers(0, 16, (R, 1))}, mask=(rng.random((R, 1)) < 0.3).astype(np.float32),
               rgb=rng.random((R, 3)).astype(np.float32))
  objective = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1)
  res = {}
  for mode in ('1', '0', '1 again'):
    monkeypatch.setenv('NERFDS_TRAIN_FUSED_FWD', mode[0])
    tr = Trainer(cfg, params, max_rays=R)
    stats = tr.step(batch, EX, 0.0, mask_ratio=0.7, grads_only=True, seed=11, objective=objective)
    res[mode] = (stats, dict(tree_leaves(tr.get_grads())))
    del tr
  (sa, ga), (sb, gb) = res['1'], res['0']
  # the fused forward's hand-placed waits (vmcnt(4) stage boundaries): a second trainer gives the same gradients up to the order of the
  # float atomics that sum them
  for k, v in res['1 again'][1].items():
    assert float(np.abs(v - ga[k]).max()) <= 1e-5 * max(float(np.abs(ga[k]).max()), 1e-12) + 1e-9, k
  for k in ('loss/fine', 'loss/coarse', 'loss/total'):
    if k in sb:
      assert abs(sa[k] - sb[k]) <= 1e-5 * max(1.0, abs(sb[k])), (k, sa[k], sb[k])
  # per leaf, relative L2 with the denominator floored as in the oracle test above (near-zero leaves compare absolutely); each forward is
  # within 1.5e-2 of the fp64 oracle under this objective (the warp-field leaves, through the posenc backward), so two of
  # them are within 3e-2 of each other; the median leaf agrees to 2e-3
  gmax = max(float(np.abs(v).max()) for v in gb.values())
  errs = sorted((float(np.linalg.norm(ga[k] - gb[k]) / max(np.linalg.norm(gb[k]), 1e-3 * gmax * np.sqrt(gb[k].size))), k) for k in gb)
  # (measured 2.5e-2 on warp_field/trunk/hidden_1/bias at 301 rays x 24 samples: the layer-by-layer path - every dX rounded to split
  # bf16 again on its way back from HBM - is the looser of the two; against the oracle the fused path measures 2.5e-3 under this objective)
  assert errs[-1][0] <= 3e-2 and errs[len(errs) // 2][0] <= 2e-3, errs[-3:]


@pytest.mark.gpu
def test_trainer_rejects_what_it_cannot_do():
  from nerfds_amd import static_config
  from nerfds_amd.training import Trainer
  with pytest.raises(NotImplementedError):
    Trainer(static_config(), None, max_rays=8)
  cfg, params, batch, t, u = _problem(9, 8, 8)
  tr = Trainer(cfg, params, max_rays=8)
  with pytest.raises(RuntimeError):
    tr.step(batch, EX, 0.0, t_rand=t, u_rand=u)       # 9 rays > max_rays


def _assert_same_update(pa, pb, lr):
  """Two trainers that took the same step.  Gradients are accumulated with float atomics (weight gradients: one partial per
  workgroup; bias / embedding gradients), so two runs differ in the last bits, and Adam's first update is lr * g / (|g| + 1e-8):
  an element whose gradient is itself at rounding-noise level can move by a visible fraction of lr.  Everything else must agree."""
  d = np.abs(pa.astype(np.float64) - pb)
  assert d.max() <= 2.0 * lr
  assert (d > 1e-7).mean() < 2e-3, ((d > 1e-7).mean(), d.max())


@pytest.mark.gpu
def test_grads_only_then_apply_equals_one_step_and_views_alias():
  import torch
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(32, 8, 8)
  a = Trainer(cfg, params, max_rays=32)
  b = Trainer(cfg, params, max_rays=32)
  a.step(batch, EX, 1e-3, t_rand=t, u_rand=u)
  b.step(batch, EX, 1e-3, t_rand=t, u_rand=u, grads_only=True)
  g = b.grads_tensor()
  assert g.is_cuda and g.numel() == b.num_params
  np.testing.assert_array_equal(g.cpu().numpy(), b._download(1))           # the view aliases the library's vector
  b.apply_gradients(1e-3)
  torch.cuda.synchronize()
  _assert_same_update(a._download(0), b._download(0), 1e-3)


@pytest.mark.gpu
def test_sigma_gradient_target_norm_matches_oracle():
  """target_norm = normalize(R normalize(-d sigma / d x)) by forward-mode tangents vs the oracle's autograd."""
  from nerfds_amd.training import Trainer
  from oracle import nerfds_oracle as O
  cfg, params, batch, t, u = _problem(24, 12, 12)
  ref = O.NerfModel(cfg, params).apply(batch, EX, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=True,
                                       return_weights=True, return_points=True)
  tr = Trainer(cfg, params, max_rays=24)
  tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, sigma_gradient=True)
  for level in ('coarse', 'fine'):
    got, want = tr.target_norm(level), ref[level]['target_norm'].numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, atol=1e-5)
    cos = (got * want).sum(-1)
    # Normalising a gradient amplifies fp32 rounding where |d sigma / d x| is tiny.  Reported, not hidden: the fraction of samples
    # off by more than 1e-4 (measured 0.0 % coarse / 0.35 % fine on this case), bounded at 1 %; the median error is at fp32 level and
    # no sample is off by more than 1 - cos = 0.1 (measured worst 3.9e-2).
    frac = float((1 - cos > 1e-4).mean())
    print(f'target_norm {level}: {100 * frac:.2f} % of the samples have 1 - cos > 1e-4; median {np.median(1 - cos):.1e}, worst {1 - cos.min():.2e}', file=sys.stderr)
    assert frac < 0.01 and np.median(1 - cos) < 1e-6 and cos.min() > 0.9, (level, frac, np.median(1 - cos), cos.min())
  with pytest.raises(RuntimeError):
    tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
    tr.target_norm('fine')                      # the last step did not evaluate it


OBJECTIVE = dict(warp_reg_loss_weight=0.001, warp_reg_loss_alpha=-2.0, warp_reg_loss_scale=0.001, back_facing_reg_weight=0.1,
                 predicted_mask_loss_weight=0.1, sharp_weights_std=0.1)      # configs/nerf_ds.gin:62-63, 86-87, 108, 120-126


@pytest.mark.gpu
@pytest.mark.parametrize('sharp', [True, False])
def test_auxiliary_losses_match_the_oracle(sharp, gemm):
  """warp regulariser, back-facing regulariser, 3-D mask supervision and hyper-point regulariser of the reference objective (everything
  but the second-order norm loss): loss terms and the full gradient vs autograd."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(40, 12, 12)
  cfg = cfg.replace(use_mask_sharp_weights=sharp)
  # give the mask / normal heads something to supervise (the init regime has relu(mask logit) == 0 everywhere)
  params['mask_mlp']['MLP_0']['logit']['bias'] = np.asarray(params['mask_mlp']['MLP_0']['logit']['bias']) + 0.7      # logits span [-1.0, -0.35] at init: make relu(logit) a mix of zeros and positives
  ob = dict(OBJECTIVE, hyper_reg_loss_weight=0.01, mask_occlusion_reg_loss_weight=1.0)      # + the mask occlusion regulariser (training.py:409-417) and the hyper-point regulariser (training.py:312-321; off in nerf_ds.gin, on in HyperNeRF's own configs)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
  L0, G0, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=OBJECTIVE)
  tr = Trainer(cfg, params, max_rays=40)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  hs = np.linalg.norm(dict(tree_leaves(G))['hyper_sheet_mlp/MLP_0/logit/kernel'] - dict(tree_leaves(G0))['hyper_sheet_mlp/MLP_0/logit/kernel'])
  assert hs > 0.05 * np.linalg.norm(dict(tree_leaves(G0))['hyper_sheet_mlp/MLP_0/logit/kernel'])      # the regulariser moves the hyper sheet's gradient visibly
  assert stats['loss/hyper_reg/fine'] == 0.0 and 'hyper_reg/fine' not in L      # coarse level only (training.py:461-466)
  for level in ('fine', 'coarse'):
    for k in ('warp_reg', 'back_facing', 'predicted_mask', 'mask_occlusion_reg') + (('hyper_reg',) if level == 'coarse' else ()):
      want = L[f'{k}/{level}']
      assert abs(stats[f'loss/{k}/{level}'] - want) <= 2e-4 * max(abs(want), 1e-4), (k, level, stats[f'loss/{k}/{level}'], want)
  assert abs(stats['loss/total'] - L['total']) < 1e-4 * L['total']
  got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
    worst = max(worst, (float(l2), name))
    assert l2 < L2_TOL_AUX, (name, l2)
  print(f'auxiliary losses (sharp={sharp}): worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)
  # the normal channels of the alpha head now DO receive gradient (back-facing regulariser), the mask net too
  assert np.abs(got['nerf_mlps_fine/alpha_mlp/logit/kernel'][:, 1:]).max() > 0
  assert np.abs(got['mask_mlp/MLP_0/hidden_0/kernel']).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize('with_aux', [False, True])
def test_background_loss_matches_the_oracle(with_aux):
  """training.py:159-183, 468-479: background_loss_weight * mean general_loss(|apply_warp(x) - x|^2, -2, 0.001) over a batch of points that should not move
  (ids and noise injected - the reference draws them), added to the levels' losses: its term, the total and every gradient leaf against autograd, alone
  (the levels then keep the merged flow) and together with the per-ray auxiliary terms; a ragged point count (3 workgroup tiles + 5 rows)."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(64, 16, 16)      # (the size of the rgb-loss test: at 24 rays x 8 + 8 a ReLU flip already costs 1e-2 on a leaf)
  rng = np.random.default_rng(11)
  B = 101
  batch['background_points'] = rng.uniform(-1.0, 1.0, (B, 3)).astype(np.float32)
  batch['background_ids'] = rng.integers(0, cfg.num_warp_embeds, (B,))
  # scale 0.3 instead of the reference's default 0.001: the init-regime warp moves points by ~0.05, where the alpha = -2 loss at scale 0.001 is
  # saturated and its gradient vanishes - the check would see nothing
  ob = dict(OBJECTIVE if with_aux else {}, background_loss_weight=1.0, background_loss_scale=0.3)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
  L0, G0, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=(OBJECTIVE if with_aux else None))
  tr = Trainer(cfg, params, max_rays=64)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  assert L['background'] > 1e-3 * L['total']                       # the term is a visible part of this objective
  assert abs(stats['loss/background'] - L['background']) <= 2e-4 * L['background'], (stats['loss/background'], L['background'])
  assert abs(stats['loss/total'] - L['total']) < 1e-4 * L['total']
  got, want, base = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G)), dict(tree_leaves(G0))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
    worst = max(worst, (float(l2), name))
    assert l2 < L2_TOL_AUX, (name, l2)
  print(f'background loss (with_aux={with_aux}): worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)
  for name in ('warp_field/trunk/hidden_0/kernel', 'warp_field/branches_v/logit/bias', 'warp_embed/embed/embedding'):
    assert np.linalg.norm(want[name] - base[name]) > 0.05 * np.linalg.norm(base[name]), name      # ... and of these gradients
  with pytest.raises(ValueError):
    tr.step({k: v for k, v in batch.items() if k != 'background_points'}, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)


@pytest.mark.gpu
@pytest.mark.parametrize('method,with_rest', [('median', False), ('weight', False), ('median', True)])
def test_elastic_loss_second_order_matches_the_oracle(method, with_rest):
  """training.py:112-156 ('log_svals'), 274-295: general_loss(sum log^2 of the singular values of the warp field's Jacobian) at the median-depth sample
  (or over all samples, weighted) of the COARSE level.  Second order in the warp field's weights: forward-mode Jacobian + its backward in the trainer,
  autograd-of-autograd in the oracle; alone, and together with the first-order auxiliary terms and the second-order norm loss."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(24, 16, 16)
  rest = dict(OBJECTIVE, norm_loss_weight=0.05, hyper_reg_loss_weight=0.01) if with_rest else {}
  # (weight 1: at the init regime's warp the alpha = -2 loss sits in its flat part - 0.059 of at most 0.06 - and a typical 0.01 would move no gradient visibly)
  ob = dict(rest, elastic_loss_weight=1.0, elastic_reduce_method=method)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
  L0, G0, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=rest or None)
  tr = Trainer(cfg, params, max_rays=24)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  assert L['elastic'] > 1e-4 * L['total']
  assert abs(stats['loss/elastic'] - L['elastic']) <= 1e-3 * L['elastic'], (stats['loss/elastic'], L['elastic'])
  assert abs(stats['loss/total'] - L['total']) < 2e-4 * L['total']
  got, want, base = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G)), dict(tree_leaves(G0))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
    worst = max(worst, (float(l2), name))
    assert l2 < L2_TOL_2ND['mfma'], (name, l2)
  print(f'elastic loss ({method}, with_rest={with_rest}): worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)
  for name in ('warp_field/trunk/hidden_2/kernel', 'warp_field/branches_w/logit/kernel'):
    assert np.linalg.norm(want[name] - base[name]) > 0.05 * np.linalg.norm(base[name]), name      # the term is a visible part of these gradients
  with pytest.raises(NotImplementedError):
    tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=dict(ob, elastic_loss_type='svals'))


@pytest.mark.gpu
@pytest.mark.parametrize('only_norm', [True, False])
def test_norm_loss_second_order_matches_the_oracle(only_norm, gemm):
  """training.py:323-332: mean(w |n - target_norm|) with NO stop_gradient on target_norm = d sigma / d x, i.e. second order in
  the warp / hyper / trunk weights: backward of the forward-mode tangent pass, vs torch double-backward in the oracle."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  # (16 + 16 samples: at 8 + 8 one sample of this seed sits on a ReLU boundary of the fp32 primal pass, which flips its mask
  # w.r.t. the fp64 oracle and moves a few leaves by 1 % - inherent to comparing fp32 with fp64 at a kink, not a bug)
  cfg, params, batch, t, u = _problem(24, 16, 16)
  ob = dict(norm_loss_weight=0.05) if only_norm else dict(OBJECTIVE, norm_loss_weight=0.05, hyper_reg_loss_weight=0.01)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
  L0, G0, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective={k: v for k, v in ob.items() if k != 'norm_loss_weight'} or None)
  tr = Trainer(cfg, params, max_rays=24)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  for level in ('fine', 'coarse'):
    want = L[f'norm/{level}']
    assert abs(stats[f'loss/norm/{level}'] - want) <= 2e-3 * want, (level, stats[f'loss/norm/{level}'], want)
  got, want, base = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G)), dict(tree_leaves(G0))
  gmax = max(np.abs(v).max() for v in want.values())
  moved = 0
  worst = (0.0, '')
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    l2 = np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
    worst = max(worst, (float(l2), name))
    assert l2 < L2_TOL_2ND[gemm], (name, l2)
    # the norm loss must actually have contributed to this leaf's gradient for the check to mean something
    if np.linalg.norm(w - base[name]) > 0.05 * max(np.linalg.norm(w), 1e-12):
      moved += 1
  print(f'norm loss (only_norm={only_norm}): worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)
  assert moved >= 20, moved      # trunk, warp and hyper leaves all move (second-order path), not only the alpha head


@pytest.mark.gpu
def test_gradient_clipping_matches_utils_clip_gradients():
  """utils.clip_gradients (utils.py:32-47): by value, then by global norm; applied between the gradient and the Adam update."""
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(16, 8, 8)
  tr = Trainer(cfg, params, max_rays=16)
  tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
  g = tr._download(1).astype(np.float64)
  max_val, max_norm = float(np.abs(g).max() * 0.3), float(np.linalg.norm(g) * 0.1)
  want = np.clip(g, -max_val, max_val)
  want = want * min(1.0, max_norm / (1e-7 + np.sqrt((want ** 2).sum())))
  tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, grad_max_val=max_val, grad_max_norm=max_norm)
  got = tr._download(1)
  np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-6 * np.abs(g).max())     # two runs: float atomics, see _assert_same_update
  assert abs(np.linalg.norm(got) - max_norm) < 1e-4 * max_norm
  # and a clipped training step = clipped gradient fed to Adam
  a, b = Trainer(cfg, params, max_rays=16), Trainer(cfg, params, max_rays=16)
  a.step(batch, EX, 1e-3, t_rand=t, u_rand=u, grad_max_norm=max_norm)
  b.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, grad_max_norm=max_norm)
  b.apply_gradients(1e-3)
  _assert_same_update(a._download(0), b._download(0), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('R,Nc,Nf,elastic', [(37, 12, 20, False), (50, 24, 8, False), (37, 12, 20, True)])
def test_whole_objective_step_at_ragged_shapes(R, Nc, Nf, elastic):
  """The whole-objective step (shared networks once per position, reverse-mode norm loss) where nothing divides anything: a ray count that is no multiple
  of the workgroup's four rays, Nc != Nf (the gathers / scatters between the fine level's row order and the position rows), a trainer built for more rays
  than the batch holds.  Same objective and bound as test_norm_loss_second_order_matches_the_oracle; `elastic` adds the elastic regulariser - with it the
  norm loss runs on the three-direction tangent pass (the regulariser needs the whole warp Jacobian), both second-order terms in one step."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(R, Nc, Nf)
  ob = dict(OBJECTIVE, norm_loss_weight=0.05, hyper_reg_loss_weight=0.01)
  if elastic:
    ob.update(elastic_loss_weight=0.01, elastic_reduce_method='weight')
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
  tr = Trainer(cfg, params, max_rays=R + 7)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  assert abs(stats['loss/total'] - L['total']) < 2e-5 * max(1.0, abs(L['total']))
  got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    l2 = float(np.linalg.norm(got[name].reshape(w.shape) - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size)))
    worst = max(worst, (l2, name))
    assert l2 < L2_TOL_2ND['mfma'], (name, l2)
  print(f'ragged ({R} rays, {Nc}+{Nf}, elastic={elastic}): worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)


_FALLBACK_SNIPPET = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from tests.test_training import _problem, EX, OBJECTIVE, tree_leaves
from nerfds_amd.training import Trainer
from oracle import train_oracle as T
cfg, params, batch, t, u = _problem(24, 16, 16)
ob = dict(OBJECTIVE, norm_loss_weight=0.05, hyper_reg_loss_weight=0.01)
L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
tr = Trainer(cfg, params, max_rays=24)
stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G))
gmax = max(np.abs(v).max() for v in want.values())
worst = max(float(np.linalg.norm(got[k].reshape(w.shape) - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))) for k, w in want.items())
print('WORST %%.3e LOSSDIFF %%.3e' %% (worst, abs(stats['loss/total'] - L['total'])))
"""


@pytest.mark.gpu
@pytest.mark.parametrize('env', [{'NERFDS_TRAIN_REVERSE_SIGMA': '0'}, {'NERFDS_TRAIN_MERGED_FULL': '0'}, {'NERFDS_TRAIN_REV_FWD_F16': '0', 'NERFDS_TRAIN_TAN_BWD_F16': '0'},
                                 {'NERFDS_TRAIN_HALF_TANGENTS': '0'}],
                         ids=['three_directions', 'level_by_level', 'split_bf16_tangent_chains', 'fp32_activations_for_tangent_steps'])
def test_fallback_flows_of_the_objective_step(env):
  """The switches that restore the earlier flows of the whole-objective step (read once per process, hence a subprocess each): the three-unit-direction
  tangent pass instead of the reverse-mode one, the level-by-level flow instead of the shared networks once per position, split-bf16 tangent chains
  instead of f16 - same objective as test_norm_loss_second_order_matches_the_oracle (only_norm=False), same oracle, same bound."""
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  e = dict(os.environ, **env)
  out = subprocess.run([sys.executable, '-c', _FALLBACK_SNIPPET % root], env=e, capture_output=True, text=True, timeout=600, cwd=root)
  line = [l for l in out.stdout.splitlines() if l.startswith('WORST')]
  assert line, out.stderr[-800:]
  worst, lossdiff = float(line[0].split()[1]), float(line[0].split()[3])
  print(f'fallback {env}: worst leaf l2 {worst:.2e}', file=sys.stderr)
  assert worst < L2_TOL_2ND['mfma'] and lossdiff < 2e-5, line


@pytest.mark.gpu
@pytest.mark.parametrize('full', [False, True])
def test_gradients_match_the_oracle_at_multi_tile_size(full):
  """The oracle comparisons above run at <= 2048 samples, where every persistent workgroup of the MFMA kernels sees one
  tile.  Here 600 rays x (16 + 16) samples = 9 600 / 19 200 rows (300 / 600 tiles of 32 over 256 workgroups, a partial
  16-row tile for the weight gradient): several tiles per workgroup - and the SAME pin as above, the fp64 autograd oracle
  (5-8 s on the host), not another mode of the trainer.  What this guards against is an indexing error past the first
  tile (O(1) differences).  Bounds, rgb loss: 6e-3 per leaf, or twice the oracle's OWN fp32-against-fp64 difference on that leaf where that
  is larger (the yardstick of the small-size test, measured here too); L2_TOL_2ND with the full objective."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  R = 600
  cfg, params, batch, t, u = _problem(R, 16, 16)
  obj = dict(warp_reg_loss_weight=0.001, back_facing_reg_weight=0.1, predicted_mask_loss_weight=0.1, sharp_weights_std=0.1,
             norm_loss_weight=0.1) if full else None
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, **({'objective': obj} if full else {}))
  tr = Trainer(cfg, params, max_rays=R)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=obj)
  assert abs(stats['loss/total'] - L['total']) < 2e-5 * max(1.0, abs(L['total']))
  got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  errs = {}
  for name, w in want.items():
    g = got[name].reshape(w.shape)
    errs[name] = float(np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size)))
  top = sorted(errs.items(), key=lambda kv: -kv[1])[:5]
  print(f'multi-tile ({"full objective" if full else "rgb loss"}): worst leaves ' + ', '.join(f'{k} {v:.2e}' for k, v in top), file=sys.stderr)
  # rgb loss: every leaf but the SE(3) field's cancelling sums sits below 4.3e-3.  The warp field's bias gradients are sums of 19 200 rows that
  # cancel to ~1e-4 of their terms: there ANY fp32 evaluation is several 1e-3 off the fp64 one - the oracle itself, run in fp32, differs from its fp64
  # run by 7.3e-3 on warp_field/branches_v/logit/bias and 4.6e-3 on trunk/hidden_5/bias at this size - and which side of that noise a given build lands
  # on moves with one-ulp changes upstream (measured on these two leaves: 3.3e-3 / 3.9e-3 with one shared-network pass per level, 7.9e-3 / 5.1e-3 in
  # the merged step - deterministic, the same with fp32 or f16 g arrays and with every weight-gradient kernel).  So: 6e-3, or 2 x the fp32 oracle's own
  # difference on the leaf.
  # Full objective: the norm loss differentiates d sigma / d x of a ReLU network - piecewise constant in the parameters - so ONE hidden unit of one
  # high-weight sample whose pre-activation rounds to the other side of zero moves the loss term by 1e-4 of itself and the warp field's leaves by 2 %.
  # At this size the fp32 run of the oracle differs from its fp64 run by 2.6e-2 on warp_field/branches_w/logit/bias (1.7 - 2.0e-2 on the warp trunk),
  # and the step that evaluates the shared networks once per position (round 5, run_merged_full: the NerfMLP reads the warped point the backward
  # differentiates at) lands within 6 % of those very numbers - on the fp32 side of that unit - where the level-by-level flow happens to land on the fp64
  # side (6.2e-3).  Same yardstick as for the rgb loss: the bound, or 2 x the oracle's own fp32-against-fp64 difference on the leaf.
  import torch
  bound = L2_TOL_2ND['mfma'] if full else 6e-3
  _, G32, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, dtype=torch.float32, **({'objective': obj} if full else {}))
  w32 = dict(tree_leaves(G32))
  for name, e in errs.items():
    w = want[name]
    noise = float(np.linalg.norm(w32[name].reshape(w.shape) - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size)))
    assert e < max(bound, 2.0 * noise), (name, e, noise)


@pytest.mark.gpu
@pytest.mark.parametrize('g16', [False, True])
def test_fused_backward_chain_against_numpy_on_the_step_buffers(g16, monkeypatch):
  """Kernel-level pin of the fused backward (train_bwd_kernel.hip) on the buffers of one step, read back through
  nerfds_trainer_debug_read: the forward's ReLU bits equal (f16 activation > 0) bit for bit; the first two links of the NerfMLP
  chain - g_rgb = 1[h_rgb > 0] (d rgb_logit W_rgb^T) and g_7 = 1[h_7 > 0] (g_rgb F^T + d alpha W_alpha^T), F = the bottleneck folded
  into rgb hidden_0 - and the input gradient d_trunk_in = g_0 W_0^T + g_4 W_4[raw-input rows]^T equal a float64 numpy evaluation of
  the same expressions on the same inputs to split-bf16 accuracy.  g16=False (NERFDS_TRAIN_G16=0): the chains write g as fp32, which pins
  the chain arithmetic at 1e-4; g16=True (the default): the g arrays are the scaled f16 copies the weight-gradient kernels read - every element
  within an f16 ulp (2^-10) of the float64 value, while the input gradient (from the registers, never rounded) stays at 1e-4 of its own
  float64 value when that is computed from fp32-grade g (here: from the float64 chain)."""
  from nerfds_amd.training import Trainer
  monkeypatch.setenv('NERFDS_TRAIN_G16', '1' if g16 else '0')
  R, Nc = 37, 8                                          # 296 rows: three 128-row workgroup iterations, a ragged tail
  cfg, params, batch, t, u = _problem(R, Nc, 0)
  tr = Trainer(cfg, params, max_rays=R)

  def read_g(name, shape):
    if not g16:
      return tr.debug_read(name, shape).astype(np.float64)
    scale = float(tr.debug_read('g_scale', (1,))[0])     # the chains store g_scale * g as f16 (loss scaling, a power of two)
    assert scale >= 64.0 and np.log2(scale) == int(np.log2(scale))
    return tr.debug_read(name, shape, np.float16).astype(np.float64) / scale
  gtol = 2.0 ** -10 if g16 else 1e-4
  tr.step(batch, EX, 0.0, t_rand=t, mask_ratio=1.0, grads_only=True)
  M = R * Nc

  def bits_to_mask(bits, W):                             # u16 at [(row * 2 + half) * (W / 32) + tile], bit r <-> accumulator register r
    b = bits.reshape(M, 2, W // 32)
    m = np.zeros((M, W), bool)
    for tl in range(W // 32):
      for h in range(2):
        for r in range(16):
          m[:, 32 * tl + (r & 3) + 8 * (r >> 2) + 4 * h] = (b[:, h, tl] >> r) & 1
    return m
  P = params['nerf_mlps_coarse']
  h_rgb = tr.debug_read('rgb_h16', (M, 128), np.float16).astype(np.float64)
  def check_bits(mask, h16):        # bit = (fp32 activation > 0); the stored f16 may have underflowed to 0 for a tiny positive value
    assert np.all(mask[h16 > 0]), 'a positive stored activation without its ReLU bit'
    assert np.count_nonzero(mask & (h16 == 0)) <= 1e-4 * mask.size, 'set bits on zero activations beyond f16 underflow'
  check_bits(bits_to_mask(tr.debug_read('rgb_bits', (M * 2 * 4,), np.uint16), 128), h_rgb)
  h7 = tr.debug_read('trunk_h16_7', (M, 256), np.float16).astype(np.float64)
  check_bits(bits_to_mask(tr.debug_read('trunk_bits_7', (M * 2 * 8,), np.uint16), 256), h7)
  with pytest.raises(RuntimeError):                      # a view holds M * width elements: an oversized read is refused, not performed
    tr.debug_read('trunk_h16_7', (tr.max_rays * Nc + 1, 256), np.float16)
  d_rgb = tr.debug_read('d_rgb_logit', (M, 3)).astype(np.float64)
  d_alpha = tr.debug_read('d_alpha', (M, 4)).astype(np.float64)
  Wr = np.asarray(P['rgb_mlp']['logit']['kernel'], np.float64)
  want_rgb = (d_rgb @ Wr.T) * (h_rgb > 0)
  got_rgb = read_g('rgb_g', (M, 128))
  assert np.abs(got_rgb - want_rgb).max() <= gtol * np.abs(want_rgb).max()
  K = np.asarray(P['rgb_mlp']['hidden_0']['kernel'], np.float64)      # rows [bottleneck 256 | viewdir 24 | trunk_out 256 | normal 24]
  F = np.asarray(P['bottleneck']['kernel'], np.float64) @ K[:256] + K[280:536]
  Wa = np.asarray(P['alpha_mlp']['logit']['kernel'], np.float64)
  want7 = (want_rgb @ F.T + d_alpha @ Wa.T) * (h7 > 0)
  got7 = read_g('trunk_g_7', (M, 256))
  assert np.abs(got7 - want7).max() <= gtol * np.abs(want7).max()
  # the input gradient of a chain: d_trunk_in = g_0 W_0^T + g_4 W_4[256:]^T
  g0, g4 = read_g('trunk_g_0', (M, 256)), read_g('trunk_g_4', (M, 256))
  W0 = np.asarray(P['trunk_mlp']['hidden_0']['kernel'], np.float64)
  W4 = np.asarray(P['trunk_mlp']['hidden_4']['kernel'], np.float64)
  want_in = g0 @ W0.T + g4 @ W4[256:].T
  got_in = tr.debug_read('d_trunk_in', (M, 52))
  # (g16: want_in is built from the ROUNDED copies of g_0 / g_4 while the kernel used its registers: 2^-12 per term, averaging down)
  assert np.abs(got_in - want_in).max() <= (1e-4 if not g16 else 5e-4) * np.abs(want_in).max()


@pytest.mark.gpu
def test_nonfinite_gradient_skips_the_update_and_is_reported():
  """A gradient that is non-finite in fp32 arithmetic too - here a NaN parameter: the reference's step would write NaN into every parameter
  (training.py:494-508 has no check) - skips the update as a whole (parameters, Adam moments and step count untouched) and is raised, after the
  overflow ladder has climbed to its last rung (the fp32 step) and found the same NaN there."""
  import copy
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(16, 8, 8, seed=4)
  bad = copy.deepcopy(params)
  k = np.array(bad['nerf_mlps_coarse']['trunk_mlp']['hidden_1']['kernel'], np.float32)
  k[3, 5] = np.nan
  bad['nerf_mlps_coarse']['trunk_mlp']['hidden_1']['kernel'] = k
  tr = Trainer(cfg, bad, max_rays=16)
  before = tr.get_params()
  with pytest.raises(FloatingPointError) as ei:
    tr.step(batch, EX, 1e-3, t_rand=t, u_rand=u, mask_ratio=1.0)
  assert 'fp32_step = True' in str(ei.value), str(ei.value)      # raised from the last rung, not before
  assert tr.nonfinite() and tr.optimizer_step == 0
  after = tr.get_params()
  for (ka, a), (kb, b) in zip(tree_leaves(before), tree_leaves(after)):
    assert ka == kb and np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), ka
  tr.set_params(params)                                   # a healthy step afterwards clears the flag
  tr.step(batch, EX, 1e-3, t_rand=t, u_rand=u, mask_ratio=1.0)
  assert not tr.nonfinite()


@pytest.mark.gpu
@pytest.mark.parametrize('source', ['activation', 'primal_g', 'second_order'])
def test_overflow_ladder_attributes_the_source_and_moves_only_its_knob(source):
  """The reference's fp32 step cannot overflow (training.py:494-508); this trainer's f16 storage can, in four places.  A skipped update is DIAGNOSED
  (nerfds_trainer_overflow_sources: which stored array holds the inf) and the attempt repeated under a policy that removes exactly that source:
    activation beyond 65504   -> the fp32 step (fp32 activations and g, layer by layer); the loss scales are left alone
    loss-scaled primal g      -> the primal loss scale, and nothing else
    second-order terms' f16   -> split-bf16 chains / the tangent scale; the PRIMAL loss scale is left alone (round 5 lowered only that one, 8 times,
                                 to no effect, and left it 16 binades down)
  In every case ONE update is applied, with a finite gradient."""
  import copy
  from nerfds_amd import training as TR
  cfg, params, batch, t, u = _problem(32, 8, 8, seed=7)
  kw = dict(t_rand=t, u_rand=u, mask_ratio=1.0)
  if source == 'activation':
    big = copy.deepcopy(params)
    big['nerf_mlps_coarse']['trunk_mlp']['hidden_1']['kernel'] = np.asarray(big['nerf_mlps_coarse']['trunk_mlp']['hidden_1']['kernel']) * 3e5
    tr = TR.Trainer(cfg, big, max_rays=32)
    tr.step(batch, EX, 1e-3, **kw)
    assert tr.fp32_step and tr.loss_scale_adjust == 0 and tr.tangent_scale_adjust == 0 and not tr.split_chains
    assert tr.overflow_events[-1][1] & TR.OVF_ACTIVATION and tr.overflow_events[-1][2] == 'fp32 step', tr.overflow_events
  elif source == 'primal_g':
    tr = TR.Trainer(cfg, params, max_rays=32)
    tr.loss_scale_adjust = 16                                  # 2^27 x a head gradient of ~1e-3: beyond f16
    tr.step(batch, EX, 1e-3, **kw)
    assert tr.loss_scale_adjust < 16 and tr.tangent_scale_adjust == 0 and not tr.split_chains and not tr.fp32_step
    assert tr.overflow_events and all(e[1] == TR.OVF_PRIMAL_G for e in tr.overflow_events), tr.overflow_events
  else:
    tr = TR.Trainer(cfg, params, max_rays=32)
    tr.tangent_scale_adjust = 12                               # cotangents aimed at 2^17: beyond f16 before any layer has amplified them
    tr.step(batch, EX, 1e-3, objective=dict(OBJECTIVE, norm_loss_weight=0.05), **kw)
    assert tr.loss_scale_adjust == 0 and not tr.fp32_step, (tr.loss_scale_adjust, tr.fp32_step, tr.overflow_events)
    assert tr.tangent_scale_adjust < 12 and tr.overflow_events, tr.overflow_events
    assert all(e[1] & (TR.OVF_TANGENT | TR.OVF_COTANGENT) for e in tr.overflow_events), tr.overflow_events      # (what lies behind them - the primal g too - is their consequence)
  assert tr.optimizer_step == 1 and not tr.nonfinite()
  assert all(np.isfinite(v).all() for _, v in tree_leaves(tr.get_grads()))
  # the policy relaxes one notch per `loss_scale_growth_interval` clean steps
  tr.loss_scale_growth_interval = 1
  for _ in range(24):
    tr.step(batch, EX, 0.0, **kw)
  assert tr.loss_scale_adjust >= 0 and tr.tangent_scale_adjust >= 0 and not tr.split_chains


@pytest.mark.gpu
@pytest.mark.parametrize('policy', ['fp32_step', 'split_chains'])
def test_policies_of_the_overflow_ladder_match_the_oracle(policy):
  """The rungs of the overflow ladder are full implementations of the step, not degraded modes: the whole objective (first-order auxiliary losses +
  the second-order norm loss) under the fp32 step (fp32 activations and g, layer-by-layer backward, three-direction tangent pass) and under
  split-bf16 chains - same oracle, same bound as test_norm_loss_second_order_matches_the_oracle."""
  from nerfds_amd.training import Trainer
  from oracle import train_oracle as T
  cfg, params, batch, t, u = _problem(24, 16, 16)
  ob = dict(OBJECTIVE, norm_loss_weight=0.05, hyper_reg_loss_weight=0.01)
  L, G, _ = T.loss_and_grads(cfg, params, batch, batch['rgb'], EX, t, u, objective=ob)
  tr = Trainer(cfg, params, max_rays=24)
  setattr(tr, policy, True)
  stats = tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True, objective=ob)
  assert abs(stats['loss/total'] - L['total']) < 2e-5 * max(1.0, abs(L['total']))
  got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(G))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    l2 = float(np.linalg.norm(got[name].reshape(w.shape) - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size)))
    worst = max(worst, (l2, name))
    assert l2 < L2_TOL_2ND['mfma'], (name, l2)
  print(f'{policy}: worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)
  # and the plain rgb step: the rung computes what the default policy computes (24 rays of this seed put a sample on a ReLU boundary of the fp32 primal
  # pass - 6.3e-3 against the fp64 oracle under EVERY policy, the default included, tools/cmp_policy.py - so the yardstick here is the default step)
  ref = Trainer(cfg, params, max_rays=24)
  ref.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
  tr.step(batch, EX, 0.0, t_rand=t, u_rand=u, grads_only=True)
  got, want = dict(tree_leaves(tr.get_grads())), dict(tree_leaves(ref.get_grads()))
  gmax = max(np.abs(v).max() for v in want.values())
  worst = (0.0, '')
  for name, w in want.items():
    l2 = float(np.linalg.norm(got[name].reshape(w.shape) - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size)))
    worst = max(worst, (l2, name))
    assert l2 < 2e-3, (name, l2)
  print(f'{policy}, rgb loss only, against the default policy: worst leaf l2 {worst[0]:.2e} ({worst[1]})', file=sys.stderr)


@pytest.mark.gpu
@pytest.mark.parametrize('path', ['fused', 'clip'])
def test_dynamic_loss_scaling_and_step_count(path):
  """A g that leaves f16's range is a property of the loss scale, not of the problem (the reference's g is fp32): the step is re-run at a lower
  scale - same seed, same samples - until the gradient is finite, on the fused path (Adam inside nerfds_trainer_step) and on the deferred one
  (clip / data-parallel: Adam in nerfds_trainer_apply, where the skip used to go unnoticed).  The optimizer's step count is the number of updates
  APPLIED: a skipped update leaves parameters, moments and the count alone (flax OptimizerState.step)."""
  import copy
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(32, 8, 8, seed=7)
  kw = dict(t_rand=t, u_rand=u, mask_ratio=1.0, **({'grad_max_norm': 1e3} if path == 'clip' else {}))
  ref = Trainer(cfg, params, max_rays=32)
  p0 = dict(tree_leaves(ref.get_params()))
  ref.step(batch, EX, 1e-3, **kw)
  assert ref.optimizer_step == 1 and ref.loss_scale_adjust == 0
  tr = Trainer(cfg, params, max_rays=32)
  tr.loss_scale_adjust = 16                                  # 2^27 x a head gradient of ~1e-3: beyond f16 -> overflow on the first attempts
  tr.step(batch, EX, 1e-3, **kw)
  assert tr.loss_scale_adjust < 16 and tr.loss_scale_adjust % 2 == 0 and not tr.nonfinite()
  assert tr.optimizer_step == 1                              # ONE update applied, however many attempts it took
  pa, pb = dict(tree_leaves(ref.get_params())), dict(tree_leaves(tr.get_params()))
  moved, differ, total = 0.0, 0, 0
  for k, v in p0.items():
    moved = max(moved, float(np.abs(pa[k] - v).max()))
    differ += int((np.abs(pa[k] - pb[k]) > 1e-4).sum())       # Adam's first step is -lr sign(g) per element: the two runs differ only where
    total += v.size                                           # g is so close to 0 that the f16 rounding of g at another exponent flips its sign
  assert moved > 5e-4 and differ <= 0.02 * total, (moved, differ, total)
  # only the primal loss scale moved (its array was the one that overflowed)
  assert tr.tangent_scale_adjust == 0 and not tr.split_chains and not tr.fp32_step


@pytest.mark.gpu
def test_gradient_scale_of_the_f16_g_arrays(monkeypatch):
  """The chains store g_scale * g as f16 (loss scaling) and the weight-gradient kernels undo the power of two: the gradients do not depend on
  it while g stays in range (2^8 .. 2^22 here: every leaf within 2e-3 of the default scale's, the differences being f16 roundings of g at another
  exponent and - at the small end - g values that drop below f16's denormals), and a scale that pushes g beyond 65504 is REPORTED: the weight gradients
  become inf, the update is skipped and the step raises (NERFDS_ENONFINITE) instead of training on garbage."""
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(32, 8, 8, seed=7)
  kw = dict(t_rand=t, u_rand=u, mask_ratio=1.0, grads_only=True)

  def grads(log2):
    if log2 is None:
      monkeypatch.delenv('NERFDS_TRAIN_G_SCALE_LOG2', raising=False)
    else:
      monkeypatch.setenv('NERFDS_TRAIN_G_SCALE_LOG2', str(log2))
    tr = Trainer(cfg, params, max_rays=32)
    tr.step(batch, EX, 0.0, **kw)
    scale = float(tr.debug_read('g_scale', (1,))[0])
    return dict(tree_leaves(tr.get_grads())), scale
  ref, s0 = grads(None)
  assert s0 == 2.0 ** 11                                 # 64 x 32 rays
  gmax = max(np.abs(v).max() for v in ref.values())
  for log2 in (8, 16, 22):
    got, sc = grads(log2)
    assert sc == 2.0 ** log2
    for name, w in ref.items():
      l2 = np.linalg.norm(got[name] - w) / max(np.linalg.norm(w), 1e-3 * gmax * np.sqrt(w.size))
      assert l2 < 2e-3, (log2, name, l2)
  monkeypatch.setenv('NERFDS_TRAIN_G_SCALE_LOG2', '40')  # 2^40 x a head gradient of ~1e-3: far beyond f16
  tr = Trainer(cfg, params, max_rays=32)
  with pytest.raises(FloatingPointError):
    tr.step(batch, EX, 1e-3, t_rand=t, u_rand=u, mask_ratio=1.0)
  assert tr.nonfinite()


@pytest.mark.gpu
def test_resume_from_a_checkpoint_with_adam_state(tmp_path):
  """training.save_checkpoint / restore (training.py:59-66, train.py:335-338): parameters + Adam moments + step count through the flax-msgpack file;
  a run resumed from it continues like the uninterrupted one (up to the order of the float atomics in the gradient sums)."""
  from nerfds_amd import checkpoint as ck
  from nerfds_amd.training import Trainer
  cfg, params, batch, t, u = _problem(32, 8, 8, seed=6)
  kw = dict(t_rand=t, u_rand=u, mask_ratio=1.0)
  a = Trainer(cfg, params, max_rays=32)
  for _ in range(3):
    a.step(batch, EX, 1e-3, **kw)
  ck.save_checkpoint(str(tmp_path), a.get_params(), dict(nerf_alpha=8.0), 3, opt_state=a.get_opt_state())
  for _ in range(2):
    a.step(batch, EX, 1e-3, **kw)
  p, extra, step = ck.restore_checkpoint(str(tmp_path))
  ema, sq, ostep = ck.restore_optimizer_state(str(tmp_path))
  assert step == ostep == 3 and extra == {'nerf_alpha': 8.0}
  b = Trainer(cfg, p, max_rays=32)
  b.set_opt_state(ema, sq, ostep)
  for _ in range(2):
    b.step(batch, EX, 1e-3, **kw)
  pa, pb = dict(tree_leaves(a.get_params())), dict(tree_leaves(b.get_params()))
  p0 = dict(tree_leaves(params))
  flat = lambda d: np.concatenate([np.asarray(d[k], np.float32).ravel() for k in pa])
  va, vb, v0 = flat(pa), flat(pb), flat(p0)
  moved = np.linalg.norm(va - v0)
  # Adam normalises every element's step to ~lr: an element whose gradient is at the noise level of the float atomics can take a different sign
  # in two runs of the SAME steps, so the comparison is on the whole vector
  same = np.linalg.norm(va - vb) / moved
  assert same < 0.05, same
  # without the moments the continuation is a different one (moments and bias correction restart): the check above is not vacuous
  c = Trainer(cfg, p, max_rays=32)
  for _ in range(2):
    c.step(batch, EX, 1e-3, **kw)
  other = np.linalg.norm(va - flat(dict(tree_leaves(c.get_params())))) / moved
  assert other > 4 * same and other > 0.1, (same, other)
