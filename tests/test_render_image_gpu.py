"""The drop-in top level on the GPU: evaluation.render_image + make_model_fn(model) with the real fused kernel
(hypernerf/evaluation.py:53-149, render.py:139-174), on one rank and - BASELINE config 3 - with a chunk split over two ranks
and exchanged by ONE RCCL all-gather of the per-ray records; the trainer's on-chip sampling jitter; target_norm on the
render surface."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.join(os.path.dirname(__file__), '..')
sys.path.insert(0, os.path.join(ROOT, 'nerf-ds_amd'))
sys.path.insert(0, ROOT)
from nerfds_amd import init_params, nerf_ds_config                  # noqa: E402

pytestmark = pytest.mark.gpu
EXTRA = dict(nerf_alpha=8., warp_alpha=4., hyper_alpha=1., hyper_sheet_alpha=6., norm_input_alpha=4.)


def _frame_rays(H, W, n_ids, seed):
  rng = np.random.default_rng(seed)
  d = rng.normal(size=(H, W, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  return dict(origins=(rng.normal(size=(H, W, 3)) * 0.1).astype(np.float32), directions=d.astype(np.float32), viewdirs=d.astype(np.float32),
              metadata={'warp': rng.integers(0, n_ids, (H, W, 1))}, mask=(rng.random((H, W, 1)) < 0.3).astype(np.float32))


def _setup(stratified):
  cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=16, num_fine_samples=16, use_stratified_sampling=stratified)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  return cfg, params


def test_render_image_top_level_matches_model_apply_and_the_oracle():
  """render_image(state, rays[H, W], make_model_fn(model), device_count=1, chunk) with the real kernel: equals one model.apply
  over the frame bit for bit (deterministic sampling; chunking must not change a ray), and the oracle on a sub-grid within 1e-4."""
  from nerfds_amd.evaluation import TrainState, make_model_fn, render_image
  from nerfds_amd.model import NerfModel
  from oracle import nerfds_oracle as O
  cfg, params = _setup(False)
  H, W = 37, 53                                        # 1961 rays: chunks of 512 leave a ragged last chunk
  rays = _frame_rays(H, W, 4, 0)
  model = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32')
  state = TrainState.create(params, **EXTRA)
  fn = make_model_fn(model, precision='f32')
  out = render_image(state, rays, fn, device_count=1, rng=np.array([0, 3]), chunk=512, cfg=cfg)
  assert out['rgb'].shape == (H, W, 3) and out['med_points'].shape == (H, W, 1, 5) and not out['rgb'].is_cuda   # host copy, once
  whole = model.apply({'params': params}, rays, EXTRA, use_predicted_norm=True, precision='f32')['fine']
  for k in ('rgb', 'depth', 'med_depth', 'acc', 'ray_delta_x', 'ray_predicted_mask', 'med_points'):
    assert torch.equal(out[k], whole[k].cpu()), k
  dev_out = render_image(state, rays, fn, device_count=1, rng=np.array([0, 3]), chunk=512, cfg=cfg, to_host=False)
  assert dev_out['rgb'].is_cuda and torch.equal(dev_out['rgb'].cpu(), out['rgb'])
  coarse = render_image(state, rays, fn, device_count=1, rng=None, chunk=700, cfg=cfg, default_ret_key='coarse')
  grid = lambda v: v[::6, ::7].reshape(-1, v.shape[-1])
  sub = {k: (grid(v) if k != 'metadata' else {'warp': grid(v['warp'])}) for k, v in rays.items()}
  ref = O.NerfModel(cfg, params).apply(sub, EXTRA, use_predicted_norm=True, compute_sigma_gradient=False)
  for level, got in (('fine', out), ('coarse', coarse)):
    e = float((grid(got['rgb']) - ref[level]['rgb'].float()).abs().max() / ref[level]['rgb'].abs().max())
    assert e <= 1e-4, (level, e)


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _rank_worker(rank, world, port, q, backend, shape=(23, 41), chunk=300):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
  import torch.distributed as dist
  gpu = rank if backend == 'nccl' else 0
  torch.cuda.set_device(gpu)
  if backend == 'nccl':
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', gpu))
  else:
    dist.init_process_group(backend, rank=rank, world_size=world)
  try:
    from nerfds_amd.evaluation import TrainState, make_model_fn, render_image
    from nerfds_amd.model import NerfModel
    cfg, params = _setup(False)
    rays = _frame_rays(shape[0], shape[1], 4, 1)         # default 943 rays: chunk 300 -> odd chunk sizes, padding on the last chunk
    model = NerfModel(cfg, device=torch.device('cuda', gpu), precision='f32')
    out = render_image(TrainState.create(params, **EXTRA), rays, make_model_fn(model, precision='f32'), device_count=world,
                       rng=np.array([0, 1]), chunk=chunk, cfg=cfg)
    q.put((rank, out['rgb'].numpy(), out['depth'].numpy()))
  finally:
    dist.destroy_process_group()


def _two_rank_frame(backend, world=2, shape=(23, 41), chunk=300):
  import torch.multiprocessing as mp
  from nerfds_amd.model import NerfModel
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_rank_worker, args=(r, world, port, q, backend, shape, chunk)) for r in range(world)]
  for p in procs:
    p.start()
  try:
    got = dict((r, (a, b)) for r, a, b in (q.get(timeout=600) for _ in range(world)))
  finally:
    for p in procs:
      p.join(60)
      if p.is_alive():
        p.kill()
  cfg, params = _setup(False)
  rays = _frame_rays(shape[0], shape[1], 4, 1)
  whole = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32').apply({'params': params}, rays, EXTRA, use_predicted_norm=True,
                                                                                 precision='f32')['fine']
  for r in range(world):     # every rank holds the whole frame, identical to the unsharded render
    assert np.array_equal(got[r][0], whole['rgb'].cpu().numpy()) and np.array_equal(got[r][1], whole['depth'].cpu().numpy())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='BASELINE config 3 needs two GPUs (ray shard + RCCL all-gather)')
def test_render_image_two_ranks_nccl():
  _two_rank_frame('nccl')


def test_render_image_two_ranks_sharing_one_gpu():
  """The N > 1 code path of render_image on a one-GPU box: two processes on cuda:0, chunk shards, staging buffers, side-stream
  exchange, padding of the ragged chunk - everything but RCCL itself (the exchange goes through gloo on host copies)."""
  _two_rank_frame('gloo')


def test_config3_eight_ranks_sharing_one_gpu():
  """BASELINE configs[2] at its own numbers without an 8-GPU node: chunk_size = 65 536 rays split over EIGHT ranks in contiguous blocks of
  8 192 (evaluation.py:97-129, render.py:155), two chunks (131 072 rays), every rank a process on cuda:0 with the exchange staged through
  gloo - the frame every rank ends up with equals the 1-rank render bit for bit (fp32 kernel)."""
  _two_rank_frame('gloo', world=8, shape=(256, 512), chunk=65536)


def test_bench_config3_eight_ranks_sharing_one_gpu():
  """`bench.py --gpus 8 --strong --rays 131072 --chunk 65536` as the driver launches it, eight ranks on cuda:0 over gloo: one JSON line."""
  import json
  import subprocess
  env = dict(os.environ, NERFDS_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1',
         '--rays', '131072', '--chunk', '65536', '--no-cpu-baseline', '--strong']
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, r.stdout
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 8 and rec['scaling'] == 'strong' and rec['value'] > 0
  assert abs(rec['value'] - 131072 / (rec['ms_per_step'] * 1e-3)) <= 1e-3 * rec['value']


@pytest.mark.parametrize('mode', ['weak', 'strong'])
def test_bench_two_ranks_sharing_one_gpu(mode):
  """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on cuda:0 with the
  exchange staged through gloo: the run finishes and rank 0 prints ONE JSON line with the aggregate of both ranks."""
  import json
  import subprocess
  env = dict(os.environ, NERFDS_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
         '--rays', '20000', '--chunk', '8192', '--no-cpu-baseline'] + (['--strong'] if mode == 'strong' else [])
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
  assert len(lines) == 1, r.stdout
  rec = json.loads(lines[0])
  assert rec['n_gpus'] == 2 and rec['scaling'] == mode and rec['value'] > 0
  total = 20000 * (2 if mode == 'weak' else 1)
  assert abs(rec['value'] - total / (rec['ms_per_step'] * 1e-3)) <= 1e-3 * rec['value']


def test_chunked_philox_frame_equals_one_call():
  """nerfds_rand.first_ray: a frame rendered in chunks with ray_offset draws the jitter one call over all rays draws."""
  from nerfds_amd.model import NerfModel
  cfg, params = _setup(True)
  rays = {k: (v.reshape(-1, v.shape[-1]) if k != 'metadata' else {'warp': v['warp'].reshape(-1, 1)}) for k, v in _frame_rays(9, 11, 4, 2).items()}
  m = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32')
  kw = dict(use_predicted_norm=True, precision='f32', rngs={'coarse': 5, 'fine': 6})
  whole = m.apply({'params': params}, rays, EXTRA, **kw)['fine']['rgb']
  parts = []
  for lo in range(0, 99, 40):
    sl = {k: (v[lo:lo + 40] if k != 'metadata' else {'warp': v['warp'][lo:lo + 40]}) for k, v in rays.items()}
    parts.append(m.apply({'params': params}, sl, EXTRA, ray_offset=lo, **kw)['fine']['rgb'])
  assert torch.equal(torch.cat(parts), whole)


def test_large_launch_and_64_bit_ray_counter():
  """One launch of 300 001 rays (more workgroup iterations than CUs, a ragged last group) starting at Philox ray counter 2^40 + 5
  equals the same rays rendered as two launches split at an odd ray: per-ray results do not depend on the launch they are in, and the
  ray counter is 64 bits wide end to end (a 32-bit counter would alias first_ray = 2^40 + 5 with 5)."""
  from nerfds_amd.model import NerfModel
  cfg, params = _setup(True)
  R = 300001
  g = torch.Generator(device='cpu').manual_seed(3)
  d = torch.nn.functional.normalize(torch.randn(R, 3, generator=g), dim=-1)
  rays = dict(origins=torch.randn(R, 3, generator=g) * 0.1, directions=d, viewdirs=d, metadata={'warp': torch.randint(0, 4, (R, 1), generator=g)})
  m = NerfModel(cfg, device=torch.device('cuda', 0), precision='bf16')
  kw = dict(use_predicted_norm=True, precision='bf16', rngs={'coarse': 5, 'fine': 6})
  base = (1 << 40) + 5
  whole = m.apply({'params': params}, rays, EXTRA, ray_offset=base, **kw)['fine']
  cut = 123457
  parts = []
  for lo, hi in ((0, cut), (cut, R)):
    sl = {k: (v[lo:hi] if k != 'metadata' else {'warp': v['warp'][lo:hi]}) for k, v in rays.items()}
    parts.append(m.apply({'params': params}, sl, EXTRA, ray_offset=base + lo, **kw)['fine'])
  for k in ('rgb', 'depth', 'acc'):
    assert torch.equal(torch.cat([p[k] for p in parts]), whole[k]), k
  low = m.apply({'params': params}, {k: (v[:64] if k != 'metadata' else {'warp': v['warp'][:64]}) for k, v in rays.items()}, EXTRA, ray_offset=5, **kw)['fine']
  assert not torch.equal(low['rgb'], whole['rgb'][:64])
  assert bool(torch.isfinite(whole['rgb']).all())


def test_trainer_draws_the_stratified_jitter_on_chip():
  """Trainer.step without injected uniforms (the reference always draws them, model_utils.py:84,217): the Philox stream of
  csrc/philox.h - every step other depths, the same depths as the render kernel for the same seed, one sample per stratum."""
  from nerfds_amd.model import NerfModel
  from nerfds_amd.training import Trainer
  cfg, params = _setup(True)
  R = 64
  f = _frame_rays(8, 8, 4, 3)
  batch = {k: (v.reshape(R, -1) if k != 'metadata' else {'warp': v['warp'].reshape(R, 1)}) for k, v in f.items()}
  batch['rgb'] = np.random.default_rng(0).random((R, 3)).astype(np.float32)
  tr = Trainer(cfg, params, max_rays=R)
  a = tr.step(batch, EXTRA, 0.0, grads_only=True)['loss/total']
  b = tr.step(batch, EXTRA, 0.0, grads_only=True)['loss/total']
  c = tr.step(batch, EXTRA, 0.0, grads_only=True, seed=1234)['loss/total']
  d = tr.step(batch, EXTRA, 0.0, grads_only=True, seed=1234)['loss/total']
  assert a != b and c == d                                # new jitter every step unless the seed is pinned
  # same seed, same rays -> the fused render kernel composites the same depths: its MSE equals the trainer's loss
  m = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32')
  import nerfds_amd.model as M
  old = M._seed_from_rngs
  M._seed_from_rngs = lambda rngs: 1234
  try:
    out = m.apply({'params': params}, batch, EXTRA, use_predicted_norm=True, precision='f32', rngs=None, return_samples=True)
  finally:
    M._seed_from_rngs = old
  tgt = torch.as_tensor(batch['rgb'], device='cuda')
  mse = float(((out['fine']['rgb'] - tgt) ** 2).mean() + ((out['coarse']['rgb'] - tgt) ** 2).mean())
  assert abs(mse - c) < 2e-5 * max(1.0, mse), (mse, c)
  z = out['coarse']['z_vals'].cpu().numpy()
  edges = np.linspace(cfg.near, cfg.far, 16)
  mids = 0.5 * (edges[1:] + edges[:-1])
  assert np.all(z >= np.r_[edges[0], mids] - 1e-6) and np.all(z <= np.r_[mids, edges[-1]] + 1e-6) and np.ptp(z, axis=0).min() > 0


def test_target_norm_on_the_render_surface():
  """NerfModel.apply(return_target_norm=True): out[level]['target_norm'] (models.py:1065-1077, 1328) through the trainer's reverse pass, in blocks,
  against the oracle's autograd.  The reference's keyword use_sigma_gradient=True means something else (the rgb branch reads stop_gradient(d sigma /
  d x) and asserts not use_predicted_norm, models.py:1107-1112): it is rejected, not re-purposed."""
  from nerfds_amd.model import NerfModel
  from oracle import nerfds_oracle as O
  cfg = nerf_ds_config(num_warp_embeds=4, num_coarse_samples=12, num_fine_samples=12)
  params = init_params(cfg, 3, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  f = _frame_rays(5, 6, 4, 4)
  rng = np.random.default_rng(1)
  t, u = rng.random((30, 12)), rng.random((30, 12))
  flat = {k: (v.reshape(30, -1) if k != 'metadata' else {'warp': v['warp'].reshape(30, 1)}) for k, v in f.items()}
  ref = O.NerfModel(cfg, params).apply(flat, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=True)
  m = NerfModel(cfg, device=torch.device('cuda', 0), precision='f32')
  m.sigma_gradient_block = 16                            # two blocks of 16 / 14 rays
  with pytest.raises(NotImplementedError):
    m.apply({'params': params}, f, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, use_sigma_gradient=True, precision='f32')
  out = m.apply({'params': params}, f, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, return_target_norm=True, precision='f32')
  for level, S in (('coarse', 12), ('fine', 24)):
    got = out[level]['target_norm'].cpu().numpy()
    assert got.shape == (5, 6, S, 3)
    cos = (got.reshape(30, S, 3) * ref[level]['target_norm'].numpy()).sum(-1)
    # quantile bound (measured: q99 = 6e-8, worst sample 1.3e-2 - a sample whose gradient is tiny, where normalisation amplifies fp32 rounding)
    assert np.quantile(1 - cos, 0.99) < 1e-6 and float((1 - cos > 1e-4).mean()) <= 0.01 and float((1 - cos).max()) < 5e-2, level
    assert float((out[level]['rgb'].cpu() - ref[level]['rgb'].reshape(5, 6, 3).float()).abs().max()) <= 1e-4


def test_parity_modes_on_a_badly_conditioned_scene():
  """north_star's 1e-4 beyond the bench's own frame (round 6, tools/parity_sweep.py): scene (seed 15, 309 GLO rows, near 0.3, far 2.0) of BASELINE configs[4]'s seven has a
  few dozen rays on which the fp32-MFMA kernel itself sits at 8e-5 of the fp64 oracle - split bf16 (16 significand bits per operand) leaves the tolerance there (1.2e-3 on the
  coarse level), split f16 (22 bits, NERFDS_PREC_F16X3, the same three MFMAs per product) holds it on both levels over every ray of the 800 x 600 frame."""
  import bench
  from nerfds_amd.model import NerfModel
  dev = torch.device('cuda', 0)
  cfg = nerf_ds_config(near=0.3, far=2.0, num_warp_embeds=309)
  params = init_params(cfg, 15, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 480000
  rays = bench.synth_rays(R, 309, 15, dev)
  m = NerfModel(cfg, device=dev, precision='f32')
  rec = {p: {lv: torch.empty((R, 26), device=dev) for lv in ('fine', 'coarse')} for p in ('f32', 'bf16x3', 'f16x3')}
  for p in rec:
    for lo in range(0, R, 65536):
      hi = min(lo + 65536, R)
      sl = {k: (v[lo:hi] if not isinstance(v, dict) else {'warp': v['warp'][lo:hi]}) for k, v in rays.items()}
      m.apply({'params': params}, sl, EXTRA, rngs={'coarse': 15, 'fine': 515}, ray_offset=lo, use_predicted_norm=True, precision=p,
              records_out={lv: rec[p][lv][lo:hi] for lv in ('fine', 'coarse')})
  torch.cuda.synchronize()
  err = {}
  for p in ('bf16x3', 'f16x3'):
    for lv in ('fine', 'coarse'):
      ref, got = rec['f32'][lv][:, :3], rec[p][lv][:, :3]
      assert bool(torch.isfinite(got).all())
      d = (got - ref).abs().max(dim=1).values / ref.abs().max()
      err[p, lv] = (float(d.max()), int((d > 1e-4).sum()))
      print(f'scene 15, {lv}: {p} vs f32 kernel over {R} rays: max {err[p, lv][0]:.3e}, {err[p, lv][1]} rays over 1e-4')
  assert err['f16x3', 'fine'][0] <= 1e-4 and err['f16x3', 'coarse'][0] <= 1e-4, err
  # (recorded, not required: this is what split bf16 does on the scene - the reason the mode above exists)
  assert err['bf16x3', 'coarse'][0] > err['f16x3', 'coarse'][0]


def test_full_frame_error_of_the_parity_path():
  """north_star: "within 1e-4 rel on composited RGB".  The split-bf16 kernel against the fp32-MFMA kernel over EVERY ray of the
  800x600 frame the metric is quoted on (the max-over-rays statistic grows with the ray count), trained-regime weights, on-chip
  Philox jitter - and both against the CPU oracle on a 2048-ray sub-sample of that frame with the same injected uniforms."""
  import bench
  from nerfds_amd.model import NerfModel
  from oracle import nerfds_oracle as O
  dev = torch.device('cuda', 0)
  cfg = nerf_ds_config(near=0.3, far=1.7, num_warp_embeds=256)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 480000
  rays = bench.synth_rays(R, cfg.num_warp_embeds, 100, dev)
  m = NerfModel(cfg, device=dev, precision='bf16x3')
  rec = {p: {lv: torch.empty((R, 26), device=dev) for lv in ('fine', 'coarse')} for p in ('f32', 'bf16x3', 'f16x3')}
  for p in rec:
    for lo in range(0, R, 65536):
      hi = min(lo + 65536, R)
      sl = {k: (v[lo:hi] if not isinstance(v, dict) else {'warp': v['warp'][lo:hi]}) for k, v in rays.items()}
      m.apply({'params': params}, sl, EXTRA, rngs={'coarse': 7, 'fine': 507}, ray_offset=lo, use_predicted_norm=True, precision=p,
              records_out={lv: rec[p][lv][lo:hi] for lv in ('fine', 'coarse')})
  torch.cuda.synchronize()
  for p in ('bf16x3', 'f16x3'):
    for lv in ('fine', 'coarse'):
      ref, got = rec['f32'][lv][:, :3], rec[p][lv][:, :3]
      assert bool(torch.isfinite(got).all())
      err = float((got - ref).abs().max() / ref.abs().max())
      print(f'full-frame {lv}: {p} vs f32 kernel over {R} rays: {err:.3e}')
      assert err <= 1e-4, (p, lv, err)
  idx = torch.arange(0, R, R // 2048, device=dev)[:2048]
  sub = {k: (v[idx] if not isinstance(v, dict) else {'warp': v['warp'][idx]}) for k, v in rays.items()}
  rng = np.random.default_rng(0)
  t, u = rng.random((2048, 64)), rng.random((2048, 64))
  cpu = {k: (v.cpu() if not isinstance(v, dict) else {'warp': v['warp'].cpu()}) for k, v in sub.items()}
  torch.set_num_threads(min(bench.available_cores(), 64))
  ref = O.NerfModel(cfg, params, torch.float32).apply(cpu, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=False)
  for p in ('f32', 'bf16x3', 'f16x3'):
    out = m.apply({'params': params}, sub, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, precision=p)
    for lv in ('fine', 'coarse'):
      err = float((out[lv]['rgb'].cpu() - ref[lv]['rgb']).abs().max() / ref[lv]['rgb'].abs().max())
      print(f'oracle sub-sample {lv}: {p} {err:.3e}')
      assert err <= 1e-4, (p, lv, err)


@pytest.mark.gpu
def test_fine_level_parity_with_a_one_mfma_coarse_nerfmlp():
  """precision='bf16x3_fine' (NERFDS_PREC_BF16X3_FINE): split bf16 everywhere except the COARSE level's NerfMLP, which runs one f16 MFMA per product.
  The fine level - the one render_fn returns (evaluation.py:121-124) - sees of the coarse NerfMLP only the compositing weights its depths are drawn
  from (model_utils.py:193-269): held to north_star's 1e-4 over EVERY ray of the 800 x 600 frame against the fp32-MFMA kernel, and against the CPU
  oracle on a 2048-ray sub-sample with injected uniforms; the fine DEPTHS themselves move by what an f16-grade pdf moves them.  The coarse level's
  own composited RGB is f16-grade and bounded as such (RTOL of the f16 kernel, tests/test_gpu_parity.py) - it is NOT a parity-grade output in this
  mode, and the test says so.  The level-independent networks stay split bf16 (their results at the coarse positions are reused by the fine level)."""
  import bench
  from nerfds_amd.model import NerfModel
  from oracle import nerfds_oracle as O
  dev = torch.device('cuda', 0)
  cfg = nerf_ds_config(near=0.3, far=1.7, num_warp_embeds=256)
  params = init_params(cfg, 0, warp_head_scale=5e-2, small_head_scale=0.3, bias_scale=0.1)
  R = 480000
  rays = bench.synth_rays(R, cfg.num_warp_embeds, 100, dev)
  m = NerfModel(cfg, device=dev, precision='bf16x3_fine')
  rec = {p: {lv: torch.empty((R, 26), device=dev) for lv in ('fine', 'coarse')} for p in ('f32', 'bf16x3_fine')}
  for p in rec:
    for lo in range(0, R, 65536):
      hi = min(lo + 65536, R)
      sl = {k: (v[lo:hi] if not isinstance(v, dict) else {'warp': v['warp'][lo:hi]}) for k, v in rays.items()}
      m.apply({'params': params}, sl, EXTRA, rngs={'coarse': 7, 'fine': 507}, ray_offset=lo, use_predicted_norm=True, precision=p,
              records_out={lv: rec[p][lv][lo:hi] for lv in ('fine', 'coarse')})
  torch.cuda.synchronize()
  errs = {}
  for lv in ('fine', 'coarse'):
    ref, got = rec['f32'][lv][:, :3], rec['bf16x3_fine'][lv][:, :3]
    assert bool(torch.isfinite(got).all())
    errs[lv] = float((got - ref).abs().max() / ref.abs().max())
    print(f'full-frame {lv}: bf16x3_fine vs f32 kernel over {R} rays: {errs[lv]:.3e}')
  assert errs['fine'] <= 1e-4, errs                   # what render_fn returns: parity grade
  assert errs['coarse'] <= 2.4e-3, errs               # f16-grade, as the f16 kernel's bound
  idx = torch.arange(0, R, R // 2048, device=dev)[:2048]
  sub = {k: (v[idx] if not isinstance(v, dict) else {'warp': v['warp'][idx]}) for k, v in rays.items()}
  rng = np.random.default_rng(0)
  t, u = rng.random((2048, 64)), rng.random((2048, 64))
  cpu = {k: (v.cpu() if not isinstance(v, dict) else {'warp': v['warp'].cpu()}) for k, v in sub.items()}
  torch.set_num_threads(min(bench.available_cores(), 64))
  ref = O.NerfModel(cfg, params, torch.float32).apply(cpu, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, compute_sigma_gradient=False)
  out = m.apply({'params': params}, sub, EXTRA, t_rand=t, u_rand=u, use_predicted_norm=True, precision='bf16x3_fine')
  for lv, bound in (('fine', 1e-4), ('coarse', 2.4e-3)):
    for k in ('rgb', 'depth', 'acc'):
      err = float((out[lv][k].cpu() - ref[lv][k]).abs().max() / ref[lv][k].abs().max())
      print(f'oracle sub-sample {lv} {k}: bf16x3_fine {err:.3e}')
      assert err <= (bound if k == 'rgb' else 10 * bound), (lv, k, err)
  # a single-level model has no coarse pass to run cheaply: the mode is plain split bf16 there, bit for bit
  from nerfds_amd import static_config
  cfg1 = static_config()
  p1 = init_params(cfg1, 3, bias_scale=0.1)
  m1 = NerfModel(cfg1, device=dev)
  r1 = {k: (v[:512] if not isinstance(v, dict) else {'warp': v['warp'][:512]}) for k, v in rays.items()}
  t1 = rng.random((512, cfg1.num_coarse_samples))
  a = m1.apply({'params': p1}, r1, EXTRA, t_rand=t1, use_predicted_norm=False, precision='bf16x3_fine')
  b = m1.apply({'params': p1}, r1, EXTRA, t_rand=t1, use_predicted_norm=False, precision='bf16x3')
  assert torch.equal(a['coarse']['rgb'], b['coarse']['rgb'])
