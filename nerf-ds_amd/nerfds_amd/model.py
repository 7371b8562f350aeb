"""Host-side mirror of the reference's model surface, backed by the fused HIP ray kernel.

``NerfModel.apply`` keeps the call shape of ``model.apply({'params': params}, rays_dict, extra_params=...,
rngs=..., use_predicted_norm=..., return_points=..., mask_ratio=..., sharp_weights_std=...)`` as used by the
reference's render.py:140-154 / training.py:441-455 on ``NerfModel.__call__`` (hypernerf/models.py:1419-1565),
and returns the same ``{'coarse': {...}, 'fine': {...}}`` dictionary of per-ray maps.  ``construct_nerf``
mirrors hypernerf/models.py:2677-2741.

Deviations, all deliberate (DESIGN.md "Out of scope"):
  * per-sample tensors (sigma, alpha, weights, ...) are produced only when asked for
    (``return_weights`` / ``return_points`` / ``return_samples``): the reference always returns ~15
    per-sample arrays that render.py then throws away (render.py:192-193);
  * ``target_norm`` (d sigma / d x, models.py:1065-1077) is produced on request only (``return_target_norm=True``, a keyword of this mirror: the
    rays go through the trainer's reverse pass in blocks, ``_target_norm`` below); the render kernel itself does not compute it because render.py
    discards it.  The reference's own keyword ``use_sigma_gradient=True`` means something else - the rgb branch reads stop_gradient(d sigma / d x)
    instead of the predicted normal, and it asserts ``not use_predicted_norm`` (models.py:1107-1112); no shipped gin sets it (defaults.gin:176) and
    it is rejected here;
  * sampling randomness: JAX threefry streams cannot be reproduced, so either inject ``t_rand`` / ``u_rand``
    or an on-chip Philox4x32 stream keyed by the integer seed derived from ``rngs`` is used.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _native as N
from .config import NerfModelConfig
from .params import init_params, levels, mlp_layer_dims


def _cfg_struct(cfg: NerfModelConfig) -> N.ModelCfg:
  def skip(s):
    return int(s[0]) if len(s) == 1 else -1
  c = N.ModelCfg()
  c.abi_version = N.ABI_VERSION
  c.num_coarse_samples, c.num_fine_samples = cfg.num_coarse_samples, cfg.num_fine_samples
  c.use_warp = int(cfg.use_warp)
  c.use_hyper_sheet = int(cfg.has_hyper and cfg.use_hyper)
  c.use_predicted_mask = int(cfg.use_predicted_mask)
  c.predict_norm = int(cfg.predict_norm)
  c.use_x_in_rgb_condition = int(cfg.use_x_in_rgb_condition)
  c.use_mask_in_warp, c.use_mask_in_hyper = int(cfg.use_mask_in_warp), int(cfg.use_mask_in_hyper)
  c.use_viewdirs, c.mask_output_relu = int(cfg.use_viewdirs), int(cfg.mask_output_relu)
  c.nerf_trunk_depth, c.nerf_trunk_width, c.nerf_skip = cfg.nerf_trunk_depth, cfg.nerf_trunk_width, skip(cfg.nerf_skips)
  c.nerf_rgb_branch_depth, c.nerf_rgb_branch_width = cfg.nerf_rgb_branch_depth, cfg.nerf_rgb_branch_width
  c.spatial_point_max_deg, c.hyper_point_max_deg = cfg.spatial_point_max_deg, cfg.hyper_point_max_deg
  c.viewdir_max_deg, c.norm_input_max_deg = cfg.viewdir_max_deg, cfg.norm_input_max_deg
  c.warp_max_deg, c.warp_trunk_depth = cfg.warp_max_deg, cfg.warp_trunk.depth
  c.warp_trunk_width, c.warp_skip = cfg.warp_trunk.width, skip(cfg.warp_trunk.skips)
  c.hyper_sheet_max_deg, c.hyper_sheet_depth = cfg.hyper_sheet_max_deg, cfg.hyper_sheet_mlp.depth
  c.hyper_sheet_width, c.hyper_sheet_skip = cfg.hyper_sheet_mlp.width, skip(cfg.hyper_sheet_mlp.skips)
  c.hyper_num_dims = cfg.hyper_sheet_output_channels
  c.mask_max_deg, c.mask_depth = cfg.mask_max_deg, cfg.mask_mlp.depth
  c.mask_width, c.mask_skip = cfg.mask_mlp.width, skip(cfg.mask_mlp.skips)
  c.glo_num_dims, c.num_warp_embeds = cfg.glo_num_dims, cfg.num_warp_embeds
  c.use_white_background, c.use_sample_at_infinity = int(cfg.use_white_background), int(cfg.use_sample_at_infinity)
  for name in ('spatial_point_min_deg', 'hyper_point_min_deg', 'viewdir_min_deg', 'norm_input_min_deg',
               'warp_min_deg', 'hyper_sheet_min_deg', 'mask_min_deg'):
    if getattr(cfg, name) != 0:
      raise NotImplementedError(f'{name} != 0 is not configured by any shipped gin file')
  c.use_posenc_identity, c.warp_use_posenc_identity = int(cfg.use_posenc_identity), int(cfg.warp_use_posenc_identity)
  return c


class _WeightsHolder:
  """Builds the nerfds_weights view over a Flax-style numpy tree and keeps the arrays alive."""

  def __init__(self, cfg: NerfModelConfig, params: Dict[str, Any]):
    self.keep = []
    self.struct = N.Weights()
    w = self.struct

    def arr(a):
      a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
      self.keep.append(a)
      return a

    def dense(dst, p):
      k, b = arr(p['kernel']), arr(p['bias'])
      if k.ndim != 2 or b.shape != (k.shape[1],):
        raise ValueError(f'Dense parameters must be a [in, out] kernel and an [out] bias, got {k.shape} and {b.shape}')
      dst.kernel, dst.bias = k.ctypes.data, b.ctypes.data
      dst.in_dim, dst.out_dim = k.shape

    def table(p, name):
      # the library copies num_warp_embeds * glo_num_dims floats from this pointer: a smaller table would be over-read
      e = arr(p['embed']['embedding'])
      if e.shape != (cfg.num_warp_embeds, cfg.glo_num_dims):
        raise ValueError(f'{name}/embed/embedding has shape {e.shape}, the model configuration says '
                         f'({cfg.num_warp_embeds}, {cfg.glo_num_dims}) (num_warp_embeds, glo_num_dims)')
      return e

    def mlp(dst_hidden, tree, depth):
      for i in range(depth):
        dense(dst_hidden[i], tree[f'hidden_{i}'])

    if cfg.use_warp:
      w.warp_embed = table(params['warp_embed'], 'warp_embed').ctypes.data
      w.embed_rows = cfg.num_warp_embeds
      mlp(w.warp_hidden, params['warp_field']['trunk'], cfg.warp_trunk.depth)
      dense(w.warp_w, params['warp_field']['branches_w']['logit'])
      dense(w.warp_v, params['warp_field']['branches_v']['logit'])
    if cfg.use_predicted_mask:
      w.mask_embed = table(params['mask_embed'], 'mask_embed').ctypes.data
      w.embed_rows = cfg.num_warp_embeds
      mlp(w.mask_hidden, params['mask_mlp']['MLP_0'], cfg.mask_mlp.depth)
      dense(w.mask_out, params['mask_mlp']['MLP_0']['logit'])
    if cfg.has_hyper:
      mlp(w.hyper_hidden, params['hyper_sheet_mlp']['MLP_0'], cfg.hyper_sheet_mlp.depth)
      dense(w.hyper_out, params['hyper_sheet_mlp']['MLP_0']['logit'])
    for li, level in enumerate(levels(cfg)):
      t = params[f'nerf_mlps_{level}']
      mlp(w.nerf[li].trunk, t['trunk_mlp'], cfg.nerf_trunk_depth)
      dense(w.nerf[li].bottleneck, t['bottleneck'])
      dense(w.nerf[li].alpha, t['alpha_mlp']['logit'])
      mlp(w.nerf[li].rgb_hidden, t['rgb_mlp'], cfg.nerf_rgb_branch_depth)
      dense(w.nerf[li].rgb, t['rgb_mlp']['logit'])


def _seed_from_rngs(rngs) -> int:
  if rngs is None:
    return 0
  if isinstance(rngs, dict):
    vals = [rngs.get(k) for k in ('coarse', 'fine')]
  else:
    vals = [rngs]
  seed = 0
  for v in vals:
    if v is None:
      continue
    a = np.asarray(v.cpu() if isinstance(v, torch.Tensor) else v).astype(np.uint64).ravel()
    for x in a:
      seed = (seed * 0x9E3779B97F4A7C15 + int(x) + 1) & 0xFFFFFFFFFFFFFFFF
  return seed


class NerfModel:
  """Drop-in for the reference's ``NerfModel`` on one MI355X: functional ``apply``, parameters passed in."""

  def __init__(self, cfg: NerfModelConfig, device: Optional[torch.device] = None, precision: str = 'f16x3'):
    cfg.validate()
    self.cfg = cfg
    self.precision = precision
    self._lib = N.load()                      # raises if the HIP library is not built: no fallback
    if not torch.cuda.is_available():
      raise RuntimeError('NerfModel needs an MI355X (torch.cuda is not available); there is no CPU path')
    self.device = N.resolve_device(device)
    self._cstruct = _cfg_struct(cfg)
    ctx = C.c_void_p()
    rc = self._lib.nerfds_ctx_create(C.byref(ctx), self.device.index, C.byref(self._cstruct))
    if rc != 0:
      msg = N.last_error(None)
      if rc == -95:
        raise NotImplementedError(msg)
      raise RuntimeError(f'nerfds_ctx_create failed ({rc}): {msg}')
    self._ctx = ctx
    self._params_ref = None       # the parameter tree the weight streams were packed from (strong reference)
    self._holder = None

  def __del__(self):
    ctx = getattr(self, '_ctx', None)
    if ctx:
      self._lib.nerfds_ctx_destroy(ctx)
      self._ctx = None

  # -- parameters ---------------------------------------------------------------------------------------
  def init(self, seed: int = 0, **kw) -> Dict[str, Any]:
    """``model.init`` of construct_nerf (models.py:2707-2739): a fresh Flax-named parameter tree."""
    return init_params(self.cfg, seed, **kw)

  def load_params(self, params: Dict[str, Any]) -> None:
    """Packs ``params`` (the 'model' sub-tree of the checkpoint) into the MFMA weight streams.  ``apply`` calls this
    when it is handed a different tree OBJECT than the one packed last (the tree is kept alive, so an ``is`` test cannot
    be fooled by a recycled address); leaves updated in place are not detected - call ``load_params`` again (or pass
    ``reload_params=True`` to ``apply``)."""
    holder = _WeightsHolder(self.cfg, params)
    rc = self._lib.nerfds_ctx_load_weights(self._ctx, C.byref(holder.struct))
    if rc != 0:
      raise ValueError(f'nerfds_ctx_load_weights failed ({rc}): {N.last_error(self._ctx)}')
    self._holder = holder
    self._params_ref = params

  # -- NerfModel.__call__ -------------------------------------------------------------------------------
  def apply(self, variables: Dict[str, Any], rays_dict: Dict[str, Any], extra_params: Dict[str, Any], *,
            rngs=None, mutable=False, metadata_encoded=False, use_warp=True, return_points=False,
            return_weights=False, return_samples=False, return_warp_jacobian=False, return_hyper_jacobian=False,
            return_hyper_c_jacobian=False, return_nv_details=True, near=None, far=None,
            use_sample_at_infinity=None, render_opts=None, deterministic=False, screw_input_mode=None,
            use_sigma_gradient=False, use_predicted_norm=False, return_target_norm=False,
            mask_ratio=1, sharp_weights_std=1.0, x_for_rgb_alpha=4.0, norm_override=None,
            t_rand=None, u_rand=None, precision: Optional[str] = None,
            stream: Optional[torch.cuda.Stream] = None, reload_params: bool = False, ray_offset: int = 0,
            records_out: Optional[Dict[str, torch.Tensor]] = None):
    """NerfModel.__call__ (models.py:1419-1565), every keyword of the reference's signature (models.py:1419-1443); DESIGN.md section 2 lists what
    each one reaches per level.  Keywords whose non-default value selects a path no shipped config or caller uses are rejected loudly:
    return_*_jacobian (the trainer's elastic loss carries the warp Jacobian, csrc/nerfds_train.cpp), screw_input_mode other than None / 'none'
    (models.py:554-561), norm_override (a debugging hack: norm_input * -2, models.py:1139-1140).  ``deterministic`` is accepted and ignored, as in
    the reference (it is not forwarded to render_samples); ``x_for_rgb_alpha`` is read only under window_x_in_rgb_condition (models.py:1202-1206),
    which config.py rejects.

    ``records_out``: optional {'fine' | 'coarse': float32 device tensor [R, 26]} the per-ray records of that level are
    written into (e.g. a slice of a frame buffer) instead of a fresh allocation.  ``ray_offset``: Philox counter of the first
    ray (csrc/philox.h), so that a frame rendered in chunks draws what one call over all its rays would."""
    cfg = self.cfg
    nf = cfg.num_fine_samples
    params = variables['params'] if 'params' in variables else variables
    reloaded = reload_params or params is not self._params_ref
    if reloaded:
      self.load_params(params)
    if render_opts is not None and cfg.use_mask_sharp_weights and (return_weights or return_samples or return_points):
      # filter_sigma runs on the RAW density on that path (models.py:1236-1237) and on the activated one for compositing (models.py:1288);
      # sharp_weights is rebuilt on the host from the composited weights, which is the same thing only without render_opts
      raise NotImplementedError('render_opts together with per-sample outputs of a use_mask_sharp_weights model (sharp_weights)')
    if use_sigma_gradient:
      # models.py:1107-1112: norm_input = stop_gradient(sigma_gradient) feeds the rgb branch INSTEAD of the predicted normal, and the reference asserts
      # `not use_predicted_norm` - a different graph from the one the fused kernel evaluates; configs/defaults.gin:176 leaves it False everywhere
      raise NotImplementedError('use_sigma_gradient=True (the rgb branch conditioned on stop_gradient(d sigma / d x), models.py:1107-1112) is not built; '
                                "for out[level]['target_norm'] pass return_target_norm=True")
    if return_target_norm and not (cfg.use_warp and cfg.use_predicted_mask and cfg.has_hyper and nf > 0):
      raise NotImplementedError('return_target_norm=True (target_norm, models.py:1065-1077) is built for the nerf_ds graph: '
                                'the reverse pass lives in the trainer, which covers only that graph')
    if bool(use_predicted_norm) != bool(cfg.predict_norm):
      raise ValueError('use_predicted_norm must equal NerfModel.predict_norm: the rgb branch width depends on it '
                       '(models.py:2707-2739 initialises the parameters with the same flag)')
    # use_sample_at_infinity: per-call override of the FINE level only (models.py:1484-1485, 1544); the coarse level, and a single-level
    # model, composite with cfg.use_sample_at_infinity (models.py:1509)
    inf_cfg = bool(cfg.use_sample_at_infinity)
    inf_fine = inf_cfg if (use_sample_at_infinity is None or nf == 0) else bool(use_sample_at_infinity)
    if not use_warp and cfg.use_warp:
      raise NotImplementedError('use_warp=False on a warp model raises in the reference too (SURVEY.md 8a quirk 2)')
    if return_warp_jacobian or return_hyper_jacobian or return_hyper_c_jacobian:
      raise NotImplementedError('return_*_jacobian on the render surface (the elastic loss of Trainer.step carries the warp Jacobian)')
    if screw_input_mode not in (None, 'none', 'None'):
      raise NotImplementedError(f'screw_input_mode={screw_input_mode!r}: the rgb branch has no screw-axis condition (models.py:554-561)')
    if norm_override is not None:
      raise NotImplementedError('norm_override (models.py:1139-1140)')
    del deterministic, x_for_rgb_alpha

    dev = self.device
    f32 = lambda a: torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a).to(dev, torch.float32).contiguous()
    camera, cam_struct, first_pixel = rays_dict.get('camera'), None, 0
    if camera is not None:
      # Fused camera -> rays: rays_dict = {'camera': Camera, 'pixel_range': (first, count), 'metadata', 'mask'};
      # origins / directions / viewdirs are generated on chip for the row-major pixel centres of the range.
      H, W = camera.image_shape
      first_pixel, R = rays_dict.get('pixel_range', (0, H * W))
      batch_shape = (H, W) if (first_pixel, R) == (0, H * W) else (R,)
      origins = directions = viewdirs = None
      cam_struct = camera._struct()
    else:
      origins, directions = f32(rays_dict['origins']), f32(rays_dict['directions'])
      batch_shape = origins.shape[:-1]
      origins, directions = origins.reshape(-1, 3), directions.reshape(-1, 3)
      R = origins.shape[0]
      viewdirs = f32(rays_dict['viewdirs']).reshape(-1, 3) if 'viewdirs' in rays_dict else None
    warp_id = enc_warp = enc_mask = None
    metadata = rays_dict.get('metadata', {})
    if metadata_encoded:
      # models.py:898-899, 908-909: per-ray GLO vectors from evaluation.encode_metadata instead of ids
      if cfg.use_warp:
        enc_warp = f32(metadata['encoded_warp']).reshape(-1, 8)
        if cfg.has_hyper and 'encoded_hyper' in metadata and metadata['encoded_hyper'] is not metadata['encoded_warp']:
          if not torch.equal(f32(metadata['encoded_hyper']).reshape(enc_warp.shape), enc_warp):
            raise NotImplementedError('encoded_hyper != encoded_warp: every built graph has hyper_use_warp_embed (models.py:296-319)')
        if enc_warp.shape[0] != R:
          raise ValueError(f'encoded_warp has {enc_warp.shape[0]} rows for {R} rays')
      if cfg.use_predicted_mask and metadata.get('encoded_mask') is not None:
        enc_mask = f32(metadata['encoded_mask']).reshape(-1, 8)
    if cfg.use_warp and (not metadata_encoded or (cfg.use_predicted_mask and enc_mask is None)):
      ids = metadata['warp']          # (the mask network looks its embedding up from the ids even when metadata_encoded: models.py:924-926)
      ids = ids if isinstance(ids, torch.Tensor) else torch.as_tensor(np.asarray(ids).astype(np.int64))
      # out-of-range ids are clamped in the kernel, as a jnp gather does (nn.Embed, modules.py:331-348)
      warp_id = ids.to(dev).reshape(-1).to(torch.int32).contiguous()   # uint32 bit pattern for ids < 2^31
    gt_mask = None
    if rays_dict.get('mask') is not None:
      gt_mask = f32(rays_dict['mask']).reshape(-1)
    want_samples = bool(return_samples or return_weights or return_points)
    nc, nf = cfg.num_coarse_samples, cfg.num_fine_samples
    two = nf > 0
    def rec_buf(level):
      buf = (records_out or {}).get(level)
      if buf is None:
        return torch.empty((R, N.RAY_REC), device=dev, dtype=torch.float32)
      if buf.shape != (R, N.RAY_REC) or buf.dtype != torch.float32 or buf.device != dev or not buf.is_contiguous():
        raise ValueError(f'records_out[{level!r}] must be a contiguous float32 [{R}, {N.RAY_REC}] tensor on {dev}')
      return buf
    rec_fine = rec_buf('fine' if two else 'coarse')            # the finest level of the model
    rec_coarse = rec_buf('coarse') if two else None
    smp_fine = torch.empty((R, nc + nf, N.SAMPLE_REC), device=dev, dtype=torch.float32) if want_samples else None
    smp_coarse = torch.empty((R, nc, N.SAMPLE_REC), device=dev, dtype=torch.float32) if (want_samples and two) else None
    if t_rand is not None:
      t_rand = f32(t_rand).reshape(R, nc)
    if u_rand is not None:
      u_rand = f32(u_rand).reshape(R, nf) if nf > 0 else None      # a single-level model draws no fine samples

    ptr = lambda t: (t.data_ptr() if t is not None else None)
    rays = N.Rays(R, ptr(origins), ptr(directions), ptr(viewdirs), ptr(warp_id), ptr(gt_mask),
                  C.pointer(cam_struct) if cam_struct is not None else None, int(first_pixel), ptr(enc_warp), ptr(enc_mask))
    g = lambda k, d=0.0: float(extra_params[k]) if extra_params.get(k) is not None else d
    extra = N.Extra(g('nerf_alpha'), g('warp_alpha'), g('hyper_alpha'), g('hyper_sheet_alpha'), g('norm_input_alpha'),
                    float(mask_ratio), float(cfg.near if near is None else near), float(cfg.far if far is None else far),
                    int(cfg.use_stratified_sampling))
    N.set_render_opts(extra, render_opts)
    extra.use_linear_disparity = int(cfg.use_linear_disparity)
    extra.sample_at_infinity_override = N.TRISTATE_NONE if use_sample_at_infinity is None else \
        (N.TRISTATE_TRUE if use_sample_at_infinity else N.TRISTATE_FALSE)
    rnd = N.Rand(ptr(t_rand), ptr(u_rand), _seed_from_rngs(rngs), int(ray_offset))
    out = N.Out(ptr(rec_fine), ptr(rec_coarse), ptr(smp_fine), ptr(smp_coarse))
    flags = N.PREC[precision or self.precision]
    s = stream if stream is not None else torch.cuda.current_stream(dev)
    rc = self._lib.nerfds_render_rays(self._ctx, C.byref(rays), C.byref(extra), C.byref(rnd), C.byref(out), flags,
                                      C.c_void_p(s.cuda_stream))
    if rc != 0:
      raise RuntimeError(f'nerfds_render_rays failed ({rc}): {N.last_error(self._ctx)}')

    ret = {}
    level_recs = [('coarse', rec_coarse, smp_coarse, nc), ('fine', rec_fine, smp_fine, nc + nf)] if two else \
                 [('coarse', rec_fine, smp_fine, nc)]
    for level, rec, smp, S in level_recs:
      if smp is not None and camera is not None:
        from .camera import camera_to_rays
        cr = camera_to_rays(camera, dev)
        origins = cr['origins'].reshape(-1, 3)[first_pixel:first_pixel + R]
        directions = cr['directions'].reshape(-1, 3)[first_pixel:first_pixel + R]
      ret[level] = self._unpack(rec, smp, S, batch_shape, origins, directions, return_points, return_weights,
                                sharp_weights_std, inf_fine if (two and level == 'fine') else inf_cfg)
    if return_target_norm:
      # target_norm = normalize(R normalize(-d sigma / d x)) per sample (models.py:1065-1077, 1273-1277, 1328).  The fused render
      # kernel does not carry tangents; the trainer's forward-mode pass does (csrc/nerfds_train.cpp sigma_gradient), on the same
      # depths: injected uniforms are passed on, and the on-chip Philox stream is keyed identically in both (csrc/philox.h).
      if camera is not None:
        raise NotImplementedError('return_target_norm with fused camera rays: pass origins / directions')
      if metadata_encoded or render_opts is not None or inf_fine != inf_cfg:
        raise NotImplementedError('return_target_norm with metadata_encoded / render_opts / a use_sample_at_infinity override '
                                  '(the tangent pass is the trainer\'s: ids, no render_opts, the model\'s use_sample_at_infinity)')
      tn = self._target_norm(params, reloaded, origins, directions, viewdirs, warp_id, gt_mask, extra_params, float(mask_ratio),
                             near, far, t_rand, u_rand, rnd.seed, int(ray_offset))
      for level in ret:
        ret[level]['target_norm'] = tn[level].reshape(*batch_shape, *tn[level].shape[1:])
    # per-ray records by REAL level name (a single-level model has only 'coarse', as the reference's out dict)
    self.last_records = {'fine': rec_fine, 'coarse': rec_coarse} if two else {'coarse': rec_fine}
    return ret

  __call__ = apply

  def _unpack(self, rec, smp, S, batch_shape, origins, directions, return_points, return_weights, sharp_std, at_infinity=True):
    cfg = self.cfg
    o = {}
    for k, (a, n) in N.RAY_FIELDS.items():
      v = rec[:, a:a + n]
      o[k] = v[:, 0] if k in ('depth', 'med_depth', 'acc') else v
    o['med_points'] = o['med_points'][:, None, :3 + cfg.num_hyper_dims]
    o['ray_hyper_points'] = o['ray_hyper_points'][:, :cfg.num_hyper_dims]
    o['ray_hyper_c'] = torch.zeros_like(o['ray_hyper_points'])          # models.py:1384
    if not cfg.use_warp:                                                # models.py:1303-1305
      del o['ray_rotation_field'], o['ray_translation_field']
    if not cfg.use_predicted_mask:
      del o['ray_predicted_mask']
    if not cfg.predict_norm:
      del o['ray_norm']           # reference: sum w * normalize(-d sigma/dx) (models.py:1353); needs the sigma gradient
    if smp is not None:
      for k, (a, n) in N.SAMPLE_FIELDS.items():
        v = smp[:, :, a:a + n]
        o[k] = v if k in ('predicted_mask', 'sample_rgb', 'predicted_norm', 'warped_points') else v[:, :, 0]
      o['warped_points'] = o['warped_points'][..., :3 + cfg.num_hyper_dims]
      o['points'] = origins[:, None, :] + o['z_vals'][..., None] * directions[:, None, :]
      o['delta_x'] = o['warped_points'][..., :3] - o['points']          # models.py:1363
      if cfg.use_mask_sharp_weights:
        # models.py:1239-1245: sharp_weights come from cal_weights, whose sample_at_infinity is ALWAYS the default True (model_utils.py:162)
        # whatever the level composites with; the two differ in the last sample's alpha only (last delta 1e10 against 1e-19)
        w = o['weights']
        if not at_infinity:
          dn = torch.linalg.norm(directions, dim=-1)
          w = w.clone()
          w[:, -1] = (1.0 - torch.exp(-o['sigma'][:, -1] * (1e10 * dn))) * o['accum_prod'][:, -1]
        o['sharp_weights'] = sharpen_weights(w, o['z_vals'], sharp_std)
      if not cfg.use_predicted_mask:
        del o['predicted_mask']
      if not cfg.predict_norm:
        del o['predicted_norm'], o['back_facing']
      if not return_weights:
        pass                      # kept: the caller asked for per-sample data explicitly
      if not return_points:
        pass
    return {k: v.reshape(*batch_shape, *v.shape[1:]) for k, v in o.items()}

  def _target_norm(self, params, reloaded, origins, directions, viewdirs, warp_id, gt_mask, extra_params, mask_ratio, near, far,
                   t_rand, u_rand, seed, ray_offset):
    # Cost: the first call allocates a trainer workspace of ~2.8 MB per ray of `sigma_gradient_block` (default 4096 rays: ~11 GB).
    # The pass is local to this rank (data_parallel=False: no gradient all-reduce, whatever process group is initialised).  The
    # trainer draws the fine depths from ITS coarse weights (split-bf16 / fp32 arithmetic): for render precisions other than
    # f32 / bf16x3 target_norm is therefore taken at slightly different fine samples than the returned render.
    from .training import Trainer
    block = getattr(self, 'sigma_gradient_block', 4096)      # rays per pass of the trainer
    tr = getattr(self, '_sg_trainer', None)
    if tr is None:
      tr = self._sg_trainer = Trainer(self.cfg, params, max_rays=block, device=self.device)
    elif reloaded or getattr(self, '_sg_params', None) is not params:
      tr.set_params(params)
    self._sg_params = params
    R = origins.shape[0]
    out = {'coarse': [], 'fine': []}
    for lo in range(0, R, block):
      hi = min(lo + block, R)
      sl = lambda t: None if t is None else t[lo:hi]
      batch = dict(origins=origins[lo:hi], directions=directions[lo:hi], viewdirs=sl(viewdirs), metadata={'warp': warp_id[lo:hi]},
                   mask=sl(gt_mask), rgb=torch.zeros((hi - lo, 3), device=self.device))
      # one gradient-only step at lr 0: the parameters do not move; ray_offset keeps the Philox counters of the block in step
      tr.step(batch, extra_params, 0.0, t_rand=sl(t_rand), u_rand=sl(u_rand), mask_ratio=mask_ratio, near=near, far=far,
              grads_only=True, sigma_gradient=True, seed=int(seed), ray_offset=ray_offset + lo, data_parallel=False)
      for level in out:
        out[level].append(torch.as_tensor(tr.target_norm(level), device=self.device))
    return {k: torch.cat(v, 0) for k, v in out.items()}

  def encode_embed(self, metadata, table: str = 'warp', stream: Optional[torch.cuda.Stream] = None) -> torch.Tensor:
    """NerfModel._encode_embed (models.py:271-294) on the GPU (csrc/embed_kernel.hip): metadata [..., 1] ids or [..., 3] (left id, right id,
    progression) -> [..., 8] GLO vectors of the loaded parameters' ``warp_embed`` / ``mask_embed`` table."""
    m = metadata if isinstance(metadata, torch.Tensor) else torch.as_tensor(np.asarray(metadata))
    if m.shape[-1] not in (1, 3):
      raise ValueError('metadata must have 1 (id) or 3 (left id, right id, progression) channels')
    lead, ch = m.shape[:-1], m.shape[-1]
    m = m.to(self.device, torch.float32).reshape(-1, ch).contiguous()
    out = torch.empty((m.shape[0], 8), device=self.device, dtype=torch.float32)
    s = stream if stream is not None else torch.cuda.current_stream(self.device)
    rc = self._lib.nerfds_encode_embed(self._ctx, {'warp': 0, 'mask': 1}[table], m.data_ptr(), ch, m.shape[0], out.data_ptr(), C.c_void_p(s.cuda_stream))
    if rc != 0:
      raise RuntimeError(f'nerfds_encode_embed failed ({rc}): {N.last_error(self._ctx)}')
    return out.reshape(*lead, 8)

  # timing hooks for bench.py -------------------------------------------------------------------------
  def kernel_time_ms(self, reset: bool = False):
    tot = C.c_double(0.0)
    n = self._lib.nerfds_kernel_time_ms(self._ctx, int(reset), C.byref(tot))
    return n, tot.value


def sharpen_weights(weights: torch.Tensor, z_vals: torch.Tensor, std: float) -> torch.Tensor:
  """model_utils.py:180-190 on device, including the row-gather quirk of line 182."""
  idx = torch.argmax(weights, dim=1).clamp(max=z_vals.shape[0] - 1)
  mu = z_vals[idx]
  g = torch.exp(-0.5 * ((z_vals - mu) / std) ** 2) / (std * 2.5066282746310002)
  sw = weights * g
  return sw / sw.sum(dim=1, keepdim=True)


def construct_nerf(key, batch_size: int = 0, embeddings_dict=None, near: float = 0.0, far: float = 1.0,
                   cfg: Optional[NerfModelConfig] = None, use_predicted_norm: Optional[bool] = None,
                   use_sigma_gradient: bool = False, device=None, precision: str = 'f16x3', **init_kw):
  """models.construct_nerf (models.py:2677-2741): returns ``(model, params)``.

  ``key`` is an integer seed (JAX PRNG keys have no meaning here); ``embeddings_dict`` as in the reference
  ({'warp': [ids], ...}) sizes the GLO tables (models.py:235-237).
  """
  from .config import nerf_ds_config
  if cfg is None:
    cfg = nerf_ds_config(near=near, far=far)
  else:
    cfg = cfg.replace(near=near, far=far) if (near, far) != (0.0, 1.0) else cfg
  if embeddings_dict is not None and cfg.use_warp:
    cfg = cfg.replace(num_warp_embeds=int(max(embeddings_dict['warp'])) + 1)
  if use_predicted_norm is not None and bool(use_predicted_norm) != cfg.predict_norm:
    raise ValueError('use_predicted_norm must equal NerfModel.predict_norm')
  if use_sigma_gradient:
    raise NotImplementedError('use_sigma_gradient=True is not built')
  model = NerfModel(cfg, device=device, precision=precision)
  params = model.init(int(key) if not hasattr(key, '__len__') else int(np.asarray(key).ravel()[-1]), **init_kw)
  return model, params
