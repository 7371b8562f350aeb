"""A reader for the subset of gin-config syntax the reference's config files use, enough to resolve the static
render graph (``NerfModelConfig``) from ``configs/*.gin`` or from the ``config.gin`` that train.py writes into an
experiment directory (train.py:335-338) and render.py re-parses (render.py:52-56).

Supported: ``include 'file'``, macros (``name = value`` / ``%name``), bindings ``Configurable.field = value`` and
scoped bindings ``scope/Configurable.field = value``, python literals (numbers, strings, tuples, lists, dicts,
None/True/False), configurable references ``@name`` / ``@scope/name`` (kept as strings), comments.  gin resolves
macros lazily, so a later ``warp_max_deg = 4`` overrides an earlier definition for every ``%warp_max_deg`` use —
reproduced here by substituting macros only after the whole file tree has been read.
"""
from __future__ import annotations

import ast
import os
import re
from typing import Any, Dict, Tuple

from .config import MLPSpec, NerfModelConfig

_REF = re.compile(r'@([A-Za-z_][\w./]*)(\(\))?')
_MACRO = re.compile(r'%([A-Za-z_]\w*)')


def _logical_lines(text: str):
  """Joins continuation lines (open brackets) and strips comments."""
  buf, depth = '', 0
  for raw in text.splitlines():
    line = re.sub(r'(?<![\'"])#.*$', '', raw).rstrip()
    if not line.strip() and depth == 0:
      continue
    buf = (buf + ' ' + line.strip()) if buf else line.strip()
    depth = sum(buf.count(c) for c in '([{') - sum(buf.count(c) for c in ')]}')
    if depth <= 0:
      yield buf
      buf, depth = '', 0
  if buf:
    yield buf


def parse_gin(path: str, _seen=None) -> Tuple[Dict[str, str], Dict[str, str]]:
  """Returns (macros, bindings) as raw right-hand-side strings; later definitions override earlier ones."""
  macros: Dict[str, str] = {}
  bindings: Dict[str, str] = {}
  _seen = _seen or set()
  path = os.path.abspath(path)
  if path in _seen:
    return macros, bindings
  _seen.add(path)
  base_dirs = [os.path.dirname(path), os.path.dirname(os.path.dirname(path)), os.getcwd()]
  for line in _logical_lines(open(path).read()):
    m = re.match(r"include\s+['\"](.+)['\"]", line)
    if m:
      for d in base_dirs:
        cand = os.path.join(d, m.group(1))
        if os.path.exists(cand):
          im, ib = parse_gin(cand, _seen)
          macros.update(im)
          bindings.update(ib)
          break
      else:
        raise FileNotFoundError(f'gin include {m.group(1)!r} not found from {path}')
      continue
    if line.startswith('import '):
      continue
    if '=' not in line:
      raise ValueError(f'cannot parse gin line: {line!r}')
    lhs, rhs = (x.strip() for x in line.split('=', 1))
    (bindings if '.' in lhs else macros)[lhs] = rhs
  return macros, bindings


def _evaluate(rhs: str, macros: Dict[str, str], depth: int = 0) -> Any:
  if depth > 32:
    raise RecursionError('cyclic gin macro')
  rhs = _REF.sub(lambda m: repr('@' + m.group(1)), rhs)                      # configurable refs -> strings
  def sub(m):
    name = m.group(1)
    if name not in macros:
      raise KeyError(f'undefined gin macro %{name}')
    return repr(_evaluate(macros[name], macros, depth + 1))
  return ast.literal_eval(_MACRO.sub(sub, rhs))


def resolve(path: str) -> Dict[str, Any]:
  macros, bindings = parse_gin(path)
  out = {}
  for k, v in bindings.items():
    try:
      out[k] = _evaluate(v, macros)
    except KeyError:
      continue            # bindings that use macros never defined for this experiment (e.g. %data_dir) are not ours
  return out


def _schedule_final(s) -> float:
  """Final value of a schedule config (schedules.py:37-48) - what a fully trained checkpoint carries."""
  if isinstance(s, (int, float)):
    return float(s)
  if isinstance(s, tuple):
    kind, *args = s
    s = {'constant': {'type': 'constant', 'value': args[0]} if kind == 'constant' else None}.get(kind) or \
        {'type': kind, 'args': args}
  t = s.get('type')
  if t == 'constant':
    return float(s['value'])
  if t in ('linear', 'exponential', 'cosine_easing'):
    return float(s['final_value'] if 'final_value' in s else s['args'][1])
  if t == 'piecewise':
    return _schedule_final(s['schedules'][-1][1])
  if t == 'delayed':
    return _schedule_final(s['base_schedule'])
  raise ValueError(f'unknown schedule {s!r}')


def config_from_gin(path: str, near: float = 0.0, far: float = 1.0, num_warp_embeds: int = 1) -> NerfModelConfig:
  """Builds the static render-graph config from a gin file (defaults = the dataclass defaults of models.py:116-229)."""
  b = resolve(path)
  g = lambda key, default: b.get(key, default)
  act = g('MaskMLP.output_activation', None)
  skips = lambda key, d: tuple(g(key, d))
  cfg = NerfModelConfig(
      near=near, far=far, num_warp_embeds=num_warp_embeds,
      use_viewdirs=g('NerfModel.use_viewdirs', True),
      nerf_trunk_depth=g('NerfModel.nerf_trunk_depth', 8), nerf_trunk_width=g('NerfModel.nerf_trunk_width', 256),
      nerf_rgb_branch_depth=g('NerfModel.nerf_rgb_branch_depth', 1), nerf_rgb_branch_width=g('NerfModel.nerf_rgb_branch_width', 128),
      nerf_skips=skips('NerfModel.nerf_skips', (4,)),
      num_coarse_samples=g('NerfModel.num_coarse_samples', 196), num_fine_samples=g('NerfModel.num_fine_samples', 196),
      use_stratified_sampling=g('NerfModel.use_stratified_sampling', True),
      use_white_background=g('NerfModel.use_white_background', False),
      use_linear_disparity=g('NerfModel.use_linear_disparity', False),
      use_sample_at_infinity=g('NerfModel.use_sample_at_infinity', True),
      spatial_point_min_deg=g('NerfModel.spatial_point_min_deg', 0), spatial_point_max_deg=g('NerfModel.spatial_point_max_deg', 10),
      hyper_point_min_deg=g('NerfModel.hyper_point_min_deg', 0), hyper_point_max_deg=g('NerfModel.hyper_point_max_deg', 4),
      viewdir_min_deg=g('NerfModel.viewdir_min_deg', 0), viewdir_max_deg=g('NerfModel.viewdir_max_deg', 4),
      use_posenc_identity=g('NerfModel.use_posenc_identity', True),
      hyper_slice_method=g('NerfModel.hyper_slice_method', 'none'), use_hyper=g('NerfModel.use_hyper', True),
      hyper_use_warp_embed=g('NerfModel.hyper_use_warp_embed', True), use_hyper_for_sigma=g('NerfModel.use_hyper_for_sigma', True),
      hyper_sheet_min_deg=g('HyperSheetMLP.min_deg', 0), hyper_sheet_max_deg=g('HyperSheetMLP.max_deg', 1),
      hyper_sheet_output_channels=g('HyperSheetMLP.output_channels', 2),
      hyper_sheet_mlp=MLPSpec(g('HyperSheetMLP.depth', 6), g('HyperSheetMLP.width', 64), skips('HyperSheetMLP.skips', (4,))),
      use_warp=g('NerfModel.use_warp', False),
      warp_min_deg=g('SE3Field.min_deg', 0), warp_max_deg=g('SE3Field.max_deg', 8),
      warp_use_posenc_identity=g('SE3Field.use_posenc_identity', False),
      warp_trunk=MLPSpec(g('SE3Field.trunk_depth', 6), g('SE3Field.trunk_width', 128), skips('SE3Field.skips', (4,))),
      glo_num_dims=g('warp/GLOEmbed.num_dims', 8),
      predict_norm=g('NerfModel.predict_norm', False), norm_supervision_type=g('NerfModel.norm_supervision_type', 'warped'),
      norm_input_posenc=g('NerfModel.norm_input_posenc', True),
      norm_input_min_deg=g('NerfModel.norm_input_min_deg', 0), norm_input_max_deg=g('NerfModel.norm_input_max_deg', 4),
      use_x_in_rgb_condition=g('NerfModel.use_x_in_rgb_condition', False),
      window_x_in_rgb_condition=g('NerfModel.window_x_in_rgb_condition', False),
      use_mask_in_warp=g('NerfModel.use_mask_in_warp', False), use_mask_in_hyper=g('NerfModel.use_mask_in_hyper', False),
      use_mask_in_rgb=g('NerfModel.use_mask_in_rgb', False), use_predicted_mask=g('NerfModel.use_predicted_mask', False),
      use_mask_embed=g('NerfModel.use_mask_embed', True), use_3d_mask=g('NerfModel.use_3d_mask', False),
      use_mask_sharp_weights=g('NerfModel.use_mask_sharp_weights', False),
      mask_min_deg=g('MaskMLP.min_deg', 0), mask_max_deg=g('MaskMLP.max_deg', 6),
      mask_mlp=MLPSpec(g('MaskMLP.depth', 6), g('MaskMLP.width', 64), skips('MaskMLP.skips', (4,))),
      mask_output_relu=(act is not None and 'relu' in str(act)),
  )
  for unsupported in ('use_hyper_c', 'use_bone', 'use_ref_radiance', 'use_nerf_embed', 'use_delta_x_in_rgb_condition',
                      'use_hyper_for_rgb', 'use_viewdirs_in_hyper', 'use_mask_scaled_weights', 'use_rgb_sharp_weights',
                      'clamp_predicted_mask', 'use_coarse_depth_for_mask'):
    if g('NerfModel.' + unsupported, False):
      raise NotImplementedError(f'NerfModel.{unsupported}=True has no HIP kernel')
  return cfg


def extra_params_from_gin(path: str) -> Dict[str, float]:
  """The alphas a fully trained state carries (final values of the schedules in TrainConfig / SpecularConfig)."""
  b = resolve(path)
  out = {}
  for key, name in (('TrainConfig.nerf_alpha_schedule', 'nerf_alpha'), ('TrainConfig.warp_alpha_schedule', 'warp_alpha'),
                    ('TrainConfig.hyper_alpha_schedule', 'hyper_alpha'),
                    ('TrainConfig.hyper_sheet_alpha_schedule', 'hyper_sheet_alpha'),
                    ('SpecularConfig.norm_input_alpha_schedule', 'norm_input_alpha')):
    if key in b:
      out[name] = _schedule_final(b[key])
  return out


def objective_from_gin(path: str, step: int = 0, honour_hyper_reg_loss_weight: bool = False) -> Dict[str, Any]:
  """The ``objective`` dict of ``Trainer.step`` at training step ``step``, from the loss switches and weights train.py reads out of
  ``TrainConfig`` / ``SpecularConfig`` (train.py:313-355: ``scalar_params``, the static flags of ``training.train_step``, ``state.norm_loss_weight``;
  defaults as configs.py:40-110, 222-252 and training.py:36-56).  A switched-off loss contributes no key, so ``{}`` (falsy) selects the plain rgb step.

  ``TrainConfig.hyper_reg_loss_weight``: train.py builds ``ScalarParams`` without it (train.py:312-325, and no ``.replace`` in the loop sets it), so
  the reference trains with the dataclass default 0.0 (training.py:49) whatever the gin file says - ``use_hyper_reg_loss=True`` only adds the
  statistic.  The default here reproduces the reference AS IT RUNS (weight 0, no key); ``honour_hyper_reg_loss_weight=True`` takes the gin value."""
  from . import sched
  b = resolve(path)
  g = lambda k, d: b.get(k, d)
  at = lambda key, default: sched.build(b[key] if key in b else default)(step)
  ob: Dict[str, Any] = {}
  if g('TrainConfig.use_warp_reg_loss', False):
    ob.update(warp_reg_loss_weight=float(g('TrainConfig.warp_reg_loss_weight', 0.0)), warp_reg_loss_alpha=float(g('TrainConfig.warp_reg_loss_alpha', -2.0)),
              warp_reg_loss_scale=float(g('TrainConfig.warp_reg_loss_scale', 0.001)))
  if g('TrainConfig.use_hyper_reg_loss', False) and honour_hyper_reg_loss_weight:
    ob['hyper_reg_loss_weight'] = float(g('TrainConfig.hyper_reg_loss_weight', 0.0))
  if g('TrainConfig.use_background_loss', False):      # batch['background_points'] comes from the data source (train.py; batch size
    ob.update(background_loss_weight=float(g('TrainConfig.background_loss_weight', 0.0)),              # TrainConfig.background_points_batch_size)
              background_noise_std=float(g('TrainConfig.background_noise_std', 0.001)))
  if g('TrainConfig.use_elastic_loss', False):
    ob.update(elastic_loss_weight=float(at('TrainConfig.elastic_loss_weight_schedule', None) or 0.0),
              elastic_reduce_method=g('TrainConfig.elastic_reduce_method', 'weight'), elastic_loss_type=g('TrainConfig.elastic_loss_type', 'log_svals'))
  if g('SpecularConfig.use_back_facing_reg', False):
    ob['back_facing_reg_weight'] = float(g('SpecularConfig.back_facing_reg_weight', 0.0))
  if g('SpecularConfig.use_predicted_norm', False):    # training.py:323-332 with state.norm_loss_weight (train.py:401-427)
    ob['norm_loss_weight'] = float(at('SpecularConfig.norm_loss_weight_schedule', {'type': 'constant', 'value': 0.001}))
  if g('NerfModel.use_predicted_mask', False) and float(g('SpecularConfig.predicted_mask_loss_weight', 0.0)) != 0.0:
    ob['predicted_mask_loss_weight'] = float(g('SpecularConfig.predicted_mask_loss_weight', 0.0))
    ob['sharp_weights_std'] = float(at('SpecularConfig.sharp_mask_std_schedule', {'type': 'constant', 'value': 1.0}))
    if g('SpecularConfig.use_mask_occlusion_reg_loss', False):      # training.py:409-417 (inside the 3-D mask branch)
      ob['mask_occlusion_reg_loss_weight'] = float(g('SpecularConfig.mask_occlusion_reg_loss_weight', 1.0))
  return ob if any(k.endswith('_weight') and v != 0.0 for k, v in ob.items()) else {}

