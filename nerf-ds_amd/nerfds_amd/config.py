"""Static configuration of the NeRF-DS render graph.

Field names follow the reference's gin-configurable ``NerfModel`` dataclass
(/root/reference/hypernerf/models.py:116-229) and its sub-modules
(``SE3Field`` warping.py:139-157, ``HyperSheetMLP`` modules.py:354-365,
``MaskMLP`` modules.py:396-407) so that a ``config.gin`` written by the
reference's train.py maps onto it one to one (see ``gin_subset.py``).

Only the fields the hot path reads are kept; every feature the reference
model has but no shipped gin file enables (hyper_c, bone warp, ref radiance,
nerf_embed conditions, ...) is rejected loudly in ``validate``.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class MLPSpec:
  """depth/width/skips of a reference ``modules.MLP`` (modules.py:44-83)."""
  depth: int
  width: int
  skips: Tuple[int, ...] = (4,)


@dataclass
class NerfModelConfig:
  # Scene bounds (models.py:117-118).
  near: float = 0.0
  far: float = 1.0
  # Number of rows in the GLO tables (max(embeddings_dict[key]) + 1, models.py:235-237).
  num_warp_embeds: int = 1

  # NeRF architecture (models.py:121-127).
  use_viewdirs: bool = True
  nerf_trunk_depth: int = 8
  nerf_trunk_width: int = 256
  nerf_rgb_branch_depth: int = 1
  nerf_rgb_branch_width: int = 128
  nerf_skips: Tuple[int, ...] = (4,)

  # Rendering (models.py:130-135; nerf_ds.gin:14-17).
  num_coarse_samples: int = 64
  num_fine_samples: int = 64
  use_stratified_sampling: bool = True
  use_white_background: bool = False
  use_linear_disparity: bool = False
  use_sample_at_infinity: bool = True

  # Positional encodings (models.py:137-143; nerf_ds.gin:22-30; defaults.gin:111).
  spatial_point_min_deg: int = 0
  spatial_point_max_deg: int = 8
  hyper_point_min_deg: int = 0
  hyper_point_max_deg: int = 1
  viewdir_min_deg: int = 0
  viewdir_max_deg: int = 4
  use_posenc_identity: bool = False

  alpha_channels: int = 1
  rgb_channels: int = 3

  # Hyper slicing (models.py:158-167; nerf_ds.gin:35-44).
  hyper_slice_method: str = 'bendy_sheet'   # 'none' | 'bendy_sheet'
  use_hyper: bool = True
  hyper_use_warp_embed: bool = True
  use_hyper_for_sigma: bool = True
  hyper_sheet_min_deg: int = 0
  hyper_sheet_max_deg: int = 6
  hyper_sheet_output_channels: int = 2
  hyper_sheet_mlp: MLPSpec = field(default_factory=lambda: MLPSpec(6, 64, (4,)))

  # Warp field = SE3Field (models.py:170-174; warping.py:139-157; nerf_ds.gin:47-49).
  use_warp: bool = True
  warp_min_deg: int = 0
  warp_max_deg: int = 4
  warp_use_posenc_identity: bool = False
  warp_trunk: MLPSpec = field(default_factory=lambda: MLPSpec(6, 128, (4,)))
  glo_num_dims: int = 8   # warp/GLOEmbed.num_dims (defaults.gin:134)

  # NeRF-DS additions (models.py:177-186; nerf_ds.gin:89-98).
  predict_norm: bool = True
  norm_supervision_type: str = 'warped'
  stop_norm_gradient: bool = True
  norm_input_posenc: bool = True
  norm_input_min_deg: int = 0
  norm_input_max_deg: int = 4
  use_x_in_rgb_condition: bool = True
  window_x_in_rgb_condition: bool = False

  # Mask (models.py:202-214; nerf_ds.gin:105-118).
  use_mask_in_warp: bool = True
  use_mask_in_hyper: bool = True
  use_mask_in_rgb: bool = False
  use_predicted_mask: bool = True
  use_mask_embed: bool = True
  use_3d_mask: bool = True
  use_mask_sharp_weights: bool = True
  mask_min_deg: int = 0
  mask_max_deg: int = 6
  mask_mlp: MLPSpec = field(default_factory=lambda: MLPSpec(8, 128, (4,)))
  mask_output_relu: bool = True   # MaskMLP.output_activation = @jax.nn.relu (nerf_ds.gin:118)

  # ---- derived widths -------------------------------------------------
  @property
  def has_hyper(self) -> bool:
    return self.hyper_slice_method != 'none'

  def posenc_dim(self, channels: int, min_deg: int, max_deg: int, identity: bool) -> int:
    return 2 * (max_deg - min_deg) * channels + (channels if identity else 0)

  @property
  def mask_in_dim(self) -> int:
    d = self.posenc_dim(3, self.mask_min_deg, self.mask_max_deg, False)
    return d + (self.glo_num_dims if self.use_mask_embed else 0)

  @property
  def warp_in_dim(self) -> int:
    d = self.posenc_dim(3, self.warp_min_deg, self.warp_max_deg, self.warp_use_posenc_identity)
    return d + self.glo_num_dims + (1 if self.use_mask_in_warp else 0)

  @property
  def hyper_in_dim(self) -> int:
    d = self.posenc_dim(3, self.hyper_sheet_min_deg, self.hyper_sheet_max_deg, False)
    return d + self.glo_num_dims + (1 if self.use_mask_in_hyper else 0)

  @property
  def num_hyper_dims(self) -> int:
    return self.hyper_sheet_output_channels if (self.has_hyper and self.use_hyper and self.use_hyper_for_sigma) else 0

  @property
  def trunk_in_dim(self) -> int:
    d = self.posenc_dim(3, self.spatial_point_min_deg, self.spatial_point_max_deg, self.use_posenc_identity)
    if self.num_hyper_dims:
      d += self.posenc_dim(self.num_hyper_dims, self.hyper_point_min_deg, self.hyper_point_max_deg, False)
    return d

  @property
  def viewdir_dim(self) -> int:
    if not self.use_viewdirs:
      return 0
    return self.posenc_dim(3, self.viewdir_min_deg, self.viewdir_max_deg, self.use_posenc_identity)

  @property
  def norm_feat_dim(self) -> int:
    if not self.predict_norm:
      return 0
    if not self.norm_input_posenc:
      return 3
    return self.posenc_dim(3, self.norm_input_min_deg, self.norm_input_max_deg, self.use_posenc_identity)

  @property
  def alpha_out_dim(self) -> int:
    return self.alpha_channels + (3 if self.predict_norm else 0)

  @property
  def rgb_in_dim(self) -> int:
    """modules.py:296-310: [bottleneck|trunk_out, rgb_condition, extra, norm]."""
    d = self.nerf_trunk_width
    d += self.viewdir_dim
    if self.use_x_in_rgb_condition:
      d += self.nerf_trunk_width      # points_feat is rebound to trunk_output (models.py:1046,1208)
    d += self.norm_feat_dim
    return d

  def validate(self) -> None:
    if self.hyper_slice_method not in ('none', 'bendy_sheet'):
      raise RuntimeError(f'Unknown hyper slice method {self.hyper_slice_method}.')   # models.py:315
    if self.norm_supervision_type != 'warped' and self.predict_norm:
      raise NotImplementedError('only norm_supervision_type="warped" (nerf_ds.gin:90) is built')
    if self.use_mask_in_rgb:
      raise NotImplementedError('use_mask_in_rgb=True is not configured by any shipped gin file')
    if self.window_x_in_rgb_condition:
      raise NotImplementedError('window_x_in_rgb_condition=True is not configured by any shipped gin file')
    if self.use_predicted_mask and not self.use_warp:
      raise ValueError('use_predicted_mask needs the warp GLO ids (models.py:335-336,925)')
    if self.has_hyper and not self.hyper_use_warp_embed:
      raise NotImplementedError('separate hyper embedding is not configured by any shipped gin file')
    if self.has_hyper and not self.use_warp:
      raise NotImplementedError('bendy_sheet without use_warp has no warp GLO table to share')
    if self.alpha_channels != 1 or self.rgb_channels != 3:
      raise NotImplementedError('alpha_channels/rgb_channels other than 1/3')

  def replace(self, **kw) -> 'NerfModelConfig':
    return dataclasses.replace(self, **kw)


def nerf_ds_config(near: float = 0.3, far: float = 1.7, num_warp_embeds: int = 256, **kw) -> NerfModelConfig:
  """configs/nerf_ds.gin over configs/defaults.gin (SURVEY.md section 8 'Configuration resolved')."""
  cfg = NerfModelConfig(near=near, far=far, num_warp_embeds=num_warp_embeds)
  return cfg.replace(**kw) if kw else cfg


def hypernerf_config(near: float = 0.3, far: float = 1.7, num_warp_embeds: int = 256, **kw) -> NerfModelConfig:
  """configs/base.gin over configs/defaults.gin (the HyperNeRF graph; BASELINE config 5 per SURVEY 8d): SE(3) warp with 6 bands
  and posenc identity, bendy-sheet hyper slicing, posenc identity on x' and the view direction, 128 + 128 samples; no mask
  network, no predicted normal, no trunk output in the rgb condition."""
  cfg = NerfModelConfig(
      near=near, far=far, num_warp_embeds=num_warp_embeds, num_coarse_samples=128, num_fine_samples=128,
      use_posenc_identity=True, warp_use_posenc_identity=True, warp_max_deg=6,
      predict_norm=False, use_x_in_rgb_condition=False, use_mask_in_warp=False, use_mask_in_hyper=False,
      use_predicted_mask=False, use_3d_mask=False, use_mask_sharp_weights=False,
      mask_mlp=MLPSpec(6, 64, (4,)), mask_output_relu=False)      # MaskMLP defaults (unused: no mask network)
  return cfg.replace(**kw) if kw else cfg


def static_config(near: float = 2.0, far: float = 6.0, num_coarse_samples: int = 64, **kw) -> NerfModelConfig:
  """BASELINE.json configs[0]: static scene, coarse only, warp/hyper/mask/normal disabled.

  The reference model cannot literally run this; it is defined as
  identity warp, no hyper points, trunk in = posenc_8(x), rgb in = [bottleneck, posenc_4(viewdir)].
  """
  cfg = NerfModelConfig(
      near=near, far=far, num_warp_embeds=1,
      num_coarse_samples=num_coarse_samples, num_fine_samples=0,
      hyper_slice_method='none', use_warp=False,
      predict_norm=False, use_x_in_rgb_condition=False,
      use_mask_in_warp=False, use_mask_in_hyper=False, use_predicted_mask=False,
      use_3d_mask=False, use_mask_sharp_weights=False)
  return cfg.replace(**kw) if kw else cfg
