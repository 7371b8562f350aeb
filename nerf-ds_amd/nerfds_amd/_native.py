"""ctypes binding of include/nerfds.h (libnerfds_hip.so, built in-tree by csrc/Makefile).

There is no fallback: if the shared library is missing or fails to load, importing the product path
raises.  (The CPU oracle under /oracle is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NERFDS_LIB', os.path.join(_HERE, '_lib', 'libnerfds_hip.so'))   # NERFDS_LIB: development builds

ABI_VERSION = 7


def resolve_device(device=None):
  """A torch.device with an explicit index: 'cuda' / torch.device('cuda') mean the current device (tensors allocated on
  'cuda' report cuda:<current>, and an unindexed device compares unequal to them)."""
  import torch
  d = torch.device(device) if device is not None else torch.device('cuda')
  if d.type != 'cuda':
    raise ValueError(f'the HIP path runs on a cuda (ROCm) device, got {d}')
  return d if d.index is not None else torch.device('cuda', torch.cuda.current_device())

MAX_DEPTH = 16
RAY_REC = 26
SAMPLE_REC = 18
PREC = {'bf16': 0, 'bf16x3': 1, 'f32': 2, 'f16': 3, 'mixed': 4, 'bf16x3_fine': 5, 'f16x3': 6}

# per-ray record slices (enum nerfds_ray_field)
RAY_FIELDS = {
    'rgb': (0, 3), 'depth': (3, 1), 'med_depth': (4, 1), 'acc': (5, 1), 'ray_norm': (6, 3),
    'ray_rotation_field': (9, 3), 'ray_translation_field': (12, 3), 'ray_delta_x': (15, 3),
    'ray_hyper_points': (18, 2), 'ray_predicted_mask': (20, 1), 'med_points': (21, 5),
}
SAMPLE_FIELDS = {
    'z_vals': (0, 1), 'sigma': (1, 1), 'alpha': (2, 1), 'accum_prod': (3, 1), 'weights': (4, 1),
    'predicted_mask': (5, 1), 'sample_rgb': (6, 3), 'predicted_norm': (9, 3), 'warped_points': (12, 5),
    'back_facing': (17, 1),
}


class ModelCfg(C.Structure):
  _fields_ = [(n, C.c_int32) for n in (
      'abi_version', 'num_coarse_samples', 'num_fine_samples',
      'use_warp', 'use_hyper_sheet', 'use_predicted_mask', 'predict_norm', 'use_x_in_rgb_condition',
      'use_mask_in_warp', 'use_mask_in_hyper', 'use_viewdirs', 'mask_output_relu',
      'nerf_trunk_depth', 'nerf_trunk_width', 'nerf_skip', 'nerf_rgb_branch_depth', 'nerf_rgb_branch_width',
      'spatial_point_max_deg', 'hyper_point_max_deg', 'viewdir_max_deg', 'norm_input_max_deg',
      'warp_max_deg', 'warp_trunk_depth', 'warp_trunk_width', 'warp_skip',
      'hyper_sheet_max_deg', 'hyper_sheet_depth', 'hyper_sheet_width', 'hyper_sheet_skip', 'hyper_num_dims',
      'mask_max_deg', 'mask_depth', 'mask_width', 'mask_skip',
      'glo_num_dims', 'num_warp_embeds', 'use_white_background', 'use_sample_at_infinity',
      'use_posenc_identity', 'warp_use_posenc_identity')]


class Dense(C.Structure):
  _fields_ = [('kernel', C.c_void_p), ('bias', C.c_void_p), ('in_dim', C.c_int32), ('out_dim', C.c_int32)]


class NerfMlp(C.Structure):
  _fields_ = [('trunk', Dense * MAX_DEPTH), ('bottleneck', Dense), ('alpha', Dense),
              ('rgb_hidden', Dense * MAX_DEPTH), ('rgb', Dense)]


class Weights(C.Structure):
  _fields_ = [('warp_embed', C.c_void_p), ('mask_embed', C.c_void_p),
              ('mask_hidden', Dense * MAX_DEPTH), ('mask_out', Dense),
              ('warp_hidden', Dense * MAX_DEPTH), ('warp_w', Dense), ('warp_v', Dense),
              ('hyper_hidden', Dense * MAX_DEPTH), ('hyper_out', Dense),
              ('nerf', NerfMlp * 2), ('embed_rows', C.c_int32)]


class CameraStruct(C.Structure):
  _fields_ = [('orientation', C.c_float * 9), ('position', C.c_float * 3), ('focal_length', C.c_float),
              ('principal_point', C.c_float * 2), ('skew', C.c_float), ('pixel_aspect_ratio', C.c_float),
              ('radial_distortion', C.c_float * 3), ('tangential_distortion', C.c_float * 2),
              ('image_width', C.c_int32), ('image_height', C.c_int32)]


class Rays(C.Structure):
  _fields_ = [('num_rays', C.c_int64), ('origins', C.c_void_p), ('directions', C.c_void_p),
              ('viewdirs', C.c_void_p), ('warp_id', C.c_void_p), ('gt_mask', C.c_void_p),
              ('camera', C.POINTER(CameraStruct)), ('first_pixel', C.c_int64),
              ('encoded_warp', C.c_void_p), ('encoded_mask', C.c_void_p)]


class Extra(C.Structure):
  _fields_ = [('nerf_alpha', C.c_float), ('warp_alpha', C.c_float), ('hyper_alpha', C.c_float),
              ('hyper_sheet_alpha', C.c_float), ('norm_input_alpha', C.c_float), ('mask_ratio', C.c_float),
              ('near', C.c_float), ('far', C.c_float), ('use_stratified_sampling', C.c_int32),
              ('render_opt_flags', C.c_uint32), ('dust_threshold', C.c_float), ('bounding_box', C.c_float * 6),
              ('use_linear_disparity', C.c_int32), ('sample_at_infinity_override', C.c_int32)]


TRISTATE_NONE, TRISTATE_TRUE, TRISTATE_FALSE = 0, 1, 2
OPT_DUST_THRESHOLD, OPT_BOUNDING_BOX = 1, 2


def set_render_opts(extra: 'Extra', render_opts) -> None:
  """render_opts dict of NerfModel.__call__ (filter_sigma, models.py:38-66) -> the nerfds_extra fields."""
  if render_opts is None:
    return
  unknown = set(render_opts) - {'dust_threshold', 'bounding_box'}
  if unknown:
    raise ValueError(f'unknown render_opts keys {sorted(unknown)} (filter_sigma knows dust_threshold and bounding_box)')
  if 'dust_threshold' in render_opts:
    extra.render_opt_flags |= OPT_DUST_THRESHOLD
    extra.dust_threshold = float(render_opts['dust_threshold'])
  if 'bounding_box' in render_opts:
    box = [float(v) for v in render_opts['bounding_box']]
    if len(box) != 6:
      raise ValueError('bounding_box = (xmin, xmax, ymin, ymax, zmin, zmax)')
    extra.render_opt_flags |= OPT_BOUNDING_BOX
    extra.bounding_box = (C.c_float * 6)(*box)


class Rand(C.Structure):
  _fields_ = [('t_rand', C.c_void_p), ('u_rand', C.c_void_p), ('seed', C.c_uint64), ('first_ray', C.c_int64)]


class Out(C.Structure):
  _fields_ = [('ray_fine', C.c_void_p), ('ray_coarse', C.c_void_p),
              ('sample_fine', C.c_void_p), ('sample_coarse', C.c_void_p)]


# every symbol include/nerfds.h declares
SYMBOLS = ('nerfds_abi_version', 'nerfds_struct_size', 'nerfds_precision_plan', 'nerfds_precision_plan_level', 'nerfds_ctx_create', 'nerfds_ctx_load_weights', 'nerfds_render_rays',
           'nerfds_encode_embed', 'nerfds_ctx_destroy', 'nerfds_last_error', 'nerfds_kernel_time_ms', 'nerfds_pack_stream_bytes', 'nerfds_pack_stream_bytes_level',
           'nerfds_pack_bias_floats', 'nerfds_pack_tile_pair', 'nerfds_pack_stream', 'nerfds_debug_mfma', 'nerfds_camera_to_rays',
           'nerfds_frame_images', 'nerfds_trainer_create', 'nerfds_trainer_destroy', 'nerfds_trainer_param_count',
           'nerfds_trainer_num_leaves', 'nerfds_trainer_leaf', 'nerfds_trainer_params', 'nerfds_trainer_grads',
           'nerfds_trainer_download', 'nerfds_trainer_upload', 'nerfds_trainer_reset_optimizer', 'nerfds_trainer_step', 'nerfds_trainer_apply', 'nerfds_trainer_clip_gradients', 'nerfds_trainer_target_norm',
           'nerfds_trainer_last_error', 'nerfds_trainer_debug_read', 'nerfds_trainer_nonfinite', 'nerfds_trainer_set_step', 'nerfds_trainer_get_step', 'nerfds_trainer_set_loss_scale_adjust', 'nerfds_trainer_set_numerics', 'nerfds_trainer_get_numerics', 'nerfds_trainer_overflow_sources', 'nerfds_trainer_forward', 'nerfds_render_rays_bwd',
           'nerfds_debug_lds_attr_first_use')

_lib = None


def load():
  """Loads the HIP library; raises if it is not built (python __graft_entry__.py / make -C nerf-ds_amd/csrc)."""
  global _lib
  if _lib is not None:
    return _lib
  # torch FIRST: it ships its own HIP runtime (torch/lib/libamdhip64.so) and the library must bind to THAT copy - the one that owns the
  # tensors and streams it is handed.  Loaded the other way round the process holds two HIP runtimes (this library's from /opt/rocm, then
  # torch's) and neither finds the GPU afterwards ("no ROCm-capable device is detected": build() + smoke() in one process, round 5).
  import torch  # noqa: F401
  if not os.path.exists(LIB_PATH):
    raise RuntimeError(f'{LIB_PATH} is not built: run `make -C nerf-ds_amd/csrc -j8` (there is no CPU fallback)')
  lib = C.CDLL(LIB_PATH)
  lib.nerfds_abi_version.restype = C.c_int
  lib.nerfds_precision_plan.argtypes = [C.c_uint32, C.POINTER(C.c_int32)]
  lib.nerfds_precision_plan_level.argtypes = [C.c_uint32, C.c_int32, C.POINTER(C.c_int32)]
  lib.nerfds_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(ModelCfg)]
  lib.nerfds_ctx_load_weights.argtypes = [C.c_void_p, C.POINTER(Weights)]
  lib.nerfds_render_rays.argtypes = [C.c_void_p, C.POINTER(Rays), C.POINTER(Extra), C.POINTER(Rand), C.POINTER(Out),
                                     C.c_uint32, C.c_void_p]
  lib.nerfds_encode_embed.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
  lib.nerfds_ctx_destroy.argtypes = [C.c_void_p]
  lib.nerfds_last_error.argtypes = [C.c_void_p]
  lib.nerfds_last_error.restype = C.c_char_p
  lib.nerfds_kernel_time_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
  lib.nerfds_pack_stream_bytes.argtypes = [C.POINTER(ModelCfg), C.c_int, C.c_uint32]
  lib.nerfds_pack_stream_bytes.restype = C.c_int64
  lib.nerfds_pack_stream_bytes_level.argtypes = [C.POINTER(ModelCfg), C.c_int, C.c_int, C.c_uint32]
  lib.nerfds_pack_stream_bytes_level.restype = C.c_int64
  lib.nerfds_pack_tile_pair.argtypes = [C.POINTER(ModelCfg), C.c_uint32]
  lib.nerfds_pack_bias_floats.argtypes = [C.POINTER(ModelCfg), C.c_int]
  lib.nerfds_pack_bias_floats.restype = C.c_int64
  lib.nerfds_pack_stream.argtypes = [C.POINTER(ModelCfg), C.POINTER(Weights), C.c_int, C.c_int, C.c_uint32,
                                     C.c_void_p, C.c_void_p]
  lib.nerfds_debug_mfma.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  lib.nerfds_camera_to_rays.argtypes = [C.c_int, C.POINTER(CameraStruct), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
  lib.nerfds_frame_images.argtypes = [C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]
  lib.nerfds_debug_lds_attr_first_use.argtypes = [C.c_uint64, C.c_int]
  if lib.nerfds_abi_version() != ABI_VERSION:
    raise RuntimeError('libnerfds_hip.so ABI version mismatch: rebuild')
  lib.nerfds_struct_size.argtypes = [C.c_int]
  lib.nerfds_struct_size.restype = C.c_int64
  for which, st in enumerate((ModelCfg, Weights, CameraStruct, Rays, Extra, Rand, Out)):      # the ctypes mirrors against the library's own sizeof
    if lib.nerfds_struct_size(which) != C.sizeof(st):
      raise RuntimeError(f'{st.__name__}: ctypes declares {C.sizeof(st)} bytes, libnerfds_hip.so has {lib.nerfds_struct_size(which)} (include/nerfds.h changed?)')
  _lib = lib
  return lib


def last_error(ctx=None) -> str:
  msg = load().nerfds_last_error(ctx)
  return msg.decode() if msg else ''
