"""TEST INFRASTRUCTURE - CPU oracle (numpy) of the frame output path that follows the gather: the per-frame
post-processing of render.py:231-262 (rgb / median-depth / normal / mask / delta_x / median-point tiles), the depth
colourisation of hypernerf/visualization.py:178-235 (scale_values, interpolate_colormap, colorize) and the uint8
conversion of hypernerf/image_utils.py:124-131 (image_to_uint8).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline may import this file.

PARITY: the depth colourisation (get_colormap, scale_values, interpolate_colormap, colorize) is PINNED BY REFERENCE-RUN VECTORS:
hypernerf/visualization.py needs only numpy + matplotlib and runs in the build container; tests/golden/make_frame_golden.py
executes it from /root/reference and tests/golden/frame_colorize_ref.npz holds its outputs for the real 'magma' table that
render.py:263 uses (tests/test_frames.py::test_oracle_matches_reference_run_colorize, bit for bit).  The rest of the frame
assembly (render.py:231-268, image_utils.image_to_uint8) imports cv2 / mediapy / jax and cannot run here: those lines stay
restated and unpinned.  The oracle restates the numpy arithmetic INCLUDING its dtypes, because the outputs are bytes and must
match bit for bit:
  * the tiles that come from the model are float32; matplotlib colormaps are float64 [256, 3], so ``depth_viz`` is
    float64 and np.concatenate promotes the whole debug frame to float64 before ``* 255`` (render.py:265-268), while the
    stand-alone rgb frame stays float32 (render.py:269);
  * scale_values / ``1 - x`` / ``values * 255`` / floor / the interpolation fraction are float32 (weak python scalars),
    the table lookup + lerp is float64.
``normalize_vector`` (model_utils.py:438-442) is a jnp call in the reference; its 3-term sum is restated in float32,
left to right.
"""
import numpy as np

RAY_REC = 26
F_RGB, F_MED_DEPTH, F_NORM, F_DELTA_X, F_MASK, F_MED_POINTS = 0, 4, 6, 15, 20, 21


def sinebow(h):                                                       # visualization.py:168-170
  f = lambda x: np.sin(np.pi * x) ** 2
  return np.stack([f(3 / 6 - h), f(5 / 6 - h), f(7 / 6 - h)], -1)


def get_colormap(name, num_bins=256):                                  # visualization.py:158-164, 173-183
  if name == 'sinebow':
    return np.array([sinebow(i) for i in np.linspace(0, 1, num_bins)])
  if name == 'gray':                                                   # (not a reference map: an analytic table for known answers)
    g = np.linspace(0, 1, num_bins)
    return np.stack([g, g, g], -1)
  # _build_colormap: the matplotlib map sampled at num_bins points, rebuilt as a LinearSegmentedColormap and sampled again
  from matplotlib import cm
  from matplotlib.colors import LinearSegmentedColormap
  base = cm.get_cmap(name)
  color_list = base(np.linspace(0, 1, num_bins))
  colormap = LinearSegmentedColormap.from_list(base.name + str(num_bins), color_list, num_bins)
  return colormap(np.linspace(0, 1, num_bins))[:, :3]


def scale_values(values, vmin, vmax, eps=1e-6):                        # visualization.py:195-196
  return (values - np.float32(vmin)) / np.float32(max(vmax - vmin, eps))


def interpolate_colormap(values, colormap):                            # visualization.py:186-192
  a = np.floor(values * np.float32(255))
  b = (a + np.float32(1)).clip(max=255)
  f = values * np.float32(255.0) - a
  with np.errstate(invalid='ignore'):
    a = np.nan_to_num(a, nan=0.0).clip(0, 65535).astype(np.uint16).clip(0, 255)    # out-of-range pixels are overwritten below
    b = np.nan_to_num(b, nan=0.0).clip(0, 65535).astype(np.uint16).clip(0, 255)
  return colormap[a] + (colormap[b] - colormap[a]) * f[..., np.newaxis]


def colorize(array, cmin=None, cmax=None, colormap='magma', eps=1e-6, invert=False, clip=False):     # visualization.py:199-235
  array = np.asarray(array, np.float32)
  if cmin is None:
    cmin = array.min()
  if cmax is None:
    cmax = array.max()
  if isinstance(colormap, str):
    colormap = get_colormap(colormap)
  x = scale_values(array, cmin, cmax, eps)
  if clip:
    x = np.clip(x, 0.0, 1.0)
  colorized = interpolate_colormap(np.float32(1.0) - x if invert else x, np.asarray(colormap, np.float64))
  colorized[x > 1.0] = 0.0 if invert else 1.0
  colorized[x < 0.0] = 1.0 if invert else 0.0
  return colorized


def image_to_uint8(image):                                             # image_utils.py:124-131
  if image.dtype == np.uint8:
    return image
  if not issubclass(image.dtype.type, np.floating):
    raise ValueError(f'Input image should be a floating type but is of type {image.dtype!r}')
  return (image * 255).clip(0.0, 255).astype(np.uint8)


def normalize_vector(v):                                               # model_utils.py:438-442
  v = np.asarray(v, np.float32)
  eps = np.finfo(np.float32).eps
  n2 = (v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1]) + v[..., 2] * v[..., 2]
  return v / np.sqrt(np.maximum(n2, eps))[..., None]


def frame_images(records, height, width, near, far, colormap):
  """records: [H*W, 26] float32 ray records (include/nerfds.h) -> (rgb uint8 [H, W, 3], debug uint8 [2H, 3W, 3])."""
  r = np.asarray(records, np.float32).reshape(height, width, RAY_REC)
  rgb = r[..., F_RGB:F_RGB + 3]
  depth_med = r[..., F_MED_DEPTH]
  ray_norm = normalize_vector(r[..., F_NORM:F_NORM + 3]) / np.float32(2.0) + np.float32(0.5)            # render.py:237-239
  ray_delta_x = np.abs(r[..., F_DELTA_X:F_DELTA_X + 3]) * np.float32(10)                                 # render.py:241-243
  med_points = ((r[..., F_MED_POINTS:F_MED_POINTS + 5] + np.float32(1.5)) / np.float32(3))[..., :3]      # render.py:245-246, 264
  mask = np.broadcast_to(r[..., F_MASK:F_MASK + 1], rgb.shape)                                           # render.py:248-250
  depth_viz = colorize(depth_med, cmin=near, cmax=far, colormap=colormap, invert=True)                   # render.py:263
  row1 = np.concatenate([rgb, depth_viz, ray_norm], axis=1)                                              # render.py:266
  row2 = np.concatenate([mask, ray_delta_x, med_points], axis=1)                                         # render.py:267
  debug = np.concatenate([row1, row2], axis=0)                                                           # render.py:268
  return image_to_uint8(rgb), image_to_uint8(debug)
