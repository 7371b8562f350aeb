"""Generates tests/golden/frame_colorize_ref.npz: REFERENCE-RUN vectors for the depth colourisation of the frame output path.

    python tests/golden/make_frame_golden.py          (build container only: needs /root/reference and matplotlib)

This is the one piece of the hot path's surroundings that the reference can execute here: hypernerf/visualization.py
needs only numpy + matplotlib.  The script loads THAT FILE from /root/reference (nothing of it is copied), calls its own
``get_colormap``, ``scale_values``, ``interpolate_colormap`` and ``colorize`` (visualization.py:173-235) on seeded float32
median-depth maps with the arguments render.py:263 uses (cmin = near, cmax = far, cmap = the default 'magma', invert = True)
plus the variants the function supports (no inversion, clip, 'turbo', auto range), and stores inputs and outputs.  Only
the .npz travels; tests/test_frames.py compares the oracle (CPU) and nerfds_frame_images (GPU, byte-exact) with it.
"""
import importlib.util
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference/hypernerf/visualization.py'


def load_reference():
  spec = importlib.util.spec_from_file_location('ref_visualization', REF)
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def depth_map(rng, H, W, near, far):
  """float32 median depths: mostly inside [near, far], some outside on both sides, the exact end points, zeros (med_depth of an
  empty ray, model_utils.py:316-317) and the values that land exactly on table entries."""
  d = (rng.random((H, W)) * (far - near) * 1.3 + near - 0.15 * (far - near)).astype(np.float32)
  flat = d.reshape(-1)
  flat[:8] = np.array([near, far, 0.0, near - 1, far + 1, (near + far) / 2, near + (far - near) / 255, far - (far - near) / 255], np.float32)
  flat[8:8 + 256] = (near + (far - near) * np.arange(256) / 255).astype(np.float32)[:max(0, min(256, flat.size - 8))]
  return d


def main():
  viz = load_reference()
  import matplotlib
  rng = np.random.default_rng(20240930)
  out = {'matplotlib_version': np.array(matplotlib.__version__), 'numpy_version': np.array(np.__version__)}
  for name in ('magma', 'turbo', 'sinebow'):
    out[f'table_{name}'] = np.asarray(viz.get_colormap(name), np.float64)
  cases = []
  with warnings.catch_warnings():
    warnings.simplefilter('ignore', RuntimeWarning)      # the uint16 cast of out-of-range pixels (overwritten at visualization.py:231-232)
    for ci, (H, W, near, far) in enumerate([(24, 40, 0.3, 1.7), (16, 24, 2.0, 6.0), (8, 40, 0.05, 0.051)]):
      d = depth_map(rng, H, W, near, far)
      out[f'c{ci}_depth'] = d
      out[f'c{ci}_near_far'] = np.array([near, far], np.float64)
      out[f'c{ci}_scaled'] = viz.scale_values(d, near, far)
      out[f'c{ci}_magma_inv'] = viz.colorize(d, cmin=near, cmax=far, invert=True)                 # render.py:263
      out[f'c{ci}_magma'] = viz.colorize(d, cmin=near, cmax=far)
      out[f'c{ci}_turbo_inv'] = viz.colorize(d, cmin=near, cmax=far, cmap='turbo', invert=True)
      out[f'c{ci}_magma_inv_clip'] = viz.colorize(d, cmin=near, cmax=far, invert=True, clip=True)
      out[f'c{ci}_magma_auto'] = viz.colorize(d)                                                   # cmin / cmax from the data
      cases.append(ci)
    # interpolate_colormap on its own grid (values in [0, 1] incl. both ends and every table knot +- one ulp)
    k = (np.arange(256) / 255).astype(np.float32)
    v = np.concatenate([k, np.nextafter(k, np.float32(2)), np.nextafter(k, np.float32(-1)).clip(0, 1), rng.random(512).astype(np.float32)])
    out['interp_values'] = v
    out['interp_magma'] = viz.interpolate_colormap(v, viz.get_colormap('magma'))
  out['cases'] = np.array(cases)
  path = os.path.join(HERE, 'frame_colorize_ref.npz')
  np.savez_compressed(path, **out)
  print(path, os.path.getsize(path), 'bytes;', {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.startswith('c0_')})


if __name__ == '__main__':
  main()
