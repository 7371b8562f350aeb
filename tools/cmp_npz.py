"""Development aid: bitwise comparison of two .npz dumps of tools/dump_render.py."""
import sys, numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
  if k not in b.files:
    continue
  x, y = a[k], b[k]
  same = x.shape == y.shape and np.array_equal(x.view(np.uint8) if x.dtype != object else x, y.view(np.uint8) if y.dtype != object else y)
  if not same:
    bad += 1
    d = np.abs(x.astype(np.float64) - y.astype(np.float64))
    print(f'DIFF {k}: max abs {np.nanmax(d):.3e}, {np.count_nonzero(d > 0)} of {d.size} elements, nan {np.isnan(x).sum()} / {np.isnan(y).sum()}')
print(f'{len(a.files) - bad} of {len(a.files)} arrays bit-identical')
sys.exit(1 if bad else 0)
