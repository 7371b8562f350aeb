import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
np.set_printoptions(linewidth=250)
for k in a.files:
    x, y = a[k], b[k]
    e = np.abs(x - y).reshape(7, -1, x.shape[-1] if x.ndim == 3 else 1).max(-1)
    blk = (e.reshape(7, -1, 32).max(-1) > 1e-3).astype(int)
    print(k, 'max %.1e' % e.max(), 'n_bad', int((e > 1e-3).sum()), 'blocks(ray x 32-sample):', blk.tolist() if e.max() > 1e-3 else '')
