"""Repo-root pytest bootstrap: make the in-tree package (nerf-ds_amd/nerfds_amd) importable."""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(_ROOT, 'nerf-ds_amd'), _ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)
