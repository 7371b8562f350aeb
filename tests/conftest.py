import os
import sys

import pytest

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(_ROOT, 'nerf-ds_amd'), _ROOT):
  if p not in sys.path:
    sys.path.insert(0, p)


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
  """GPU tests are skipped (not failed) when no device is visible, so a bare `pytest tests/` works anywhere."""
  try:
    import torch
    have_gpu = torch.cuda.is_available()
  except Exception:   # pragma: no cover
    have_gpu = False
  if have_gpu:
    return
  skip = pytest.mark.skip(reason='no GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)
