"""Known answers for the step -> scalar curves (hypernerf/schedules.py), incl. the schedules configs/nerf_ds.gin really uses."""
import math
import os

import pytest

from nerfds_amd.sched import build


def test_known_answers():
  assert build(None)(5) is None and build(('constant', 8))(123) == 8.0
  lin = build({'type': 'linear', 'initial_value': 0, 'final_value': 4, 'num_steps': 50000})       # nerf_ds.gin:50-55
  assert lin(0) == 0 and lin(25000) == 2.0 and lin(50000) == 4.0 and lin(10 ** 6) == 4.0
  ex = build(('exponential', 1, 0.1, 30000))
  assert ex(0) == 1 and abs(ex(29999) - 0.1) < 1e-12 and ex(30000) == 0.1 and abs(ex(15000) - 10 ** (-15000 / 29999)) < 1e-12
  with pytest.raises(ValueError):
    build(('exponential', 0.1, 1.0, 10))
  ce = build(('cosine_easing', 0.01, 1e-8, 100000))
  assert abs(ce(0) - 0.01) < 1e-15 and abs(ce(50000) - 0.5 * (0.01 + 1e-8)) < 1e-12 and abs(ce(200000) - 1e-8) < 1e-15
  st = build({'type': 'step', 'initial_value': 1.0, 'decay_interval': 10, 'decay_factor': 0.5, 'max_decays': 3})
  assert [st(s) for s in (0, 9, 10, 29, 30, 1000)] == [1.0, 1.0, 0.5, 0.25, 0.125, 0.125]
  # nerf_ds.gin:120-126 sharp_mask_std: exponential 1 -> 0.1 over 30k steps, then constant 0.1 (milestones are durations)
  pw = build({'type': 'piecewise', 'schedules': [(30000, ('exponential', 1, 0.1, 30000)), (220000, ('constant', 0.1))]})
  assert pw(0) == 1 and abs(pw(29999) - 0.1) < 1e-12 and pw(30000) == 0.1 and pw(10 ** 6) == 0.1
  # x_for_rgb_alpha (nerf_ds.gin:129-135): the second piece restarts its own clock at the milestone
  pw2 = build({'type': 'piecewise', 'schedules': [(50000, ('constant', 0)), (50000, ('linear', 0, 4.0, 50000)), (150000, ('constant', 4.0))]})
  assert pw2(49999) == 0 and pw2(50000) == 0 and pw2(75000) == 2.0 and pw2(100000) == 4.0
  dl = build({'type': 'delayed', 'delay_steps': 2500, 'delay_mult': 0.01, 'base_schedule': ('constant', 1e-3)})
  assert abs(dl(0) - 1e-5) < 1e-18 and abs(dl(2500) - 1e-3) < 1e-18 and abs(dl(1250) - 1e-3 * (0.01 + 0.99 * math.sin(math.pi / 4))) < 1e-15


@pytest.mark.skipif(not os.path.exists('/root/reference/configs/nerf_ds.gin'), reason='reference tree only exists in the authoring container')
def test_every_schedule_of_the_reference_gin_files_builds():
  from nerfds_amd.gin_subset import resolve
  n = 0
  for name in ('nerf_ds.gin', 'base.gin'):
    for k, v in resolve('/root/reference/configs/' + name).items():
      if k.endswith('_schedule') or k.endswith('_sched'):
        f = build(v)
        for step in (0, 1, 1000, 250000):
          r = f(step)
          assert r is None or math.isfinite(r), (k, step, r)
        n += 1
  assert n >= 8
