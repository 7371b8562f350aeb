"""CPU: the decision table of the trainer's overflow ladder (nerfds_amd/training.py Trainer._escalate / _relax; include/nerfds.h NERFDS_OVF_*).
No device work: the Trainer object is built without its constructor and only the policy fields are touched."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'nerf-ds_amd'))
from nerfds_amd import training as TR      # noqa: E402


def _bare():
  t = TR.Trainer.__new__(TR.Trainer)
  t.loss_scale_adjust, t.tangent_scale_adjust, t.split_chains, t.fp32_step = 0, 0, False, False
  return t


def _policy(t):
  return (t.loss_scale_adjust, t.tangent_scale_adjust, t.split_chains, t.fp32_step)


def test_earliest_cause_decides():
  # activation (or an fp32 value of the forward pass) -> the fp32 step, whatever else is set behind it
  for src in (TR.OVF_ACTIVATION, TR.OVF_ACTIVATION | TR.OVF_PRIMAL_G | TR.OVF_COTANGENT, TR.OVF_FP32 | TR.OVF_PRIMAL_G, TR.OVF_FP32):
    t = _bare()
    assert t._escalate(src, 0) == 'fp32 step' and _policy(t) == (0, 0, False, True), src
  # tangents / cotangents -> the second-order knobs; the primal scale is NOT touched even though the primal g (their consequence) overflowed too
  t = _bare()
  t._escalate(TR.OVF_COTANGENT | TR.OVF_PRIMAL_G | TR.OVF_FP32_BACKWARD | TR.OVF_FP32_SECOND_ORDER, 0)
  assert _policy(t) == (0, -2, True, False)
  t._escalate(TR.OVF_TANGENT, 0)
  assert _policy(t) == (0, -4, True, False)
  # primal g alone -> the primal scale alone
  t = _bare()
  t._escalate(TR.OVF_PRIMAL_G, 0)
  assert _policy(t) == (-2, 0, False, False)
  # an fp32 cotangent with no f16 array in front of it: no power of two helps
  for src in (TR.OVF_FP32_BACKWARD, TR.OVF_FP32_SECOND_ORDER, TR.OVF_FP32_BACKWARD | TR.OVF_PRIMAL_G):
    t = _bare()
    assert t._escalate(src, 0) == 'fp32 step' and t.fp32_step, src


def test_knobs_are_bounded_and_end_in_the_fp32_step():
  t = _bare()
  for _ in range(12):
    t._escalate(TR.OVF_PRIMAL_G, 0)
  assert t.loss_scale_adjust == -24 and not t.fp32_step
  t._escalate(TR.OVF_PRIMAL_G, 0)
  assert t.fp32_step
  t = _bare()
  for _ in range(8):
    t._escalate(TR.OVF_COTANGENT, 0)
  assert t.tangent_scale_adjust == -16 and t.split_chains and not t.fp32_step
  t._escalate(TR.OVF_COTANGENT, 0)
  assert t.fp32_step


def test_unattributed_alternates_then_gives_up_on_f16():
  t = _bare()
  acts = [t._escalate(0, turn) for turn in range(5)]
  assert _policy(t) == (-4, -4, True, True), (acts, _policy(t))
  assert acts[-1] == 'fp32 step' and 'loss scale' in acts[1] and 'tangent scale' in acts[0]


def test_relax_walks_back_one_notch_at_a_time():
  t = _bare()
  t.loss_scale_adjust, t.tangent_scale_adjust, t.split_chains, t.fp32_step = -2, -1, True, True
  seen = []
  for _ in range(5):
    t._relax()
    seen.append(_policy(t))
  assert seen == [(-2, -1, True, False), (-1, 0, True, False), (0, 0, True, False), (0, 0, False, False), (0, 0, False, False)], seen


def test_names():
  assert 'activation' in TR.overflow_names(TR.OVF_ACTIVATION) and 'unattributed' in TR.overflow_names(0)
  assert TR.overflow_names(TR.OVF_PRIMAL_G | TR.OVF_COTANGENT).count(',') == 1
