"""The step -> scalar curves train.py feeds into the model every iteration (``state.*_alpha``, ``scalar_params``; train.py:401-427),
i.e. the configs of ``hypernerf/schedules.py`` as plain functions.  ``build(cfg)`` accepts what gin hands to ``schedules.from_config``
(schedules.py:26-51): ``None``, a ``(kind, *args)`` tuple, or a dict with a ``'type'`` key, and returns ``f(step) -> float | None``.
Kinds and formulas (schedules.py): constant (71-82), linear (85-99), exponential (102-124), cosine_easing (127-141),
step (144-167), piecewise of (duration, schedule) pairs (170-183), delayed (186-199)."""
from __future__ import annotations

import math
from typing import Any, Callable, Optional


def _constant(value):
  return lambda step: None if value is None else float(value)


def _linear(initial_value, final_value, num_steps):
  def f(step):
    if num_steps == 0:
      return float(final_value)
    a = min(step / num_steps, 1.0)
    return (1.0 - a) * initial_value + a * final_value
  return f


def _exponential(initial_value, final_value, num_steps, eps=1e-10):
  if initial_value <= final_value:
    raise ValueError('Final value must be less than initial value.')
  def f(step):
    if step >= num_steps:
      return float(final_value)
    return initial_value * (max(final_value, eps) / initial_value) ** (step / (num_steps - 1))
  return f


def _cosine_easing(initial_value, final_value, num_steps):
  def f(step):
    x = min(max(min(step / num_steps, 1.0), 0.0), 1.0)
    return initial_value + (final_value - initial_value) * 0.5 * (1 + math.cos(math.pi * x + math.pi))
  return f


def _step(initial_value, decay_interval, decay_factor, max_decays, final_value=None):
  last = initial_value * decay_factor ** max_decays if final_value is None else final_value
  def f(step):
    phase = step // decay_interval
    return last if phase >= max_decays else initial_value * decay_factor ** phase
  return f


def _piecewise(schedules):
  parts = [build(s) for _, s in schedules]
  ends, tot = [], 0
  for duration, _ in schedules:
    tot += duration
    ends.append(tot)
  ends = ends[:-1]                                    # milestones = cumsum(durations)[:-1]
  def f(step):
    idx = sum(1 for e in ends if e <= step)           # searchsorted(milestones, step, side='right')
    return parts[idx](step - (ends[idx - 1] if idx >= 1 else 0))
  return f


def _delayed(base_schedule, delay_steps, delay_mult):
  base = build(base_schedule)
  def f(step):
    rate = delay_mult + (1 - delay_mult) * math.sin(0.5 * math.pi * min(max(step / delay_steps, 0.0), 1.0))
    return rate * base(step)
  return f


_KINDS = {'constant': _constant, 'linear': _linear, 'exponential': _exponential, 'cosine_easing': _cosine_easing, 'step': _step,
          'piecewise': _piecewise, 'delayed': _delayed}


def build(cfg: Any) -> Callable[[int], Optional[float]]:
  if cfg is None:
    return lambda step: None
  if callable(cfg):
    return cfg
  if isinstance(cfg, (tuple, list)):
    kind, *args = cfg
    return _KINDS[kind](*args)
  if isinstance(cfg, dict):
    d = dict(cfg)
    return _KINDS[d.pop('type')](**d)
  raise ValueError(f'Unknown type {type(cfg)}.')
