"""The > 64 KiB dynamic-LDS launch attribute is set once per (kernel, device) pair (csrc/lds_attr.h): the function handle behind a kernel
symbol is per-device state, so neither a process-wide flag nor a per-device flag shared by all kernels is enough.  Host logic, no GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'nerf-ds_amd'))
from nerfds_amd import _native as N


def test_guard_is_keyed_by_kernel_and_device():
  first = N.load().nerfds_debug_lds_attr_first_use
  k1, k2 = 0xfeed0000beef0010, 0xfeed0000beef0020        # keys no real kernel address can take
  assert first(k1, 0) == 1 and first(k1, 0) == 0
  assert first(k1, 1) == 1 and first(k1, 1) == 0         # the second device of one process still gets the attribute
  assert first(k2, 0) == 1 and first(k2, 1) == 1         # ... and so does a second kernel on a device already seen
  assert first(k2, 0) == 0 and first(k1, 7) == 1


def test_no_process_wide_attribute_flags_in_the_sources():
  """Every hipFuncSetAttribute of the library goes through lds_attr.h."""
  csrc = os.path.join(os.path.dirname(__file__), '..', 'nerf-ds_amd', 'csrc')
  for name in os.listdir(csrc):
    if name.endswith(('.hip', '.cpp', '.h')) and name != 'lds_attr.h':
      text = open(os.path.join(csrc, name)).read()
      assert 'hipFuncSetAttribute(' not in text, name
