// "More than 64 KiB of dynamic LDS" launch attribute, set once per (kernel, device).
// The function handle behind a __global__ symbol is per-device state: a process-wide "done" flag (round 3 / 4: `static bool attr`
// in train_gemm.hip, a per-device but kernel-blind table in launch.h) leaves the second device of a single-process multi-GPU
// trainer - or the second kernel on a device - without the attribute.  One table per library, keyed by both (lds_attr.cpp).
#pragma once
#include <hip/hip_runtime.h>

namespace nerfds {
// true exactly once per (kernel, device) pair; thread-safe.  Host logic only (tests/test_launch_guard.py drives it without a GPU
// through nerfds_debug_lds_attr_first_use).
bool lds_attr_first_use(const void* kernel, int device);

inline void allow_dynamic_lds(const void* kernel, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds_attr_first_use(kernel, dev)) (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace nerfds
