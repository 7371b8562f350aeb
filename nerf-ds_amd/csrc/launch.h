// Launch-side helpers shared by the three fused-kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#ifndef NERFDS_GRAPH
#error "compile with -DNERFDS_GRAPH=<GraphNerfDS|GraphStatic|GraphHyperNeRF> -DNERFDS_NAME=<suffix> (render: -DNERFDS_PREC=<P_BF16|P_BF16X3|P_F32|P_F16> or -DNERFDS_MIXED)"
#endif
#define NERFDS_CAT2(a, b) a##b
#define NERFDS_CAT(a, b) NERFDS_CAT2(a, b)

namespace nerfds {
// Kernels that use more than 64 KiB of dynamic LDS need the attribute set once PER DEVICE (the function handle is per-device state: a
// process-wide "done" flag would leave the second device of a multi-GPU process without it - ADVICE r3).
inline void allow_dynamic_lds(const void* kernel, int bytes) {
  static bool done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !done[dev]) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (dev >= 0 && dev < 64) done[dev] = true;
  }
}
}  // namespace nerfds
