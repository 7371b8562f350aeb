#include <hip/hip_runtime.h>
__global__ void k1(float* p, float v) { unsafeAtomicAdd(p + threadIdx.x, v); }
__global__ void k2(float* p, float v) { __hip_atomic_fetch_add(p + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__global__ void k3(float* p, float v) { __hip_atomic_fetch_add(p + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k4(float* p, float v, int* o) { int x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); o[0] = x; }
