// Micro-benchmark (development aid): what does a 4-waves-per-CU persistent kernel get out of its global stores, by access pattern?
// Every wave writes 32-row x 64-byte tiles of a row-major [M][W] 16-bit array (the training forward's activation stores), W = 256.
//   0: 4 x 8-byte pieces per lane   (lane l: row l & 31, features 8g + 4 (l >> 5))         - round 2 / first half of round 3
//   1: 2 x 16-byte pieces per lane  (lane l: row l & 31, features 16 (l >> 5) + 8k)        - shipped
//   2: whole 128-byte lines: lane l writes 16 bytes at row l >> 3, byte (l & 7) * 16 (8 rows per instruction)
//   3: whole lines: lane l writes 16 bytes at row l >> 2, byte (l & 3) * 16 of the tile's 64-byte row piece (16 rows per instruction)
//   4: 1 KiB contiguous per instruction (no row structure: the ceiling)
// usage: store_bench <pattern> [valu_per_store]   (valu_per_store: dependent v_fma filler between stores, to look at overlap)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
extern __shared__ char smem[];
template <int PAT> __global__ __launch_bounds__(256) void k(unsigned short* base, long long M, int iters, int filler) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long wid = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
  constexpr int W = 256;
  u32x4 v = {(unsigned)lane, (unsigned)wave, 3u, 4u};
  float f = lane;
  for (int it = 0; it < iters; ++it) {
    const long long row0 = ((wid + (long long)it * nw) * 32) % (M - 32);
    for (int tile = 0; tile < W / 32; ++tile) {
      for (int q = 0; q < filler; ++q) f = __builtin_fmaf(f, 1.0001f, 0.5f);
      v[0] += (unsigned)f;
      if (PAT == 0) {
        unsigned short* p = base + (row0 + (lane & 31)) * W + 32 * tile + 4 * (lane >> 5);
        for (int g = 0; g < 4; ++g) *reinterpret_cast<u32x2*>(p + 8 * g) = u32x2{v[0], v[1] + (unsigned)g};
      } else if (PAT == 1) {
        unsigned short* p = base + (row0 + (lane & 31)) * W + 32 * tile + 16 * (lane >> 5);
        *reinterpret_cast<u32x4*>(p) = v; *reinterpret_cast<u32x4*>(p + 8) = v;
      } else if (PAT == 2) {     // as 3 but 8 rows x 128 bytes: two adjacent tiles per instruction (whole 128-byte lines)
        unsigned short* p = base + (row0 + 8 * (tile & 3) + (lane >> 3)) * W + 64 * (tile >> 2) + 8 * (lane & 7);
        *reinterpret_cast<u32x4*>(p) = v; *reinterpret_cast<u32x4*>(p + 128) = v;
      } else if (PAT == 3) {
        for (int hh = 0; hh < 2; ++hh) {
          unsigned short* p = base + (row0 + 16 * hh + (lane >> 2)) * W + 32 * tile + 8 * (lane & 3);
          *reinterpret_cast<u32x4*>(p) = v;
        }
      } else if (PAT == 5) {     // no store at all: the filler alone
        if (v[0] == 0xdeadbeefu) base[lane] = 1;
      } else {
        unsigned short* p = base + row0 * W + (tile * 2) * 512 + lane * 8;
        *reinterpret_cast<u32x4*>(p) = v; *reinterpret_cast<u32x4*>(p + 512) = v;
      }
    }
  }
  if (f == 12345.f) base[0] = 1;
}
int main(int argc, char** argv) {
  const int pat = argc > 1 ? atoi(argv[1]) : 1, filler = argc > 2 ? atoi(argv[2]) : 0;
  const long long M = 1 << 21;                 // 2 M rows x 512 B = 1 GiB
  unsigned short* d; (void)hipMalloc(&d, M * 256 * 2);
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount, iters = 64;
  auto launch = [&](int n) {
    const int lds = 100 * 1024;                // one workgroup per CU
#define L(P) { (void)hipFuncSetAttribute((const void*)k<P>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipLaunchKernelGGL(k<P>, dim3(grid), dim3(256), lds, 0, d, M, n, filler); }
    switch (pat) { case 0: L(0) break; case 1: L(1) break; case 2: L(2) break; case 3: L(3) break; case 5: L(5) break; default: L(4) }
  };
  launch(4); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0); launch(iters); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)grid * 4 * iters * 8 * 2048;
  printf("pattern %d filler %d: %.3f ms  %.0f GB/s  (%.0f cycles per KiB-store per wave at 2.2 GHz)\n", pat, filler, ms, bytes / ms / 1e6,
         ms * 1e-3 * 2.2e9 / (iters * 8 * 2));
  return 0;
}
