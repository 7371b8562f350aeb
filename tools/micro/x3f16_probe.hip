// one wave: D = A * B with v_mfma_f32_32x32x16_f16 on split-f16 operands (hi + lo), against the fp64 product - does the split-f16 product work on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void probe(const float* A /*[32][16]*/, const float* B /*[16][32]*/, float* C /*[32 rows][32 cols]*/, int mode) {
  const int l = threadIdx.x, m = l & 31, h = l >> 5;
  f16x8 ah, al, bh, bl;
  for (int i = 0; i < 8; ++i) {
    const float a = A[m * 16 + 8 * h + i], b = B[(8 * h + i) * 32 + m];
    ah[i] = (_Float16)a; al[i] = (_Float16)(a - (float)ah[i]);
    bh[i] = (_Float16)b; bl[i] = (_Float16)(b - (float)bh[i]);
  }
  f32x16 acc = {0};
  if (mode >= 1) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0); }
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + m] = acc[r];
}
int main() {
  float hA[512], hB[512], hC[1024];
  unsigned s = 1;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
  for (int i = 0; i < 512; ++i) { hA[i] = 0.2f * rnd(); hB[i] = 2.0f * rnd(); }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode);
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    double worst = 0, mx = 0;
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
      double ref = 0; for (int k = 0; k < 16; ++k) ref += (double)hA[r * 16 + k] * hB[k * 32 + c];
      worst = fmax(worst, fabs(ref - hC[r * 32 + c])); mx = fmax(mx, fabs(ref));
    }
    printf("mode %d (%s): max |ref| %.4f, max abs err %.3e (rel %.3e), C[0][0] %.6f\n", mode, mode ? "hi lo + lo hi + hi hi" : "hi hi only", mx, worst, worst / mx, hC[0]);
  }
  return 0;
}
