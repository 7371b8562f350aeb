// NerfModel._encode_embed (hypernerf/models.py:271-294), the body of evaluation.encode_metadata (evaluation.py:29-50): per-ray metadata ->
// GLO vectors.  One id channel: the table row (GLOEmbed, modules.py:336-348); three channels (left id, right id, progression):
// (1 - progression) * row(left) + progression * row(right).  HBM-bound (4 - 12 B in, 32 B out per ray), one thread per output float.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerfds {

__global__ void encode_embed_kernel(const float* __restrict__ table, int rows, const float* __restrict__ meta, int channels, long long n,
                                    float* __restrict__ out) {
  constexpr int D = 8;        // glo_num_dims of every built graph
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n * D; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / D;
    const int c = (int)(i % D);
    auto row = [&](float id) {        // astype(uint32), then the clamp of a jnp gather
      uint32_t k = (uint32_t)id;
      k = k < (uint32_t)rows ? k : (uint32_t)(rows - 1);
      return table[(size_t)k * D + c];
    };
    if (channels == 1) {
      out[i] = row(meta[r]);
    } else {
      const float p = meta[3 * r + 2];
      // the reference's fp32 expression, rounded product by product (no fma contraction)
      out[i] = __fadd_rn(__fmul_rn(__fsub_rn(1.0f, p), row(meta[3 * r])), __fmul_rn(p, row(meta[3 * r + 1])));
    }
  }
}

}  // namespace nerfds

extern "C" void nerfds_launch_encode_embed(const float* table, int rows, const float* meta, int channels, long long n, float* out, void* stream) {
  if (n <= 0) return;
  const int block = 256;
  const long long want = (n * 8 + block - 1) / block;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(nerfds::encode_embed_kernel, dim3(grid), dim3(block), 0, static_cast<hipStream_t>(stream), table, rows, meta, channels, n, out);
}
