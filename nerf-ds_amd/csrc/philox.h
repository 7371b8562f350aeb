// Sampling uniforms drawn on chip when the caller injects none (nerfds_rand.t_rand / u_rand == NULL).  JAX's threefry
// streams (jax.random.uniform, model_utils.py:84,217) cannot be reproduced without JAX; the contract here is a
// counter-based Philox4x32-10 stream (Salmon et al. 2011) keyed by nerfds_rand.seed, shared by the fused render kernel
// and the trainer so that both draw the same jitter for the same (seed, ray, sample):
//   coarse sample i of ray r : philox4(lo32(r), 0, i >> 2, hi32(r), seed)[i & 3]
//   fine   sample k of ray r : philox4(lo32(r), 1, k >> 2, hi32(r), seed)[k & 3]        r = nerfds_rand.first_ray + ray index in the call
#pragma once
#include <stdint.h>

namespace nerfds {

__device__ __forceinline__ void philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, float (&u)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  u[0] = (c0 >> 8) * 5.9604645e-8f; u[1] = (c1 >> 8) * 5.9604645e-8f;
  u[2] = (c2 >> 8) * 5.9604645e-8f; u[3] = (c3 >> 8) * 5.9604645e-8f;
}
// level: 0 = coarse jitter (model_utils.py:84), 1 = fine inverse-CDF uniforms (model_utils.py:217)
__device__ __forceinline__ float sample_uniform(uint64_t seed, long long ray, int level, int index) {
  float r4[4];
  philox4((uint32_t)ray, (uint32_t)level, (uint32_t)(index >> 2), (uint32_t)((unsigned long long)ray >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), r4);
  return r4[index & 3];
}

}  // namespace nerfds
