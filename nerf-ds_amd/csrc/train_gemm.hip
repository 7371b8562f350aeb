// The dense layers of the training step (nerfds_train.cpp), hand-written for gfx950.  Reference: the nn.Dense calls of
// hypernerf/modules.py:56-88 (MLP), 228-289 (NerfMLP) under jax.value_and_grad (training.py:494).  Three kernels:
//
//   k_dense_dma   Y[M x N] = epilogue(concat_s X_s[M x K_s] . W): forward layers and, on a transposed fragment pack, the data
//                 gradient dX = dZ . W^T.  Weight stationary: M = rays x samples (up to 524 288), K and N <= 560 / 256, so the layer's
//                 whole weight matrix fits the register file of ONE workgroup (256 x 256 as split bf16 = 256 KiB of the CU's
//                 512 KiB): wave w keeps output tile w (32 features) as MFMA A-fragments and the persistent workgroup walks over
//                 32-sample tiles of X, which arrive by LDS-DMA (fp32, 3-4 tiles deep) and are split into bf16 hi / lo on the way
//                 to the MFMAs (3 MFMAs per fragment; P3: hi / mid / lo, 6 MFMAs, fp32-level products).
//   k_dense_ws    the same layer with the X tile staged through registers (rows that are not 16-byte aligned: 33 / 45-wide inputs).
//   k_wgrad       dW[K x N] += X^T dZ: 16-sample tiles of both operands by LDS-DMA, transposed + split once per workgroup into
//                 fragment images, output tiles accumulated in registers for the whole kernel, float atomics at the end.
//   k_wgrad_tr    the same product with BOTH operands 16-bit (f16 activations, loss-scaled f16 g - the fused training step): the DMA'd stage is the
//                 operand store, transposed reads (ds_read_b64_tr_b16), one MFMA per product; several layers of an MLP per launch.
//   k_wgrad_head  heads (f16 X, fp32 dZ of <= 6 columns): VALU, bound by the read of X.
//
// Epilogue of the first two: bias, ReLU | accumulate, the consumer's ReLU mask (dX . 1[y_prev > 0]), column sums (bias gradient).
// Every layer is HBM bound (X read once, Y written once); DESIGN.md section 8.1 has the measurements and what was tried.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "train_gemm.h"
#include "lds_attr.h"

namespace nerfds_train {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

extern __shared__ __attribute__((aligned(16))) char g_tile[];      // the one LDS object of every kernel here (a second one de-pipelines LDS-DMA code)

__device__ __forceinline__ unsigned short bf16_rne(float f) {
  unsigned u = __builtin_bit_cast(unsigned, f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// Packs the weights of one layer into fragment order: fragment (tile ot, chunk kc, part hi|lo), lane l = (h << 5) | m holds, for
// output feature 32 ot + m, the 8 k-slots 16 kc + 8 h + i.  transpose = 0: value(k, n) = W[(row0 + k) * ldw + n] (forward:
// `in` = rows of W); transpose = 1: value(k, n) = W[(row0 + n) * ldw + k] (backward data: in = columns of W, out = its rows).
__device__ __forceinline__ void pack_one(const float* __restrict__ W, int ldw, int row0, int in_dim, int out_dim, int transpose, int parts, u32x4* __restrict__ out,
                                         long long idx) {
  const int KC = (in_dim + 15) / 16, tiles = (out_dim + 31) / 32;
  if (idx >= (long long)tiles * KC * 64) return;
  const int lane = (int)(idx % 64), kc = (int)((idx / 64) % KC), ot = (int)(idx / (64LL * KC));
  const int m = lane & 31, h = lane >> 5, n = 32 * ot + m;
  unsigned short hi[8], mid[8], lo[8];
  for (int i = 0; i < 8; ++i) {
    const int k = 16 * kc + 8 * h + i;
    float v = 0.f;
    if (k < in_dim && n < out_dim) v = transpose ? W[(size_t)(row0 + n) * ldw + k] : W[(size_t)(row0 + k) * ldw + n];
    hi[i] = bf16_rne(v);
    const float r = v - bf16_f32(hi[i]);
    mid[i] = bf16_rne(r);                                   // parts == 2: this is the "lo" of the two-way split
    lo[i] = bf16_rne(r - bf16_f32(mid[i]));
  }
  u32x4 a, b, c;
  for (int i = 0; i < 4; ++i) {
    a[i] = hi[2 * i] | ((unsigned)hi[2 * i + 1] << 16);
    b[i] = mid[2 * i] | ((unsigned)mid[2 * i + 1] << 16);
    c[i] = lo[2 * i] | ((unsigned)lo[2 * i + 1] << 16);
  }
  const size_t base = ((size_t)(ot * KC + kc) * parts) * 64;
  out[base + lane] = a;
  out[base + 64 + lane] = b;
  if (parts == 3) out[base + 128 + lane] = c;
}

__global__ void k_pack_frags(const float* __restrict__ W, int ldw, int row0, int in_dim, int out_dim, int transpose, int parts, u32x4* __restrict__ out) {
  pack_one(W, ldw, row0, in_dim, out_dim, transpose, parts, out, blockIdx.x * (long long)blockDim.x + threadIdx.x);       // (ot, kc, lane)
}
// every pack of a training step in one launch: blockIdx.y = entry
__global__ void k_pack_frags_all(const float* __restrict__ theta, const PackEntry* __restrict__ e, char* __restrict__ arena) {
  const PackEntry E = e[blockIdx.y];
  pack_one(theta + E.woff, E.ldw, E.row0, E.in_dim, E.out_dim, E.transpose, E.parts, reinterpret_cast<u32x4*>(arena + E.dst),
           blockIdx.x * (long long)blockDim.x + threadIdx.x);
}

// Epilogue: this lane holds sample `row`, features 32 ot + 8 q + 4 h + j of the accumulator (q = 0..3, j = 0..3).
// MODE: the 8-wave kernels have no registers to spare, so they are compiled twice - 1 = forward epilogue only (bias, ReLU),
// 2 = backward epilogue only (accumulate, mask, column sums); 0 = everything (4-wave kernels).
template <int MODE>
__device__ __forceinline__ void store_tile(const DenseArgs& A, const f32x16& acc, long long row, int ot, int h, f32x16& csum) {
  if (row >= A.M) return;
  constexpr bool FWD = MODE != 2, BWD = MODE != 1;
  const size_t mrow = (BWD && A.mask_y != nullptr) ? (size_t)(A.mask_div == 3 ? row / 3 : row) * A.ld_mask : 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n0 = 32 * ot + 8 * q + 4 * h;
    f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
    if (FWD && A.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
    if (A.vec_out) {
      if (n0 < A.n_out) {                                      // n_out is a multiple of 4 on this path
        f32x4* dst = reinterpret_cast<f32x4*>(A.y + (size_t)row * A.ldy + n0);
        if (BWD && A.accumulate) v += *dst;                   // the mask and the column sums apply to the TOTAL
        if (BWD && A.mask_y != nullptr) {
          const f32x4 y = *reinterpret_cast<const f32x4*>(A.mask_y + mrow + n0);
#pragma unroll
          for (int j = 0; j < 4; ++j) if (!(y[j] > 0.f)) v[j] = 0.f;
        }
        if constexpr (BWD) { csum[4 * q] += v[0]; csum[4 * q + 1] += v[1]; csum[4 * q + 2] += v[2]; csum[4 * q + 3] += v[3]; }
        *dst = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + j;
        if (n >= A.n_out) continue;
        float x = v[j];
        float* dst = A.y + (size_t)row * A.ldy + n;
        if (BWD && A.accumulate) x += *dst;
        if (BWD && A.mask_y != nullptr && !(A.mask_y[mrow + n] > 0.f)) x = 0.f;
        if constexpr (BWD) csum[4 * q + j] += x;
        *dst = x;
      }
    }
  }
}

// End of the kernel: the per-lane column sums of this wave's 32 features (16 per lane, over the samples the lane stored) are
// reduced over the 32 lanes of a half and added to colsum[] (one atomic per feature and wave).
__device__ __forceinline__ void flush_colsum(const DenseArgs& A, f32x16 csum, int ot, int h, int m) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = csum[r];
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    const int n = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (m == 0 && n < A.n_out) atomicAdd(A.colsum + (A.nrep > 1 ? (size_t)(blockIdx.x % A.nrep) * A.rep_stride : 0) + n, v);
  }
}

// Position of sample s inside the 32 x 16-byte row group of (chunk c, half hh): rotated by 2 c + hh so that the tile WRITE (lanes
// along k: same s, different (c, hh)) spreads over the LDS banks; the B-operand READ (lanes along s) stays a permutation.
__device__ __forceinline__ int slot_of(int c, int hh, int s) { return (s + 2 * c + hh) & 31; }

template <int KC, int WAVES, bool P3>
__global__ __launch_bounds__(64 * WAVES) void k_dense_ws(const DenseArgs A) {
  constexpr int PARTS = P3 ? 3 : 2;            // bf16 pieces per operand (P3: hi / mid / lo = fp32-level products, 6 MFMAs)
  constexpr int R = 32 / WAVES;                  // rows of the X tile this wave fetches
  constexpr int QPR = (KC * 4 + 63) / 64;        // float4 per lane per row
  constexpr int TILE = KC * 1024 * PARTS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 31, h = lane >> 5;
  const int n_tiles = (A.n_out + 31) >> 5;
  const bool computes = wave < n_tiles;
  const int ot = computes ? wave : 0;
  // this wave's weight slice, resident for the whole kernel
  bf16x8 wh[KC], wl[KC], wm[P3 ? KC : 1];
  const u32x4* wfrag = static_cast<const u32x4*>(A.wfrag);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const size_t base = ((size_t)(ot * KC + kc) * PARTS) * 64;
    wh[kc] = __builtin_bit_cast(bf16x8, wfrag[base + lane]);
    if constexpr (P3) { wm[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 64 + lane]); wl[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 128 + lane]); }
    else wl[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 64 + lane]);
  }
  const int K = A.k_total;
  const long long tiles = (A.M + 31) / 32;
  f32x16 csum = {};

  // Where this lane's quads of a tile row come from does not depend on the row: resolved once (static indices only - a
  // dynamically indexed kernel-argument array would be copied to scratch).  Segment starts are multiples of 4 (host check),
  // so a quad never straddles two segments; nv = how many of its 4 floats exist.
  const float* qbase[QPR];
  int qld[QPR], qnv[QPR];
#pragma unroll
  for (int q = 0; q < QPR; ++q) {
    int k = 4 * (lane + 64 * q);
    const float* p = nullptr;
    int ld = 0, nv = 0;
    bool done = k >= K;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (!done && g < A.nseg) {
        if (k < A.seg[g].k) { p = A.seg[g].x + k; ld = A.seg[g].ld; nv = A.seg[g].k - k < 4 ? A.seg[g].k - k : 4; done = true; }
        else k -= A.seg[g].k;
      }
    }
    qbase[q] = p; qld[q] = ld; qnv[q] = nv;
  }
  f32x16 bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
    bias[r] = (A.bias != nullptr && n < A.n_out) ? A.bias[n] : 0.f;
  }

  f32x4 pf[R][QPR];
  auto fetch = [&](long long tile) {             // HBM -> registers (fp32, row major)
    const long long m0 = tile * 32;
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const long long row = m0 + wave + WAVES * j;
#pragma unroll
      for (int q = 0; q < QPR; ++q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (qnv[q] > 0 && row < A.M) {
          const float* p = qbase[q] + (size_t)row * qld[q];
          if (A.vec_in) v = *reinterpret_cast<const f32x4*>(p);
          else {
            v[0] = p[0];
            if (qnv[q] > 1) v[1] = p[1];
            if (qnv[q] > 2) v[2] = p[2];
            if (qnv[q] > 3) v[3] = p[3];
          }
        }
        pf[j][q] = v;
      }
    }
  };
  auto stage = [&](char* buf) {                  // registers -> LDS (bf16 hi / lo, B-operand order, rotated slots)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int s = wave + WAVES * j;
#pragma unroll
      for (int q = 0; q < QPR; ++q) {
        const int kk = 4 * (lane + 64 * q);
        if (kk < KC * 16) {
          unsigned short hi[4], mid[4], lo[4];       // two-way split: hi, mid (= its "lo")
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            hi[e] = bf16_rne(pf[j][q][e]);
            const float r = pf[j][q][e] - bf16_f32(hi[e]);
            mid[e] = bf16_rne(r);
            lo[e] = bf16_rne(r - bf16_f32(mid[e]));
          }
          const int c = kk >> 4, hh = (kk >> 3) & 1;
          const int off = c * 1024 * PARTS + hh * 512 + slot_of(c, hh, s) * 16 + (kk & 7) * 2;
          typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
          u32x2 a = {hi[0] | ((unsigned)hi[1] << 16), hi[2] | ((unsigned)hi[3] << 16)};
          u32x2 b = {mid[0] | ((unsigned)mid[1] << 16), mid[2] | ((unsigned)mid[3] << 16)};
          *reinterpret_cast<u32x2*>(buf + off) = a;
          *reinterpret_cast<u32x2*>(buf + off + 1024) = b;
          if constexpr (P3) {
            u32x2 c3 = {lo[0] | ((unsigned)lo[1] << 16), lo[2] | ((unsigned)lo[3] << 16)};
            *reinterpret_cast<u32x2*>(buf + off + 2048) = c3;
          }
        }
      }
    }
  };

  long long tile = blockIdx.x;
  if (tile < tiles) { fetch(tile); stage(g_tile); }
  __syncthreads();
  int cur = 0;
  for (; tile < tiles; tile += gridDim.x, cur ^= 1) {
    const long long next = tile + gridDim.x;
    if (next < tiles) fetch(next);                // in flight during the MFMAs below
    const char* buf = g_tile + cur * TILE;
    if (computes) {
      // ---- 32 output features x 32 samples ----
      f32x16 acc = bias;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const int off = kc * 1024 * PARTS + h * 512 + slot_of(kc, h, m) * 16;
        const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(buf + off));
        const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(buf + off + 1024 * (PARTS - 1)));
        if constexpr (P3) {
          const bf16x8 xm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(buf + off + 1024));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[kc], xm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xh, acc, 0, 0, 0);
        } else {
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xh, acc, 0, 0, 0);
        }
      }
      store_tile<0>(A, acc, tile * 32 + m, ot, h, csum);
    }
    if (next < tiles) stage(g_tile + (cur ^ 1) * TILE);
    __syncthreads();
  }
  if (A.colsum != nullptr && computes) flush_colsum(A, csum, ot, h, m);
}

// ---- the same layer with the X tile brought in by LDS-DMA ---------------------------------------------------------
// The register-staged kernel above keeps one tile (16 .. 40 KiB) in flight per CU: a quarter of what HBM latency x bandwidth
// needs.  Here the fp32 tile goes HBM -> LDS without passing through registers (global_load_lds_dwordx4), NS tiles deep, and
// every wave converts the fp32 B-operand to bf16 hi / lo as it reads it (v_cvt_pk_bf16_f32: VALU work that hides under the
// other wave's MFMAs).  LDS image of a tile: 16-byte slots, slot L = row * STRIDE + quad, STRIDE = K/4 + 1 (odd -> the read
// "one quad pair of 32 different rows" touches every bank once).  The DMA writes 64 consecutive slots per instruction; which
// (row, quad) a lane fetches is free, so the padding slot, rows past M and quads past K are fetched from a zero line.
template <int KC, int WAVES> struct DmaShape {
  // Bank spreading of "one quad pair of 32 different rows": K a multiple of 64 floats -> no padding, the quad index is XORed with
  // row & 15 (the swizzle goes on the DMA's SOURCE address, the LDS image stays lane-linear; 256 x 256: 32 instructions per tile instead
  // of 33, which is what lets a fourth stage fit); otherwise an odd row stride (one padding slot per row).
  static constexpr bool SWZ = KC % 4 == 0;
  static constexpr int KQ = KC * 4, STRIDE = SWZ ? KQ : KQ + 1, SLOTS = 32 * STRIDE;
  static constexpr int NI = (SLOTS + 63) / 64, IPW = (NI + WAVES - 1) / WAVES;
  static constexpr int TILE_BYTES = IPW * WAVES * 1024;
  // 8 waves would each convert the whole tile to bf16 hi / lo on their own (VALU time ~ MFMA time): with COOP the workgroup
  // converts it ONCE into a bf16 image (the register-staged kernel's layout) between two barriers, and the waves read that.
  static constexpr bool COOP = WAVES == 8;
  static constexpr int IMG_BYTES = COOP ? KC * 2048 : 0;
  static constexpr int NS_FIT = (163840 - IMG_BYTES) / TILE_BYTES;
  static constexpr int NS = NS_FIT > 4 ? 4 : NS_FIT;
  static_assert(NS >= 2, "two stages of the X tile must fit the LDS");
  static_assert((NS - 2) * IPW < 64, "vmcnt is a 6-bit counter");
};

template <int N> __device__ __forceinline__ void wait_vm_lgkm0() {      // vmcnt(N) lgkmcnt(0), expcnt untouched
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4) | (0 << 8));
}

template <int KC, int WAVES, bool P3, int MODE>
__global__ __launch_bounds__(64 * WAVES) void k_dense_dma(const DenseArgs A) {
  constexpr bool FWD = MODE != 2, BWD = MODE != 1;      // see store_tile
  static_assert(!P3 || WAVES == 4, "the three-way split is built for the 4-wave shape only");
  constexpr int PARTS = P3 ? 3 : 2;
  typedef DmaShape<KC, WAVES> S;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int n_tiles = (A.n_out + 31) >> 5;
  const bool computes = wave < n_tiles;
  const int ot = computes ? wave : 0;
  bf16x8 wh[KC], wl[KC], wm[P3 ? KC : 1];          // P3: hi / mid / lo, else hi / lo
  const u32x4* wfrag = static_cast<const u32x4*>(A.wfrag);
#pragma unroll
  for (int kc = 0; kc < KC; ++kc) {
    const size_t base = ((size_t)(ot * KC + kc) * PARTS) * 64;
    wh[kc] = __builtin_bit_cast(bf16x8, wfrag[base + lane]);
    if constexpr (P3) {
      wm[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 64 + lane]);
      wl[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 128 + lane]);
    } else {
      wl[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 64 + lane]);
    }
  }
  f32x16 bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
    bias[r] = (FWD && A.bias != nullptr && n < A.n_out) ? A.bias[n] : 0.f;
  }
  const int K = A.k_total;
  const long long tiles = (A.M + 31) / 32;
  const int grid = gridDim.x;
  f32x16 csum = {};

  // per DMA instruction of this wave: source of this lane's slot in the workgroup's first tile, and its advance per tile
  const char* src[S::IPW];
  int step[S::IPW], rowu[S::IPW];
#pragma unroll
  for (int u = 0; u < S::IPW; ++u) {
    const int L = 64 * (wave + WAVES * u) + lane;
    const int row = L / S::STRIDE, j = L - row * S::STRIDE;
    int k = 4 * (S::SWZ ? (j ^ (row & 15)) : j);        // the quad this slot holds
    const float* p = nullptr;
    int ld = 0;
    bool done = !(row < 32 && j < S::KQ && k < K);
    bool found = false;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (!done && g < A.nseg) {
        if (k < A.seg[g].k) { p = A.seg[g].x + k; ld = A.seg[g].ld; done = true; found = true; }
        else k -= A.seg[g].k;
      }
    }
    src[u] = found ? reinterpret_cast<const char*>(p + ((size_t)blockIdx.x * 32 + row) * ld) : nullptr;
    step[u] = found ? grid * 32 * ld * 4 : 0;
    rowu[u] = found ? row : 0x3fffffff;
  }
  auto issue = [&](long long tile, int stage) {
#pragma unroll
    for (int u = 0; u < S::IPW; ++u) {
      const bool ok = tile * 32 + rowu[u] < A.M;
      const char* g = ok ? src[u] : reinterpret_cast<const char*>(A.zeros);
      const int off = __builtin_amdgcn_readfirstlane(stage * S::TILE_BYTES + (wave + WAVES * u) * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(g_tile + off), 16, 0, 0);
      src[u] += step[u];
    }
  };

  wait_vm_lgkm0<0>();                              // the weight / bias loads above: nothing but DMA on the VM counter from here
  long long tile = blockIdx.x;
#pragma unroll
  for (int s = 0; s < S::NS - 1; ++s) issue(tile + (long long)s * grid, s);
  int it = 0;
  for (; tile < tiles; tile += grid, ++it) {
    // tile `it` of this workgroup: my DMAs of it have landed (NS - 2 younger tiles may be in flight), my LDS reads of the
    // previous tile are done; past the barrier the same holds for every wave, so its slot can be refilled.
    wait_vm_lgkm0<(S::NS - 2) * S::IPW>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(tile + (long long)(S::NS - 1) * grid, (it + S::NS - 1) % S::NS);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S::COOP) {
      // fp32 stage -> bf16 image, 32 rows x 2 KC octets, lanes along the rows on both sides (odd row stride / rotated slots)
      const char* stg = g_tile + (it % S::NS) * S::TILE_BYTES;
      char* img = g_tile + S::NS * S::TILE_BYTES;
#pragma unroll
      for (int j = 0; j < (2 * KC + 2 * WAVES - 1) / (2 * WAVES); ++j) {
        const int pair = (threadIdx.x >> 5) + 2 * WAVES * j;
        if (pair < 2 * KC) {
          const int kc = pair >> 1, hh = pair & 1;
          const int q0 = 4 * kc + 2 * hh, sw = S::SWZ ? (m & 15) : 0;
          const char* p = stg + m * S::STRIDE * 16;
          const f32x4 f0 = *reinterpret_cast<const f32x4*>(p + (q0 ^ sw) * 16), f1 = *reinterpret_cast<const f32x4*>(p + ((q0 + 1) ^ sw) * 16);
          bf16x8 xh, xl;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 a = (__bf16)f0[e], b = (__bf16)f1[e];
            xh[e] = a; xh[4 + e] = b;
            xl[e] = (__bf16)(f0[e] - (float)a); xl[4 + e] = (__bf16)(f1[e] - (float)b);
          }
          char* q = img + kc * 2048 + hh * 512 + slot_of(kc, hh, m) * 16;
          *reinterpret_cast<u32x4*>(q) = __builtin_bit_cast(u32x4, xh);
          *reinterpret_cast<u32x4*>(q + 1024) = __builtin_bit_cast(u32x4, xl);
        }
      }
      wait_vm_lgkm0<63>();                           // lgkmcnt(0): my image writes are done (the VM counter is not waited on)
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (computes) {
        f32x16 acc = FWD ? bias : f32x16{};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
          const int off = kc * 2048 + h * 512 + slot_of(kc, h, m) * 16;
          const bf16x8 xh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + off));
          const bf16x8 xl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + off + 1024));
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xh, acc, 0, 0, 0);
        }
        store_tile<MODE>(A, acc, tile * 32 + m, ot, h, csum);
      }
    } else
    if (computes) {
      const char* buf = g_tile + (it % S::NS) * S::TILE_BYTES + m * S::STRIDE * 16;
      const int sw = S::SWZ ? (m & 15) : 0;
      f32x16 acc = FWD ? bias : f32x16{};
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(buf + ((4 * kc + 2 * h) ^ sw) * 16);
        const f32x4 f1 = *reinterpret_cast<const f32x4*>(buf + ((4 * kc + 2 * h + 1) ^ sw) * 16);
        bf16x8 xh, xl;
        if constexpr (P3) {
          // three-way split (24 mantissa bits on both operands), every product that is not below 2^-24: fp32-level layer
          bf16x8 xm;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = e < 4 ? f0[e] : f1[e - 4];
            const __bf16 a = (__bf16)f;
            const float r = f - (float)a;
            const __bf16 b = (__bf16)r;
            xh[e] = a; xm[e] = b; xl[e] = (__bf16)(r - (float)b);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[kc], xm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xm, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xh, acc, 0, 0, 0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const __bf16 a = (__bf16)f0[e], b = (__bf16)f1[e];
            xh[e] = a; xh[4 + e] = b;
            xl[e] = (__bf16)(f0[e] - (float)a); xl[4 + e] = (__bf16)(f1[e] - (float)b);
          }
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xl, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kc], xh, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xh, acc, 0, 0, 0);
        }
      }
      store_tile<MODE>(A, acc, tile * 32 + m, ot, h, csum);
    }
  }
  wait_vm_lgkm0<0>();                              // DMAs issued past the last tile (zero line) before the LDS is released
  if constexpr (BWD) if (A.colsum != nullptr && computes) flush_colsum(A, csum, ot, h, m);
}

// ---- weight gradient: dW[K x N] = X^T dY, contraction over the samples ------------------------------------------------
// The contraction index is the SAMPLE, so both MFMA operands want 8 consecutive samples of one feature per lane - the
// transpose of how X and dY lie in HBM (row major, one sample per row).  A 16-sample tile of both ([16][K] and [16][N] fp32)
// comes in by LDS-DMA (NS deep); between two barriers the workgroup turns it ONCE into bf16 hi / lo fragment images (thread =
// one feature x 8 samples: 8 conflict-free 4-byte reads down a column, one 16-byte write each for hi and lo); every wave then
// reads whole fragments and accumulates its share of the (K/32) x (N/32) output tiles in registers for the whole kernel.
// Each workgroup ends with a full K x N partial sum, written to part[blockIdx]; the caller adds the partials.
// FULL: every one of the 8 IPW DMA instructions of a tile carries data (256 x 256: 32 instructions), so the stage needs no spare KiB for
// idle ones - and then a fourth stage fits the 160 KiB next to the images (3 tiles = 96 KiB in flight per CU instead of 2).
template <int TPW, int IPW, bool FULL = false> struct WgShape {
  static constexpr int STAGE_BYTES = IPW * 8 * 1024 + (FULL ? 0 : 1024);   // 8 waves x IPW DMA instructions x 1 KiB (+ a spare KiB for idle instructions)
  // fragments of hi | lo in one image: KT + NT with KT NT <= 8 TPW and both <= 8
  static constexpr int IMG_FRAGS = TPW == 1 ? 9 : TPW == 2 ? 10 : TPW == 4 ? 12 : 16;
  static constexpr int IMG_BYTES = IMG_FRAGS * 2048;
  static constexpr int NS_FIT = (163840 - 2 * IMG_BYTES) / STAGE_BYTES;    // two images: tile t + 1 is converted while tile t is multiplied
  static constexpr int NS = NS_FIT > 4 ? 4 : NS_FIT;
  static_assert(NS >= 2 && (NS - 1) * IPW < 64, "ring depth");
  static constexpr int LDS = NS * STAGE_BYTES + 2 * IMG_BYTES;
};

// ROW: the TPW output tiles of a wave lie in one row of tiles (N/32 is a multiple of TPW), so its X^T fragment is read once per
// sample tile instead of once per output tile (a compile-time fact: as a run-time branch the two loop bodies cost 4x in spills).
// DYH: dY is f16 [M x ldy] (the scaled g arrays of the fused backward, WgradArgs::dy_half): half the bytes, its fragment image is the
// transposed tile itself (no conversion, no lo part), X goes into f16 hi + lo and a product is two f16 MFMAs (X hi * g, X lo * g) instead of three
// bf16 ones.  (With an f16 X as well the launch goes to k_wgrad_tr below; this path then serves the fp32 inputs of the first / skip layers.)
template <int TPW, int IPW, bool ROW, bool FULL, bool DYH>
__global__ __launch_bounds__(512) void k_wgrad(const WgradArgs A) {
  typedef WgShape<TPW, IPW, FULL> S;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int K = A.k, N = A.n, KQ = K >> 2, NQ = N >> 2;
  const int KT = (K + 31) >> 5, NT = (N + 31) >> 5, TT = KT * NT;
  const long long tiles = (A.M + 15) / 16;
  const int grid = gridDim.x;

  // DMA descriptors.  The stage holds [16][K] floats of X, then [16][N] floats of dY; each part is fetched either in 16-byte
  // pieces (rows 16-byte aligned) or, for the odd shapes (K = 33, N = 1 .. 6), float by float - one kind per instruction,
  // so a part is rounded up to whole instructions.  Lanes past the end of a part / of M read the zero line.
  const int XH = A.x_half;                                                   // X is f16: 8 elements per 16-byte piece, 2 bytes each
  const int NX = A.x_scalar ? (16 * K + 63) >> 6 : (XH ? (2 * K + 63) >> 6 : (16 * KQ + 63) >> 6);
  const int NY = A.dy_scalar ? (16 * N + 63) >> 6 : (DYH ? (2 * N + 63) >> 6 : (16 * NQ + 63) >> 6);
  const int XBYTES = NX * (A.x_scalar ? 256 : 1024);
  const char* src[IPW];
  int step[IPW], rowu[IPW], ldsoff[IPW];
  bool scalar[IPW];
#pragma unroll
  for (int u = 0; u < IPW; ++u) {
    const int i = wave + 8 * u;
    const char* p = nullptr;
    int st = 0, row = 0x3fffffff, off = S::STAGE_BYTES - 1024;            // spare KiB at the end of the stage: nobody reads it
    bool sc = false;
    if (i < NX + NY) {
      const bool isx = i < NX;
      const int ii = isx ? i : i - NX, W = isx ? K : N, ld = isx ? A.ldx : A.ldy;
      const char* base = reinterpret_cast<const char*>(isx ? A.x : A.dy);
      sc = isx ? A.x_scalar != 0 : A.dy_scalar != 0;
      const int esz = (isx ? XH != 0 : DYH) ? 2 : 4;                        // bytes per element
      const int e = (64 * ii + lane) * (sc ? 1 : 16 / esz);                 // first element of this lane's piece, in [16][W]
      off = (isx ? 0 : XBYTES) + ii * (sc ? 256 : 1024);
      if (e < 16 * W) {
        row = e / W;
        p = base + (((size_t)blockIdx.x * 16 + row) * ld + (e - row * W)) * esz;
        st = grid * 16 * ld * esz;
      }
    }
    src[u] = p; step[u] = st; rowu[u] = row; ldsoff[u] = __builtin_amdgcn_readfirstlane(off); scalar[u] = __builtin_amdgcn_readfirstlane(sc);
  }
  auto issue = [&](long long tile, int stage) {
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const bool ok = tile * 16 + rowu[u] < A.M;
      const char* g = ok ? src[u] : reinterpret_cast<const char*>(A.zeros);
      const int off = __builtin_amdgcn_readfirstlane(stage * S::STAGE_BYTES + ldsoff[u]);
      if (scalar[u])
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(g_tile + off), 4, 0, 0);
      else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(g_tile + off), 16, 0, 0);
      src[u] += step[u];
    }
  };
  // conversion items of this thread: (feature f, sample octet o) of X (idx < 2 K) or dY; 2 (K + N) <= 1024 items.
  // f16 X: the item's 8 samples come by two ds_read_b64_tr_b16 (a 16-lane group reads a [4 samples][16 features] block of the row-major
  // stage, lane i supplies the 8-byte piece (row i / 4, features 4 (i % 4) ..) and receives column i) instead of 8 two-byte reads.
  // (bf16 dY: its items start at a multiple of 64, so that whole waves - and whole 16-lane transpose groups - lie on one side)
  const int KI = DYH ? ((2 * K + 63) & ~63) : 2 * K;
  int c_src[2], c_dst[2], c_stride[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = threadIdx.x + 512 * j;
    c_src[j] = -1; c_dst[j] = 0; c_stride[j] = 0;
    if (idx < 2 * K) {
      const int o = idx / K, f = idx - o * K, esz = XH ? 2 : 4;
      c_src[j] = (8 * o * K + f) * esz; c_stride[j] = K * esz;
      if (XH) { const int i = lane & 15; c_src[j] = ((8 * o + (i >> 2)) * K + (f - i) + 4 * (i & 3)) * 2; }
      c_dst[j] = (f >> 5) * 2048 + ((o << 5) | (f & 31)) * 16;
    } else if (idx >= KI && idx < KI + 2 * N) {
      const int i2 = idx - KI, o = i2 / N, f = i2 - o * N;
      c_src[j] = XBYTES + (8 * o * N + f) * 4; c_stride[j] = N * 4;
      if (DYH) { const int i = lane & 15; c_src[j] = XBYTES + ((8 * o + (i >> 2)) * N + (f - i) + 4 * (i & 3)) * 2; c_stride[j] = N * 2; }
      c_dst[j] = (KT + (f >> 5)) * 2048 + ((o << 5) | (f & 31)) * 16;
    }
  }
  f32x16 acc[TPW];
#pragma unroll
  for (int j = 0; j < TPW; ++j) acc[j] = f32x16{};
  char* img0 = g_tile + S::NS * S::STAGE_BYTES;
  float cs[2] = {0.f, 0.f};            // column sums of this thread's dY items (feature f, sample octet o) over the workgroup's tiles

  // stage -> image: bf16 hi / lo fragment images of one 16-sample tile
  auto convert = [&](const char* stg, char* img) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (c_src[j] >= 0) {
        float f[8];
        if (DYH && threadIdx.x + 512 * j >= KI) {       // f16 item of dY: the transposed 8 samples ARE the fragment piece
          u32x2 q0, q1;
          const unsigned a0 = (unsigned)(size_t)(stg + c_src[j]), a1 = a0 + 4 * c_stride[j];
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1) : "v"(a0), "v"(a1) : "memory");
          const u32x4 pk = {q0[0], q0[1], q1[0], q1[1]};
          auto lo16 = [](unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); };
          auto hi16 = [](unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); };
          cs[j] += ((lo16(pk[0]) + hi16(pk[0])) + (lo16(pk[1]) + hi16(pk[1]))) + ((lo16(pk[2]) + hi16(pk[2])) + (lo16(pk[3]) + hi16(pk[3])));
          *reinterpret_cast<u32x4*>(img + c_dst[j]) = pk;
          continue;
        }
        if (XH && threadIdx.x + 512 * j < 2 * K) {       // f16 item of X (whole waves take one side: 2 K is a multiple of 64 on this path)
          // (inline asm: behind the builtin hipcc orders the read after EVERY outstanding LDS-DMA - s_waitcnt vmcnt(0) - and the ring is gone)
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          h4 q0, q1;
          const unsigned a0 = (unsigned)(size_t)(stg + c_src[j]), a1 = a0 + 4 * c_stride[j];
          asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(q0), "=&v"(q1) : "v"(a0), "v"(a1) : "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) { f[i] = (float)q0[i]; f[4 + i] = (float)q1[i]; }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const float*>(stg + c_src[j] + i * c_stride[j]);
        }
        if (threadIdx.x + 512 * j >= KI) cs[j] += ((f[0] + f[1]) + (f[2] + f[3])) + ((f[4] + f[5]) + (f[6] + f[7]));
        if constexpr (DYH) {                              // f16 dY: X as f16 hi + lo (an f16 X: lo = 0)
          f16x8 xh, xl;
          const float xsc = A.x_scale != 0.f ? A.x_scale : 1.f;      // (an fp32 X beyond f16's range - a raw TANGENT input - would become inf here: WgradArgs::x_scale)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float v = f[i] * xsc;
            const _Float16 a = (_Float16)v;
            xh[i] = a;
            xl[i] = (_Float16)(v - (float)a);
          }
          *reinterpret_cast<u32x4*>(img + c_dst[j]) = __builtin_bit_cast(u32x4, xh);
          *reinterpret_cast<u32x4*>(img + c_dst[j] + 1024) = __builtin_bit_cast(u32x4, xl);
        } else {
          bf16x8 xh, xl;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const __bf16 a = (__bf16)f[i];
            xh[i] = a;
            xl[i] = (__bf16)(f[i] - (float)a);
          }
          *reinterpret_cast<u32x4*>(img + c_dst[j]) = __builtin_bit_cast(u32x4, xh);
          *reinterpret_cast<u32x4*>(img + c_dst[j] + 1024) = __builtin_bit_cast(u32x4, xl);
        }
      }
    }
  };

  // Software pipeline, ONE barrier per tile: iteration t converts tile t + 1 into the other image while it multiplies tile t, so the
  // conversion (LDS reads, ~40 VALU, LDS writes per item) sits under the MFMAs of the same and of the SIMD's other wave instead of between
  // two barriers of its own.  After the barrier of iteration t: every wave's DMA of tile t + 1 has landed, tile t is converted (its stage
  // is free: tile t + NS goes there), and nobody reads the image of tile t - 1 any more.
  long long tile = blockIdx.x;
#pragma unroll
  for (int s = 0; s < S::NS; ++s) issue(tile + (long long)s * grid, s);
  wait_vm_lgkm0<(S::NS - 1) * IPW>();
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  convert(g_tile, img0);
  int it = 0, st = 0;                  // st = it % NS
  for (; tile < tiles; tile += grid, ++it) {
    wait_vm_lgkm0<(S::NS - 2) * IPW>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(tile + (long long)S::NS * grid, st);
    __builtin_amdgcn_sched_barrier(0);
    st = st + 1 == S::NS ? 0 : st + 1;
    const char* img = img0 + (it & 1) * S::IMG_BYTES;
    auto frag = [&](int f, int lo) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + f * 2048 + lo * 1024 + lane * 16)); };
    auto mm = [](const bf16x8& a, const bf16x8& b, const f32x16& c) {     // the images hold f16 bits when dY is f16
      if constexpr (DYH) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    };
    if constexpr (ROW) {
      // all TPW tiles of the wave share kt and exist together (NT is a multiple of TPW): one uniform test, then a branch-free body whose
      // fragment reads the scheduler can batch ahead of the MFMAs (a test per tile left every read directly in front of its MFMA)
      const int t0 = wave * TPW, kt = t0 / NT, nt0 = t0 - kt * NT;
      bf16x8 ah, al, bh[TPW], bl[TPW];
      if (t0 < TT) {
        ah = frag(kt, 0); al = frag(kt, 1);
#pragma unroll
        for (int j = 0; j < TPW; ++j) { bh[j] = frag(KT + nt0 + j, 0); if constexpr (!DYH) bl[j] = frag(KT + nt0 + j, 1); }
      }
      convert(g_tile + st * S::STAGE_BYTES, img0 + ((it + 1) & 1) * S::IMG_BYTES);
      if (t0 < TT) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
          if constexpr (!DYH) acc[j] = mm(ah, bl[j], acc[j]);
          acc[j] = mm(al, bh[j], acc[j]);
          acc[j] = mm(ah, bh[j], acc[j]);
        }
      }
    } else {
      convert(g_tile + st * S::STAGE_BYTES, img0 + ((it + 1) & 1) * S::IMG_BYTES);
#pragma unroll
      for (int j = 0; j < TPW; ++j) {
        const int t = wave * TPW + j;
        if (t < TT) {
          const int kt = t / NT, nt = t - kt * NT;
          const bf16x8 ah = frag(kt, 0), al = frag(kt, 1), bh = frag(KT + nt, 0);
          if constexpr (!DYH) { const bf16x8 bl = frag(KT + nt, 1); acc[j] = mm(ah, bl, acc[j]); }
          acc[j] = mm(al, bh, acc[j]);
          acc[j] = mm(ah, bh, acc[j]);
        }
      }
    }
  }
  wait_vm_lgkm0<0>();
  const float osc = ((DYH && A.out_scale != 0.f) ? A.out_scale : 1.f) * (A.out_scale_dev != nullptr ? *A.out_scale_dev : 1.f);       // f16 dY carries the chains' power-of-two scale
  if (A.colsum != nullptr) {           // bias gradient: each dY item adds its 8-sample sums (two items per feature and workgroup)
    float* cdst = A.colsum + (A.nrep > 1 ? (size_t)(blockIdx.x % A.nrep) * A.rep_stride : 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int idx = threadIdx.x + 512 * j;
      if (idx >= KI && idx < KI + 2 * N) unsafeAtomicAdd(cdst + (idx - KI) % N, cs[j] * osc);
    }
  }
  // this workgroup's partial: lane holds column n = 32 nt + (lane & 31), rows k = 32 kt + (r & 3) + 8 (r >> 2) + 4 h
  float* part = A.dw != nullptr ? A.dw + (A.nrep > 1 ? (size_t)(blockIdx.x % A.nrep) * A.rep_stride : 0) : A.part + (size_t)blockIdx.x * K * N;
  const bool atomic = A.dw != nullptr;
  const int m = lane & 31, h = lane >> 5;
#pragma unroll
  for (int j = 0; j < TPW; ++j) {
    const int t = wave * TPW + j;
    if (t < TT) {
      const int kt = t / NT, nt = t - kt * NT, n = 32 * nt + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < K && n < N) {
          if (atomic) unsafeAtomicAdd(part + (size_t)k * N + n, acc[j][r] * osc);
          else part[(size_t)k * N + n] = acc[j][r] * osc;
        }
      }
    }
  }
}

// ---- weight gradient of the fused training step: both operands 16-bit, read TRANSPOSED straight from the stage (round 4) ----
// X is f16 [M x K] (the forward's stored activations), dY is f16 [M x N] (the chains' g, scaled by a power of two - WgradArgs::out_scale
// undoes it), K and N multiples of 32.  Both MFMA operands of dW = X^T dY want "lane = feature, 8 consecutive samples": exactly what
// ds_read_b64_tr_b16 delivers from a row-major [sample][feature] block.  So the DMA'd stage IS the operand store: no conversion pass, no
// fragment images, one v_mfma_f32_32x32x16_f16 per product (k_wgrad: an LDS -> VALU -> LDS transpose + hi / lo split of every tile between two
// barriers and two bf16 MFMAs per product; its 16-sample tile cost ~2700 cycles of latency against 1024 of MFMA work).
// Stage = SPS samples x (K + N) f16 = 24 - 32 KiB, NS = 4 stages (2 - 3 in flight per CU), one barrier per stage.  The 16-byte granules of a row
// are swizzled (granule p of sample s sits at p ^ sw(s)) by the DMA's per-lane source addresses - the LDS side of an LDS-DMA is always
// 1 KiB contiguous per instruction - so that the 8 rows x 64 B of a transposed read cover every bank group exactly twice.
// Waves: 8, as a GK x GN grid over the KT x NT output tiles (x SS sample halves when there are fewer than 8 tiles), WK x WN tiles each.
template <int KT_, int NT_> struct WtShape {
  static constexpr int KT = KT_, NT = NT_, K = 32 * KT, N = 32 * NT;
  static constexpr int PRX = K / 8, PRN = N / 8;                     // 16-byte granules per row
  static constexpr int ROWB = (K + N) * 2;
  static constexpr int SPS = (32768 / ROWB) / 16 * 16;               // samples per stage
  static constexpr int XBYTES = SPS * K * 2, STAGE = SPS * ROWB;
  static constexpr int NS = 4, LDS = NS * STAGE;
  static constexpr int NI = STAGE / 1024, IPW = NI / 8;
  static_assert(STAGE % 8192 == 0 && XBYTES % 1024 == 0, "whole DMA instructions per wave and per operand");
  static constexpr int TT = KT * NT, SS = TT >= 8 ? 1 : 8 / TT;
  static constexpr int GK = KT >= 4 ? 4 : KT, GN = 8 / SS / GK, WK = KT / GK, WN = NT / GN;
  static_assert(GK * GN * SS == 8 && WK * GK == KT && WN * GN == NT && WK >= 1 && WN >= 1, "wave grid");
  static constexpr int SUB = SPS / 16;                                // 16-sample MFMA steps per stage
  static_assert(SUB % SS == 0, "sample halves");
  static constexpr int swz(int s, int pr) { return pr >= 16 ? (s & 3) << 2 : ((s >> 1) & 1) << 2; }
};

template <int KT, int NT>
__global__ __launch_bounds__(512) void k_wgrad_tr(const WgradArgs A) {
  typedef WtShape<KT, NT> S;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long tiles = (A.M + S::SPS - 1) / S::SPS;
  // several layers in one launch (WgradArgs::nl): this workgroup is number `wg` of `grid` on layer `layer`
  const int nl = A.nl > 1 ? A.nl : 1, layer = blockIdx.x % nl, wg = blockIdx.x / nl, grid = gridDim.x / nl;
  const char* const xbase = reinterpret_cast<const char*>(A.nl > 1 ? A.mx[layer] : A.x);
  const char* const ybase = reinterpret_cast<const char*>(A.nl > 1 ? A.mdy[layer] : A.dy);
  float* const dwbase = A.nl > 1 ? A.mdw[layer] : A.dw;
  float* const csbase = A.nl > 1 ? A.mcs[layer] : A.colsum;
  // DMA: instruction c = wave + 8 u fills granules 64 c .. 64 c + 63 of the stage ([SPS][K] of X, then [SPS][N] of dY, rows swizzled)
  const char* src[S::IPW];
  long long step[S::IPW];
  int rowu[S::IPW];
#pragma unroll
  for (int u = 0; u < S::IPW; ++u) {
    int slot = (wave + 8 * u) * 64 + lane;
    const bool isx = slot < S::SPS * S::PRX;
    if (!isx) slot -= S::SPS * S::PRX;
    const int pr = isx ? S::PRX : S::PRN, s = slot / pr, p = (slot - s * pr) ^ S::swz(s, pr);
    const int ld = isx ? A.ldx : A.ldy;
    const char* base = isx ? xbase : ybase;
    src[u] = base + (((size_t)wg * S::SPS + s) * ld + 8 * p) * 2;
    step[u] = (long long)grid * S::SPS * ld * 2;
    rowu[u] = s;
  }
  auto issue = [&](long long tile, int stage) {
#pragma unroll
    for (int u = 0; u < S::IPW; ++u) {
      const bool ok = tile * S::SPS + rowu[u] < A.M;
      const char* g = ok ? src[u] : reinterpret_cast<const char*>(A.zeros);
      const int off = __builtin_amdgcn_readfirstlane(stage * S::STAGE + (wave + 8 * u) * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(g_tile + off), 16, 0, 0);
      src[u] += step[u];
    }
  };
  // operand addresses: a 16-lane group reads a [4 samples][16 features] block, lane i supplies the 8-byte piece (row i / 4, features 4 (i % 4) ..)
  // and receives feature i of the block at those 4 samples; lanes 16-31 the tile's second 16 features, lanes 32-63 samples 8 .. 15
  const int i16 = lane & 15, fh = (lane >> 4) & 1, o8 = 8 * (lane >> 5);
  const int ws = wave / (S::GK * S::GN), wr = wave - ws * (S::GK * S::GN), wa = wr / S::GN, wb = wr - wa * S::GN;
  const int piece = (2 * fh + ((i16 & 3) >> 1)) * 16 + (i16 & 1) * 8;
  const int xk = S::PRX >= 16 ? (i16 >> 2) : (i16 >> 3) & 1, xn = S::PRN >= 16 ? (i16 >> 2) : (i16 >> 3) & 1;
  const unsigned lds0 = (unsigned)(size_t)g_tile;
  unsigned adA[S::WK], adB[S::WN];
#pragma unroll
  for (int j = 0; j < S::WK; ++j) adA[j] = lds0 + (16 * ws + o8 + (i16 >> 2)) * (S::PRX * 16) + 64 * ((wa * S::WK + j) ^ xk) + piece;
#pragma unroll
  for (int j = 0; j < S::WN; ++j) adB[j] = lds0 + S::XBYTES + (16 * ws + o8 + (i16 >> 2)) * (S::PRN * 16) + 64 * ((wb * S::WN + j) ^ xn) + piece;
  f32x16 acc[S::WK][S::WN];
#pragma unroll
  for (int j = 0; j < S::WK; ++j)
#pragma unroll
    for (int n = 0; n < S::WN; ++n) acc[j][n] = f32x16{};
  float cs[S::WN];
#pragma unroll
  for (int n = 0; n < S::WN; ++n) cs[n] = 0.f;
  const bool sums = csbase != nullptr && wa == 0;                   // the waves of tile row 0 see every dY fragment of their sample half once

  long long tile = wg;
#pragma unroll
  for (int s = 0; s < S::NS - 1; ++s) issue(tile + (long long)s * grid, s);
  int st = 0;
  for (; tile < tiles; tile += grid) {
    wait_vm_lgkm0<(S::NS - 2) * S::IPW>();                          // this wave's pieces of the stage have landed ...
    __builtin_amdgcn_s_barrier();                                    // ... everybody's have, and nobody reads the previous stage any more
    __builtin_amdgcn_sched_barrier(0);
    issue(tile + (long long)(S::NS - 1) * grid, st == 0 ? S::NS - 1 : st - 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned so = st * S::STAGE;
    // software pipeline over the stage's 16-sample steps: the reads of step s + 1 are issued before the MFMAs of step s; the wait in front of
    // a step's MFMAs lets exactly those newer reads stay outstanding (LDS returns in order), and the step's registers pass THROUGH the wait
    // statement, so nothing of theirs is touched before it
    constexpr int NSUB = S::SUB / S::SS, NRD = 2 * (S::WK + S::WN);
    u32x2 qa[2][S::WK][2], qb[2][S::WN][2];
    auto reads = [&](int sub, int b) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < S::WK; ++j)
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                     : "=&v"(qa[b][j][0]), "=&v"(qa[b][j][1]) : "v"(adA[j] + (so + sub * S::SS * 16 * S::PRX * 16)), "n"(4 * S::PRX * 16) : "memory");
#pragma unroll
      for (int n = 0; n < S::WN; ++n)
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3"
                     : "=&v"(qb[b][n][0]), "=&v"(qb[b][n][1]) : "v"(adB[n] + (so + sub * S::SS * 16 * S::PRN * 16)), "n"(4 * S::PRN * 16) : "memory");
    };
#define NERFDS_WT_WAIT(CNT)                                                                                                                            \
    do {                                                                                                                                               \
      if constexpr (S::WK == 2 && S::WN == 4)                                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(%12)" : "+v"(qa[b][0][0]), "+v"(qa[b][0][1]), "+v"(qa[b][1][0]), "+v"(qa[b][1][1]), "+v"(qb[b][0][0]), "+v"(qb[b][0][1]), \
                     "+v"(qb[b][1][0]), "+v"(qb[b][1][1]), "+v"(qb[b][2][0]), "+v"(qb[b][2][1]), "+v"(qb[b][3][0]), "+v"(qb[b][3][1]) : "n"(CNT));  \
      else if constexpr (S::WK == 2 && S::WN == 2)                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(qa[b][0][0]), "+v"(qa[b][0][1]), "+v"(qa[b][1][0]), "+v"(qa[b][1][1]), "+v"(qb[b][0][0]), "+v"(qb[b][0][1]), \
                     "+v"(qb[b][1][0]), "+v"(qb[b][1][1]) : "n"(CNT));                                                                               \
      else if constexpr (S::WK == 1 && S::WN == 2)                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(qa[b][0][0]), "+v"(qa[b][0][1]), "+v"(qb[b][0][0]), "+v"(qb[b][0][1]), "+v"(qb[b][1][0]), "+v"(qb[b][1][1]) : "n"(CNT)); \
      else {                                                                                                                                           \
        static_assert((S::WK == 2 || S::WK == 1) && (S::WN == 4 || S::WN == 2 || S::WN == 1), "wave block shapes with a written-out wait");            \
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(qa[b][0][0]), "+v"(qa[b][0][1]), "+v"(qb[b][0][0]), "+v"(qb[b][0][1]) : "n"(CNT));              \
      }                                                                                                                                                \
    } while (0)
    reads(0, 0);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      const int b = sub & 1;
      if (sub + 1 < NSUB) { reads(sub + 1, b ^ 1); NERFDS_WT_WAIT(NRD); }
      else NERFDS_WT_WAIT(0);
      f16x8 fa[S::WK], fb[S::WN];
#pragma unroll
      for (int j = 0; j < S::WK; ++j) fa[j] = __builtin_bit_cast(f16x8, u32x4{qa[b][j][0][0], qa[b][j][0][1], qa[b][j][1][0], qa[b][j][1][1]});
#pragma unroll
      for (int n = 0; n < S::WN; ++n) fb[n] = __builtin_bit_cast(f16x8, u32x4{qb[b][n][0][0], qb[b][n][0][1], qb[b][n][1][0], qb[b][n][1][1]});
#pragma unroll
      for (int j = 0; j < S::WK; ++j)
#pragma unroll
        for (int n = 0; n < S::WN; ++n) acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[j], fb[n], acc[j][n], 0, 0, 0);
      if (sums) {
        const f16x2 one = {(_Float16)1.f, (_Float16)1.f};
#pragma unroll
        for (int n = 0; n < S::WN; ++n)
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[n] = __builtin_amdgcn_fdot2(f16x2{fb[n][2 * q], fb[n][2 * q + 1]}, one, cs[n], false);
      }
    }
#undef NERFDS_WT_WAIT
    __builtin_amdgcn_sched_barrier(0);
    st = st + 1 == S::NS ? 0 : st + 1;
  }
  wait_vm_lgkm0<0>();
  const float osc = (A.out_scale != 0.f ? A.out_scale : 1.f) * (A.out_scale_dev != nullptr ? *A.out_scale_dev : 1.f);
  const size_t rep = A.nrep > 1 ? (size_t)(wg % A.nrep) * A.rep_stride : 0;
  const int m = lane & 31, h = lane >> 5;
  if (sums) {
#pragma unroll
    for (int n = 0; n < S::WN; ++n) unsafeAtomicAdd(csbase + rep + 32 * (wb * S::WN + n) + m, cs[n] * osc);
  }
  float* part = dwbase + rep;                                        // (the launcher requires dw: every workgroup adds with float atomics)
  const bool atomic = true;
#pragma unroll
  for (int j = 0; j < S::WK; ++j)
#pragma unroll
    for (int n = 0; n < S::WN; ++n) {
      const int kt = wa * S::WK + j, nn = 32 * (wb * S::WN + n) + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (atomic) unsafeAtomicAdd(part + (size_t)k * S::N + nn, acc[j][n][r] * osc);
        else part[(size_t)k * S::N + nn] = acc[j][n][r] * osc;
      }
    }
}

// ---- weight gradient of a HEAD: X f16 [M x K] (the last hidden layer, K = 64 / 128 / 256), dY fp32 [M x N], N <= 6 -------------------------
// 768 FMAs per row at most: VALU work, the kernel is the read of X.  A thread owns 8 features (one 16-byte load per row) x N outputs in registers,
// K / 8 threads share a row, the rows of a block iteration are reduced by shuffles and through LDS, one float atomic per output and block.
// (k_wgrad ran these shapes with scalar-DMA dY tiles and MFMAs on 32-padded N: 60 - 110 us per launch for 134 - 268 MB, eight launches per step.)
template <int K, int N>
__global__ __launch_bounds__(256) void k_wgrad_head(const WgradArgs A) {
  constexpr int TPR = K / 8, RPI = 256 / TPR;                            // threads per row; rows per block iteration
  const int f = threadIdx.x % TPR, rr = threadIdx.x / TPR, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const _Float16* X = reinterpret_cast<const _Float16*>(A.x);
  float acc[8][N];
  double cs[N];                    // the bias gradient is a sum of CANCELLING rows (|sum| << sum of |terms|): summed in double per block
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int n = 0; n < N; ++n) acc[i][n] = 0.f;
#pragma unroll
  for (int n = 0; n < N; ++n) cs[n] = 0.0;
#pragma unroll 4
  for (long long row = (long long)blockIdx.x * RPI + rr; row < A.M; row += (long long)gridDim.x * RPI) {
    const f16x8 x = *reinterpret_cast<const f16x8*>(X + (size_t)row * A.ldx + 8 * f);
    float d[N];
#pragma unroll
    for (int n = 0; n < N; ++n) d[n] = A.dy[(size_t)row * A.ldy + n];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xi = (float)x[i];
#pragma unroll
      for (int n = 0; n < N; ++n) acc[i][n] += xi * d[n];
    }
    if (f == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) cs[n] += (double)d[n];
    }
  }
  // rows of one wave: lanes f, f + TPR, ... hold the same features
#pragma unroll
  for (int sft = TPR; sft < 64; sft <<= 1) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < N; ++n) acc[i][n] += __shfl_xor(acc[i][n], sft, 64);
#pragma unroll
    for (int n = 0; n < N; ++n) cs[n] += __shfl_xor(cs[n], sft, 64);
  }
  __shared__ double scs[4][8];
  float* sm = reinterpret_cast<float*>(g_tile);                         // [4 waves][K x N + N]
  if (lane < TPR) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int n = 0; n < N; ++n) sm[wave * (K * N + N) + (8 * f + i) * N + n] = acc[i][n];
    if (f == 0) {
#pragma unroll
      for (int n = 0; n < N; ++n) scs[wave][n] = cs[n];
    }
  }
  __syncthreads();
  const size_t rep = A.nrep > 1 ? (size_t)(blockIdx.x % A.nrep) * A.rep_stride : 0;
  // (X may carry a power-of-two scale - the stored f16 tangents of the second-order terms: out_scale / out_scale_dev undo it)
  const float osc = (A.out_scale != 0.f ? A.out_scale : 1.f) * (A.out_scale_dev != nullptr ? *A.out_scale_dev : 1.f);
  for (int e = threadIdx.x; e < K * N; e += 256)
    unsafeAtomicAdd(A.dw + rep + e, osc * ((sm[e] + sm[(K * N + N) + e]) + (sm[2 * (K * N + N) + e] + sm[3 * (K * N + N) + e])));
  if (A.colsum != nullptr && threadIdx.x < N)
    unsafeAtomicAdd(A.colsum + rep + threadIdx.x, (float)((scs[0][threadIdx.x] + scs[1][threadIdx.x]) + (scs[2][threadIdx.x] + scs[3][threadIdx.x])));
}

// ---- backward of a narrow hidden layer in one pass ---------------------------------------------------------------------
// k_wgrad and the data-gradient kernel of a layer both read dZ, and the latter reads X again as its ReLU mask: 5 array passes per
// layer.  For the 64 / 128-wide networks (mask MLP, warp field, hyper sheet: 34 layers per step) everything fits one 4-wave
// workgroup - W^T as A-fragments (64 registers per wave), the K x N weight-gradient tiles (64 registers per wave) - so one kernel
// reads X and dZ once (LDS-DMA, 32-sample tiles) and produces dX (masked, with its column sums) and dW: 3 passes.
// LDS: NS stages of [32][N] + [32][K] fp32 (odd row strides), then the transposed bf16 hi / lo images of both for the dW product.
template <int KCN, int IPW> struct BfShape {
  static constexpr int STAGE_BYTES = IPW * 4 * 1024;
  static constexpr int IMG_BYTES = 8 * 2 * 2048;                 // (K/32 + N/32 <= 8 feature tiles) x 2 sample chunks x (hi | lo)
  static constexpr int NS = (3 * STAGE_BYTES + IMG_BYTES <= 163840) ? 3 : 2;
  static_assert((NS - 2) * IPW < 64, "vmcnt is a 6-bit counter");
  static constexpr int LDS = NS * STAGE_BYTES + IMG_BYTES;
};

template <int KCN, int IPW>
__global__ __launch_bounds__(256) void k_bwd_fused(const BwdFusedArgs A) {
  typedef BfShape<KCN, IPW> S;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = lane & 31, h = lane >> 5;
  const int K = A.k, N = A.n;                                    // N == 16 KCN
  const int SN = N / 4 + 1, SK = K / 4 + 1;                      // row strides in 16-byte slots (odd)
  const int KT = K >> 5, NT = N >> 5, TT = KT * NT;              // dW tiles; TPW = 4 per wave at most
  const int DY_SLOTS = 32 * SN;
  const long long tiles = (A.M + 31) / 32;
  const int grid = gridDim.x;
  const bool dx_wave = wave < KT;                                // this wave owns dX output tile `wave`
  const int ot = dx_wave ? wave : 0;
  // W^T slice of this wave's dX tile
  bf16x8 wh[KCN], wl[KCN];
  const u32x4* wfrag = static_cast<const u32x4*>(A.wfrag);
#pragma unroll
  for (int kc = 0; kc < KCN; ++kc) {
    const size_t base = ((size_t)(ot * KCN + kc) * 2) * 64;
    wh[kc] = __builtin_bit_cast(bf16x8, wfrag[base + lane]);
    wl[kc] = __builtin_bit_cast(bf16x8, wfrag[base + 64 + lane]);
  }
  // DMA descriptors: slots [0, 32 SN) = dZ rows, then 32 SK slots of X rows; padding slots / rows >= M read the zero line
  const char* src[IPW];
  int step[IPW], rowu[IPW];
#pragma unroll
  for (int u = 0; u < IPW; ++u) {
    const int L = 64 * (wave + 4 * u) + lane;
    const char* p = nullptr;
    int st = 0, row = 0x3fffffff;
    if (L < DY_SLOTS) {
      const int r = L / SN, j = L - r * SN;
      if (j < N / 4) { row = r; p = reinterpret_cast<const char*>(A.dz + ((size_t)blockIdx.x * 32 + r) * A.lddz + 4 * j); st = grid * 32 * A.lddz * 4; }
    } else if (L < DY_SLOTS + 32 * SK) {
      const int L2 = L - DY_SLOTS, r = L2 / SK, j = L2 - r * SK;
      if (j < K / 4) { row = r; p = reinterpret_cast<const char*>(A.x + ((size_t)blockIdx.x * 32 + r) * A.ldx + 4 * j); st = grid * 32 * A.ldx * 4; }
    }
    src[u] = p; step[u] = st; rowu[u] = row;
  }
  auto issue = [&](long long tile, int stage) {
#pragma unroll
    for (int u = 0; u < IPW; ++u) {
      const bool ok = tile * 32 + rowu[u] < A.M;
      const char* g = ok ? src[u] : reinterpret_cast<const char*>(A.zeros);
      const int off = __builtin_amdgcn_readfirstlane(stage * S::STAGE_BYTES + (wave + 4 * u) * 1024);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(g_tile + off), 16, 0, 0);
      src[u] += step[u];
    }
  };
  // conversion items (transposed images for dW): feature f of dZ (f < N) or X, sample octet o of the 32-sample tile; 4 (K + N) items
  constexpr int CI = 4;                                           // items per thread: 4 (128 + 128) / 256
  int c_src[CI], c_dst[CI], c_stride[CI];
#pragma unroll
  for (int j = 0; j < CI; ++j) {
    const int idx = threadIdx.x + 256 * j;
    c_src[j] = -1; c_dst[j] = 0; c_stride[j] = 0;
    if (idx < 4 * N) {
      const int o = idx / N, f = idx - o * N;
      c_src[j] = (8 * o * SN + (f >> 2)) * 16 + (f & 3) * 4; c_stride[j] = SN * 16;
      c_dst[j] = ((KT + (f >> 5)) * 2 + (o >> 1)) * 2048 + (((o & 1) << 5) | (f & 31)) * 16;
    } else if (idx < 4 * (N + K)) {
      const int i2 = idx - 4 * N, o = i2 / K, f = i2 - o * K;
      c_src[j] = (DY_SLOTS + 8 * o * SK + (f >> 2)) * 16 + (f & 3) * 4; c_stride[j] = SK * 16;
      c_dst[j] = ((f >> 5) * 2 + (o >> 1)) * 2048 + (((o & 1) << 5) | (f & 31)) * 16;
    }
  }
  f32x16 wacc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wacc[j] = f32x16{};
  f32x16 csum = {};
  char* img = g_tile + S::NS * S::STAGE_BYTES;

  wait_vm_lgkm0<0>();
  long long tile = blockIdx.x;
#pragma unroll
  for (int s0 = 0; s0 < S::NS - 1; ++s0) issue(tile + (long long)s0 * grid, s0);
  int it = 0;
  for (; tile < tiles; tile += grid, ++it) {
    wait_vm_lgkm0<(S::NS - 2) * IPW>();              // this tile's DMAs (NS - 2 younger tiles may still be in flight)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    issue(tile + (long long)(S::NS - 1) * grid, (it + S::NS - 1) % S::NS);
    __builtin_amdgcn_sched_barrier(0);
    const char* stg = g_tile + (it % S::NS) * S::STAGE_BYTES;
    // ---- transposed bf16 images of dZ and X (for dW) ----
#pragma unroll
    for (int j = 0; j < CI; ++j) {
      if (c_src[j] >= 0) {
        float f[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = *reinterpret_cast<const float*>(stg + c_src[j] + i * c_stride[j]);
        bf16x8 xh, xl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const __bf16 a = (__bf16)f[i];
          xh[i] = a;
          xl[i] = (__bf16)(f[i] - (float)a);
        }
        *reinterpret_cast<u32x4*>(img + c_dst[j]) = __builtin_bit_cast(u32x4, xh);
        *reinterpret_cast<u32x4*>(img + c_dst[j] + 1024) = __builtin_bit_cast(u32x4, xl);
      }
    }
    wait_vm_lgkm0<63>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- dW tiles of this wave: t = 4 wave + j -> (kt, nt), two 16-sample chunks ----
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int t = wave * 4 + j;
      if (t < TT) {
        const int kt = t / NT, nt = t - kt * NT;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const char* pa = img + (kt * 2 + c) * 2048 + lane * 16;
          const char* pb = img + ((KT + nt) * 2 + c) * 2048 + lane * 16;
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pa));
          const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pa + 1024));
          const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pb));
          const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pb + 1024));
          wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, wacc[j], 0, 0, 0);
          wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, wacc[j], 0, 0, 0);
          wacc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, wacc[j], 0, 0, 0);
        }
      }
    }
    // ---- dX tile of this wave: 32 features x 32 samples, masked with X > 0 (read from the stage), column sums, store ----
    if (dx_wave) {
      const char* bdz = stg + (m * SN + 2 * h) * 16;
      f32x16 acc = {};
#pragma unroll
      for (int kc = 0; kc < KCN; ++kc) {
        const f32x4 f0 = *reinterpret_cast<const f32x4*>(bdz + kc * 64);
        const f32x4 f1 = *reinterpret_cast<const f32x4*>(bdz + kc * 64 + 16);
        bf16x8 xh, xl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 a = (__bf16)f0[e], b = (__bf16)f1[e];
          xh[e] = a; xh[4 + e] = b;
          xl[e] = (__bf16)(f0[e] - (float)a); xl[4 + e] = (__bf16)(f1[e] - (float)b);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kc], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kc], xh, acc, 0, 0, 0);
      }
      const long long row = tile * 32 + m;
      if (row < A.M) {
        const char* bx = stg + (DY_SLOTS + m * SK) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n0 = 32 * ot + 8 * q + 4 * h;
          const f32x4 y = *reinterpret_cast<const f32x4*>(bx + n0 * 4);
          f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
#pragma unroll
          for (int j = 0; j < 4; ++j) if (!(y[j] > 0.f)) v[j] = 0.f;
          csum[4 * q] += v[0]; csum[4 * q + 1] += v[1]; csum[4 * q + 2] += v[2]; csum[4 * q + 3] += v[3];
          *reinterpret_cast<f32x4*>(A.dx + (size_t)row * A.lddx + n0) = v;
        }
      }
    }
  }
  wait_vm_lgkm0<0>();
  if (A.colsum != nullptr && dx_wave) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = csum[r];
#pragma unroll
      for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
      if (m == 0) atomicAdd(A.colsum + (A.nrep > 1 ? (size_t)(blockIdx.x % A.nrep) * A.rep_stride : 0) + 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h, v);
    }
  }
  float* dwr = A.dw + (A.nrep > 1 ? (size_t)(blockIdx.x % A.nrep) * A.rep_stride : 0);
  if (A.dw != nullptr)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = wave * 4 + j;
    if (t < TT) {
      const int kt = t / NT, nt = t - kt * NT, n = 32 * nt + m;
#pragma unroll
      for (int r = 0; r < 16; ++r) unsafeAtomicAdd(dwr + (size_t)(32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h) * N + n, wacc[j][r]);
    }
  }
}

void pack_frags(hipStream_t st, const float* W, int ldw, int row0, int in_dim, int out_dim, int transpose, void* out, int parts) {
  const int KC = (in_dim + 15) / 16, tiles = (out_dim + 31) / 32;
  const long long n = (long long)tiles * KC * 64;
  hipLaunchKernelGGL(k_pack_frags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, ldw, row0, in_dim, out_dim, transpose,
                     parts, static_cast<u32x4*>(out));
}
void pack_frags_all(hipStream_t st, const float* theta, const PackEntry* entries_dev, int n, int max_frag_lanes, void* arena) {
  if (n < 1) return;
  hipLaunchKernelGGL(k_pack_frags_all, dim3((unsigned)((max_frag_lanes + 255) / 256), (unsigned)n), dim3(256), 0, st, theta, entries_dev, static_cast<char*>(arena));
}
size_t frag_bytes(int in_dim, int out_dim, int parts) { return (size_t)((out_dim + 31) / 32) * ((in_dim + 15) / 16) * 1024 * parts; }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int KC, int WAVES, bool P3, int MODE> static void launch_dma(hipStream_t st, const DenseArgs& A, int num_cus, long long tiles) {
  typedef DmaShape<KC, WAVES> S;
  const int lds = S::NS * S::TILE_BYTES + S::IMG_BYTES;
  auto kern = k_dense_dma<KC, WAVES, P3, MODE>;
  static int per_cu = 0;                                        // resident workgroups per CU (registers / LDS decide: the same on every device of a node)
  nerfds::allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  if (per_cu == 0) {
    int n = 0;
    per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * WAVES, lds) == hipSuccess && n > 0) ? n : 1;
  }
  const long long want = (long long)num_cus * per_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < want ? tiles : want)), dim3(64 * WAVES), lds, st, A);
}

template <int KC, int WAVES, bool P3 = false> static void launch(hipStream_t st, DenseArgs A, int num_cus) {
  const long long tiles = (A.M + 31) / 32;
  // 16-byte fetches: rows 16-byte aligned and every segment a whole number of quads - or, for the LAST segment, a row stride that
  // covers its last quad (the pad floats exist and are finite; their weights are the zero padding of the fragment pack)
  A.vec_in = 1;
  for (int g = 0; g < A.nseg; ++g) {
    const bool tail_ok = g == A.nseg - 1 && A.seg[g].ld >= ((A.seg[g].k + 3) & ~3);
    if ((A.seg[g].k % 4 && !tail_ok) || A.seg[g].ld % 4 || !aligned16(A.seg[g].x)) A.vec_in = 0;
  }
  A.vec_out = (A.n_out % 4 == 0 && A.ldy % 4 == 0 && aligned16(A.y) && (A.mask_y == nullptr || (A.ld_mask % 4 == 0 && aligned16(A.mask_y)))) ? 1 : 0;
  const bool dma = A.vec_in && A.zeros != nullptr && getenv("NERFDS_WS_NODMA") == nullptr;
  if (dma) {
    // 8 waves: one instantiation per epilogue kind (registers); a layer has a forward epilogue or a backward one
    const bool bwd = A.mask_y != nullptr || A.accumulate || A.colsum != nullptr;
    if constexpr (WAVES == 8) {
      if (bwd) launch_dma<KC, WAVES, P3, 2>(st, A, num_cus, tiles); else launch_dma<KC, WAVES, P3, 1>(st, A, num_cus, tiles);
    } else {
      launch_dma<KC, WAVES, P3, 0>(st, A, num_cus, tiles);
    }
    return;
  }
  const int lds = 2 * KC * 1024 * (P3 ? 3 : 2);
  auto kern = k_dense_ws<KC, WAVES, P3>;
  static int per_cu = 0;
  nerfds::allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  if (per_cu == 0) {
    int n = 0;
    per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 64 * WAVES, lds) == hipSuccess && n > 0) ? n : 1;
  }
  const long long want = (long long)num_cus * per_cu;
  const int grid = (int)(tiles < want ? tiles : want);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, st, A);
}

bool dense_ws_supported(const DenseArgs& A) {
  if (A.n_out > 256 || A.n_out < 1 || A.M <= 0 || A.nseg < 1 || A.nseg > 4) return false;
  const int KC = (A.k_total + 15) / 16;
  if (KC > 21 && A.n_out > 128) return false;      // the slice of a long layer needs a 512-register wave: at most 4 waves
  if (A.mask_div != 1 && A.mask_div != 3) return false;
  if (A.n_out > 128 && (A.mask_y != nullptr || A.accumulate || A.colsum != nullptr) && (A.bias != nullptr || A.relu))
    return false;                                               // 8-wave kernels: forward epilogue or backward epilogue, not both
  if (A.precise) {                                              // three-way split: 4-wave LDS-DMA shape, K <= 192
    if (A.n_out > 128 || KC > 12) return false;
  }
  for (int g = 0, k0 = 0; g < A.nseg; ++g) {        // a quad of 4 consecutive k never straddles two segments
    if (k0 % 4) return false;
    k0 += A.seg[g].k;
  }
  return (KC >= 1 && KC <= 21) || KC == 35;
}

bool dense_ws(hipStream_t st, const DenseArgs& A, int num_cus) {
  if (!dense_ws_supported(A)) return false;
  const int KC = (A.k_total + 15) / 16;
  const bool wide = A.n_out > 128;
  if (A.precise) {
#define NERFDS_KC3(n) case n: launch<n, 4, true>(st, A, num_cus); return true;
    switch (KC) {
      NERFDS_KC3(1) NERFDS_KC3(2) NERFDS_KC3(3) NERFDS_KC3(4) NERFDS_KC3(5) NERFDS_KC3(6) NERFDS_KC3(7) NERFDS_KC3(8) NERFDS_KC3(9) NERFDS_KC3(10)
      NERFDS_KC3(11) NERFDS_KC3(12)
      default: return false;
    }
#undef NERFDS_KC3
  }
#define NERFDS_KC(n) case n: if (wide) launch<n, 8>(st, A, num_cus); else launch<n, 4>(st, A, num_cus); return true;
  switch (KC) {
    NERFDS_KC(1) NERFDS_KC(2) NERFDS_KC(3) NERFDS_KC(4) NERFDS_KC(5) NERFDS_KC(6) NERFDS_KC(7) NERFDS_KC(8) NERFDS_KC(9) NERFDS_KC(10)
    NERFDS_KC(11) NERFDS_KC(12) NERFDS_KC(13) NERFDS_KC(14) NERFDS_KC(15) NERFDS_KC(16) NERFDS_KC(17) NERFDS_KC(18) NERFDS_KC(19)
    NERFDS_KC(20) NERFDS_KC(21)
    case 35: launch<35, 4>(st, A, num_cus); return true;
    default: return false;
  }
#undef NERFDS_KC
}

static void wgrad_kinds(const WgradArgs& A, int& xs, int& ys, int& ninstr) {
  xs = (!A.x_half && (A.k % 4 || A.ldx % 4 || !aligned16(A.x))) ? 1 : 0;
  ys = (!A.dy_half && (A.n % 4 || A.ldy % 4 || !aligned16(A.dy))) ? 1 : 0;
  ninstr = (xs ? (16 * A.k + 63) / 64 : ((A.x_half ? 2 : 4) * A.k + 63) / 64) + (ys ? (16 * A.n + 63) / 64 : ((A.dy_half ? 2 : 4) * A.n + 63) / 64);
}
// development switches (A/B), read once: NERFDS_WGRAD_TR_OFF=1 sends every shape to k_wgrad, NERFDS_WGRAD_HEAD_OFF=1 the heads too
static bool wgrad_tr_off() { static const bool v = getenv("NERFDS_WGRAD_TR_OFF") != nullptr; return v; }
static bool wgrad_head_off() { static const bool v = getenv("NERFDS_WGRAD_HEAD_OFF") != nullptr; return v; }
// the shapes k_wgrad_tr is built for; several layers per launch exist on that kernel only
bool wgrad_multi_supported(const WgradArgs& A) {
  if (!(A.x_half && A.dy_half && A.dw != nullptr) || A.nl > 8 || wgrad_tr_off()) return false;
  const int kt = A.k / 32, nt = A.n / 32;
  if (!((kt == 8 && nt == 8) || (kt == 8 && nt == 4) || (kt == 4 && nt == 4) || (kt == 2 && nt == 2))) return false;
  for (int i = 0; i < A.nl; ++i)
    if (!aligned16(A.mx[i]) || !aligned16(A.mdy[i]) || A.mdw[i] == nullptr) return false;
  return A.nl < 2 || (A.mx[0] == A.x && A.mdy[0] == A.dy && A.mdw[0] == A.dw && A.mcs[0] == A.colsum);
}
bool wgrad_supported(const WgradArgs& A) {
  if (A.k < 1 || A.n < 1 || A.k > 256 || A.n > 256 || A.M <= 0 || A.zeros == nullptr) return false;
  if (A.x_half && (A.k % 32 || A.ldx % 8 || !aligned16(A.x))) return false;       // f16 rows: whole 16-byte pieces, whole waves of items
  if (A.dy_half && (A.n % 32 || A.ldy % 8 || !aligned16(A.dy) || ((2 * A.k + 63) & ~63) + 2 * A.n > 1024)) return false;   // bf16 rows likewise
  int xs, ys, ninstr;
  wgrad_kinds(A, xs, ys, ninstr);
  return ninstr <= 32;                                              // at most 4 DMA instructions per wave and tile
}
int wgrad_grid(const WgradArgs& A, int num_cus) {
  const long long tiles = (A.M + 15) / 16;
  // one workgroup per CU for the big shapes (128 KiB of LDS); the small ones fit several, but more workgroups = more partials
  const long long want = num_cus;
  return (int)(tiles < want ? tiles : want);
}
template <int TPW, int IPW, bool ROW, bool FULL, bool DYH> static void launch_wgrad_t(hipStream_t st, const WgradArgs& A, int grid) {
  auto kern = k_wgrad<TPW, IPW, ROW, FULL, DYH>;
  typedef WgShape<TPW, IPW, FULL> S;
  nerfds::allow_dynamic_lds(reinterpret_cast<const void*>(kern), S::LDS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), S::LDS, st, A);
}
template <int TPW, int IPW, bool ROW, bool FULL = false> static void launch_wgrad(hipStream_t st, const WgradArgs& A, int grid) {
  if (A.dy_half) launch_wgrad_t<TPW, IPW, ROW, FULL, true>(st, A, grid);
  else launch_wgrad_t<TPW, IPW, ROW, FULL, false>(st, A, grid);
}
template <int KT, int NT> static void launch_wgrad_tr(hipStream_t st, const WgradArgs& A, int grid) {
  typedef WtShape<KT, NT> S;
  auto kern = k_wgrad_tr<KT, NT>;
  nerfds::allow_dynamic_lds(reinterpret_cast<const void*>(kern), S::LDS);
  const long long tiles = (A.M + S::SPS - 1) / S::SPS;
  const int nl = A.nl > 1 ? A.nl : 1;
  long long per = grid / nl;                                          // workgroups per layer
  if (per < 1) per = 1;
  if (per > tiles) per = tiles;
  hipLaunchKernelGGL(kern, dim3((unsigned)(per * nl)), dim3(512), S::LDS, st, A);
}
bool wgrad(hipStream_t st, const WgradArgs& A0, int grid) {
  if (!wgrad_supported(A0) || grid < 1) return false;
  WgradArgs A = A0;
  if (A.nl > 1 && !wgrad_multi_supported(A)) return false;
  if (A.x_half && A.dy_half && A.dw != nullptr && !wgrad_tr_off()) {        // both operands 16-bit: transposed reads straight from the stage
    const int kt = A.k / 32, nt = A.n / 32;
#define NERFDS_WT(KT, NT) if (kt == KT && nt == NT) { launch_wgrad_tr<KT, NT>(st, A, grid); return true; }
    NERFDS_WT(8, 8) NERFDS_WT(8, 4) NERFDS_WT(4, 4) NERFDS_WT(2, 2)
#undef NERFDS_WT
  }
  if (A.x_half && !A.dy_half && A.dw != nullptr && A.n <= 6 && (A.k == 64 || A.k == 128 || A.k == 256) && !wgrad_head_off()) {
    // a head on an f16 hidden layer: VALU kernel, the read of X is the cost
    const long long want = 4LL * grid, rows = (A.M + 255 / (A.k / 8)) / (256 / (A.k / 8));
    const dim3 g((unsigned)(rows < want ? (rows < 1 ? 1 : rows) : want));
    const size_t lds = 4 * (size_t)(A.k * A.n + A.n) * sizeof(float);
#define NERFDS_WH(KK, NN) if (A.k == KK && A.n == NN) { hipLaunchKernelGGL((k_wgrad_head<KK, NN>), g, dim3(256), lds, st, A); return true; }
#define NERFDS_WHK(KK) NERFDS_WH(KK, 1) NERFDS_WH(KK, 2) NERFDS_WH(KK, 3) NERFDS_WH(KK, 4) NERFDS_WH(KK, 6)
    NERFDS_WHK(64) NERFDS_WHK(128) NERFDS_WHK(256)
#undef NERFDS_WHK
#undef NERFDS_WH
  }
  int ninstr;
  wgrad_kinds(A, A.x_scalar, A.dy_scalar, ninstr);
  const int TT = ((A.k + 31) / 32) * ((A.n + 31) / 32);
  const int tpw = (TT + 7) / 8;                                     // 1 .. 8 output tiles per wave
  const int ipw = (ninstr + 7) / 8;                                 // DMA instructions per wave and 16-sample tile: 1 .. 4
  const int NT = (A.n + 31) / 32;
  if (tpw == 8 && ipw == 4 && ninstr == 32) {                       // 256 x 256, fp32 X: no idle DMA instruction
    if (NT % 8 == 0) launch_wgrad<8, 4, true, true>(st, A, grid); else launch_wgrad<8, 4, false, true>(st, A, grid);
    return true;
  }
  if (tpw == 8 && ipw == 2 && ninstr == 16) {                       // 256 x 256, f16 X and bf16 dY: no idle DMA instruction
    if (NT % 8 == 0) launch_wgrad<8, 2, true, true>(st, A, grid); else launch_wgrad<8, 2, false, true>(st, A, grid);
    return true;
  }
  if (tpw == 8 && ipw == 3 && ninstr == 24) {                       // 256 x 256, f16 X: no idle DMA instruction, four stages next to two images
    if (NT % 8 == 0) launch_wgrad<8, 3, true, true>(st, A, grid); else launch_wgrad<8, 3, false, true>(st, A, grid);
    return true;
  }
#define NERFDS_WG(T, I) if (tpw <= T && ipw <= I) { if (T > 1 && NT % T == 0) launch_wgrad<T, I, true>(st, A, grid); else launch_wgrad<T, I, false>(st, A, grid); return true; }
  NERFDS_WG(1, 1) NERFDS_WG(1, 2) NERFDS_WG(1, 4)
  NERFDS_WG(2, 2) NERFDS_WG(2, 4)
  NERFDS_WG(4, 2) NERFDS_WG(4, 3) NERFDS_WG(4, 4)
  NERFDS_WG(8, 2) NERFDS_WG(8, 3) NERFDS_WG(8, 4)
#undef NERFDS_WG
  return false;
}

bool bwd_fused_supported(const BwdFusedArgs& A) {
  if (!((A.k == 64 || A.k == 128) && (A.n == 64 || A.n == 128)) || A.M <= 0 || A.zeros == nullptr) return false;
  if (A.ldx % 4 || A.lddz % 4 || A.lddx % 4 || !aligned16(A.x) || !aligned16(A.dz) || !aligned16(A.dx)) return false;
  const int slots = 32 * (A.n / 4 + 1 + A.k / 4 + 1);
  const int ipw = (slots + 255) / 256;
  return (A.n == 128 && ipw == 9) || (A.n == 64 && ipw <= 7);
}
template <int KCN, int IPW> static void launch_bwd_fused(hipStream_t st, const BwdFusedArgs& A, int num_cus) {
  typedef BfShape<KCN, IPW> S;
  auto kern = k_bwd_fused<KCN, IPW>;
  static int per_cu = 0;
  nerfds::allow_dynamic_lds(reinterpret_cast<const void*>(kern), S::LDS);
  if (per_cu == 0) {
    int n = 0;
    per_cu = (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, S::LDS) == hipSuccess && n > 0) ? n : 1;
  }
  const long long tiles = (A.M + 31) / 32, want = (long long)num_cus * per_cu;
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles < want ? tiles : want)), dim3(256), S::LDS, st, A);
}
bool bwd_fused(hipStream_t st, const BwdFusedArgs& A, int num_cus) {
  if (!bwd_fused_supported(A)) return false;
  const int ipw = (32 * (A.n / 4 + 1 + A.k / 4 + 1) + 255) / 256;
  if (A.n == 128) launch_bwd_fused<8, 9>(st, A, num_cus);
  else if (ipw <= 5) launch_bwd_fused<4, 5>(st, A, num_cus);
  else launch_bwd_fused<4, 7>(st, A, num_cus);
  return true;
}

}  // namespace nerfds_train
