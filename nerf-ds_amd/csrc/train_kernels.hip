// Element-wise / per-ray kernels of the training step (BASELINE config 4): everything of the
// nerf_ds graph that is not a dense layer, forward AND backward, in fp32.  The dense layers themselves are plain
// [samples x width] GEMMs over HBM-resident activations (hand-written MFMA kernels: train_gemm.hip, see nerfds_train.cpp): a training step has to keep
// every layer's activations for dW anyway, so this path is HBM-resident by design, unlike the fused render kernel.
// One thread per sample unless stated; sample m = ray * S + s.  Reference lines are cited per kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "train_kernels.h"
#include "philox.h"

namespace nerfds_train {

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// posenc feature g of a C-vector, layout [band][sin, cos][channel] (model_utils.py:398-417), windowed (420-436)
template <int C> __device__ __forceinline__ float posenc_val(int g, const float* x, const float* win) {
  const int band = g / (2 * C), sc = (g % (2 * C)) / C, ch = g % C;
  return win[band] * sinf(x[ch] * (float)(1 << band) + (sc ? 1.57079637f : 0.0f));
}
// d feature / d x[ch]
template <int C> __device__ __forceinline__ float posenc_dval(int g, const float* x, const float* win) {
  const int band = g / (2 * C), sc = (g % (2 * C)) / C, ch = g % C;
  return win[band] * (float)(1 << band) * cosf(x[ch] * (float)(1 << band) + (sc ? 1.57079637f : 0.0f));
}

// ---- row tiles through LDS ---------------------------------------------------------------------------------------------
// The per-sample kernels below own one ROW of a [samples][W] fp32 array per thread (W = 48 .. 60 floats): read or written straight from
// the thread, every wave instruction touches 64 different cache lines.  Instead the block's rows go through an LDS tile with an odd row
// stride (thread = row: conflict-free) and cross HBM as one contiguous run, consecutive threads = consecutive floats.
constexpr int TILE_ROWS = 128;                                   // block size of the tiled kernels
__device__ __forceinline__ int tile_stride(int W) { return W | 1; }
static inline size_t tile_bytes(int W) { return (size_t)TILE_ROWS * (W | 1) * sizeof(float); }
template <bool STORE> __device__ __forceinline__ void tile_io(float* __restrict__ g, int nrows, int W, float* __restrict__ tile, int stride) {
  const int n = nrows * W;
  int r = threadIdx.x / W, c = threadIdx.x - r * W;
  const int dr = blockDim.x / W, dc = blockDim.x - dr * W;
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    if (STORE) g[e] = tile[r * stride + c]; else tile[r * stride + c] = g[e];
    r += dr; c += dc;
    if (c >= W) { c -= W; ++r; }
  }
}
// the calling block's rows [m0, m0 + blockDim.x) of g[M][W] <- tile (after every thread wrote its row) / -> tile (before it reads it)
__device__ __forceinline__ void tile_store(float* g, long long m0, long long M, int W, float* tile) {
  __syncthreads();
  const long long left = M - m0;
  tile_io<true>(g + m0 * W, (int)(left < (long long)blockDim.x ? left : (long long)blockDim.x), W, tile, tile_stride(W));
  __syncthreads();
}
__device__ __forceinline__ void tile_load(const float* g, long long m0, long long M, int W, float* tile) {
  const long long left = M - m0;
  tile_io<false>(const_cast<float*>(g) + m0 * W, (int)(left < (long long)blockDim.x ? left : (long long)blockDim.x), W, tile, tile_stride(W));
  __syncthreads();
}
extern __shared__ float g_rows[];

// ---- sampling (model_utils.py:55-92) ---------------------------------------------------------------------------
// t_rand == nullptr with stratified sampling: the on-chip Philox stream of philox.h (the reference always draws, model_utils.py:84)
__global__ void k_coarse_z(int R, int Nc, float near_, float far_, int stratified, int lindisp, const float* __restrict__ t_rand, uint64_t seed, long long first_ray, float* __restrict__ z) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)R * Nc) return;
  const int s = (int)(i % Nc);
  auto zlin = [&](int q) {
    const float t = (Nc > 1) ? (float)q / (float)(Nc - 1) : 0.f;
    if (lindisp) return 1.0f / (1.0f / near_ * (1.0f - t) + 1.0f / far_ * t);
    return near_ * (1.0f - t) + far_ * t;
  };
  float v = zlin(s);
  if (stratified) {
    const float lo = s > 0 ? 0.5f * (v + zlin(s - 1)) : v, hi = s + 1 < Nc ? 0.5f * (zlin(s + 1) + v) : v;
    v = lo + (hi - lo) * (t_rand ? t_rand[i] : nerfds::sample_uniform(seed, first_ray + i / Nc, 0, s));
  }
  z[i] = v;
}

// ---- inverse-CDF resample + sorted union (model_utils.py:193-269), one thread per ray; no gradient (line 241) -----
__global__ void k_resample(int R, int Nc, int Nf, const float* __restrict__ zc, const float* __restrict__ wc, int stratified,
                           const float* __restrict__ u_rand, uint64_t seed, long long first_ray, float* __restrict__ zf, float* __restrict__ scratch) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* z = zc + (size_t)r * Nc;
  const float* w = wc + (size_t)r * Nc;
  const int nb = Nc - 1, nw = Nc - 2;
  float* cdf = scratch + (size_t)r * (2 * Nc + Nf);       // [nb]
  float* bins = cdf + Nc;                                  // [nb]
  float* zs = bins + Nc;                                   // [Nf]
  float tot = 0.f;
  for (int i = 0; i < nw; ++i) tot += w[i + 1] + 1e-5f;
  cdf[0] = 0.f;
  float c = 0.f;
  for (int i = 0; i < nw; ++i) { c += (w[i + 1] + 1e-5f) / tot; cdf[i + 1] = c; }
  for (int i = 0; i < nb; ++i) bins[i] = 0.5f * (z[i + 1] + z[i]);
  for (int k = 0; k < Nf; ++k) {
    const float u = stratified ? (u_rand ? u_rand[(size_t)r * Nf + k] : nerfds::sample_uniform(seed, first_ray + r, 1, k))
                               : (Nf > 1 ? (float)k / (float)(Nf - 1) : 0.f);
    int lo = 0, hi = nb;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
    int k0 = lo - 1; if (k0 < 0) k0 = 0;
    const int k1 = (k0 + 1 < nb) ? k0 + 1 : nb - 1;
    const float b0 = fminf(bins[k0], bins[nb - 2]), b1 = fmaxf(bins[k1], bins[1]);
    const float c0 = fminf(cdf[k0], cdf[nb - 2]), c1 = fmaxf(cdf[k1], cdf[1]);
    float denom = c1 - c0; if (denom < 1e-5f) denom = 1.0f;
    zs[k] = b0 + (u - c0) / denom * (b1 - b0);
  }
  // sort(concat(z_coarse, z_fine)): rank sort (stable)
  float* out = zf + (size_t)r * (Nc + Nf);
  const int n = Nc + Nf;
  for (int i = 0; i < n; ++i) {
    const float v = i < Nc ? z[i] : zs[i - Nc];
    int rank = 0;
    for (int q = 0; q < n; ++q) { const float o = q < Nc ? z[q] : zs[q - Nc]; rank += (o < v || (o == v && q < i)) ? 1 : 0; }
    out[rank] = v;
  }
}

// The same, one wavefront per ray (Nc, Nf <= 256): the running sums stay serial (lane 0: the cdf must round exactly as the
// sequential cumsum does), the Nf inverse-cdf look-ups and the rank sort of the Nc + Nf depths are spread over the lanes.
// z_new / src (optional, the merged step): the Nf NEW depths of the ray in draw order, and for every slot of the sorted union the POSITION ROW its
// sample lives in - coarse sample i of ray r: r * Nc + i; new sample k: R * Nc + r * Nf + k (the shared networks run once per position row)
__global__ __launch_bounds__(64) void k_resample_wave(int R, int Nc, int Nf, const float* __restrict__ zc, const float* __restrict__ wc, int stratified,
                                                      const float* __restrict__ u_rand, uint64_t seed, long long first_ray, float* __restrict__ zf,
                                                      float* __restrict__ z_new, int* __restrict__ src) {
  __shared__ float cdf[256], bins[256], zall[512];
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* z = zc + (size_t)r * Nc;
  const float* w = wc + (size_t)r * Nc;
  const int nb = Nc - 1, nw = Nc - 2, n = Nc + Nf;
  for (int i = lane; i < nb; i += 64) bins[i] = 0.5f * (z[i + 1] + z[i]);
  for (int i = lane; i < Nc; i += 64) zall[i] = z[i];
  if (lane == 0) {
    float tot = 0.f;
    for (int i = 0; i < nw; ++i) tot += w[i + 1] + 1e-5f;
    cdf[0] = 0.f;
    float c = 0.f;
    for (int i = 0; i < nw; ++i) { c += (w[i + 1] + 1e-5f) / tot; cdf[i + 1] = c; }
  }
  __syncthreads();
  for (int k = lane; k < Nf; k += 64) {
    const float u = stratified ? (u_rand ? u_rand[(size_t)r * Nf + k] : nerfds::sample_uniform(seed, first_ray + r, 1, k))
                               : (Nf > 1 ? (float)k / (float)(Nf - 1) : 0.f);
    int lo = 0, hi = nb;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
    int k0 = lo - 1; if (k0 < 0) k0 = 0;
    const int k1 = (k0 + 1 < nb) ? k0 + 1 : nb - 1;
    const float b0 = fminf(bins[k0], bins[nb - 2]), b1 = fmaxf(bins[k1], bins[1]);
    const float c0 = fminf(cdf[k0], cdf[nb - 2]), c1 = fmaxf(cdf[k1], cdf[1]);
    float denom = c1 - c0; if (denom < 1e-5f) denom = 1.0f;
    zall[Nc + k] = b0 + (u - c0) / denom * (b1 - b0);
    if (z_new) z_new[(size_t)r * Nf + k] = zall[Nc + k];
  }
  __syncthreads();
  float* out = zf + (size_t)r * n;
  for (int i = lane; i < n; i += 64) {
    const float v = zall[i];
    int rank = 0;
    for (int q = 0; q < n; ++q) { const float o = zall[q]; rank += (o < v || (o == v && q < i)) ? 1 : 0; }
    out[rank] = v;
    if (src) src[(size_t)r * n + rank] = i < Nc ? r * Nc + i : R * Nc + r * Nf + (i - Nc);
  }
}

// merged step: the shared networks' per-sample results in the fine level's (sorted-union) row order, and the way back for their gradients
__global__ void k_gather_rows(long long M, const int* __restrict__ src, const float* __restrict__ xw, const float* __restrict__ wamb, const float* __restrict__ wv,
                              float* __restrict__ xw_f, float* __restrict__ wamb_f, float* __restrict__ wv_f) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= M) return;
  const long long s = src[i];
  for (int c = 0; c < 3; ++c) xw_f[3 * i + c] = xw[3 * s + c];
  for (int c = 0; c < 2; ++c) wamb_f[2 * i + c] = wamb[2 * s + c];
  for (int c = 0; c < 6; ++c) wv_f[6 * i + c] = wv[6 * s + c];
}
// every position row appears exactly once in the union: rows below `add_below` (the coarse positions, which already hold the coarse level's
// gradient) are added to, the others written
__global__ void k_scatter_rows(long long M, const int* __restrict__ src, long long add_below, const float* __restrict__ dxw_f, const float* __restrict__ dwamb_f,
                               float* __restrict__ dxw, float* __restrict__ dwamb) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= M) return;
  const long long s = src[i];
  const bool add = s < add_below;
  for (int c = 0; c < 3; ++c) dxw[3 * s + c] = (add ? dxw[3 * s + c] : 0.f) + dxw_f[3 * i + c];
  for (int c = 0; c < 2; ++c) dwamb[2 * s + c] = (add ? dwamb[2 * s + c] : 0.f) + dwamb_f[2 * i + c];
}

// the same for any per-position array of `rp` rows x C floats per sample (rp = 3: the tangent rows 3 m + j): one thread per element
__global__ void k_gather_cols(long long M, const int* __restrict__ src, int rp, int C, const float* __restrict__ in, float* __restrict__ out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int W = rp * C;
  if (i >= M * W) return;
  const long long m = i / W;
  out[i] = in[(long long)src[m] * W + (i - m * W)];
}
__global__ void k_scatter_cols(long long M, const int* __restrict__ src, long long add_below, int rp, int C, const float* __restrict__ in_f, float* __restrict__ out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int W = rp * C;
  if (i >= M * W) return;
  const long long m = i / W, s = src[m], o = s * W + (i - m * W);
  out[o] = (s < add_below ? out[o] : 0.f) + in_f[i];
}

// ---- inputs of the three shared nets (models.py:931-975, 729-732; modules.py:367-434; warping.py:200-237) --------
__global__ void k_encode_inputs(Dims D, int R, int S, const float* __restrict__ o, const float* __restrict__ d, const float* __restrict__ z,
                                const uint32_t* __restrict__ warp_id, int n_embeds, const float* __restrict__ warp_tbl,
                                const float* __restrict__ mask_tbl, Windows W, float* __restrict__ x, float* __restrict__ mask_in,
                                float* __restrict__ warp_in, float* __restrict__ hyper_in) {
  const long long M = (long long)R * S, m0 = blockIdx.x * (long long)blockDim.x, m = m0 + threadIdx.x;
  const bool live = m < M;
  float p[3] = {0.f, 0.f, 0.f};
  uint32_t id = 0u;
  if (live) {
    const int r = (int)(m / S);
    for (int c = 0; c < 3; ++c) { p[c] = o[3 * r + c] + z[m] * d[3 * r + c]; x[3 * m + c] = p[c]; }
    id = warp_id ? warp_id[r] : 0u;
    if (id >= (uint32_t)n_embeds) id = n_embeds - 1;      // jnp gathers clamp
  }
  {
    float* mi = g_rows + threadIdx.x * tile_stride(D.mask_in);
    for (int g = 0; g < 6 * D.mask_bands; ++g) mi[g] = posenc_val<3>(g, p, W.mask);
    for (int g = 0; g < 8; ++g) mi[6 * D.mask_bands + g] = mask_tbl[id * 8 + g];
    tile_store(mask_in, m0, M, D.mask_in, g_rows);
  }
  {
    float* wi = g_rows + threadIdx.x * tile_stride(D.warp_ld);
    for (int g = 0; g < 6 * D.warp_bands; ++g) wi[g] = posenc_val<3>(g, p, W.warp);
    for (int g = 0; g < 8; ++g) wi[6 * D.warp_bands + g] = warp_tbl[id * 8 + g];
    for (int g = D.warp_in - 1; g < D.warp_ld; ++g) wi[g] = 0.f;        // the mask column D.warp_in - 1 (k_mask_post fills it in) and the pad columns
    tile_store(warp_in, m0, M, D.warp_ld, g_rows);
  }
  {
    float* hi = g_rows + threadIdx.x * tile_stride(D.hyper_ld);
    for (int g = 0; g < 6 * D.hyp_bands; ++g) hi[g] = posenc_val<3>(g, p, W.hyp);
    for (int g = 0; g < 8; ++g) hi[6 * D.hyp_bands + g] = warp_tbl[id * 8 + g];
    for (int g = D.hyper_in - 1; g < D.hyper_ld; ++g) hi[g] = 0.f;
    tile_store(hyper_in, m0, M, D.hyper_ld, g_rows);
  }
}

__global__ void k_bias_act(float* __restrict__ y, const float* __restrict__ b, long long M, int N, int ld, int relu) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= M * N) return;
  const long long row = i / N;
  const int col = (int)(i % N);
  float v = y[row * ld + col] + b[col];
  y[row * ld + col] = relu ? fmaxf(v, 0.f) : v;
}

// mask = relu(logit) * ratio + gt * (1 - ratio) (models.py:975) appended to the warp / hyper inputs (models.py:729-732)
__global__ void k_mask_post(Dims D, int R, int S, const float* __restrict__ logit, const float* __restrict__ gt, float ratio,
                            float* __restrict__ warp_in, float* __restrict__ hyper_in) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= (long long)R * S) return;
  const float pm = fmaxf(logit[m], 0.f);
  const float g = gt ? gt[m / S] : 0.f;
  const float v = pm * ratio + g * (1.0f - ratio);
  warp_in[m * D.warp_ld + D.warp_in - 1] = v;
  hyper_in[m * D.hyper_ld + D.hyper_in - 1] = v;
}

// ---- SE(3): x' = exp_se3(w / |w|, v / |w|, |w|) x (warping.py:219-237, rigid_body.py:59-101), generic in the scalar -----
struct Dual {      // value + gradient w.r.t. (w0 w1 w2 v0 v1 v2)
  float v, g[6];
};
__device__ __forceinline__ Dual dconst(float c) { Dual r; r.v = c; for (int i = 0; i < 6; ++i) r.g[i] = 0.f; return r; }
__device__ __forceinline__ Dual operator+(const Dual& a, const Dual& b) { Dual r; r.v = a.v + b.v; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] + b.g[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a, const Dual& b) { Dual r; r.v = a.v - b.v; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] - b.g[i]; return r; }
__device__ __forceinline__ Dual operator-(const Dual& a) { Dual r; r.v = -a.v; for (int i = 0; i < 6; ++i) r.g[i] = -a.g[i]; return r; }
__device__ __forceinline__ Dual operator*(const Dual& a, const Dual& b) { Dual r; r.v = a.v * b.v; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] * b.v + a.v * b.g[i]; return r; }
__device__ __forceinline__ Dual operator/(const Dual& a, const Dual& b) {
  Dual r; const float inv = 1.0f / b.v; r.v = a.v * inv;
  for (int i = 0; i < 6; ++i) r.g[i] = (a.g[i] - r.v * b.g[i]) * inv;
  return r;
}
__device__ __forceinline__ Dual operator*(const Dual& a, float b) { Dual r; r.v = a.v * b; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] * b; return r; }
__device__ __forceinline__ Dual dsqrt(const Dual& a) { Dual r; r.v = sqrtf(a.v); const float k = 0.5f / r.v; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] * k; return r; }
__device__ __forceinline__ float dsqrt(float a) { return sqrtf(a); }
__device__ __forceinline__ float tconst(float c, const float*) { return c; }
__device__ __forceinline__ Dual tconst(float c, const Dual*) { return dconst(c); }
// sin / cos of a (nested) dual whose innermost value is a0, given s = sinf(a0), c = cosf(a0): the same numbers dsin / dcos produce (every level of a
// nested dual evaluates sinf / cosf on the SAME a0), with the two libm calls made once by the caller instead of once per level and component
__device__ __forceinline__ float base_of(float a) { return a; }
__device__ __forceinline__ float base_of(const Dual& a) { return a.v; }
__device__ __forceinline__ float dsin_sc(float, float s, float) { return s; }
__device__ __forceinline__ float dcos_sc(float, float, float c) { return c; }
__device__ __forceinline__ Dual dsin_sc(const Dual& a, float s, float c) { Dual r; r.v = s; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] * c; return r; }
__device__ __forceinline__ Dual dcos_sc(const Dual& a, float s, float c) { Dual r; r.v = c; const float k = -s; for (int i = 0; i < 6; ++i) r.g[i] = a.g[i] * k; return r; }

// R (row major) and p of exp_se3 for raw head outputs (w, v)
template <class T> __device__ void se3_Rp(const T (&w_raw)[3], const T (&v_raw)[3], T (&Rm)[9], T (&p)[3]) {
  const T* tag = nullptr;
  const T theta = dsqrt(w_raw[0] * w_raw[0] + w_raw[1] * w_raw[1] + w_raw[2] * w_raw[2]);      // no epsilon, as the reference
  T w[3], v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { w[i] = w_raw[i] / theta; v[i] = v_raw[i] / theta; }
  const float th0 = base_of(theta), s0 = sinf(th0), c0 = cosf(th0);
  const T st = dsin_sc(theta, s0, c0), ct = dcos_sc(theta, s0, c0);
  const T one = tconst(1.0f, tag), zero = tconst(0.0f, tag);
  const T omc = one - ct, tms = theta - st;
  const T W[9] = {zero, -w[2], w[1], w[2], zero, -w[0], -w[1], w[0], zero};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    T g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T w2 = W[3 * r] * W[c] + W[3 * r + 1] * W[3 + c] + W[3 * r + 2] * W[6 + c];
      const T id = (r == c) ? one : zero;
      Rm[3 * r + c] = id + st * W[3 * r + c] + omc * w2;                                     // rigid_body.py:59-74
      g[c] = ((r == c) ? theta : zero) + omc * W[3 * r + c] + tms * w2;                       // rigid_body.py:94-95
    }
    p[r] = g[0] * v[0] + g[1] * v[1] + g[2] * v[2];
  }
}

__global__ void k_se3_fwd(long long M, const float* __restrict__ wv, const float* __restrict__ x, float* __restrict__ xw) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float w[3] = {wv[6 * m], wv[6 * m + 1], wv[6 * m + 2]}, v[3] = {wv[6 * m + 3], wv[6 * m + 4], wv[6 * m + 5]};
  float Rm[9], p[3];
  se3_Rp<float>(w, v, Rm, p);
  for (int r = 0; r < 3; ++r) xw[3 * m + r] = Rm[3 * r] * x[3 * m] + Rm[3 * r + 1] * x[3 * m + 1] + Rm[3 * r + 2] * x[3 * m + 2] + p[r];
}

// d loss / d (w, v) from d loss / d x'   (x is a constant: the observation-space sample point)
__global__ void k_se3_bwd(long long M, const float* __restrict__ wv, const float* __restrict__ x, const float* __restrict__ dxw,
                          const float* __restrict__ dwv_extra, float* __restrict__ dwv) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  Dual w[3], v[3];
  for (int i = 0; i < 3; ++i) { w[i] = dconst(wv[6 * m + i]); w[i].g[i] = 1.f; v[i] = dconst(wv[6 * m + 3 + i]); v[i].g[3 + i] = 1.f; }
  Dual Rm[9], p[3];
  se3_Rp<Dual>(w, v, Rm, p);
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < 3; ++r) {
    const Dual xr = Rm[3 * r] * x[3 * m] + Rm[3 * r + 1] * x[3 * m + 1] + Rm[3 * r + 2] * x[3 * m + 2] + p[r];
    for (int i = 0; i < 6; ++i) acc[i] += dxw[3 * m + r] * xr.g[i];
  }
  for (int i = 0; i < 6; ++i) dwv[6 * m + i] = acc[i] + (dwv_extra ? dwv_extra[6 * m + i] : 0.f);
}

// ---- NerfMLP trunk input: posenc(x') | posenc(ambient coords) (models.py:493-523) and its backward -------------------
__global__ void k_trunk_in(Dims D, long long M, const float* __restrict__ xw, const float* __restrict__ wamb, Windows W, float* __restrict__ tin) {
  const long long m0 = blockIdx.x * (long long)blockDim.x, m = m0 + threadIdx.x, mc = m < M ? m : M - 1;
  const float p[3] = {xw[3 * mc], xw[3 * mc + 1], xw[3 * mc + 2]}, a[2] = {wamb[2 * mc], wamb[2 * mc + 1]};
  float* t = g_rows + threadIdx.x * tile_stride(D.trunk_in);
  for (int g = 0; g < 6 * D.sp_bands; ++g) t[g] = posenc_val<3>(g, p, W.sp);
  for (int g = 0; g < 4 * D.hp_bands; ++g) t[6 * D.sp_bands + g] = posenc_val<2>(g, a, W.hp);
  tile_store(tin, m0, M, D.trunk_in, g_rows);
}
__global__ void k_trunk_in_bwd(Dims D, long long M, const float* __restrict__ dtin, const float* __restrict__ xw, const float* __restrict__ wamb,
                               Windows W, const float* __restrict__ dxw_extra, const float* __restrict__ dwamb_extra, float* __restrict__ dxw,
                               float* __restrict__ dwamb) {
  const long long m0 = blockIdx.x * (long long)blockDim.x, m = m0 + threadIdx.x;
  tile_load(dtin, m0, M, D.trunk_in, g_rows);
  if (m >= M) return;
  const float p[3] = {xw[3 * m], xw[3 * m + 1], xw[3 * m + 2]}, a[2] = {wamb[2 * m], wamb[2 * m + 1]};
  const float* t = g_rows + threadIdx.x * tile_stride(D.trunk_in);
  for (int c = 0; c < 3; ++c) {
    float acc = 0.f;
    for (int bs = 0; bs < 2 * D.sp_bands; ++bs) acc += t[3 * bs + c] * posenc_dval<3>(3 * bs + c, p, W.sp);
    dxw[3 * m + c] = acc + (dxw_extra ? dxw_extra[3 * m + c] : 0.f);
  }
  for (int c = 0; c < 2; ++c) {
    float acc = 0.f;
    for (int bs = 0; bs < 2 * D.hp_bands; ++bs) acc += t[6 * D.sp_bands + 2 * bs + c] * posenc_dval<2>(2 * bs + c, a, W.hp);
    dwamb[2 * m + c] = acc + (dwamb_extra ? dwamb_extra[2 * m + c] : 0.f);
  }
}

// ---- after the alpha head: sigma (models.py:577) and the rgb condition [posenc(viewdir) | posenc(R^T n)] ---------------
// (models.py:401-405, 1124-1150; the normal branch carries a stop_gradient, models.py:1132-1133 -> no backward)
__global__ void k_alpha_post(Dims D, int R, int S, const float* __restrict__ alpha, const float* __restrict__ wv, const float* __restrict__ viewdirs,
                             Windows W, float* __restrict__ sigma, float* __restrict__ cond) {
  const long long M = (long long)R * S, m0 = blockIdx.x * (long long)blockDim.x, ml = m0 + threadIdx.x, m = ml < M ? ml : M - 1;
  const int r = (int)(m / S);
  if (ml < M) sigma[m] = softplus_f(alpha[4 * m]);
  float n[3] = {alpha[4 * m + 1], alpha[4 * m + 2], alpha[4 * m + 3]};
  auto normalize = [](float (&v)[3]) {
    const float inv = 1.0f / sqrtf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], 1.1920929e-07f));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
  };
  normalize(n);
  const float w[3] = {wv[6 * m], wv[6 * m + 1], wv[6 * m + 2]}, v[3] = {wv[6 * m + 3], wv[6 * m + 4], wv[6 * m + 5]};
  float Rm[9], p[3];
  se3_Rp<float>(w, v, Rm, p);
  float nin[3];
  for (int c = 0; c < 3; ++c) nin[c] = Rm[c] * n[0] + Rm[3 + c] * n[1] + Rm[6 + c] * n[2];       // R^T n (inverse warp of a vector)
  normalize(nin);
  const float vd[3] = {viewdirs[3 * r], viewdirs[3 * r + 1], viewdirs[3 * r + 2]};
  const int CW = 6 * D.vd_bands + 6 * D.nm_bands;
  float* c = g_rows + threadIdx.x * tile_stride(CW);
  const float ones[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  for (int g = 0; g < 6 * D.vd_bands; ++g) c[g] = posenc_val<3>(g, vd, ones);
  for (int g = 0; g < 6 * D.nm_bands; ++g) c[6 * D.vd_bands + g] = posenc_val<3>(g, nin, W.nm);
  tile_store(cond, m0, M, CW, g_rows);
}

// ---- sigma gradient (models.py:1035-1077): forward-mode tangents of sigma_raw w.r.t. the observation-space
// point, 3 directions per sample, tangent row 3 m + j <-> d / d x_j.  The mask is a constant input (cal_single_pt_sigma takes
// it as an argument), the GLO columns have zero tangents.
// (one thread per (sample, column) of the wider of the two inputs: one cosf per feature for the three rows, coalesced stores)
__global__ void k_encode_tangents(Dims D, long long M, const float* __restrict__ x, Windows W, float* __restrict__ t_warp_in,
                                  float* __restrict__ t_hyper_in) {
  const int LD = D.warp_ld > D.hyper_ld ? D.warp_ld : D.hyper_ld;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= M * LD) return;
  const long long m = i / LD;
  const int g = (int)(i - m * LD);
  const float p[3] = {x[3 * m], x[3 * m + 1], x[3 * m + 2]};
  if (g < D.warp_ld) {
    const float dv = (g < 6 * D.warp_bands) ? posenc_dval<3>(g, p, W.warp) : 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) t_warp_in[(3 * m + j) * D.warp_ld + g] = (g < 6 * D.warp_bands && g % 3 == j) ? dv : 0.f;
  }
  if (g < D.hyper_ld) {
    const float dv = (g < 6 * D.hyp_bands) ? posenc_dval<3>(g, p, W.hyp) : 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) t_hyper_in[(3 * m + j) * D.hyper_ld + g] = (g < 6 * D.hyp_bands && g % 3 == j) ? dv : 0.f;
  }
}
// tangent through a ReLU: t[3 m + j][n] = 0 where y[m][n] <= 0
__global__ void k_relu_mask3(float* __restrict__ t, const float* __restrict__ y, long long M, int N) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= 3 * M * N) return;
  const long long row = i / N;
  const int n = (int)(i % N);
  if (!(y[(row / 3) * N + n] > 0.f)) t[i] = 0.f;
}
// d x' = R e_j + (d x' / d (w, v)) d(w, v)_j
// (launch bounds: without them hipcc budgets 128 registers - a 1024-thread block - and spills 194 / 1128 values of the dual numbers to scratch)
__global__ __launch_bounds__(256) void k_se3_jvp(long long M, const float* __restrict__ wv, const float* __restrict__ x, const float* __restrict__ t_wv,
                          float* __restrict__ t_xw) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  Dual w[3], v[3];
  for (int i = 0; i < 3; ++i) { w[i] = dconst(wv[6 * m + i]); w[i].g[i] = 1.f; v[i] = dconst(wv[6 * m + 3 + i]); v[i].g[3 + i] = 1.f; }
  Dual Rm[9], p[3];
  se3_Rp<Dual>(w, v, Rm, p);
  for (int r = 0; r < 3; ++r) {
    const Dual xr = Rm[3 * r] * x[3 * m] + Rm[3 * r + 1] * x[3 * m + 1] + Rm[3 * r + 2] * x[3 * m + 2] + p[r];
    for (int j = 0; j < 3; ++j) {
      float acc = Rm[3 * r + j].v;
      for (int i = 0; i < 6; ++i) acc += xr.g[i] * t_wv[(3 * m + j) * 6 + i];
      t_xw[(3 * m + j) * 3 + r] = acc;
    }
  }
}
// One thread per (sample, feature): the derivative factor of a feature - one cosf - serves the sample's three tangent rows, and consecutive threads
// write consecutive floats of a row.  (Through round 4: one thread per tangent ROW, 52 cosf each and every store instruction 64 rows apart: 0.75 ms
// per launch on 524 288 samples where the bytes take 0.1.)  Same arithmetic per element, same bits.
// rp: tangent rows per sample - 3 (unit directions, row 3 m + j) or 1 (ONE direction per sample: the reverse-mode second-order path)
__global__ void k_trunk_in_jvp(Dims D, long long M, const float* __restrict__ xw, const float* __restrict__ wamb, const float* __restrict__ t_xw,
                               const float* __restrict__ t_wamb, Windows W, float* __restrict__ t_tin, int rp) {
  const int TI = D.trunk_in, NS = 6 * D.sp_bands;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= M * TI) return;
  const long long m = i / TI;
  const int g = (int)(i - m * TI);
  float dv;
  int ch;
  const float* tsrc;
  int tld;
  if (g < NS) {
    const float p[3] = {xw[3 * m], xw[3 * m + 1], xw[3 * m + 2]};
    dv = posenc_dval<3>(g, p, W.sp); ch = g % 3; tsrc = t_xw; tld = 3;
  } else {
    const float a[2] = {wamb[2 * m], wamb[2 * m + 1]};
    dv = posenc_dval<2>(g - NS, a, W.hp); ch = (g - NS) % 2; tsrc = t_wamb; tld = 2;
  }
  if (rp == 1) { t_tin[m * TI + g] = dv * tsrc[m * tld + ch]; return; }
#pragma unroll
  for (int j = 0; j < 3; ++j) t_tin[(3 * m + j) * TI + g] = dv * tsrc[(3 * m + j) * tld + ch];
}
// sigma_gradient = normalize(-grad) (models.py:1069, 1077); target_norm = normalize(R sigma_gradient) ('warped', models.py:1273-1277)
__global__ void k_target_norm(long long M, const float* __restrict__ t_alpha, const float* __restrict__ wv, float* __restrict__ target_norm) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  float g[3] = {-t_alpha[(3 * m) * 4], -t_alpha[(3 * m + 1) * 4], -t_alpha[(3 * m + 2) * 4]};
  auto normalize = [](float (&v)[3]) {
    const float inv = 1.0f / sqrtf(fmaxf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2], 1.1920929e-07f));
    v[0] *= inv; v[1] *= inv; v[2] *= inv;
  };
  normalize(g);
  const float w[3] = {wv[6 * m], wv[6 * m + 1], wv[6 * m + 2]}, v[3] = {wv[6 * m + 3], wv[6 * m + 4], wv[6 * m + 5]};
  float Rm[9], p[3];
  se3_Rp<float>(w, v, Rm, p);
  float r[3];
  for (int c = 0; c < 3; ++c) r[c] = Rm[3 * c] * g[0] + Rm[3 * c + 1] * g[1] + Rm[3 * c + 2] * g[2];
  normalize(r);
  for (int c = 0; c < 3; ++c) target_norm[3 * m + c] = r[c];
}

// ---- norm loss (training.py:323-332): mean(w * |n - target_norm|), w = stop_gradient(weights), n the RAW predicted normal.  The
// reference does not stop the gradient at target_norm, so the loss is second order: its backward runs through the tangent pass.
// This kernel: the loss term, d / d n (into d_alpha[:, 1:4]), and d / d target_norm pulled back through the two normalisations
// and the rotation to  d g (tangent of the alpha head, column 0 of t_alpha)  and to the pair (du, ghat) from which the SE(3)
// kernel forms d / d (w, v) of <du, R ghat>.
__device__ __forceinline__ void normalize_bwd(const float (&v)[3], const float (&dy)[3], float (&dv)[3]) {   // y = v / sqrt(max(|v|^2, eps))
  const float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  if (n2 > 1.1920929e-07f) {
    const float inv = 1.0f / sqrtf(n2);
    const float dot = (v[0] * dy[0] + v[1] * dy[1] + v[2] * dy[2]) * inv * inv;
    for (int c = 0; c < 3; ++c) dv[c] = (dy[c] - v[c] * dot) * inv;
  } else {
    const float inv = 1.0f / sqrtf(1.1920929e-07f);
    for (int c = 0; c < 3; ++c) dv[c] = dy[c] * inv;
  }
}
__global__ void k_norm_loss(int R, int S, float weight, const float* __restrict__ weights, const float* __restrict__ alpha,
                            const float* __restrict__ t_alpha, const float* __restrict__ wv, const float* __restrict__ target_norm,
                            float* __restrict__ term, float* __restrict__ d_alpha, float* __restrict__ d_t_alpha, float* __restrict__ du_out,
                            float* __restrict__ ghat_out) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= (long long)R * S) return;
  const float k = weight * weights[m] / ((float)R * (float)S);
  float diff[3], nrm = 0.f;
  for (int c = 0; c < 3; ++c) { diff[c] = alpha[4 * m + 1 + c] - target_norm[3 * m + c]; nrm += diff[c] * diff[c]; }
  nrm = sqrtf(nrm);
  atomicAdd(term, k * nrm);
  float dt[3];
  for (int c = 0; c < 3; ++c) {
    const float gdir = nrm > 0.f ? diff[c] / nrm : 0.f;
    d_alpha[4 * m + 1 + c] += k * gdir;
    dt[c] = -k * gdir;
  }
  // target_norm = N(u), u = R ghat, ghat = N(hv), hv = -g
  float hv[3] = {-t_alpha[(3 * m) * 4], -t_alpha[(3 * m + 1) * 4], -t_alpha[(3 * m + 2) * 4]};
  float ghat[3];
  {
    const float inv = 1.0f / sqrtf(fmaxf(hv[0] * hv[0] + hv[1] * hv[1] + hv[2] * hv[2], 1.1920929e-07f));
    for (int c = 0; c < 3; ++c) ghat[c] = hv[c] * inv;
  }
  const float w[3] = {wv[6 * m], wv[6 * m + 1], wv[6 * m + 2]}, v[3] = {wv[6 * m + 3], wv[6 * m + 4], wv[6 * m + 5]};
  float Rm[9], p[3];
  se3_Rp<float>(w, v, Rm, p);
  float u[3];
  for (int r = 0; r < 3; ++r) u[r] = Rm[3 * r] * ghat[0] + Rm[3 * r + 1] * ghat[1] + Rm[3 * r + 2] * ghat[2];
  float du[3], dgh[3], dhv[3];
  normalize_bwd(u, dt, du);
  for (int c = 0; c < 3; ++c) dgh[c] = Rm[c] * du[0] + Rm[3 + c] * du[1] + Rm[6 + c] * du[2];       // R^T du
  normalize_bwd(hv, dgh, dhv);
  for (int j = 0; j < 3; ++j) {
    d_t_alpha[(3 * m + j) * 4] = -dhv[j];
    d_t_alpha[(3 * m + j) * 4 + 1] = 0.f; d_t_alpha[(3 * m + j) * 4 + 2] = 0.f; d_t_alpha[(3 * m + j) * 4 + 3] = 0.f;
  }
  for (int c = 0; c < 3; ++c) { du_out[3 * m + c] = du[c]; ghat_out[3 * m + c] = ghat[c]; }
}

// backward of k_trunk_in_jvp: d tangent(x'), d tangent(w), and - because the features' derivative factors depend on x' and w
// themselves - second-derivative contributions to the PRIMAL gradients of x' and the ambient coordinates
// One thread per (sample, channel) - three spatial channels, two ambient ones: the channel's 2 x bands features share their angles (sc = 1 is the
// same angle + pi/2), so a thread evaluates cosf / sinf once per feature for the sample's THREE tangent rows.  (Through round 4: one thread per
// sample looping over rows, channels and features - 2 x 52 x 3 libm calls - 0.8 ms per launch on 524 288 samples.)  Same arithmetic per term and
// the same order of the sums over the features of a channel; the second-derivative terms are summed per row first, then over the three rows.
__global__ void k_trunk_in_jvp_bwd(Dims D, long long M, const float* __restrict__ d_t_tin, const float* __restrict__ xw, const float* __restrict__ wamb,
                                   const float* __restrict__ t_xw, const float* __restrict__ t_wamb, Windows W, float* __restrict__ d_t_xw,
                                   float* __restrict__ d_t_wamb, float* __restrict__ dxw_extra, float* __restrict__ dwamb_extra, int rp) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= 5 * M) return;
  const long long m = i / 5;
  const int c5 = (int)(i - 5 * m);
  const bool sp = c5 < 3;
  const int c = sp ? c5 : c5 - 3, C = sp ? 3 : 2, nb = sp ? D.sp_bands : D.hp_bands, g0 = sp ? 0 : 6 * D.sp_bands;
  const float pc = sp ? xw[3 * m + c] : wamb[2 * m + c];
  const float* win = sp ? W.sp : W.hp;
  const float* tt = sp ? t_xw : t_wamb;
  float* dtt = sp ? d_t_xw : d_t_wamb;
  float acc[3] = {0.f, 0.f, 0.f}, ex[3] = {0.f, 0.f, 0.f};
  for (int bs = 0; bs < 2 * nb; ++bs) {
    const int g = C * bs + c, band = g / (2 * C), sc = (g % (2 * C)) / C;
    const float sc2 = (float)(1 << band), arg = pc * sc2 + (sc ? 1.57079637f : 0.0f);
    const float fc = win[band] * sc2 * cosf(arg), fs = -win[band] * sc2 * sc2 * sinf(arg);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < rp) {
        const float dt = d_t_tin[(rp * m + j) * D.trunk_in + g0 + g];
        acc[j] += dt * fc;
        ex[j] += dt * tt[(rp * m + j) * C + c] * fs;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) if (j < rp) dtt[(rp * m + j) * C + c] = acc[j];
  if (sp) dxw_extra[3 * m + c] += (ex[0] + ex[1]) + ex[2];
  else dwamb_extra[2 * m + c] = (ex[0] + ex[1]) + ex[2];
}

// single-direction dual over any scalar T (float or Dual): the nested type D1<Dual> differentiates a directional derivative
template <class T> struct D1 { T v, d; };
template <class T> __device__ __forceinline__ D1<T> operator+(const D1<T>& a, const D1<T>& b) { return {a.v + b.v, a.d + b.d}; }
template <class T> __device__ __forceinline__ D1<T> operator-(const D1<T>& a, const D1<T>& b) { return {a.v - b.v, a.d - b.d}; }
template <class T> __device__ __forceinline__ D1<T> operator-(const D1<T>& a) { return {-a.v, -a.d}; }
template <class T> __device__ __forceinline__ D1<T> operator*(const D1<T>& a, const D1<T>& b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
template <class T> __device__ __forceinline__ D1<T> operator/(const D1<T>& a, const D1<T>& b) {
  const T q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
template <class T> __device__ __forceinline__ D1<T> operator*(const D1<T>& a, float b) { return {a.v * b, a.d * b}; }
template <class T> __device__ __forceinline__ D1<T> dsqrt(const D1<T>& a) { const T r = dsqrt(a.v); return {r, a.d / (r + r)}; }
template <class T> __device__ __forceinline__ float base_of(const D1<T>& a) { return base_of(a.v); }
template <class T> __device__ __forceinline__ D1<T> dsin_sc(const D1<T>& a, float s, float c) { return {dsin_sc(a.v, s, c), a.d * dcos_sc(a.v, s, c)}; }
template <class T> __device__ __forceinline__ D1<T> dcos_sc(const D1<T>& a, float s, float c) { return {dcos_sc(a.v, s, c), -(a.d * dsin_sc(a.v, s, c))}; }
template <class T> __device__ __forceinline__ D1<T> tconst(float c, const D1<T>*) { return {tconst(c, (const T*)nullptr), tconst(0.f, (const T*)nullptr)}; }

// backward of k_se3_jvp (t_xw_j = R e_j + J t_wv_j) for upstream a_j = d t_xw_j, plus the rotation used by target_norm (<du, R ghat>):
//   d t_wv_j = J^T a_j,   d (w, v) += grad_{(w,v)} [ sum_j <a_j, R e_j + DF[t_wv_j]> + <du, R ghat> ]     (second derivatives of exp_se3)
// One thread per (sample, direction i of (w, v)): component i of every gradient above is a derivative along e_i, so the thread runs exp_se3 on
// single-direction duals - D1<float> seeded with e_i for the first-order pieces, D1<D1<float>> (outer: along t_wv_j, inner: along e_i) for the
// second-order one - 2 and 4 floats per scalar.  (Until round 5's end one thread per sample carried all six directions at once, 7 and 14 floats per
// scalar: 256 registers, 403 spilled, 1.5 KiB of scratch per lane, one wave per SIMD.)
// Measured alone (one stream, both levels of a 4096-ray step): 1.53 ms before, 0.67 ms now; held to two or four waves per SIMD the compiler spills to
// scratch (228 - 908 bytes per lane) and the kernel is SLOWER than before (1.63 ms at two) - it is left the whole register file.
__global__ __launch_bounds__(256) void k_se3_jvp_bwd(long long M, const float* __restrict__ wv, const float* __restrict__ x, const float* __restrict__ t_wv,
                              const float* __restrict__ d_t_xw, const float* __restrict__ du, const float* __restrict__ ghat,
                              float* __restrict__ d_t_wv, float* __restrict__ dwv_extra, const float* __restrict__ extra_in, const float* __restrict__ dir) {
  // du / ghat null: no rotation term here (the merged step forms it per level, k_se3_rot_bwd, and hands the sum in as extra_in)
  // dir non-null: ONE tangent row per sample, t_xw = R dir + J t_wv (k_se3_jvp_dir), instead of the three unit directions e_j
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= 6 * M) return;
  const long long m = idx / 6;
  const int i = (int)(idx - 6 * m);
  const float xs[3] = {x[3 * m], x[3 * m + 1], x[3 * m + 2]};
  float wvs[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) wvs[k] = wv[6 * m + k];
  float extra = 0.f;
  {  // first-order pieces: d / d (w, v)_i of x' and of R
    D1<float> w[3], v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { w[k] = {wvs[k], i == k ? 1.f : 0.f}; v[k] = {wvs[3 + k], i == 3 + k ? 1.f : 0.f}; }
    D1<float> Rm[9], p[3];
    se3_Rp<D1<float>>(w, v, Rm, p);
    float xd[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) xd[r] = (Rm[3 * r] * xs[0] + Rm[3 * r + 1] * xs[1] + Rm[3 * r + 2] * xs[2] + p[r]).d;
    const int rp = dir != nullptr ? 1 : 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < rp) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float a = d_t_xw[(rp * m + j) * 3 + r];
          const float rd = dir != nullptr ? Rm[3 * r].d * dir[3 * m] + Rm[3 * r + 1].d * dir[3 * m + 1] + Rm[3 * r + 2].d * dir[3 * m + 2] : Rm[3 * r + j].d;
          acc += a * xd[r]; extra += a * rd;                                                        // J^T a_j ; grad <a_j, R e_j> (<a, R dir>)
        }
        d_t_wv[(rp * m + j) * 6 + i] = acc;
      }
    }
    if (du != nullptr) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) extra += du[3 * m + r] * ghat[3 * m + c] * Rm[3 * r + c].d;         // grad <du, R ghat>
    }
  }
  const int rp2 = dir != nullptr ? 1 : 3;
#pragma unroll 1
  for (int j = 0; j < rp2; ++j) {  // grad_{(w,v)} <a_j, DF[t_wv_j]>: the directional derivative along t_wv_j, differentiated again along e_i
    D1<D1<float>> w[3], v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      w[k].v = {wvs[k], i == k ? 1.f : 0.f};         w[k].d = {t_wv[(rp2 * m + j) * 6 + k], 0.f};
      v[k].v = {wvs[3 + k], i == 3 + k ? 1.f : 0.f}; v[k].d = {t_wv[(rp2 * m + j) * 6 + 3 + k], 0.f};
    }
    D1<D1<float>> Rm[9], p[3];
    se3_Rp<D1<D1<float>>>(w, v, Rm, p);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const D1<D1<float>> xr = Rm[3 * r] * xs[0] + Rm[3 * r + 1] * xs[1] + Rm[3 * r + 2] * xs[2] + p[r];
      extra += d_t_xw[(rp2 * m + j) * 3 + r] * xr.d.d;
    }
  }
  dwv_extra[6 * m + i] = extra + (extra_in != nullptr ? extra_in[6 * m + i] : 0.f);
}
// the rotation term alone: out[m][i] = d / d (w, v)_i <du_m, R(w, v) ghat_m> (bilinear in the LEVEL's (du, ghat): the merged step evaluates it per level
// in the level's row order and adds the levels up per sample position)
__global__ __launch_bounds__(256) void k_se3_rot_bwd(long long M, const float* __restrict__ wv, const float* __restrict__ du, const float* __restrict__ ghat,
                                                     float* __restrict__ out) {
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (idx >= 6 * M) return;
  const long long m = idx / 6;
  const int i = (int)(idx - 6 * m);
  D1<float> w[3], v[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { w[k] = {wv[6 * m + k], i == k ? 1.f : 0.f}; v[k] = {wv[6 * m + 3 + k], i == 3 + k ? 1.f : 0.f}; }
  D1<float> Rm[9], p[3];
  se3_Rp<D1<float>>(w, v, Rm, p);
  float extra = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) extra += du[3 * m + r] * ghat[3 * m + c] * Rm[3 * r + c].d;
  out[6 * m + i] = extra;
}

// ---- compositing (model_utils.py:95-159), MSE loss (training.py:265-274) and the backward of both; one WAVE per ray ----
// Lanes = samples (64 at a time): the transmittance is a wave-level exclusive product scan with a carry between chunks, the backward's
// sum over the later samples a reverse exclusive sum scan.  (Through round 4's first half: one THREAD per ray walking its S samples - 16
// workgroups for 4096 rays, every load strided by S, 76 + 149 us per step for 10 MB of data.)
__device__ __forceinline__ float wave_scan_prod(float x, int lane) {      // inclusive, lanes 0 .. 63
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const float v = __shfl_up(x, d, 64); if (lane >= d) x *= v; }
  return x;
}
__device__ __forceinline__ float wave_rscan_sum(float x, int lane) {      // inclusive from the top: lane l = sum of lanes l .. 63
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const float v = __shfl_down(x, d, 64); if (lane + d < 64) x += v; }
  return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
  return x;
}
// loss += mean over rays and channels of (rgb_ray - target)^2 (training.py:265-274): one block, every thread a fixed set of rays, a tree in LDS
__global__ __launch_bounds__(256) void k_mse_sum(int R, const float* __restrict__ rgb_ray, const float* __restrict__ target, float* __restrict__ loss) {
  __shared__ float part[256];
  float l = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    float e2 = 0.f;
    for (int c = 0; c < 3; ++c) { const float e = rgb_ray[3 * r + c] - target[3 * r + c]; e2 += e * e; }
    l += e2 / (3.0f * (float)R);
  }
  part[threadIdx.x] = l;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss += part[0];
}
// `cot` (nerfds_render_rays_bwd, a caller-defined loss): the cotangents of the level's per-ray outputs - d loss / d rgb [R][3], / d depth [R], / d acc [R],
// each nullable = zero - REPLACE the gradient of the built-in squared error (target may then be null: no loss is reported).  `out`: rgb / depth / acc of
// the level (model_utils.py:138-148) for nerfds_trainer_forward.
__global__ __launch_bounds__(256) void k_composite_loss(int R, int S, const float* __restrict__ z, const float* __restrict__ dirs, const float* __restrict__ sigma,
                                 const float* __restrict__ rgb_logit, const float* __restrict__ target, int at_infinity, int white,
                                 float* __restrict__ rgb_ray, float* __restrict__ weights, float* __restrict__ loss,
                                 float* __restrict__ d_rgb_logit, float* __restrict__ d_alpha, LevelCot cot, LevelOut out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;                                     // whole waves
  const float* zr = z + (size_t)r * S;
  const float dn = sqrtf(dirs[3 * r] * dirs[3 * r] + dirs[3 * r + 1] * dirs[3 * r + 1] + dirs[3 * r + 2] * dirs[3 * r + 2]);
  const float last = at_infinity ? 1e10f : 1e-19f;
  float carryT = 1.0f, acc[3] = {0.f, 0.f, 0.f}, wsum = 0.f, dsum = 0.f, asum = 0.f;
  for (int c0 = 0; c0 < S; c0 += 64) {
    const int s = c0 + lane;
    const bool valid = s < S;
    const size_t m = (size_t)r * S + (valid ? s : S - 1);
    const float dist = ((s >= S - 1) ? last : (zr[s + 1] - zr[s])) * dn;
    const float a = valid ? 1.0f - expf(-sigma[m] * dist) : 0.f;
    const float om = valid ? (1.0f - a) + 1e-10f : 1.0f;
    const float incl = wave_scan_prod(om, lane);
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.0f;
    const float w = a * (carryT * excl);
    carryT *= __shfl(incl, 63, 64);
    if (valid) {
      weights[m] = w;
      wsum += w;
      dsum += w * zr[s];                                            // model_utils.py:139
      asum += (at_infinity && s == S - 1) ? 0.f : w;                 // model_utils.py:141, 147-148
      for (int c = 0; c < 3; ++c) acc[c] += w / (1.0f + expf(-rgb_logit[3 * m + c]));
    }
  }
  wsum = wave_sum(wsum);
  const bool ext = cot.on != 0;
  float g[3], gsum = 0.f, l = 0.f;
  for (int c = 0; c < 3; ++c) {
    acc[c] = wave_sum(acc[c]);
    if (white) acc[c] += 1.0f - wsum;
    const float e = target != nullptr ? acc[c] - target[3 * r + c] : 0.f;
    l += e * e;
    g[c] = ext ? (cot.d_rgb != nullptr ? cot.d_rgb[3 * r + c] : 0.f) : 2.0f * e / (3.0f * (float)R);
    gsum += g[c];
  }
  const float g_depth = (ext && cot.d_depth != nullptr) ? cot.d_depth[r] : 0.f;
  const float g_acc = (ext && cot.d_acc != nullptr) ? cot.d_acc[r] : 0.f;
  if (out.rgb != nullptr || out.depth != nullptr || out.acc != nullptr) {
    dsum = wave_sum(dsum); asum = wave_sum(asum);
    if (lane == 0) {
      if (out.rgb != nullptr) for (int c = 0; c < 3; ++c) out.rgb[3 * r + c] = acc[c];
      if (out.depth != nullptr) out.depth[r] = dsum;
      if (out.acc != nullptr) out.acc[r] = asum;
    }
  }
  if (lane == 0) {
    for (int c = 0; c < 3; ++c) rgb_ray[3 * r + c] = acc[c];      // the loss is summed from these in a fixed order (k_mse_sum): a float atomic per ray
  }                                                                // from waves on different CUs would make the reported loss depend on their timing
  // backward: w_i = a_i T_i, T_i = prod_{j<i} (1 - a_j + eps);  dL/da_i = G_i T_i - (sum_{k>i} G_k w_k) / (1 - a_i + eps)
  float carryS = 0.f;
  for (int c0 = ((S - 1) / 64) * 64; c0 >= 0; c0 -= 64) {
    const int s = c0 + lane;
    const bool valid = s < S;
    const size_t m = (size_t)r * S + (valid ? s : S - 1);
    const float dist = ((s >= S - 1) ? last : (zr[s + 1] - zr[s])) * dn;
    const float sg = sigma[m];
    const float ex = expf(-sg * dist);
    const float a = 1.0f - ex;
    const float w = valid ? weights[m] : 0.f;
    const float om = (1.0f - a) + 1e-10f;
    const float Ti = (a > 0.f) ? w / a : 0.f;     // T_i (only its product with d a / d sigma matters; a == 0 only if sigma == 0)
    float G = white ? -gsum : 0.f;
    // d depth / d w_i = z_i, d acc / d w_i = 1 except for the sample at infinity (zero unless the caller's loss reads depth / acc)
    G += g_depth * zr[valid ? s : S - 1] + ((at_infinity && s >= S - 1) ? 0.f : g_acc);
    float dl[3];
    for (int c = 0; c < 3; ++c) {
      const float col = 1.0f / (1.0f + expf(-rgb_logit[3 * m + c]));
      G += g[c] * col;
      dl[c] = g[c] * w * col * (1.0f - col);
    }
    const float Gw = valid ? G * w : 0.f;
    const float incl = wave_rscan_sum(Gw, lane);
    float excl = __shfl_down(incl, 1, 64);
    if (lane == 63) excl = 0.f;
    const float suffix = carryS + excl;
    carryS += __shfl(incl, 0, 64);
    if (valid) {
      for (int c = 0; c < 3; ++c) d_rgb_logit[3 * m + c] = dl[c];
      const float dLda = G * Ti - suffix / om;
      const float dads = dist * ex;                                   // d a / d sigma
      // d sigma / d sigma_raw = sigmoid(sigma_raw) = 1 - exp(-sigma); normal channels: stop_gradient
      *reinterpret_cast<float4*>(d_alpha + 4 * m) = make_float4(dLda * dads * (1.0f - expf(-sg)), 0.f, 0.f, 0.f);
    }
  }
}

// ---- the first-order auxiliary losses of the reference objective (training.py:297-310, 334-339, 386-408), per level, one
// thread per ray; all of them read the compositing weights as constants (lax.stop_gradient(model_out['weights'])):
//   warp_reg   : general_loss_with_squared_residual(|x - x'|^2 at the median-depth sample, alpha, scale).mean()   (utils.py:208-263)
//   back_facing: mean(w * relu(n . viewdir)^2)                                                (models.py:1340-1343)
//   3-D mask   : mean((gt_mask - sum_s sw_s * predicted_mask_s)^2), sw = sharpen_weights(w, z, std) (incl. its row-gather
//                quirk, model_utils.py:180-190) or w
// Outputs: the weighted loss terms (atomicAdd into terms[0..2]) and their gradients w.r.t. x' (dxw_reg, dense, zero except the
// median sample), the raw normal (added into d_alpha[:, 1:4]) and the predicted mask (d_pm).
__device__ __forceinline__ float general_loss_sq(float x_sq, float alpha, float scale, float& dloss) {
  const float eps = 1.1920929e-07f;
  scale = fmaxf(eps, scale);
  const float inv_s2 = 1.0f / (scale * scale);
  const float loss_two = 0.5f * x_sq * inv_s2;
  if (alpha == 2.0f) { dloss = scale * 0.5f * inv_s2; return scale * loss_two; }
  if (alpha == 0.0f) { const float c = fminf(loss_two, 3e37f); dloss = scale * 0.5f * inv_s2 / (1.0f + c); return scale * log1pf(c); }
  const float a = (alpha >= 0.f ? 1.f : -1.f) * fmaxf(eps, fabsf(alpha));
  const float b = fmaxf(eps, fabsf(alpha - 2.0f));
  const float base = loss_two / (0.5f * b) + 1.0f;
  const float pw = powf(base, 0.5f * alpha);
  dloss = scale * (b / a) * (0.5f * alpha) * (pw / base) * (1.0f / (0.5f * b)) * 0.5f * inv_s2;
  return scale * (b / a) * (pw - 1.0f);
}
__global__ void k_aux_losses(int R, int S, Objective ob, const float* __restrict__ z, const float* __restrict__ weights, const float* __restrict__ x,
                             const float* __restrict__ xw, const float* __restrict__ alpha, const float* __restrict__ viewdirs,
                             const float* __restrict__ mask_logit, const float* __restrict__ gt_mask, float* __restrict__ terms,
                             float* __restrict__ dxw_reg, float* __restrict__ d_alpha, float* __restrict__ d_pm, const float* __restrict__ wamb,
                             float* __restrict__ term_hyper, float* __restrict__ dwamb_reg, float* __restrict__ term_occlusion) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* w = weights + (size_t)r * S;
  // hyper-point regulariser (training.py:312-321): (w * general_loss(|ambient|^2, alpha 0, scale 0.05)).sum(1).mean(), w constant
  if (ob.hyper_reg_weight != 0.f && wamb != nullptr) {
    const float k = ob.hyper_reg_weight / (float)R;
    float l = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t m = (size_t)r * S + s;
      const float a0 = wamb[2 * m], a1 = wamb[2 * m + 1];
      float dl;
      l += w[s] * general_loss_sq(a0 * a0 + a1 * a1, 0.0f, 0.05f, dl);
      dwamb_reg[2 * m] = k * w[s] * dl * 2.0f * a0;
      dwamb_reg[2 * m + 1] = k * w[s] * dl * 2.0f * a1;
    }
    atomicAdd(term_hyper, k * l);
  }
  // median-depth index (model_utils.py:272-299): first s with cumsum(w) >= 0.5, else 0; arg-max of the weights
  int med = 0, amax = 0;
  {
    float cum = 0.f, best = w[0];
    bool found = false;
    for (int s = 0; s < S; ++s) {
      cum += w[s];
      if (!found && cum >= 0.5f) { med = s; found = true; }
      if (w[s] > best) { best = w[s]; amax = s; }
    }
  }
  // warp regularisation
  for (int s = 0; s < S; ++s) { const size_t m = (size_t)r * S + s; dxw_reg[3 * m] = 0.f; dxw_reg[3 * m + 1] = 0.f; dxw_reg[3 * m + 2] = 0.f; }
  if (ob.warp_reg_weight != 0.f) {
    const size_t m = (size_t)r * S + med;
    float dvec[3], sq = 0.f;
    for (int c = 0; c < 3; ++c) { dvec[c] = x[3 * m + c] - xw[3 * m + c]; sq += dvec[c] * dvec[c]; }
    float dl;
    const float l = general_loss_sq(sq, ob.warp_reg_alpha, ob.warp_reg_scale, dl);
    atomicAdd(terms + 0, ob.warp_reg_weight * l / (float)R);
    for (int c = 0; c < 3; ++c) dxw_reg[3 * m + c] = ob.warp_reg_weight / (float)R * dl * (-2.0f * dvec[c]);
  }
  // back-facing regulariser on the RAW predicted normal
  if (ob.back_facing_weight != 0.f) {
    const float v[3] = {viewdirs[3 * r], viewdirs[3 * r + 1], viewdirs[3 * r + 2]};
    const float k = ob.back_facing_weight / ((float)R * (float)S);
    float l = 0.f;
    for (int s = 0; s < S; ++s) {
      const size_t m = (size_t)r * S + s;
      const float d = fmaxf(alpha[4 * m + 1] * v[0] + alpha[4 * m + 2] * v[1] + alpha[4 * m + 3] * v[2], 0.f);
      l += w[s] * d * d;
      for (int c = 0; c < 3; ++c) d_alpha[4 * m + 1 + c] += k * w[s] * 2.0f * d * v[c];
    }
    atomicAdd(terms + 1, k * l);
  }
  // 3-D mask supervision
  for (int s = 0; s < S; ++s) d_pm[(size_t)r * S + s] = 0.f;
  if (ob.mask_loss_weight != 0.f && gt_mask != nullptr) {
    const float* zr = z + (size_t)r * S;
    float rpm = 0.f, norm = 1.f;
    if (ob.use_sharp_weights) {
      const int row = amax < R ? amax : R - 1;                    // z_vals[max_weights_idx]: a ROW gather (model_utils.py:182), jnp clamps
      const float* zq = z + (size_t)row * S;
      const float inv = 1.0f / ob.sharp_weights_std, c0 = inv * 0.3989422804f;
      norm = 0.f;
      for (int s = 0; s < S; ++s) { const float t = (zr[s] - zq[s]) * inv; norm += w[s] * expf(-0.5f * t * t) * c0; }
      for (int s = 0; s < S; ++s) {
        const float t = (zr[s] - zq[s]) * inv;
        rpm += w[s] * expf(-0.5f * t * t) * c0 / norm * fmaxf(mask_logit[(size_t)r * S + s], 0.f);
      }
      const float e = rpm - gt_mask[r];
      atomicAdd(terms + 2, ob.mask_loss_weight * e * e / (float)R);
      for (int s = 0; s < S; ++s) {
        const float t = (zr[s] - zq[s]) * inv;
        d_pm[(size_t)r * S + s] = ob.mask_loss_weight * 2.0f * e / (float)R * (w[s] * expf(-0.5f * t * t) * c0 / norm);
      }
    } else {
      for (int s = 0; s < S; ++s) rpm += w[s] * fmaxf(mask_logit[(size_t)r * S + s], 0.f);
      const float e = rpm - gt_mask[r];
      atomicAdd(terms + 2, ob.mask_loss_weight * e * e / (float)R);
      for (int s = 0; s < S; ++s) d_pm[(size_t)r * S + s] = ob.mask_loss_weight * 2.0f * e / (float)R * w[s];
    }
  }
  // mask occlusion regulariser (training.py:409-417): where hardly any weight lands, the predicted mask should be 0
  if (ob.mask_occlusion_weight != 0.f && term_occlusion != nullptr) {
    const float k = ob.mask_occlusion_weight / (float)R;
    float l = 0.f;
    for (int s = 0; s < S; ++s) {
      const float low = fmaxf(0.01f - w[s], 0.f);
      l += low * fmaxf(mask_logit[(size_t)r * S + s], 0.f);        // |relu(logit)|
      d_pm[(size_t)r * S + s] += k * low;                          // (the ReLU's own mask is applied with d_pm in shared_in_bwd)
    }
    atomicAdd(term_occlusion, k * l);
  }
}

__global__ void k_relu_bwd(float* __restrict__ dy, const float* __restrict__ y, long long n) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n && !(y[i] > 0.f)) dy[i] = 0.f;
}

// db[n] += sum_m dz[m][n] (bias gradients): one block per 256 rows, one atomicAdd per column and block
__global__ void k_colsum_add(const float* __restrict__ dz, long long M, int N, int ld, float* __restrict__ db) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int per = N < 256 ? N : 256, rows_per_pass = 256 / per;
  const long long r0 = blockIdx.x * 256LL, r1 = (r0 + 256 < M) ? r0 + 256 : M;
  const int col = tid % per, sub = tid / per;
  float s = 0.f;
  if (sub < rows_per_pass)
    for (long long r = r0 + sub; r < r1; r += rows_per_pass) s += dz[r * ld + col];
  red[tid] = s;
  __syncthreads();
  if (tid < per) {
    float t = 0.f;
    for (int k = 0; k < rows_per_pass; ++k) t += red[tid + k * per];
    atomicAdd(db + tid, t);
  }
}

// ReLU backward fused with the bias gradient: dy = (y > 0) ? dy : 0 in place, db += column sums (one pass over dy, y)
__global__ void k_relu_bwd_colsum(float* __restrict__ dy, const float* __restrict__ y, long long M, int N, float* __restrict__ db) {
  __shared__ float red[256];
  const int tid = threadIdx.x;
  const int per = N < 256 ? N : 256, rows_per_pass = 256 / per;
  const long long r0 = blockIdx.x * 128LL, r1 = (r0 + 128 < M) ? r0 + 128 : M;
  const int col = tid % per, sub = tid / per;
  float s = 0.f;
  if (sub < rows_per_pass)
    for (long long r = r0 + sub; r < r1; r += rows_per_pass) {
      const long long i = r * N + col;
      const float v = (y[i] > 0.f) ? dy[i] : 0.f;
      dy[i] = v;
      s += v;
    }
  red[tid] = s;
  __syncthreads();
  if (tid < per) {
    float t = 0.f;
    for (int k = 0; k < rows_per_pass; ++k) t += red[tid + k * per];
    atomicAdd(db + tid, t);
  }
}

// gradient of the shared-net inputs: the mask column (models.py:729-732) -> mask head, the GLO columns -> embedding rows.
// One thread per SAMPLE (one per ray left 64 waves on the whole chip walking R x S rows); the 8 GLO sums of a ray's samples meet in
// LDS (ds_add_f32), then one global atomic per (ray of the block, column).
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dpp_add(float v) {      // v + (v moved by the DPP control, 0 where no lane feeds)
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_total(float v) {      // sum of the 64 lanes, valid in lane 63
  v = dpp_add<0xB1>(v); v = dpp_add<0x4E>(v); v = dpp_add<0x141>(v); v = dpp_add<0x140>(v);       // quads, half rows, rows of 16
  v = dpp_add<0x142, 0xA>(v); v = dpp_add<0x143, 0xC>(v);                                          // row_bcast:15, row_bcast:31
  return v;
}
__device__ __forceinline__ void glo_sums(const float (&e)[8], int lr, int nrays, long long r0, const uint32_t* __restrict__ warp_id, int n_embeds,
                                         float* __restrict__ d_tbl, float* acc) {
  for (int i = threadIdx.x; i < nrays * 8; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  // a wave whose 64 samples belong to one ray (S a multiple of 64: every wave) adds its DPP sums once; a mixed wave adds lane by lane
  const int lr0 = __builtin_amdgcn_readfirstlane(lr);
  if (__builtin_amdgcn_ballot_w64(lr != lr0) == 0) {
    for (int g = 0; g < 8; ++g) {
      const float t = wave_total(e[g]);
      if ((threadIdx.x & 63) == 63) unsafeAtomicAdd(acc + lr0 * 8 + g, t);
    }
  } else {
    for (int g = 0; g < 8; ++g) unsafeAtomicAdd(acc + lr * 8 + g, e[g]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nrays * 8; i += blockDim.x) {
    uint32_t id = warp_id ? warp_id[r0 + i / 8] : 0u;
    if (id >= (uint32_t)n_embeds) id = n_embeds - 1;
    unsafeAtomicAdd(d_tbl + id * 8 + (i & 7), acc[i]);
  }
}
__global__ void k_shared_in_bwd(Dims D, int R, int S, const float* __restrict__ d_warp_in, const float* __restrict__ d_hyper_in,
                                const float* __restrict__ mask_logit, float ratio, const float* __restrict__ d_pm_extra,
                                const uint32_t* __restrict__ warp_id, int n_embeds, float* __restrict__ d_warp_tbl, float* __restrict__ d_mask_logit) {
  const long long M = (long long)R * S, m0 = blockIdx.x * (long long)blockDim.x, m = m0 + threadIdx.x;
  const long long last = (m0 + blockDim.x < M ? m0 + blockDim.x : M) - 1, r0 = m0 / S;
  const int nrays = (int)(last / S - r0) + 1;
  float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int lr = 0;
  if (m < M) {
    lr = (int)(m / S - r0);
    const float* dw = d_warp_in + m * D.warp_ld;
    const float* dh = d_hyper_in + m * D.hyper_ld;
    for (int g = 0; g < 8; ++g) e[g] = dw[6 * D.warp_bands + g] + dh[6 * D.hyp_bands + g];
    const float dmask = dw[D.warp_in - 1] + dh[D.hyper_in - 1];
    d_mask_logit[m] = (mask_logit[m] > 0.f) ? dmask * ratio + (d_pm_extra ? d_pm_extra[m] : 0.f) : 0.f;
  }
  glo_sums(e, lr, nrays, r0, warp_id, n_embeds, d_warp_tbl, g_rows);
}
__global__ void k_mask_in_bwd(Dims D, int R, int S, const float* __restrict__ d_mask_in, const uint32_t* __restrict__ warp_id, int n_embeds,
                              float* __restrict__ d_mask_tbl) {
  const long long M = (long long)R * S, m0 = blockIdx.x * (long long)blockDim.x, m = m0 + threadIdx.x;
  const long long last = (m0 + blockDim.x < M ? m0 + blockDim.x : M) - 1, r0 = m0 / S;
  const int nrays = (int)(last / S - r0) + 1;
  float e[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int lr = 0;
  if (m < M) {
    lr = (int)(m / S - r0);
    const float* dm = d_mask_in + m * D.mask_in;
    for (int g = 0; g < 8; ++g) e[g] = dm[6 * D.mask_bands + g];
  }
  glo_sums(e, lr, nrays, r0, warp_id, n_embeds, d_mask_tbl, g_rows);
}

// flax.optim.Adam (flax 0.3.4): bias-corrected, no weight decay (training.py:508, train.py:297-301)
// The plain step stores activations as f16 and g as loss-scaled f16: an activation beyond 65504 becomes inf there and arrives as an
// inf / NaN weight gradient.  k_nonfinite raises a flag if ANY gradient element is not finite; k_adam then leaves parameters and moments
// untouched (the whole update is skipped, not element by element), and the host reports NERFDS_ENONFINITE at its next read-back.
__global__ void k_nonfinite(const float* __restrict__ g, long long n, unsigned* __restrict__ flag) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n && !isfinite(g[i])) atomicOr(flag, 1u);
}
// Overflow diagnosis, run by nerfds_trainer_overflow_sources AFTER a step whose gradient check failed (never inside a step): which stored array holds an
// inf / NaN.  f16: exponent field all ones; fp32: !isfinite.  One atomicOr per block that finds one.
__global__ __launch_bounds__(256) void k_scan_half(const uint16_t* __restrict__ p, long long n8, unsigned* __restrict__ flags, unsigned bit) {
  unsigned bad = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
    const uint4 v = reinterpret_cast<const uint4*>(p)[i];
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) bad |= (unsigned)(((w[k] & 0x7c00u) == 0x7c00u) | ((w[k] & 0x7c000000u) == 0x7c000000u));
  }
  if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flags, bit);
}
__global__ __launch_bounds__(256) void k_scan_float(const float* __restrict__ p, long long n, unsigned* __restrict__ flags, unsigned bit, float limit) {
  unsigned bad = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) bad |= !(fabsf(p[i]) <= limit);      // (NaN: false)
  if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(flags, bit);
}
// The optimizer's step count lives on the DEVICE (state[0] as int64; state[1], state[2] = the bias corrections 1 - b^t of the step about to be
// applied): a skipped update must not advance it - the moments did not move, so neither may the bias correction nor the checkpointed
// OptimizerState.step (flax 0.3.4 optim/adam.py: step + 1 only in apply_gradient) - and the host does not know about the skip when it enqueues.
__global__ void k_adam_prepare(const unsigned* __restrict__ skip, long long* __restrict__ step, float* __restrict__ corr, double b1, double b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0 || *skip != 0u) return;
  const long long t = *step + 1;
  *step = t;
  corr[0] = (float)(1.0 - pow(b1, (double)t));
  corr[1] = (float)(1.0 - pow(b2, (double)t));
}
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, long long n,
                       float lr, float b1, float b2, float eps, const float* __restrict__ corr, const unsigned* __restrict__ skip) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n || *skip != 0u) return;
  const float c1 = corr[0], c2 = corr[1];
  const float gi = g[i];
  const float a = (1.0f - b1) * gi + b1 * m1[i];
  const float b = (1.0f - b2) * gi * gi + b2 * m2[i];
  m1[i] = a; m2[i] = b;
  p[i] -= lr * (a / c1) / (sqrtf(b / c2) + eps);
}

__global__ void k_sum_partials(const float* __restrict__ part, int slabs, long long n, float* __restrict__ out) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < slabs; ++k) s += part[(long long)k * n + i];
  out[i] += s;
}
__global__ void k_fill(float* __restrict__ p, long long n, float v) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// utils.clip_gradients (utils.py:32-47): clip by value, then scale so that the global L2 norm is at most max_norm
__global__ void k_clip_val_sumsq(float* __restrict__ g, long long n, float max_val, float* __restrict__ sumsq) {
  __shared__ float red[256];
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  float v = 0.f;
  if (i < n) {
    v = g[i];
    if (max_val > 0.f) { v = fminf(fmaxf(v, -max_val), max_val); g[i] = v; }
  }
  red[threadIdx.x] = v * v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) atomicAdd(sumsq, red[0]);
}
__global__ void k_clip_norm(float* __restrict__ g, long long n, float max_norm, float eps, const float* __restrict__ sumsq) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float mult = fminf(1.0f, max_norm / (eps + sqrtf(*sumsq)));
  g[i] *= mult;
}

static inline dim3 grid1(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }
#define LAUNCH(kern, n, stream, ...) hipLaunchKernelGGL(kern, grid1(n), dim3(256), 0, stream, __VA_ARGS__)

void coarse_z(hipStream_t st, int R, int Nc, float near_, float far_, int stratified, int lindisp, const float* t_rand, uint64_t seed, long long first_ray, float* z) {
  LAUNCH(k_coarse_z, (long long)R * Nc, st, R, Nc, near_, far_, stratified, lindisp, t_rand, seed, first_ray, z);
}
void gather_rows(hipStream_t st, long long M, const int* src, const float* xw, const float* wamb, const float* wv, float* xw_f, float* wamb_f, float* wv_f) {
  LAUNCH(k_gather_rows, M, st, M, src, xw, wamb, wv, xw_f, wamb_f, wv_f);
}
void scatter_rows(hipStream_t st, long long M, const int* src, long long add_below, const float* dxw_f, const float* dwamb_f, float* dxw, float* dwamb) {
  LAUNCH(k_scatter_rows, M, st, M, src, add_below, dxw_f, dwamb_f, dxw, dwamb);
}
void gather_cols(hipStream_t st, long long M, const int* src, int rp, int C, const float* in, float* out) {
  LAUNCH(k_gather_cols, M * rp * C, st, M, src, rp, C, in, out);
}
void scatter_cols(hipStream_t st, long long M, const int* src, long long add_below, int rp, int C, const float* in_f, float* out) {
  LAUNCH(k_scatter_cols, M * rp * C, st, M, src, add_below, rp, C, in_f, out);
}
bool resample_has_sources(int Nc, int Nf) { return Nc <= 256 && Nf <= 256; }
void resample(hipStream_t st, int R, int Nc, int Nf, const float* zc, const float* wc, int stratified, const float* u_rand, uint64_t seed, long long first_ray, float* zf, float* scratch,
              float* z_new, int* src) {
  if (Nc <= 256 && Nf <= 256) hipLaunchKernelGGL(k_resample_wave, dim3(R), dim3(64), 0, st, R, Nc, Nf, zc, wc, stratified, u_rand, seed, first_ray, zf, z_new, src);
  else hipLaunchKernelGGL(k_resample, grid1(R, 64), dim3(64), 0, st, R, Nc, Nf, zc, wc, stratified, u_rand, seed, first_ray, zf, scratch);
}
void encode_inputs(hipStream_t st, const Dims& D, int R, int S, const float* o, const float* d, const float* z, const uint32_t* warp_id, int n_embeds,
                   const float* warp_tbl, const float* mask_tbl, const Windows& W, float* x, float* mask_in, float* warp_in, float* hyper_in) {
  const int wmax = D.mask_in > D.warp_ld ? (D.mask_in > D.hyper_ld ? D.mask_in : D.hyper_ld) : (D.warp_ld > D.hyper_ld ? D.warp_ld : D.hyper_ld);
  hipLaunchKernelGGL(k_encode_inputs, grid1((long long)R * S, TILE_ROWS), dim3(TILE_ROWS), tile_bytes(wmax), st, D, R, S, o, d, z, warp_id, n_embeds, warp_tbl, mask_tbl,
                     W, x, mask_in, warp_in, hyper_in);
}
void bias_act(hipStream_t st, float* y, const float* b, long long M, int N, int ld, int relu) { LAUNCH(k_bias_act, M * N, st, y, b, M, N, ld, relu); }
void mask_post(hipStream_t st, const Dims& D, int R, int S, const float* logit, const float* gt, float ratio, float* warp_in, float* hyper_in) {
  LAUNCH(k_mask_post, (long long)R * S, st, D, R, S, logit, gt, ratio, warp_in, hyper_in);
}
void se3_fwd(hipStream_t st, long long M, const float* wv, const float* x, float* xw) { LAUNCH(k_se3_fwd, M, st, M, wv, x, xw); }
void se3_bwd(hipStream_t st, long long M, const float* wv, const float* x, const float* dxw, const float* dwv_extra, float* dwv) {
  LAUNCH(k_se3_bwd, M, st, M, wv, x, dxw, dwv_extra, dwv);
}
void trunk_in(hipStream_t st, const Dims& D, long long M, const float* xw, const float* wamb, const Windows& W, float* tin) {
  hipLaunchKernelGGL(k_trunk_in, grid1(M, TILE_ROWS), dim3(TILE_ROWS), tile_bytes(D.trunk_in), st, D, M, xw, wamb, W, tin);
}
void trunk_in_bwd(hipStream_t st, const Dims& D, long long M, const float* dtin, const float* xw, const float* wamb, const Windows& W,
                  const float* dxw_extra, const float* dwamb_extra, float* dxw, float* dwamb) {
  hipLaunchKernelGGL(k_trunk_in_bwd, grid1(M, TILE_ROWS), dim3(TILE_ROWS), tile_bytes(D.trunk_in), st, D, M, dtin, xw, wamb, W, dxw_extra, dwamb_extra, dxw, dwamb);
}
void norm_loss(hipStream_t st, int R, int S, float weight, const float* weights, const float* alpha, const float* t_alpha, const float* wv,
               const float* target_norm, float* term, float* d_alpha, float* d_t_alpha, float* du, float* ghat) {
  LAUNCH(k_norm_loss, (long long)R * S, st, R, S, weight, weights, alpha, t_alpha, wv, target_norm, term, d_alpha, d_t_alpha, du, ghat);
}
void trunk_in_jvp_bwd(hipStream_t st, const Dims& D, long long M, const float* d_t_tin, const float* xw, const float* wamb, const float* t_xw,
                      const float* t_wamb, const Windows& W, float* d_t_xw, float* d_t_wamb, float* dxw_extra, float* dwamb_extra, int rp) {
  LAUNCH(k_trunk_in_jvp_bwd, 5 * M, st, D, M, d_t_tin, xw, wamb, t_xw, t_wamb, W, d_t_xw, d_t_wamb, dxw_extra, dwamb_extra, rp);
}
void se3_jvp_bwd(hipStream_t st, long long M, const float* wv, const float* x, const float* t_wv, const float* d_t_xw, const float* du,
                 const float* ghat, float* d_t_wv, float* dwv_extra, const float* extra_in, const float* dir) {
  LAUNCH(k_se3_jvp_bwd, 6 * M, st, M, wv, x, t_wv, d_t_xw, du, ghat, d_t_wv, dwv_extra, extra_in, dir);
}
// ---- the reverse-mode second-order path (round 5, nerfds_train.cpp run_merged_full `rev`): d sigma_raw / d x by ONE reverse pass per sample (the
// networks' data-gradient chains with the cotangent e_sigma), the norm loss's cotangent c = d L / d (grad_x sigma) from it, and then - forward over
// reverse - the tangent pass along the ONE direction c per sample and its backward: grad_theta <c, grad_x sigma> = grad_theta D_c sigma.  A third of
// the tangent rows of the three-unit-direction scheme.  The element-wise pieces between the chains:
// e_sigma as head cotangents [M][4] (scaled), or any constant first column
__global__ void k_fill_head4(long long M, const float* __restrict__ scale_dev, float value, float* __restrict__ out) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  *reinterpret_cast<float4*>(out + 4 * m) = make_float4(scale_dev != nullptr ? value * scale_dev[0] : value, 0.f, 0.f, 0.f);
}
// gx[m][j] = sum over the posenc columns g with g % 3 == j of d_in[m][g] * d posenc_g / d x_j, warp field input + hyper sheet input (the transpose of
// k_encode_tangents: same columns, same factors; the GLO / mask columns do not depend on x)
__global__ void k_posenc_rev_x(Dims D, long long M, const float* __restrict__ x, const float* __restrict__ d_warp_in, const float* __restrict__ d_hyper_in,
                               Windows W, float* __restrict__ gx) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= 3 * M) return;
  const long long m = i / 3;
  const int j = (int)(i - 3 * m);
  const float p[3] = {x[3 * m], x[3 * m + 1], x[3 * m + 2]};
  float acc = 0.f;
  for (int g = j; g < 6 * D.warp_bands; g += 3) acc += d_warp_in[m * D.warp_ld + g] * posenc_dval<3>(g, p, W.warp);
  for (int g = j; g < 6 * D.hyp_bands; g += 3) acc += d_hyper_in[m * D.hyper_ld + g] * posenc_dval<3>(g, p, W.hyp);
  gx[i] = acc;
}
// grad_x sigma = R^T a + gx (a = d sigma / d x', gx = the part through the two networks' inputs), written where the tangent pass leaves it: column 0 of
// rows 3 m + j of t_alpha [3 M][4] (k_target_norm / k_norm_loss read it there)
__global__ void k_sigma_grad_assemble(long long M, const float* __restrict__ wv, const float* __restrict__ a, const float* __restrict__ gx, float* __restrict__ t_alpha) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float w[3] = {wv[6 * m], wv[6 * m + 1], wv[6 * m + 2]}, v[3] = {wv[6 * m + 3], wv[6 * m + 4], wv[6 * m + 5]};
  float Rm[9], p[3];
  se3_Rp<float>(w, v, Rm, p);
  for (int j = 0; j < 3; ++j) {
    const float g = Rm[j] * a[3 * m] + Rm[3 + j] * a[3 * m + 1] + Rm[6 + j] * a[3 * m + 2] + gx[3 * m + j];
    *reinterpret_cast<float4*>(t_alpha + (3 * m + j) * 4) = make_float4(g, 0.f, 0.f, 0.f);
  }
}
// the direction of the tangent pass: dir[m][j] = K * d_t_alpha[3 m + j][0] (K = slot[1], a power of two picked from the largest cotangent: the pass is
// linear in the direction, and the f16 stores of its hidden tangents want O(1) values), and the cotangent of its head, [1 / K, 0, 0, 0] per sample
__global__ void k_make_dir(long long M, const float* __restrict__ d_t_alpha, const float* __restrict__ slot, float* __restrict__ dir, float* __restrict__ cot) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float K = slot[1], inv = slot[2];
  for (int j = 0; j < 3; ++j) dir[3 * m + j] = K * d_t_alpha[(3 * m + j) * 4];
  *reinterpret_cast<float4*>(cot + 4 * m) = make_float4(inv, 0.f, 0.f, 0.f);
}
// k_encode_tangents along ONE direction per sample: t_in[m][g] = d posenc_g / d x_{g % 3} * dir[m][g % 3]
__global__ void k_encode_tangent_dir(Dims D, long long M, const float* __restrict__ x, const float* __restrict__ dir, Windows W, float* __restrict__ t_warp_in,
                                     float* __restrict__ t_hyper_in) {
  const int LD = D.warp_ld > D.hyper_ld ? D.warp_ld : D.hyper_ld;
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= M * LD) return;
  const long long m = i / LD;
  const int g = (int)(i - m * LD);
  const float p[3] = {x[3 * m], x[3 * m + 1], x[3 * m + 2]};
  const float dj = dir[3 * m + g % 3];
  if (g < D.warp_ld) t_warp_in[m * D.warp_ld + g] = (g < 6 * D.warp_bands) ? posenc_dval<3>(g, p, W.warp) * dj : 0.f;
  if (g < D.hyper_ld) t_hyper_in[m * D.hyper_ld + g] = (g < 6 * D.hyp_bands) ? posenc_dval<3>(g, p, W.hyp) * dj : 0.f;
}
// k_se3_jvp along ONE direction per sample: t_xw = R dir + (d x' / d (w, v)) t_wv
__global__ __launch_bounds__(256) void k_se3_jvp_dir(long long M, const float* __restrict__ wv, const float* __restrict__ x, const float* __restrict__ dir,
                                                     const float* __restrict__ t_wv, float* __restrict__ t_xw) {
  const long long m = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (m >= M) return;
  Dual w[3], v[3];
  for (int i = 0; i < 3; ++i) { w[i] = dconst(wv[6 * m + i]); w[i].g[i] = 1.f; v[i] = dconst(wv[6 * m + 3 + i]); v[i].g[3 + i] = 1.f; }
  Dual Rm[9], p[3];
  se3_Rp<Dual>(w, v, Rm, p);
  for (int r = 0; r < 3; ++r) {
    const Dual xr = Rm[3 * r] * x[3 * m] + Rm[3 * r + 1] * x[3 * m + 1] + Rm[3 * r + 2] * x[3 * m + 2] + p[r];
    float acc = Rm[3 * r].v * dir[3 * m] + Rm[3 * r + 1].v * dir[3 * m + 1] + Rm[3 * r + 2].v * dir[3 * m + 2];
    for (int i = 0; i < 6; ++i) acc += xr.g[i] * t_wv[6 * m + i];
    t_xw[3 * m + r] = acc;
  }
}
void fill_head4(hipStream_t st, long long M, const float* scale_dev, float value, float* out) { LAUNCH(k_fill_head4, M, st, M, scale_dev, value, out); }
void posenc_rev_x(hipStream_t st, const Dims& D, long long M, const float* x, const float* d_warp_in, const float* d_hyper_in, const Windows& W, float* gx) {
  LAUNCH(k_posenc_rev_x, 3 * M, st, D, M, x, d_warp_in, d_hyper_in, W, gx);
}
void sigma_grad_assemble(hipStream_t st, long long M, const float* wv, const float* a, const float* gx, float* t_alpha) {
  LAUNCH(k_sigma_grad_assemble, M, st, M, wv, a, gx, t_alpha);
}
void make_dir(hipStream_t st, long long M, const float* d_t_alpha, const float* slot, float* dir, float* cot) { LAUNCH(k_make_dir, M, st, M, d_t_alpha, slot, dir, cot); }
void encode_tangent_dir(hipStream_t st, const Dims& D, long long M, const float* x, const float* dir, const Windows& W, float* t_warp_in, float* t_hyper_in) {
  const int LD = D.warp_ld > D.hyper_ld ? D.warp_ld : D.hyper_ld;
  LAUNCH(k_encode_tangent_dir, M * LD, st, D, M, x, dir, W, t_warp_in, t_hyper_in);
}
void se3_jvp_dir(hipStream_t st, long long M, const float* wv, const float* x, const float* dir, const float* t_wv, float* t_xw) {
  LAUNCH(k_se3_jvp_dir, M, st, M, wv, x, dir, t_wv, t_xw);
}
void se3_rot_bwd(hipStream_t st, long long M, const float* wv, const float* du, const float* ghat, float* out) {
  LAUNCH(k_se3_rot_bwd, 6 * M, st, M, wv, du, ghat, out);
}
void aux_losses(hipStream_t st, int R, int S, const Objective& ob, const float* z, const float* weights, const float* x, const float* xw,
                const float* alpha, const float* viewdirs, const float* mask_logit, const float* gt_mask, float* terms, float* dxw_reg,
                float* d_alpha, float* d_pm, const float* wamb, float* term_hyper, float* dwamb_reg, float* term_occlusion) {
  hipLaunchKernelGGL(k_aux_losses, grid1(R, 64), dim3(64), 0, st, R, S, ob, z, weights, x, xw, alpha, viewdirs, mask_logit, gt_mask, terms, dxw_reg,
                     d_alpha, d_pm, wamb, term_hyper, dwamb_reg, term_occlusion);
}
__global__ void k_add_inplace(float* __restrict__ dst, const float* __restrict__ src, long long n) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
void add_inplace(hipStream_t st, float* dst, const float* src, long long n) { LAUNCH(k_add_inplace, n, st, dst, src, n); }
__global__ void k_expand_half(const uint16_t* __restrict__ h, float* __restrict__ out, long long n8) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= n8) return;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  const h8 v = *reinterpret_cast<const h8*>(h + 8 * i);
  float4 a = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]), b = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
  *reinterpret_cast<float4*>(out + 8 * i) = a;
  *reinterpret_cast<float4*>(out + 8 * i + 4) = b;
}
// Loss scale of a tangent cotangent array, picked ON THE DEVICE (no host round trip in the middle of a step): slot = {amax bits, scale, 1 / scale,
// extra} with scale = 2^(target_log2 - floor(log2 amax)) - the cotangents of the tangent pass (norm loss, elastic regulariser) have no a-priori
// size the way the rgb loss's 2 / (3 R) has (nerfds_train.cpp g_scale).  amax == 0 or not finite: scale 1 (a non-finite cotangent reaches the
// gradient check of the update as it is).  slot[3] = 1 / (scale * x_scale): what the weight-gradient kernels multiply with when their X operand
// carries x_scale (WgradArgs::out_scale_dev).
__global__ __launch_bounds__(256) void k_amax(const float* __restrict__ x, long long n, unsigned* __restrict__ slot) {
  __shared__ float red[4];
  float a = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = fabsf(x[i]);
    if (!(v == v)) v = __uint_as_float(0x7f800000u);                     // NaN counts as inf
    a = fmaxf(a, v);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) a = fmaxf(a, __shfl_xor(a, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (a > 0.f) atomicMax(slot, __float_as_uint(a));                    // non-negative floats order like their bits; one atomic per block
  }
}
__global__ void k_pick_scale(float* __restrict__ slot, float target_log2, float x_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float a = slot[0];
  float sc = 1.f;
  if (a > 0.f && a < 3.0e38f) {
    int e;
    (void)frexpf(a, &e);                                                   // a = m 2^e, m in [0.5, 1)
    float k = target_log2 - (float)(e - 1);
    k = fminf(fmaxf(k, -60.f), 60.f);
    sc = exp2f(k);
  }
  slot[1] = sc; slot[2] = 1.f / sc; slot[3] = 1.f / (sc * x_scale);
}
void pick_scale(hipStream_t st, const float* x, long long n, float target_log2, float x_scale, float* slot) {
  (void)hipMemsetAsync(slot, 0, 4 * sizeof(float), st);
  const long long blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_amax, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st, x, n, reinterpret_cast<unsigned*>(slot));
  hipLaunchKernelGGL(k_pick_scale, dim3(1), dim3(1), 0, st, slot, target_log2, x_scale);
}
void scan_half(hipStream_t st, const uint16_t* p, long long n, unsigned* flags, unsigned bit) {      // n a multiple of 8, p 16-byte aligned
  const long long n8 = n / 8, blocks = (n8 + 255) / 256;
  if (n8 > 0) hipLaunchKernelGGL(k_scan_half, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, p, n8, flags, bit);
}
void scan_float(hipStream_t st, const float* p, long long n, unsigned* flags, unsigned bit, float limit) {
  const long long blocks = (n + 255) / 256;
  if (n > 0) hipLaunchKernelGGL(k_scan_float, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, p, n, flags, bit, limit);
}
void expand_half(hipStream_t st, const uint16_t* h16, float* out, long long n) { LAUNCH(k_expand_half, n / 8, st, h16, out, n / 8); }
// Singular values / vectors of a 3 x 3 matrix through the eigen-decomposition of J^T J (cyclic Jacobi in double: the warp Jacobian is close to a
// rotation, its singular values close to each other - any basis of a (near-)degenerate eigenspace gives the same sum over i of f'(s_i) u_i v_i^T).
__global__ void k_elastic_loss(int R, int S, float weight, int by_weight, const float* __restrict__ weights, const float* __restrict__ t_xw,
                               float* __restrict__ term, float* __restrict__ d_t_xw) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float* w = weights + (size_t)r * S;
  int med = 0;
  {
    float cum = 0.f;
    for (int s = 0; s < S; ++s) { cum += w[s]; if (cum >= 0.5f) { med = s; break; } }      // model_utils.py:272-299 (no sample reaches 0.5: index 0)
  }
  float total = 0.f;
  const float k = weight / (float)R;
  for (int s = by_weight ? 0 : med; s < (by_weight ? S : med + 1); ++s) {
    const size_t m = (size_t)r * S + s;
    double J[3][3], A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int j = 0; j < 3; ++j)
      for (int c = 0; c < 3; ++c) J[c][j] = (double)t_xw[(3 * m + j) * 3 + c];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) A[a][b] = J[0][a] * J[0][b] + J[1][a] * J[1][b] + J[2][a] * J[2][b];
    for (int sweep = 0; sweep < 12; ++sweep)
      for (int p = 0; p < 2; ++p)
        for (int q = p + 1; q < 3; ++q) {
          if (fabs(A[p][q]) < 1e-300) continue;
          const double th = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
          const double tt = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0)), cc = 1.0 / sqrt(tt * tt + 1.0), ss = tt * cc;
          for (int i = 0; i < 3; ++i) { const double x = A[i][p], y = A[i][q]; A[i][p] = cc * x - ss * y; A[i][q] = ss * x + cc * y; }
          for (int i = 0; i < 3; ++i) { const double x = A[p][i], y = A[q][i]; A[p][i] = cc * x - ss * y; A[q][i] = ss * x + cc * y; }
          for (int i = 0; i < 3; ++i) { const double x = V[i][p], y = V[i][q]; V[i][p] = cc * x - ss * y; V[i][q] = ss * x + cc * y; }
        }
    double sq = 0.0, coef[3];
    for (int i = 0; i < 3; ++i) {
      const double sv = sqrt(fmax(A[i][i], 0.0));
      const double ls = log(fmax(sv, 1e-6));
      sq += ls * ls;
      coef[i] = sv > 1e-6 ? 2.0 * ls / (sv * sv) : 0.0;            // f'(s_i) / s_i with u_i = J v_i / s_i; the clamped branch is constant
    }
    float dl;
    const float l = general_loss_sq((float)sq, -2.0f, 0.03f, dl);
    const float f = by_weight ? w[s] : 1.0f;
    total += f * l;
    // d sq / d J = sum_i coef_i (J v_i) v_i^T
    for (int c = 0; c < 3; ++c)
      for (int j = 0; j < 3; ++j) {
        double g = 0.0;
        for (int i = 0; i < 3; ++i) g += coef[i] * (J[c][0] * V[0][i] + J[c][1] * V[1][i] + J[c][2] * V[2][i]) * V[j][i];
        d_t_xw[(3 * m + j) * 3 + c] += k * f * dl * (float)g;
      }
  }
  atomicAdd(term, k * total);
}
void elastic_loss(hipStream_t st, int R, int S, float weight, int by_weight, const float* weights, const float* t_xw, float* term, float* d_t_xw) {
  hipLaunchKernelGGL(k_elastic_loss, grid1(R, 64), dim3(64), 0, st, R, S, weight, by_weight, weights, t_xw, term, d_t_xw);
}
__global__ void k_background_loss(long long B, const float* __restrict__ x, const float* __restrict__ xw, float weight, float alpha, float scale,
                                  float* __restrict__ term, float* __restrict__ dxw) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  float l = 0.f;
  if (i < B) {
    float dv[3], sq = 0.f;
    for (int c = 0; c < 3; ++c) { dv[c] = xw[3 * i + c] - x[3 * i + c]; sq += dv[c] * dv[c]; }
    float dl;
    l = general_loss_sq(sq, alpha, scale, dl) * weight / (float)B;
    for (int c = 0; c < 3; ++c) dxw[3 * i + c] = weight / (float)B * dl * 2.0f * dv[c];
  }
  // one atomic per wave
  for (int d = 32; d >= 1; d >>= 1) l += __shfl_xor(l, d, 64);
  if ((threadIdx.x & 63) == 0 && l != 0.f) atomicAdd(term, l);
}
void background_loss(hipStream_t st, long long B, const float* x, const float* xw, float weight, float alpha, float scale, float* term, float* dxw) {
  LAUNCH(k_background_loss, B, st, B, x, xw, weight, alpha, scale, term, dxw);
}
void alpha_post(hipStream_t st, const Dims& D, int R, int S, const float* alpha, const float* wv, const float* viewdirs, const Windows& W, float* sigma, float* cond) {
  hipLaunchKernelGGL(k_alpha_post, grid1((long long)R * S, TILE_ROWS), dim3(TILE_ROWS), tile_bytes(6 * D.vd_bands + 6 * D.nm_bands), st, D, R, S, alpha, wv, viewdirs, W,
                     sigma, cond);
}
void composite_loss(hipStream_t st, int R, int S, const float* z, const float* dirs, const float* sigma, const float* rgb_logit, const float* target,
                    int at_infinity, int white, float* rgb_ray, float* weights, float* loss, float* d_rgb_logit, float* d_alpha, const LevelCot& cot,
                    const LevelOut& out) {
  hipLaunchKernelGGL(k_composite_loss, dim3((R + 3) / 4), dim3(256), 0, st, R, S, z, dirs, sigma, rgb_logit, target, at_infinity, white, rgb_ray, weights,
                     loss, d_rgb_logit, d_alpha, cot, out);
  if (target != nullptr) hipLaunchKernelGGL(k_mse_sum, dim3(1), dim3(256), 0, st, R, rgb_ray, target, loss);
}
void relu_bwd(hipStream_t st, float* dy, const float* y, long long n) { LAUNCH(k_relu_bwd, n, st, dy, y, n); }
void colsum_add(hipStream_t st, const float* dz, long long M, int N, int ld, float* db) {      // N <= 256
  hipLaunchKernelGGL(k_colsum_add, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, dz, M, N, ld, db);
}
void relu_bwd_colsum(hipStream_t st, float* dy, const float* y, long long M, int N, float* db) {      // N <= 256, contiguous rows
  hipLaunchKernelGGL(k_relu_bwd_colsum, dim3((unsigned)((M + 127) / 128)), dim3(256), 0, st, dy, y, M, N, db);
}
void shared_in_bwd(hipStream_t st, const Dims& D, int R, int S, const float* d_warp_in, const float* d_hyper_in, const float* mask_logit, float ratio,
                   const float* d_pm_extra, const uint32_t* warp_id, int n_embeds, float* d_warp_tbl, float* d_mask_logit) {
  hipLaunchKernelGGL(k_shared_in_bwd, grid1((long long)R * S), dim3(256), 256 * 8 * sizeof(float), st, D, R, S, d_warp_in, d_hyper_in, mask_logit, ratio, d_pm_extra, warp_id, n_embeds,
                     d_warp_tbl, d_mask_logit);
}
void mask_in_bwd(hipStream_t st, const Dims& D, int R, int S, const float* d_mask_in, const uint32_t* warp_id, int n_embeds, float* d_mask_tbl) {
  hipLaunchKernelGGL(k_mask_in_bwd, grid1((long long)R * S), dim3(256), 256 * 8 * sizeof(float), st, D, R, S, d_mask_in, warp_id, n_embeds, d_mask_tbl);
}
void sum_partials(hipStream_t st, const float* part, int slabs, long long n, float* out) { LAUNCH(k_sum_partials, n, st, part, slabs, n, out); }
void fill(hipStream_t st, float* p, long long n, float v) { LAUNCH(k_fill, n, st, p, n, v); }
void encode_tangents(hipStream_t st, const Dims& D, long long M, const float* x, const Windows& W, float* t_warp_in, float* t_hyper_in) {
  LAUNCH(k_encode_tangents, M * (D.warp_ld > D.hyper_ld ? D.warp_ld : D.hyper_ld), st, D, M, x, W, t_warp_in, t_hyper_in);
}
void relu_mask3(hipStream_t st, float* t, const float* y, long long M, int N) { LAUNCH(k_relu_mask3, 3 * M * N, st, t, y, M, N); }
void se3_jvp(hipStream_t st, long long M, const float* wv, const float* x, const float* t_wv, float* t_xw) { LAUNCH(k_se3_jvp, M, st, M, wv, x, t_wv, t_xw); }
void trunk_in_jvp(hipStream_t st, const Dims& D, long long M, const float* xw, const float* wamb, const float* t_xw, const float* t_wamb,
                  const Windows& W, float* t_tin, int rp) {
  LAUNCH(k_trunk_in_jvp, M * D.trunk_in, st, D, M, xw, wamb, t_xw, t_wamb, W, t_tin, rp);
}
void target_norm(hipStream_t st, long long M, const float* t_alpha, const float* wv, float* out) { LAUNCH(k_target_norm, M, st, M, t_alpha, wv, out); }
void clip_gradients(hipStream_t st, float* g, long long n, float max_val, float max_norm, float* sumsq_scratch) {
  (void)hipMemsetAsync(sumsq_scratch, 0, sizeof(float), st);
  LAUNCH(k_clip_val_sumsq, n, st, g, n, max_val, sumsq_scratch);
  if (max_norm > 0.f) LAUNCH(k_clip_norm, n, st, g, n, max_norm, 1e-7f, sumsq_scratch);
}
void adam(hipStream_t st, float* p, const float* g, float* m1, float* m2, long long n, float lr, float b1, float b2, float eps, long long* step_dev,
          float* corr_dev, unsigned* nonfinite_flag) {
  LAUNCH(k_nonfinite, n, st, g, n, nonfinite_flag);
  hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(1), 0, st, nonfinite_flag, step_dev, corr_dev, (double)b1, (double)b2);
  LAUNCH(k_adam, n, st, p, g, m1, m2, n, lr, b1, b2, eps, corr_dev, nonfinite_flag);
}

// ------------------------------------------------------------------------------------------------
// Weight streams of the fused forward (train_fwd_kernel.hip) from the CURRENT parameters, every step.
// The host packs the streams once with the parameters replaced by their own indices (nerfds_train.cpp build_fused_forward):
// map[i] = 1 + index into theta (or, past P, into `fold`) of the value at float slot i of the fp32-layout stream, 0 = padding.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float map_value(const float* theta, const float* fold, long long P, int m) {
  if (m == 0) return 0.f;
  const long long i = (long long)m - 1;
  return i < P ? theta[i] : fold[i - P];
}
__device__ __forceinline__ unsigned short bf16_rne(float f) {      // pack.h f32_to_bf16_rne
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
// One thread per (fragment, lane): 8 k-slots of one output row.  The map is in the two-unit layout (2 KiB per fragment); the stream
// holds split bf16 - hi | lo units - except the fragments [x6_lo, x6_hi) (the warp field's forward): exact fp32 (wide_f32), or split three
// ways (hi | mid | lo units, graphs.h P_BF16X6) - then every fragment behind x6_lo starts one unit later per three-way fragment before it.
__device__ __forceinline__ void pack_stream_item(const float* __restrict__ theta, const float* __restrict__ fold, long long P, const int* __restrict__ map,
                                                 unsigned char* __restrict__ stream, int nfrag, int x6_lo, int x6_hi, int wide_f32, long long tid) {
  if (tid >= (long long)nfrag * 64) return;
  const int frag = (int)(tid >> 6), lane = (int)(tid & 63);
  const int4 ma = *reinterpret_cast<const int4*>(map + (size_t)frag * 512 + lane * 4);
  const int4 mb = *reinterpret_cast<const int4*>(map + (size_t)frag * 512 + 256 + lane * 4);
  const int mi[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = map_value(theta, fold, P, mi[i]);
  if (wide_f32 == 2) {                                    // ONE f16 unit per fragment (graphs.h P_F16): the tangent pass's backward chains
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 hv;
#pragma unroll
    for (int i = 0; i < 8; ++i) hv[i] = (_Float16)v[i];
    *reinterpret_cast<h8*>(stream + (size_t)frag * 1024 + lane * 16) = hv;
    return;
  }
  if (wide_f32 && frag >= x6_lo && frag < x6_hi) {        // the range keeps exact fp32: k-slots 0-3 | 4-7 (graphs.h P_F32), two units
    unsigned char* ff = stream + (size_t)frag * 2048 + lane * 16;
    *reinterpret_cast<float4*>(ff) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(ff + 1024) = make_float4(v[4], v[5], v[6], v[7]);
    return;
  }
  const int extra = wide_f32 ? 0 : (frag < x6_lo ? 0 : (frag < x6_hi ? frag - x6_lo : x6_hi - x6_lo));       // three-way fragments before this one
  unsigned char* fa = stream + ((size_t)frag * 2 + extra) * 1024 + lane * 16;
  const bool x6 = frag >= x6_lo && frag < x6_hi;
  unsigned hi[4], mid[4], lo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned short h[2], m[2], l[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float x = v[2 * i + e];
      h[e] = bf16_rne(x);
      const float r = x - __uint_as_float((unsigned)h[e] << 16);
      m[e] = bf16_rne(r);
      l[e] = bf16_rne(r - __uint_as_float((unsigned)m[e] << 16));
    }
    hi[i] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    mid[i] = (unsigned)m[0] | ((unsigned)m[1] << 16);
    lo[i] = (unsigned)l[0] | ((unsigned)l[1] << 16);
  }
  *reinterpret_cast<uint4*>(fa) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  *reinterpret_cast<uint4*>(fa + 1024) = make_uint4(mid[0], mid[1], mid[2], mid[3]);          // two-way: this is the "lo" unit
  if (x6) *reinterpret_cast<uint4*>(fa + 2048) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
// Every stream (and bias vector) of a step's packing in ONE launch: blockIdx.y picks the item (train_kernels.h PackBatch; at 512 rays the ~20 separate
// five-microsecond launches were 4 % of the step).  mode 0 / 1 / 2: pack_stream_item's wide_f32; 3: a bias vector of n values.
__global__ void k_pack_batch(const float* __restrict__ theta, const float* __restrict__ fold, long long P, const PackBatch B) {
  const PackItem I = B.it[blockIdx.y];
  const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (I.mode == 3) {
    if (tid < I.n) static_cast<float*>(I.out)[tid] = map_value(theta, fold, P, I.map[tid]);
    return;
  }
  pack_stream_item(theta, fold, P, I.map, static_cast<unsigned char*>(I.out), I.n, I.x6_lo, I.x6_hi, I.mode, tid);
}
// The activation-free bottleneck Dense folded into rgb hidden_0 (pack.h pack_nerf; modules.py:255, 296-310), in double like the host packer:
//   fold[r][c] = K[row_x + r][c] + sum_k B[r][k] K[k][c]   (r < TW),   fold[TW][c] = Kb[c] + sum_k Bb[k] K[k][c]
__global__ void k_fold_rgb(const float* __restrict__ B, const float* __restrict__ Bb, const float* __restrict__ K, const float* __restrict__ Kb,
                           int TW, int W, int row_x, float* __restrict__ fold) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (TW + 1) * W) return;
  const int r = i / W, c = i % W;
  double acc = r < TW ? (double)K[(size_t)(row_x + r) * W + c] : (double)Kb[c];
  const float* left = r < TW ? B + (size_t)r * TW : Bb;
  // (one accumulator, the order of the host packer; unrolled so that the loads of 16 steps are in flight together - as a rolled loop of dependent
  // L2 round trips the launch took 61 us, at the head of every step's critical path)
#ifndef NERFDS_EXP_FOLD_ROLLED      // (measurement build: the loop as it was)
#pragma unroll 16
#endif
  for (int k = 0; k < TW; ++k) acc += (double)left[k] * (double)K[(size_t)k * W + c];
  fold[i] = (float)acc;
}
// fused backward: see train_kernels.h bott_grads.  Three small products with double accumulation like k_fold_rgb:
//   dKb[k][n] = sum_r Wb[r][k] S[r][n] + bb[k] c[n],   dWb[r][k] = sum_n S[r][n] Kb[k][n],   dbb[k] = sum_n Kb[k][n] c[n].
// small_gemm_tile: C[i][j] = sum_r A(r, i) B(r, j) (+ u[i] v[j]) for any strides, 32 x 32 output tiles, the operands staged through LDS 32
// reduction steps at a time (as one thread per output element walking 256 operand pairs from L2 the launch took 90 us).
__device__ __forceinline__ void small_gemm_tile(float (&As)[32][33], float (&Bs)[32][33], int I, int J, int Rn, const float* __restrict__ A, long long sa_r, long long sa_i,
                                                const float* __restrict__ B, long long sb_r, long long sb_j, const float* __restrict__ u,
                                                const float* __restrict__ v, float* __restrict__ C) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  double acc[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  for (int r0 = 0; r0 < Rn; r0 += 32) {
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += 256) {
      // the faster-moving index of the load follows the smaller stride of each operand
      const int p = e & 31, q = e >> 5;
      const int ra = sa_i <= sa_r ? q : p, ia = sa_i <= sa_r ? p : q;
      As[ra][ia] = (r0 + ra < Rn && i0 + ia < I) ? A[(r0 + ra) * sa_r + (i0 + ia) * sa_i] : 0.f;
      const int rb = sb_j <= sb_r ? q : p, jb = sb_j <= sb_r ? p : q;
      Bs[rb][jb] = (r0 + rb < Rn && j0 + jb < J) ? B[(r0 + rb) * sb_r + (j0 + jb) * sb_j] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const double a0 = As[r][ty], a1 = As[r][ty + 16], b0 = Bs[r][tx], b1 = Bs[r][tx + 16];
      acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[1][0] += a1 * b0; acc[1][1] += a1 * b1;
    }
  }
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i < I && j < J) C[(size_t)i * J + j] = (float)(acc[a][b] + (u ? (double)u[i] * (double)v[j] : 0.0));
    }
}
// bott_grads of every level in ONE launch (it was three per level, the levels side by side on two streams, at the serial tail of the step): blockIdx.z = 3 * level + product
//   product 0: dKb (TW x W tiles), 1: dWb (TW x TW tiles), 2: dbb (the block (0, 0) alone, one thread per k)
__global__ __launch_bounds__(256) void k_bott_grads(const BottBatch Bt) {
  __shared__ float As[32][33], Bs[32][33];
  const BottItem L = Bt.lv[blockIdx.z / 3];
  const int prod = blockIdx.z % 3, TW = L.TW, W = L.W;
  if (prod == 0) {          // i = k, j = n, reduction r: A(r, k) = Wb[r * TW + k], B(r, n) = S[r * W + n]
    if ((int)blockIdx.x * 32 >= W || (int)blockIdx.y * 32 >= TW) return;
    small_gemm_tile(As, Bs, TW, W, TW, L.Wb, (long long)TW, 1LL, L.S, (long long)W, 1LL, L.bb, L.c, L.dKb);
  } else if (prod == 1) {   // i = r, j = k, reduction n: A(n, r) = S[r * W + n], B(n, k) = K[k * W + n]
    if ((int)blockIdx.x * 32 >= TW || (int)blockIdx.y * 32 >= TW) return;
    small_gemm_tile(As, Bs, TW, TW, W, L.S, 1LL, (long long)W, L.K, 1LL, (long long)W, nullptr, nullptr, L.dWb);
  } else {
    if (blockIdx.x != 0 || blockIdx.y != 0) return;
    for (int k = threadIdx.x; k < TW; k += blockDim.x) {
      double acc = 0.0;
#pragma unroll 16
      for (int n = 0; n < W; ++n) acc += (double)L.K[(size_t)k * W + n] * (double)L.c[n];      // (loads of 16 steps in flight: k_fold_rgb)
      L.dbb[k] = (float)acc;
    }
  }
}
void bott_grads(hipStream_t st, const BottBatch& B) {
  if (B.n <= 0) return;
  int tw = 0, w = 0;
  for (int i = 0; i < B.n; ++i) { tw = B.lv[i].TW > tw ? B.lv[i].TW : tw; w = B.lv[i].W > w ? B.lv[i].W : w; }
  const int m = tw > w ? tw : w;
  hipLaunchKernelGGL(k_bott_grads, dim3((m + 31) / 32, (tw + 31) / 32, 3 * B.n), dim3(256), 0, st, B);
}
void pack_batch(hipStream_t st, const float* theta, const float* fold, long long P, const PackBatch& B) {
  if (B.overflow) { fprintf(stderr, "nerfds_train::pack_batch: more than %d items in one batch\n", PackBatch::MAX); abort(); }
  if (B.n <= 0) return;
  long long most = 0;
  for (int i = 0; i < B.n; ++i) { const long long th = B.it[i].mode == 3 ? B.it[i].n : (long long)B.it[i].n * 64; most = th > most ? th : most; }
  if (most <= 0) return;
  hipLaunchKernelGGL(k_pack_batch, dim3((unsigned)((most + 255) / 256), (unsigned)B.n), dim3(256), 0, st, theta, fold, P, B);
}
void fold_rgb(hipStream_t st, const float* B, const float* Bb, const float* K, const float* Kb, int TW, int W, int row_x, float* fold) {
  LAUNCH(k_fold_rgb, (long long)(TW + 1) * W, st, B, Bb, K, Kb, TW, W, row_x, fold);
}

}  // namespace nerfds_train
