// Kernel-level check + micro-benchmark of the trainer's MFMA layers (csrc/train_gemm.hip) against fp64 sums on the CPU.
// Built by csrc/Makefile as nerfds_amd/_lib/train_gemm_check; `train_gemm_check [M]` prints one line per shape and exits 1 when an
// error bound is exceeded (tests/test_train_gemm.py runs it at sample counts that give partial tiles and several tiles per workgroup).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include "train_gemm.h"
using namespace nerfds_train;

static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

int main(int argc, char** argv) {
  struct Shape { int k1, k2, n; bool relu, mask, acc, precise; };
  int bad = 0;
  const Shape shapes[] = {{128, 0, 128, true, false, false, false}, {256, 0, 256, true, false, false, false}, {256, 52, 256, true, false, false, false},
                          {128, 33, 128, true, false, false, false}, {64, 0, 64, true, false, false, false}, {256, 0, 4, false, false, false, false},
                          {256, 0, 256, false, true, false, false}, {128, 0, 128, false, true, true, false}, {536, 24, 128, true, false, false, false},
                          {128, 0, 128, true, false, false, true}, {128, 0, 3, false, false, false, true},
                          {128, 33, 128, true, false, false, true}, {33, 0, 128, true, false, false, true}};
  const long long M = argc > 1 ? atoll(argv[1]) : 524288;
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  for (const Shape& s : shapes) {
    const int K = s.k1 + s.k2;
    float *x1, *x2 = nullptr, *y, *w, *b, *mk; void* frag;
    (void)hipMalloc(&x1, M * s.k1 * 4); if (s.k2) (void)hipMalloc(&x2, M * s.k2 * 4);
    (void)hipMalloc(&y, M * s.n * 4); (void)hipMalloc(&mk, M * s.n * 4); (void)hipMalloc(&w, K * s.n * 4); (void)hipMalloc(&b, s.n * 4);
    (void)hipMalloc(&frag, frag_bytes(K, s.n, 3));
    const long long Mc = M < 100 ? M : 100;     // rows checked on the CPU, spread over the whole range (+ the last row)
    std::vector<float> hx1(M * s.k1), hx2((size_t)M * s.k2), hw(K * s.n), hb(s.n), hm(M * s.n), hy0(M * s.n);
    for (auto& v : hx1) v = frand(); for (auto& v : hx2) v = frand(); for (auto& v : hw) v = frand() * 0.1f; for (auto& v : hb) v = frand();
    for (auto& v : hm) v = frand(); for (auto& v : hy0) v = frand();
    (void)hipMemcpy(x1, hx1.data(), hx1.size() * 4, hipMemcpyHostToDevice); if (s.k2) (void)hipMemcpy(x2, hx2.data(), hx2.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(mk, hm.data(), hm.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(y, hy0.data(), hy0.size() * 4, hipMemcpyHostToDevice);
    DenseArgs A{};
    A.seg[0] = {x1, s.k1, s.k1}; A.nseg = 1; if (s.k2) { A.seg[1] = {x2, s.k2, s.k2}; A.nseg = 2; }
    A.k_total = K; A.wfrag = frag; A.bias = (s.mask || s.acc) ? nullptr : b;      // (a layer has a forward epilogue or a backward one)
    A.y = y; A.ldy = s.n; A.n_out = s.n; A.M = M; A.relu = s.relu; A.mask_y = s.mask ? mk : nullptr;
    A.ld_mask = s.n; A.mask_div = 1; A.accumulate = s.acc;
    void* zeros; (void)hipMalloc(&zeros, 256); (void)hipMemset(zeros, 0, 256); A.zeros = zeros;
    A.precise = s.precise;
    pack_frags(nullptr, w, s.n, 0, K, s.n, 0, frag, s.precise ? 3 : 2);
    if (!dense_ws(nullptr, A, prop.multiProcessorCount)) { printf("shape %d+%d -> %d not supported\n", s.k1, s.k2, s.n); continue; }
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e)); }
    std::vector<float> hy((size_t)M * s.n);
    (void)hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (long long i = 0; i <= Mc; ++i) for (int n = 0; n < s.n; ++n) {
      const long long r = i == Mc ? M - 1 : i * M / Mc;
      double a = (s.mask || s.acc) ? 0.0 : hb[n];
      for (int k = 0; k < s.k1; ++k) a += (double)hx1[r * s.k1 + k] * hw[k * s.n + n];
      for (int k = 0; k < s.k2; ++k) a += (double)hx2[r * s.k2 + k] * hw[(s.k1 + k) * s.n + n];
      if (s.relu) a = a > 0 ? a : 0;
      if (s.acc) a += hy0[r * s.n + n];                 // (the mask covers the accumulated total)
      if (s.mask && !(hm[r * s.n + n] > 0)) a = 0;
      worst = std::fmax(worst, std::fabs(a - hy[r * s.n + n]));
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) dense_ws(nullptr, A, prop.multiProcessorCount);
    (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)M * (K + s.n * (1 + (s.mask ? 1 : 0) + (s.acc ? 1 : 0))) * 4;
    const double bound = s.precise ? 3e-6 : 1e-4;         // |y| ~ 3: three-way split = fp32-level, two-way split = 2^-17 per product
    if (!(worst < bound)) ++bad;
    printf("M=%lld K=%d+%d N=%d relu=%d mask=%d acc=%d split=%d: %.3f ms  %.0f GB/s  max|err|=%.2e%s\n", M, s.k1, s.k2, s.n, s.relu, s.mask, s.acc,
           s.precise ? 3 : 2, ms, bytes / ms / 1e6, worst, worst < bound ? "" : "  FAIL");
    (void)hipFree(x1); if (x2) (void)hipFree(x2); (void)hipFree(y); (void)hipFree(mk); (void)hipFree(w); (void)hipFree(b); (void)hipFree(frag);
  }
  // ---- weight gradient dW = X^T dY ----
  struct WS { int k, n; };
  const WS wshapes[] = {{256, 256}, {128, 128}, {64, 64}, {52, 256}, {33, 128}, {128, 3}};
  for (const WS& w : wshapes) {
    float *x, *dy, *dw; void* zeros;
    const int NREP = getenv("NREP") ? atoi(getenv("NREP")) : 16;      // destination replicas, as the trainer uses them
    (void)hipMalloc(&x, M * w.k * 4); (void)hipMalloc(&dy, M * w.n * 4); (void)hipMalloc(&dw, (size_t)NREP * w.k * w.n * 4); (void)hipMalloc(&zeros, 256);
    (void)hipMemset(zeros, 0, 256); (void)hipMemset(dw, 0, (size_t)NREP * w.k * w.n * 4);
    const long long Mc = M < 4096 ? M : 4096;       // rows that carry data, spread over the whole range (the rest are zero): the CPU check stays cheap
    std::vector<float> hx(Mc * w.k), hd(Mc * w.n);
    for (auto& v : hx) v = frand(); for (auto& v : hd) v = frand();
    (void)hipMemset(x, 0, M * w.k * 4); (void)hipMemset(dy, 0, M * w.n * 4);
    for (long long j = 0; j < Mc; ++j) {
      const long long r = j * M / Mc;
      (void)hipMemcpy(x + r * w.k, hx.data() + j * w.k, w.k * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dy + r * w.n, hd.data() + j * w.n, w.n * 4, hipMemcpyHostToDevice);
    }
    WgradArgs A{x, w.k, w.k, dy, w.n, w.n, M, nullptr, dw, zeros, 0, 0, (long long)w.k * w.n, NREP};
    if (!wgrad_supported(A)) { printf("wgrad %d x %d not supported\n", w.k, w.n); continue; }
    const int grid = wgrad_grid(A, prop.multiProcessorCount);
    wgrad(nullptr, A, grid);
    (void)hipDeviceSynchronize();
    std::vector<float> hw(w.k * w.n), hrep(w.k * w.n);
    (void)hipMemcpy(hw.data(), dw, hw.size() * 4, hipMemcpyDeviceToHost);
    for (int rp = 1; rp < NREP; ++rp) {            // the caller's part: sum the replicas
      (void)hipMemcpy(hrep.data(), dw + (size_t)rp * w.k * w.n, hrep.size() * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < hw.size(); ++i) hw[i] += hrep[i];
    }
    double worst = 0, big = 0;
    for (int k = 0; k < w.k; ++k) for (int n = 0; n < w.n; ++n) {
      double a = 0;
      for (long long r = 0; r < Mc; ++r) a += (double)hx[r * w.k + k] * hd[r * w.n + n];
      worst = std::fmax(worst, std::fabs(a - hw[k * w.n + n])); big = std::fmax(big, std::fabs(a));
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < reps; ++i) wgrad(nullptr, A, grid);
    (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    if (!(worst < 3e-5 * big)) ++bad;
    printf("wgrad M=%lld K=%d N=%d: %.3f ms  %.0f GB/s  max|err|=%.2e (max|dW|=%.1f)%s\n", M, w.k, w.n, ms, (double)M * (w.k + w.n) * 4 / ms / 1e6, worst, big,
           worst < 3e-5 * big ? "" : "  FAIL");
    (void)hipFree(x); (void)hipFree(dy); (void)hipFree(dw); (void)hipFree(zeros);
  }
  // ---- weight gradient of a head: X f16 (the last hidden layer), dY fp32 with N <= 6 columns in rows of `ld` floats (k_wgrad_head), bias gradient = column sums ----
  {
    struct HD { int k, n, ld, off; };
    const HD heads[] = {{256, 4, 4, 0}, {256, 3, 3, 0}, {128, 3, 6, 3}, {128, 1, 1, 0}, {64, 2, 2, 0}, {128, 6, 6, 0}};
    for (const HD& w : heads) {
      void *x, *zeros; float *dy, *dw;
      const int NREP = 16;
      const long long RS = (long long)w.k * w.n + w.n;
      (void)hipMalloc(&x, M * w.k * 2); (void)hipMalloc(&dy, M * w.ld * 4); (void)hipMalloc(&dw, (size_t)NREP * RS * 4); (void)hipMalloc(&zeros, 256);
      (void)hipMemset(zeros, 0, 256); (void)hipMemset(dw, 0, (size_t)NREP * RS * 4); (void)hipMemset(x, 0, M * w.k * 2); (void)hipMemset(dy, 0, M * w.ld * 4);
      const long long Mc = M < 4096 ? M : 4096;
      std::vector<float> hx(Mc * w.k), hd(Mc * w.ld);
      std::vector<uint16_t> x16(w.k);
      for (long long j = 0; j < Mc; ++j) {
        const long long r = j * M / Mc;
        for (int k = 0; k < w.k; ++k) { const _Float16 h = (_Float16)frand(); memcpy(&x16[k], &h, 2); hx[j * w.k + k] = (float)h; }
        for (int n = 0; n < w.ld; ++n) hd[j * w.ld + n] = frand();
        (void)hipMemcpy((char*)x + r * w.k * 2, x16.data(), w.k * 2, hipMemcpyHostToDevice);
        (void)hipMemcpy(dy + r * w.ld, hd.data() + j * w.ld, w.ld * 4, hipMemcpyHostToDevice);
      }
      WgradArgs A{(const float*)x, w.k, w.k, dy + w.off, w.ld, w.n, M, nullptr, dw, zeros, 0, 0, RS, NREP};
      A.x_half = 1; A.colsum = dw + (size_t)w.k * w.n;
      if (!wgrad_supported(A)) { printf("wgrad head %d x %d not supported  FAIL\n", w.k, w.n); ++bad; continue; }
      const int grid = wgrad_grid(A, prop.multiProcessorCount);
      wgrad(nullptr, A, grid);
      (void)hipDeviceSynchronize();
      std::vector<float> hw(RS, 0.f), hrep(RS);
      for (int rp = 0; rp < NREP; ++rp) {
        (void)hipMemcpy(hrep.data(), dw + (size_t)rp * RS, RS * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] += hrep[i];
      }
      double worst = 0, big = 0, cworst = 0, cbig = 0;
      for (int k = 0; k < w.k; ++k) for (int n = 0; n < w.n; ++n) {
        double a = 0;
        for (long long r = 0; r < Mc; ++r) a += (double)hx[r * w.k + k] * hd[r * w.ld + w.off + n];
        worst = std::fmax(worst, std::fabs(a - hw[k * w.n + n])); big = std::fmax(big, std::fabs(a));
      }
      for (int n = 0; n < w.n; ++n) {
        double a = 0;
        for (long long r = 0; r < Mc; ++r) a += hd[r * w.ld + w.off + n];
        cworst = std::fmax(cworst, std::fabs(a - hw[w.k * w.n + n])); cbig = std::fmax(cbig, std::fabs(a));
      }
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      const int reps = 20;
      (void)hipEventRecord(e0, nullptr);
      for (int i = 0; i < reps; ++i) wgrad(nullptr, A, grid);
      (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      const bool ok = worst < 3e-5 * big && cworst < 1e-5 * (cbig + Mc * 0.01);
      if (!ok) ++bad;
      printf("wgrad head M=%lld K=%d (f16) N=%d (fp32, ld %d): %.3f ms  %.0f GB/s  max|err|=%.2e (max|dW|=%.1f) colsum err %.2e%s\n", M, w.k, w.n, w.ld, ms,
             (double)M * (w.k * 2 + w.ld * 4) / ms / 1e6, worst, big, cworst, ok ? "" : "  FAIL");
      (void)hipFree(x); (void)hipFree(dy); (void)hipFree(dw); (void)hipFree(zeros);
    }
  }
  // ---- weight gradient with 16-bit operands: X as f16 (what the fused forward stores) and / or dY as scaled f16 (what the fused backward's chains
  // store), and the bias gradient = column sums of dY as a by-product.  f16 X: both operands go to the MFMAs as stored (k_wgrad_tr, one MFMA
  // per product, exact products); fp32 X: f16 hi + lo, two MFMAs.  The reference is the fp64 sum over the ROUNDED inputs, so the bound is the
  // kernel's own arithmetic, not the storage format. ----
  {
    const float GS = 64.f;
    struct HS { int k, n, xh; };
    const HS hshapes[] = {{256, 256, 1}, {128, 128, 1}, {64, 64, 1}, {256, 128, 1}, {52, 256, 0}, {36, 128, 0}, {48, 64, 0}};
    for (const HS& w : hshapes) {
      void *x, *dy, *zeros; float *dw, *cs;
      const int NREP = 16;
      const size_t xe = w.xh ? 2 : 4;
      (void)hipMalloc(&x, M * w.k * xe); (void)hipMalloc(&dy, M * w.n * 2); (void)hipMalloc(&dw, (size_t)NREP * (w.k * w.n + w.n) * 4); (void)hipMalloc(&zeros, 256);
      cs = dw + (size_t)w.k * w.n;                              // replica layout [dW | colsum], stride k n + n
      (void)hipMemset(zeros, 0, 256); (void)hipMemset(dw, 0, (size_t)NREP * (w.k * w.n + w.n) * 4);
      (void)hipMemset(x, 0, M * w.k * xe); (void)hipMemset(dy, 0, M * w.n * 2);
      const long long Mc = M < 4096 ? M : 4096;
      std::vector<float> hx(Mc * w.k), hd(Mc * w.n);
      std::vector<uint16_t> x16(w.k), d16(w.n);
      for (long long j = 0; j < Mc; ++j) {
        const long long r = j * M / Mc;
        for (int k = 0; k < w.k; ++k) {
          float v = frand();
          if (w.xh) { const _Float16 h = (_Float16)v; memcpy(&x16[k], &h, 2); v = (float)h; }
          hx[j * w.k + k] = v;
        }
        // dY as the chains store it: f16 of (g times a power of two); the kernel multiplies its sums by out_scale = 1 / that power
        for (int n = 0; n < w.n; ++n) { const _Float16 h = (_Float16)(frand() * GS); memcpy(&d16[n], &h, 2); hd[j * w.n + n] = (float)h / GS; }
        if (w.xh) (void)hipMemcpy((char*)x + r * w.k * 2, x16.data(), w.k * 2, hipMemcpyHostToDevice);
        else (void)hipMemcpy((char*)x + r * w.k * 4, hx.data() + j * w.k, w.k * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy((char*)dy + r * w.n * 2, d16.data(), w.n * 2, hipMemcpyHostToDevice);
      }
      WgradArgs A{(const float*)x, w.k, w.k, (const float*)dy, w.n, w.n, M, nullptr, dw, zeros, 0, 0, (long long)w.k * w.n + w.n, NREP};
      A.x_half = w.xh; A.dy_half = 1; A.colsum = cs; A.out_scale = 1.f / GS;
      if (!wgrad_supported(A)) { printf("wgrad16 %d x %d not supported  FAIL\n", w.k, w.n); ++bad; continue; }
      const int grid = wgrad_grid(A, prop.multiProcessorCount);
      wgrad(nullptr, A, grid);
      (void)hipDeviceSynchronize();
      std::vector<float> hw(w.k * w.n + w.n, 0.f), hrep(w.k * w.n + w.n);
      for (int rp = 0; rp < NREP; ++rp) {
        (void)hipMemcpy(hrep.data(), dw + (size_t)rp * (w.k * w.n + w.n), hrep.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < hw.size(); ++i) hw[i] += hrep[i];
      }
      double worst = 0, big = 0, cworst = 0, cbig = 0;
      for (int k = 0; k < w.k; ++k) for (int n = 0; n < w.n; ++n) {
        double a = 0;
        for (long long r = 0; r < Mc; ++r) a += (double)hx[r * w.k + k] * hd[r * w.n + n];
        worst = std::fmax(worst, std::fabs(a - hw[k * w.n + n])); big = std::fmax(big, std::fabs(a));
      }
      for (int n = 0; n < w.n; ++n) {
        double a = 0;
        for (long long r = 0; r < Mc; ++r) a += hd[r * w.n + n];
        cworst = std::fmax(cworst, std::fabs(a - hw[w.k * w.n + n])); cbig = std::fmax(cbig, std::fabs(a));
      }
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      const int reps = 20;
      (void)hipEventRecord(e0, nullptr);
      for (int i = 0; i < reps; ++i) wgrad(nullptr, A, grid);
      (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
      const bool ok = worst < 3e-5 * big && cworst < 1e-5 * (cbig + Mc * 0.01);
      if (!ok) ++bad;
      printf("wgrad16 M=%lld K=%d (%s) N=%d (f16): %.3f ms  %.0f GB/s  max|err|=%.2e (max|dW|=%.1f) colsum err %.2e%s\n", M, w.k, w.xh ? "f16" : "f32", w.n, ms,
             (double)M * (w.k * xe + w.n * 2) / ms / 1e6, worst, big, cworst, ok ? "" : "  FAIL");
      (void)hipFree(x); (void)hipFree(dy); (void)hipFree(dw); (void)hipFree(zeros);
    }
  }
  // ---- fused backward of a narrow layer: dW += X^T dZ and dX = (dZ . W^T) . 1[X > 0] with its column sums ----
  struct FS { int k, n; };
  const FS fshapes[] = {{128, 128}, {64, 64}, {128, 64}};
  for (const FS& w : fshapes) {
    float *x, *dz, *dx, *dw, *wt, *cs; void *zeros, *frag;
    (void)hipMalloc(&x, M * w.k * 4); (void)hipMalloc(&dz, M * w.n * 4); (void)hipMalloc(&dx, M * w.k * 4); (void)hipMalloc(&dw, (size_t)16 * (w.k * w.n + w.k) * 4);
    const int NREP = getenv("NREP") ? atoi(getenv("NREP")) : 16;      // destination replicas, as the trainer uses them
    const long long RS = (long long)w.k * w.n + w.k;              // replica stride: [dW | colsum]
    (void)hipMalloc(&wt, w.k * w.n * 4); (void)hipMalloc(&cs, w.k * 4); (void)hipMalloc(&zeros, 256); (void)hipMalloc(&frag, frag_bytes(w.n, w.k));
    (void)hipMemset(zeros, 0, 256); (void)hipMemset(dw, 0, (size_t)16 * (w.k * w.n + w.k) * 4); (void)hipMemset(cs, 0, w.k * 4);
    const long long Mc = M < 2048 ? M : 2048;       // rows that carry data, spread over the whole range
    std::vector<float> hx(Mc * w.k), hd(Mc * w.n), hw(w.k * w.n);
    for (auto& v : hx) v = frand(); for (auto& v : hd) v = frand(); for (auto& v : hw) v = frand() * 0.1f;
    (void)hipMemset(x, 0, M * w.k * 4); (void)hipMemset(dz, 0, M * w.n * 4);
    for (long long j = 0; j < Mc; ++j) {
      const long long r = j * M / Mc;
      (void)hipMemcpy(x + r * w.k, hx.data() + j * w.k, w.k * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dz + r * w.n, hd.data() + j * w.n, w.n * 4, hipMemcpyHostToDevice);
    }
    (void)hipMemcpy(wt, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    pack_frags(nullptr, wt, w.n, 0, w.n, w.k, 1, frag);
    BwdFusedArgs A{x, w.k, w.k, dz, w.n, w.n, frag, dx, w.k, NREP > 1 ? dw + (size_t)w.k * w.n : cs, dw, M, zeros, RS, NREP};
    if (!bwd_fused(nullptr, A, prop.multiProcessorCount)) { printf("bwd_fused %d x %d not supported\n", w.k, w.n); ++bad; continue; }
    { hipError_t e = hipDeviceSynchronize(); if (e != hipSuccess) printf("  !! %s\n", hipGetErrorString(e)); }
    std::vector<float> gw(w.k * w.n), gc(w.k), gx((size_t)M * w.k);
    (void)hipMemcpy(gw.data(), dw, gw.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(gc.data(), NREP > 1 ? dw + (size_t)w.k * w.n : cs, gc.size() * 4, hipMemcpyDeviceToHost);
    for (int rp = 1; rp < NREP; ++rp) {              // the caller's part: sum the replicas
      std::vector<float> t1(gw.size()), t2(gc.size());
      (void)hipMemcpy(t1.data(), dw + rp * RS, t1.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t2.data(), dw + rp * RS + (size_t)w.k * w.n, t2.size() * 4, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < gw.size(); ++i) gw[i] += t1[i];
      for (size_t i = 0; i < gc.size(); ++i) gc[i] += t2[i];
    }
    (void)hipMemcpy(gx.data(), dx, gx.size() * 4, hipMemcpyDeviceToHost);
    double e_w = 0, big = 0, e_x = 0, e_c = 0, bigc = 0;
    std::vector<double> cref(w.k, 0.0);
    for (long long j = 0; j < Mc; ++j) {
      const long long r = j * M / Mc;
      for (int k = 0; k < w.k; ++k) {
        double a = 0;
        for (int n = 0; n < w.n; ++n) a += (double)hd[j * w.n + n] * hw[k * w.n + n];
        if (!(hx[j * w.k + k] > 0)) a = 0;
        cref[k] += a;
        e_x = std::fmax(e_x, std::fabs(a - gx[r * w.k + k]));
      }
    }
    for (int k = 0; k < w.k; ++k) {
      e_c = std::fmax(e_c, std::fabs(cref[k] - gc[k])); bigc = std::fmax(bigc, std::fabs(cref[k]));
      for (int n = 0; n < w.n; ++n) {
        double a = 0;
        for (long long j = 0; j < Mc; ++j) a += (double)hx[j * w.k + k] * hd[j * w.n + n];
        e_w = std::fmax(e_w, std::fabs(a - gw[k * w.n + n])); big = std::fmax(big, std::fabs(a));
      }
    }
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 20;
    (void)hipEventRecord(e0, nullptr);
    if (getenv("NOATOM")) { A.dw = nullptr; A.colsum = nullptr; }
    if (getenv("NODW")) A.dw = nullptr;
    if (getenv("NOCS")) A.colsum = nullptr;
    for (int i = 0; i < reps; ++i) bwd_fused(nullptr, A, prop.multiProcessorCount);
    (void)hipEventRecord(e1, nullptr); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const bool ok = e_x < 1e-4 && e_w < 3e-5 * big && e_c < 3e-5 * (bigc + 1.0);
    if (!ok) ++bad;
    printf("bwd_fused M=%lld K=%d N=%d: %.3f ms  %.0f GB/s  max|err| dX %.2e  dW %.2e (max %.1f)  colsum %.2e (max %.1f)%s\n", M, w.k, w.n, ms,
           (double)M * (2 * w.k + w.n) * 4 / ms / 1e6, e_x, e_w, big, e_c, bigc, ok ? "" : "  FAIL");
    (void)hipFree(x); (void)hipFree(dz); (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(wt); (void)hipFree(cs); (void)hipFree(zeros); (void)hipFree(frag);
  }
  if (bad) printf("%d shape(s) out of bounds\n", bad);
  return bad ? 1 : 0;
}
