// Fused NeRF-DS ray kernel for gfx950 (MI355X / CDNA4).
//
// One wavefront renders one ray end to end: stratified sampling -> [MaskMLP -> SE(3) warp MLP + exp_se3 ->
// hyper-sheet MLP -> trunk / sigma / rgb NerfMLP] on every sample (field.h) -> exclusive-cumprod compositing ->
// inverse-CDF resample + sort -> the same networks on the fine samples -> compositing -> one 104-byte record.
// Nothing per-sample ever goes to HBM (unless the optional per-sample record is requested).
//
// Reference functions restated in this file (paths under /root/reference/hypernerf/):
//   sample_along_rays model_utils.py:55-92      volumetric_rendering model_utils.py:95-159
//   piecewise_constant_pdf / sample_pdf model_utils.py:193-269   compute_depth_* model_utils.py:272-317
//   NerfModel.__call__ models.py:1419-1565 (levels, resample, which outputs)   per-ray reductions models.py:1346-1415
//
// One translation unit per (graph, precision plan):
//   -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=P_BF16 -DNERFDS_NAME=nerfds_bf16        uniform plan
//   -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_MIXED -DNERFDS_NAME=nerfds_mixed            graphs.h NERFDS_MIX_* plan
#define NERFDS_KERNEL_KIND 0
#include "field.h"
#include "launch.h"

// N-tiles per wave of the coarse NerfMLP under NERFDS_PREC_BF16X3_FINE (1: as every other evaluation of that kernel; A/B)
#ifndef NERFDS_COARSE_NT
#define NERFDS_COARSE_NT 2
#endif

namespace nerfds {

// ------------------------------------------------------------------------------------------------
// Compositing of one level (model_utils.py:95-159, 272-317; models.py:1346-1415) -> ray record.
// ------------------------------------------------------------------------------------------------
template <class G, class LT>
DEVI void composite(const KArgs& ka, const RayConst& rc, int ray, int lane, int S, bool at_infinity, int opt_flags, LT& L,
                    float* __restrict__ rec_out, float* __restrict__ smp_out) {
  const float dnorm = sqrtf(rc.d[0] * rc.d[0] + rc.d[1] * rc.d[1] + rc.d[2] * rc.d[2]);
  const float last = at_infinity ? 1e10f : 1e-19f;
  float carryT = 1.0f, carryC = 0.0f;
  int med_idx = -1;
  float acc_v[21];
#pragma unroll
  for (int i = 0; i < 21; ++i) acc_v[i] = 0.f;
  // acc_v: 0-2 rgb, 3 depth, 4 acc, 5-7 norm, 8-10 rot, 11-13 trn, 14-16 delta_x, 17-18 hyper, 19 mask, 20 sum of all w
  for (int j = 0; j * 64 < S; ++j) {
    const int s = lane + 64 * j;
    const bool valid = s < S;
    const int sc = valid ? s : S - 1;
    const float z = L.zs[sc];
    const float zn = L.zs[(sc + 1 < S) ? sc + 1 : sc];
    float dist = (sc == S - 1) ? last : (zn - z);
    dist *= dnorm;                                                         // model_utils.py:124-128
    const float sigma = L.sv[SV_SIGMA][sc];
    float xo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xo[c] = __fadd_rn(rc.o[c], __fmul_rn(z, rc.d[c]));
    // render_opts (filter_sigma, models.py:38-66, applied at models.py:1288 to the ACTIVATED density; the per-sample 'sigma' output below is
    // the unfiltered one, models.py:1271): (sigma >= dust_threshold) * sigma, then (point inside the box) * sigma.  Uniform flags, selects.
    // opt_flags is the LEVEL's: NerfModel.__call__ hands render_opts to the fine render_samples only (models.py:1545; the coarse call,
    // models.py:1493-1517, leaves the default None), so the coarse level - and the pdf the fine depths are drawn from - is never filtered.
    float sig_f = sigma;
    if (opt_flags & 1) sig_f = (sig_f >= ka.dust_threshold) ? sig_f : 0.f * sig_f;
    if (opt_flags & 2) {
      const bool in = (xo[0] >= ka.bbox[0]) & (xo[0] <= ka.bbox[1]) & (xo[1] >= ka.bbox[2]) & (xo[1] <= ka.bbox[3]) &
                      (xo[2] >= ka.bbox[4]) & (xo[2] <= ka.bbox[5]);
      sig_f = in ? sig_f : 0.f * sig_f;
    }
    float alpha = valid ? (1.0f - expf(-sig_f * dist)) : 0.0f;             // model_utils.py:129
    float om = valid ? ((1.0f - alpha) + 1e-10f) : 1.0f;                   // model_utils.py:133
    float incl = wave_scan_mul(om, lane);
    const float excl = wave_shift_up1(incl, 1.0f);
    const float T = carryT * excl;                                         // exclusive cumprod
    carryT *= lane63(incl);
    const float w = alpha * T;                                             // model_utils.py:135
    if (valid) L.ws[s] = w;
    float cum = wave_scan_add(w, lane) + carryC;
    carryC = lane63(cum);
    unsigned long long hit = __ballot(valid && cum >= 0.5f);               // model_utils.py:285-291
    if (med_idx < 0 && hit) med_idx = 64 * j + __builtin_ctzll(hit);
    acc_v[0] += w * L.sv[SV_RGB + 0][sc];
    acc_v[1] += w * L.sv[SV_RGB + 1][sc];
    acc_v[2] += w * L.sv[SV_RGB + 2][sc];
    acc_v[3] += w * z;
    acc_v[4] += (at_infinity && sc == S - 1) ? 0.f : w;                    // model_utils.py:147-148
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      acc_v[5 + c] += w * L.sv[SV_NORM + c][sc];
      acc_v[8 + c] += w * L.sv[SV_ROT + c][sc];
      acc_v[11 + c] += w * L.sv[SV_TRN + c][sc];
      acc_v[14 + c] += w * (L.sv[SV_WP + c][sc] - xo[c]);                  // models.py:1363-1364
    }
    acc_v[17] += w * L.sv[SV_WP + 3][sc];
    acc_v[18] += w * L.sv[SV_WP + 4][sc];
    acc_v[19] += w * L.sv[SV_MASK][sc];
    acc_v[20] += w;
    if (smp_out != nullptr && valid) {
      float* r = smp_out + (size_t)s * SAMPLE_REC;
      r[0] = z; r[1] = sigma; r[2] = alpha; r[3] = T; r[4] = w; r[5] = L.sv[SV_MASK][s];
      r[6] = L.sv[SV_RGB][s]; r[7] = L.sv[SV_RGB + 1][s]; r[8] = L.sv[SV_RGB + 2][s];
      r[9] = L.sv[SV_NORM][s]; r[10] = L.sv[SV_NORM + 1][s]; r[11] = L.sv[SV_NORM + 2][s];
#pragma unroll
      for (int c = 0; c < 5; ++c) r[12 + c] = L.sv[SV_WP + c][s];
      const float* vd = (ka.viewdirs ? ka.viewdirs : ka.directions) + 3 * (size_t)ray;
      float bf = L.sv[SV_NORM][s] * vd[0] + L.sv[SV_NORM + 1][s] * vd[1] + L.sv[SV_NORM + 2][s] * vd[2];
      bf = fmaxf(bf, 0.f);
      r[17] = bf * bf;                                                     // models.py:1341-1343
    }
  }
#pragma unroll
  for (int i = 0; i < 21; ++i) acc_v[i] = wave_sum(acc_v[i]);
  if (rec_out == nullptr) return;
  const int mi = med_idx < 0 ? 0 : med_idx;                                // argmax of an all-zero mask is 0
  if (ka.white_bkgd) {                                                     // model_utils.py:144-145 (acc before :-1)
    acc_v[0] += 1.0f - acc_v[20];
    acc_v[1] += 1.0f - acc_v[20];
    acc_v[2] += 1.0f - acc_v[20];
  }
  // Select the record element for this lane (26 lanes active).
  float outv = 0.f;
  if (lane < 4) outv = lane == 0 ? acc_v[0] : lane == 1 ? acc_v[1] : lane == 2 ? acc_v[2] : acc_v[3];
  else if (lane == 4) outv = med_idx < 0 ? 0.f : L.zs[mi];                 // model_utils.py:316-317
  else if (lane <= 20) {
#pragma unroll
    for (int i = 4; i < 20; ++i) if (lane == i + 1) outv = acc_v[i];
  } else if (lane < RAY_REC) outv = L.sv[SV_WP + (lane - 21)][mi];        // models.py:1411-1415
  if (lane < RAY_REC) rec_out[lane] = outv;
}

// ------------------------------------------------------------------------------------------------
// Coarse -> fine: inverse-CDF resample + sorted union (model_utils.py:193-269; models.py:1522-1526).
// On entry zs[0..nc) = coarse z, ws[0..nc) = coarse weights.  On exit zs[0..nc+nf) = sorted union.
// ------------------------------------------------------------------------------------------------
#ifdef NERFDS_PROF
#define NERFDS_RS_MARK(k) do { unsigned tie_ = __builtin_bit_cast(unsigned, L.zn[lane]); const unsigned long long now_ = prof_now(tie_); prof_rs[k] += now_ - prof_last; prof_last = now_; } while (0)
#else
#define NERFDS_RS_MARK(k) do { } while (0)
#endif
template <class LT> DEVI void resample(const KArgs& ka, int ray, int lane, int nc, int nf, LT& L
#ifdef NERFDS_PROF
                                       , unsigned long long (&prof_rs)[4]
#endif
                                       ) {
  constexpr int MAX_SAMPLES = LT::MAX_S;
#ifdef NERFDS_PROF
  unsigned prof_tie0 = (unsigned)lane;
  unsigned long long prof_last = prof_now(prof_tie0);
#endif
  const int nb = nc - 1;          // bins = mid points (nc-1 of them); cdf has nb entries, cdf[0] = 0
  const int nw = nc - 2;          // weights[..., 1:-1]
  // pdf / cdf
  float tot = 0.f;
  for (int j = 0; j * 64 < nw; ++j) {
    int i = lane + 64 * j;
    tot += (i < nw) ? (L.ws[i + 1] + 1e-5f) : 0.f;
  }
  tot = wave_sum(tot);
  float carry = 0.f;
  if (lane == 0) L.cdf[0] = 0.f;
  for (int j = 0; j * 64 < nw; ++j) {
    int i = lane + 64 * j;
    float pdf = (i < nw) ? (L.ws[i + 1] + 1e-5f) / tot : 0.f;
    float c = wave_scan_add(pdf, lane) + carry;
    carry = lane63(c);
    if (i < nw) L.cdf[i + 1] = c;
  }
  for (int j = 0; j * 64 < nb; ++j) {
    int i = lane + 64 * j;
    if (i < nb) L.zn[i] = 0.5f * (L.zs[i + 1] + L.zs[i]);      // bins (models.py:1522)
  }
  WAVE_SYNC();
  NERFDS_RS_MARK(0);
  // inverse CDF for my fine samples
  float zf[MAX_SAMPLES / 64];
#pragma unroll
  for (int j = 0; j < MAX_SAMPLES / 64; ++j) {
    zf[j] = 0.f;
    int k = lane + 64 * j;
    if (k >= nf) continue;
    float u;
    if (ka.stratified) {
      if (ka.u_rand != nullptr) {
        u = ka.u_rand[(size_t)ray * nf + k];
      } else {
        u = sample_uniform(ka.seed, ka.first_ray + ray, 1, k);
      }
    } else {
      u = (nf > 1) ? (float)k / (float)(nf - 1) : 0.f;                       // linspace(0, 1, nf)
    }
    // count = #(cdf[i] <= u): mask = u >= cdf (model_utils.py:223)
    int lo = 0, hi = nb;                 // upper bound
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (L.cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    int k0 = lo - 1;
    if (k0 < 0) k0 = 0;
    const int k1 = (k0 + 1 < nb) ? k0 + 1 : nb - 1;
    // minmax clamps (model_utils.py:228-229)
    const float b0 = fminf(L.zn[k0], L.zn[nb - 2]), b1 = fmaxf(L.zn[k1], L.zn[1]);
    const float c0 = fminf(L.cdf[k0], L.cdf[nb - 2]), c1 = fmaxf(L.cdf[k1], L.cdf[1]);
    float denom = c1 - c0;
    if (denom < 1e-5f) denom = 1.0f;
    const float t = (u - c0) / denom;
    zf[j] = b0 + t * (b1 - b0);
  }
  WAVE_SYNC();
  // union into zn (unsorted), then rank-sort into zs.  The level-independent per-sample state of the COARSE samples (eval_shared:
  // predicted mask, warped point + ambient coordinates, rotation / translation fields, rotation) moves with them to their slots
  // in the sorted union - the fine level evaluates the mask / warp / hyper networks only on the nf NEW samples, whose slots are
  // left in L.cdf (as integers) and whose depths stay in zn[nc ..].
  const int n = nc + nf;
#pragma unroll
  for (int j = 0; j < MAX_SAMPLES / 64; ++j) {
    int k = lane + 64 * j;
    if (k < nc) L.zn[k] = L.zs[k];
    if (k < nf) L.zn[nc + k] = zf[j];
  }
  WAVE_SYNC();
  NERFDS_RS_MARK(1);
  constexpr int NJ = MAX_SAMPLES / 64, NSTATE = 1 + (SV_COUNT - SV_WP);
  int rk[NJ];
  float st[NJ][NSTATE];
  // rank of union element i = #{q : z_q < z_i or (z_q == z_i and q < i)} (the union = [coarse | new]: on ties the coarse sample first, then the
  // lower index - any order of equal depths gives the same sorted z, model_utils.py:267, and the same per-sample state).
  if (ka.near_ <= ka.far_) {
    // The coarse depths are ALREADY in ascending order (model_utils.py:75-89: z rises with t, the stratified jitter stays inside a sample's own
    // bin; equal depths - near == far - keep their index order), so only the nf NEW samples need counting:
    //   coarse i : rank = i + #{new k : z_k < z_i}
    //   new k    : rank = #{coarse i : z_i <= z_k} (a binary search in the sorted coarse depths) + #{new j : z_j < z_k or (z_j == z_k and j < k)}
    // one pass over the NEW depths for all of the lane's elements, four values per (broadcast) LDS read.  (Through round 4 every element was
    // counted against the whole union, 2 x 128 x 128 compare / select / add steps per ray: measured with the phase timers of tools/prof_phases.sh
    // at 6.0 % of the bf16 kernel and 2.4 % of the split-bf16 kernel - more than compositing both levels.)
    float vc[NJ], vn[NJ];
    int rn[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = lane + 64 * j;
      vc[j] = L.zn[i < nc ? i : nc - 1];
      vn[j] = L.zn[nc + (i < nf ? i : nf - 1)];
      rk[j] = i;                               // the coarse samples below it
      int lo = 0, hi = nc;                     // #{coarse <= vn}: upper bound in the sorted coarse depths
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.zn[mid] <= vn[j]) lo = mid + 1; else hi = mid;
      }
      rn[j] = lo;
    }
    // (uniform tests: with 64 + 64 samples a lane holds ONE live coarse and ONE live new element - the second slot of either array is past the end)
    auto count = [&](float o, int q) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (64 * j < nc) rk[j] += (o < vc[j]) ? 1 : 0;
        if (64 * j < nf) rn[j] += (o < vn[j] || (o == vn[j] && q < lane + 64 * j)) ? 1 : 0;
      }
    };
    const float* znew = L.zn + nc;             // (16-byte aligned when nc is a multiple of 4; the scalar tail loop covers the rest)
    const int n4 = (nc & 3) ? 0 : (nf & ~3);
#pragma unroll 4
    for (int q = 0; q < n4; q += 4) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(&znew[q]);
      count(o[0], q); count(o[1], q + 1); count(o[2], q + 2); count(o[3], q + 3);
    }
    for (int q = n4; q < nf; ++q) count(znew[q], q);
    // this lane's union element i = lane + 64 j is coarse sample i (i < nc) or new sample i - nc
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = lane + 64 * j;
      // new sample k = i - nc lives in lane (k & 63), slot (k >> 6) of the arrays above: fetch its rank from there (every lane takes part
      // in the shuffles - a permute reads nothing from a lane that is switched off -, the select picks)
      const int k = i >= nc ? i - nc : 0;
      int r = 0;
#pragma unroll
      for (int jj = 0; jj < NJ; ++jj) {
        const int got = __shfl(rn[jj], k & 63, 64);
        r = ((k >> 6) == jj) ? got : r;
      }
      rk[j] = i >= nc ? r : rk[j];
    }
  } else {
    // (near > far: descending coarse depths - no shipped configuration; every element against the whole union, as through round 4)
    float v[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int i = lane + 64 * j;
      v[j] = L.zn[i < n ? i : n - 1];
      rk[j] = 0;
    }
    auto count = [&](float o, int q) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) rk[j] += (o < v[j] || (o == v[j] && q < lane + 64 * j)) ? 1 : 0;
    };
    const int n4 = n & ~3;
#pragma unroll 4
    for (int q = 0; q < n4; q += 4) {
      const f32x4 o = *reinterpret_cast<const f32x4*>(&L.zn[q]);
      count(o[0], q); count(o[1], q + 1); count(o[2], q + 2); count(o[3], q + 3);
    }
    for (int q = n4; q < n; ++q) count(L.zn[q], q);
  }
#ifdef NERFDS_PROF
  { unsigned tie_ = (unsigned)rk[0]; const unsigned long long now_ = prof_now(tie_); rk[0] = (int)tie_; prof_rs[2] += now_ - prof_last; prof_last = now_; }
#endif
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    const int ic = i < nc ? i : nc - 1;
    st[j][0] = L.sv[SV_MASK][ic];
#pragma unroll
    for (int a = 0; a < SV_COUNT - SV_WP; ++a) st[j][1 + a] = L.sv[SV_WP + a][ic];
  }
  WAVE_SYNC();
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int i = lane + 64 * j;
    if (i < n) L.zs[rk[j]] = L.zn[i];
    if (i < nc) {
      L.sv[SV_MASK][rk[j]] = st[j][0];
#pragma unroll
      for (int a = 0; a < SV_COUNT - SV_WP; ++a) L.sv[SV_WP + a][rk[j]] = st[j][1 + a];
    } else if (i < n) {
      reinterpret_cast<int*>(L.cdf)[i - nc] = rk[j];
    }
  }
  WAVE_SYNC();
  NERFDS_RS_MARK(3);
}

// ------------------------------------------------------------------------------------------------
// Kernel: persistent waves, one ray per wave per iteration.
// ------------------------------------------------------------------------------------------------
// X3F16: the split arithmetic's 16-bit format is a MACRO of the translation unit (field.h NERFDS_X3_F16), not part of PL - it must still be part of the
// kernel's NAME: two translation units that build the same template arguments with different macros emit one mangled kernel name with two bodies, and
// the runtime binds both launchers to ONE of them (round 6 found the split-f16 launcher running the split-bf16 kernel on f16-packed weights: every product
// ~1e-14, every output its head's bias; round 3 lost a conclusion to the same trap in the training kernels, field.h TRAIN_TAG).
template <class G, class PL, bool WIDE, int X3F16 = NERFDS_X3_F16>
__global__ __launch_bounds__(64 * wg_waves<PL>(), wg_waves<PL>() / 4) void render_rays_kernel(const KArgs ka) {
  using SH = Shape<PL, WIDE>;
  using WaveLds = WaveLdsT<SH::MAXS>;
  constexpr int NT = SH::NT, SPLIT = SH::SPLIT, RAYS_PER_WG = SH::RAYS, WAVES = wg_waves<PL>(), BATCH = 32 * NT * SPLIT;
  using Dm = Dims<G>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // ray slot in the workgroup, share of the ray's batch.  The q == 0 waves (they run the per-ray phases alone) are waves
  // 0 .. RAYS - 1, i.e. one per SIMD; with slot = wave / SPLIT they sat two by two on two of the four SIMDs (+0.7 %).
  const int slot = wave % RAYS_PER_WG, q = wave / RAYS_PER_WG;
  WaveLds& L = *reinterpret_cast<WaveLds*>(g_smem + BIAS_OFF + bias_bytes<G>() + slot * (int)sizeof(WaveLds));
  // Waves that share a ray meet at workgroup barriers around the per-ray phases (every wave executes the same
  // barrier sequence, so these interleave consistently with the per-stage barriers of the weight pipe).
  auto ray_sync = [&]() { if constexpr (SPLIT > 1) __syncthreads(); else WAVE_SYNC(); };

  using PP = Pipe<G, PL>;
  Pipe<G, PL> pipe;
#ifdef NERFDS_PROF
  unsigned prof_k_ = (unsigned)lane;
  const unsigned long long prof_start = prof_now(prof_k_);
  unsigned long long prof_rs[4] = {0, 0, 0, 0};
#define NERFDS_EVAL(call) do { unsigned tie_ = (unsigned)lane; NERFDS_PROF_BEGIN(tie_); call; tie_ = __builtin_bit_cast(unsigned, L.sv[0][lane]); NERFDS_PROF_END(pipe.t_eval, tie_); } while (0)
#else
#define NERFDS_EVAL(call) call
#endif
#ifdef NERFDS_PROF
#define NERFDS_RAYPHASE(call) do { unsigned tie_ = (unsigned)lane; NERFDS_PROF_BEGIN(tie_); call; tie_ = __builtin_bit_cast(unsigned, L.zs[lane]); NERFDS_PROF_END(pipe.t_ray, tie_); } while (0)
#else
#define NERFDS_RAYPHASE(call) call
#endif
  // (a single-level model's one NerfMLP is the level render_fn returns: it runs in the fine plan - the host passes its stream in both slots)
  const rsrc_t rs_nerf[2] = {make_rsrc(ka.wstream[1], PP::NERF_C_PAD * 1024), make_rsrc(ka.wstream[2], PP::NERF_PAD * 1024)};
  const rsrc_t rs_shared = PP::HAS_SHARED ? make_rsrc(ka.wstream[0], PP::SHARED_PAD * 1024) : rs_nerf[0];
  pipe.cur = pipe.next = PP::HAS_SHARED ? rs_shared : rs_nerf[0];
  pipe.lane16 = lane * 16;
  pipe.wave1k = wave * 1024;
  pipe.prologue(PP::HAS_SHARED ? SEG_SHARED : SEG_NERF);     // the first NS - 1 stages of the first segment
  {  // padded biases ([tile][32 rows] from the packer) -> LDS in the bias_tile_off layout, once per workgroup
    constexpr int n0 = Dm::SHARED_BIAS_TILES * 32, n1 = Dm::NERF_BIAS_TILES * 32;     // float counts
    float* dst = reinterpret_cast<float*>(g_smem + BIAS_OFF);
    // Uniform trip count (the tail threads repeat the last element: same value to the same address): a divergent loop here is a
    // region across which `lane` and friends are live, and this hipcc has placed a live-range-split copy of `lane` at the top of
    // such a loop's join block, ahead of the exec restore (tools/isa_lint.py; seen as wild per-sample stores of the fp32 kernel).
    for (int i0 = 0; i0 < n0 + 2 * n1; i0 += 64 * WAVES) {
      const int ix = i0 + (int)threadIdx.x, i = ix < n0 + 2 * n1 ? ix : n0 + 2 * n1 - 1;
      const float v = i < n0 ? ka.bias[0][i] : (i < n0 + n1 ? ka.bias[1][i - n0] : ka.bias[2][i - n0 - n1]);
      const int t = i >> 5, m = i & 31;                      // row m = (r & 3) + 8 (r >> 2) + 4 h  ->  h = (m >> 2) & 1, r = (m & 3) + 4 (m >> 3)
      dst[bias_tile_off(t) / 4 + 128 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3)] = v;
    }
    __syncthreads();
  }

  // All waves of a workgroup walk the SAME weight stream in lockstep (one barrier per 16 KiB stage); ray slot s
  // renders ray 4 * group + s.  Tail slots re-render the last ray and drop the result.
  const int groups = (ka.num_rays + RAYS_PER_WG - 1) / RAYS_PER_WG;
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int ray_raw = grp * RAYS_PER_WG + slot;
    const bool live = (ray_raw < ka.num_rays) && (q == 0);     // the q == 0 wave of a ray owns its outputs
    const int ray = (ray_raw < ka.num_rays) ? ray_raw : ka.num_rays - 1;
    RayConst rc;
    float vdir[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rc.o[c] = ka.origins[3 * (size_t)ray + c];
      rc.d[c] = ka.directions[3 * (size_t)ray + c];
      vdir[c] = (ka.viewdirs ? ka.viewdirs : ka.directions)[3 * (size_t)ray + c];
    }
    rc.gt_mask = (ka.gt_mask != nullptr) ? ka.gt_mask[ray] : 0.f;
    if (q == 0) {  // per-ray constants -> LDS: GLO rows (modules.py:336-348) and posenc(viewdirs) (models.py:401-405, no window)
      uint32_t wid = (G::HAS_WARP && ka.warp_id != nullptr) ? ka.warp_id[ray] : 0u;
      wid = wid < (uint32_t)ka.num_embeds ? wid : (uint32_t)(ka.num_embeds - 1);     // jnp gathers clamp out-of-range ids
      if (lane < 8) {
        // metadata_encoded (models.py:898-899): the ray's pre-encoded GLO vector instead of the table row of its id (uniform selects)
        const float* wrow = ka.enc_warp != nullptr ? ka.enc_warp + (size_t)ray * 8 : ka.warp_embed + (size_t)wid * 8;
        const float* mrow = ka.enc_mask != nullptr ? ka.enc_mask + (size_t)ray * 8 : ka.mask_embed + (size_t)wid * 8;
        L.rayc[RC_WEMB + lane] = G::HAS_WARP ? wrow[lane] : 0.f;
        L.rayc[RC_MEMB + lane] = G::HAS_MASK ? mrow[lane] : 0.f;
      }
      if (lane < 24) {
        const int band = lane / 6, sc = (lane % 6) / 3, ch = lane % 3;
        const float vdc = ch == 0 ? vdir[0] : (ch == 1 ? vdir[1] : vdir[2]);
        L.rayc[RC_VDENC + lane] = sin_cw(fmaf(vdc, (float)(1 << band), sc ? 1.57079637f : 0.0f));
      }
      if (lane < 3) L.rayc[RC_VD + lane] = lane == 0 ? vdir[0] : (lane == 1 ? vdir[1] : vdir[2]);
    }

    // ---- coarse z (model_utils.py:75-89) ----
    const int nc = ka.nc, nf = ka.nf;
    for (int j = 0; j * 64 < nc; ++j) {
      const int i = lane + 64 * j;
      if (i < nc && q == 0) {
        auto zlin = [&](int q) {
          const float t = (nc > 1) ? (float)q / (float)(nc - 1) : 0.f;
          if (ka.lindisp) return 1.0f / (1.0f / ka.near_ * (1.0f - t) + 1.0f / ka.far_ * t);      // model_utils.py:75-76
          return ka.near_ * (1.0f - t) + ka.far_ * t;
        };
        float z = zlin(i);
        if (ka.stratified) {
          const float zlo = (i > 0) ? 0.5f * (z + zlin(i - 1)) : z;
          const float zhi = (i + 1 < nc) ? 0.5f * (zlin(i + 1) + z) : z;
          float t;
          if (ka.t_rand != nullptr) {
            t = ka.t_rand[(size_t)ray * nc + i];
          } else {
            t = sample_uniform(ka.seed, ka.first_ray + ray, 0, i);
          }
          z = zlo + (zhi - zlo) * t;
        }
        L.zs[i] = z;
      }
    }
    ray_sync();

    const int ln = lane & 31;
    auto samples_at = [&](int s0, int S) {       // slots s0 + 32 nt + (lane & 31) of a level with S sorted samples
      Samples<NT> sm;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int sx = s0 + 32 * nt + ln;
        sm.slot[nt] = sx < S ? sx : S - 1;
        sm.z[nt] = L.zs[sm.slot[nt]];
      }
      return sm;
    };
    // ---- coarse level: every network on every sample ----
    if constexpr (PL::HAS_C && PP::HAS_SHARED && NT == 1 && NERFDS_COARSE_NT == 2) {
      // NERFDS_PREC_BF16X3_FINE: the coarse NerfMLP runs one f16 MFMA per product, and at one MFMA per fragment a single N-tile is bound by the
      // LDS reads of the weight ring (1 KiB per MFMA and wave, four waves: the CU's 128 B / clk).  So the level-independent networks (split bf16,
      // 32 samples per wave and batch as everywhere) run first on ALL coarse samples of the ray, then the coarse NerfMLP takes TWO N-tiles per
      // wave - 64 samples, a 64 + 64 ray's whole coarse level - per walk of its stream: every fragment read feeds two MFMAs, and the stream is
      // staged (and its barriers paid) once per 64 samples instead of twice.
      for (int sb = 0; sb < nc; sb += BATCH) {
        const Samples<NT> sm = samples_at(sb + 32 * NT * q, nc);
        pipe.cur = rs_shared; pipe.next = (sb + BATCH < nc) ? rs_shared : rs_nerf[0];
        NERFDS_EVAL((eval_shared<G, PL, NT>(ka, rc, pipe, lane, sm, L)));
      }
      ray_sync();              // (wide shape: a wave's 64 NerfMLP samples include slots its partner wave has just written)
      constexpr int NTC = 2, BATCHC = 32 * NTC * SPLIT;
      for (int sb = 0; sb < nc; sb += BATCHC) {
        Samples<NTC> smc;
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
          const int sx = sb + 32 * NTC * q + 32 * nt + ln;
          smc.slot[nt] = sx < nc ? sx : nc - 1;
          smc.z[nt] = L.zs[smc.slot[nt]];
        }
        pipe.cur = rs_nerf[0]; pipe.next = (sb + BATCHC < nc) ? rs_nerf[0] : rs_shared;
        NERFDS_EVAL((eval_nerf<G, PL, NTC, WaveLds, NoTrain, true>(ka, pipe, 0, lane, smc, L)));
      }
    } else {
    for (int sb = 0; sb < nc; sb += BATCH) {
      const Samples<NT> sm = samples_at(sb + 32 * NT * q, nc);
      if constexpr (PP::HAS_SHARED) { pipe.cur = rs_shared; pipe.next = rs_nerf[0]; }
      NERFDS_EVAL((eval_shared<G, PL, NT>(ka, rc, pipe, lane, sm, L)));
      pipe.cur = rs_nerf[0];
      if constexpr (PP::HAS_SHARED) pipe.next = rs_shared;      // another coarse batch, the fine level's new samples, or the next ray group
      else pipe.next = (sb + BATCH < nc) ? rs_nerf[0] : (nf > 0 ? rs_nerf[1] : rs_nerf[0]);
      NERFDS_EVAL((eval_nerf<G, PL, NT, WaveLds, NoTrain, PL::HAS_C>(ka, pipe, 0, lane, sm, L)));
    }
    }
    ray_sync();
    if (q == 0) {
      float* rec = live ? ((nf > 0) ? ka.ray_coarse : ka.ray_fine) : nullptr;
      float* smp = live ? ((nf > 0) ? ka.smp_coarse : ka.smp_fine) : nullptr;
      // the 'coarse' render_samples call (models.py:1493-1517): the MODEL's use_sample_at_infinity (:1509), no render_opts
      NERFDS_RAYPHASE((composite<G>(ka, rc, ray, lane, nc, ka.sample_at_infinity != 0, 0, L,
                   rec ? rec + (size_t)ray * RAY_REC : nullptr,
                   smp ? smp + (size_t)ray * nc * SAMPLE_REC : nullptr)));
      WAVE_SYNC();
#ifdef NERFDS_PROF
      if (nf > 0) resample(ka, ray, lane, nc, nf, L, prof_rs);
#else
      if (nf > 0) resample(ka, ray, lane, nc, nf, L);
#endif
    }
    ray_sync();

    // ---- fine level: the level-independent networks on the nf NEW samples only (the coarse samples keep the values of the
    // coarse pass: same networks, same inputs - the reference's second evaluation, models.py:1291-1300 under models.py:1528-1546,
    // returns the same numbers), then the fine NerfMLP on the whole sorted union ----
    if (nf > 0) {
      const int n = nc + nf;
      for (int sb = 0; sb < nf; sb += BATCH) {
        Samples<NT> sm;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int kx = sb + 32 * NT * q + 32 * nt + ln, k = kx < nf ? kx : nf - 1;
          sm.z[nt] = L.zn[nc + k];
          sm.slot[nt] = reinterpret_cast<const int*>(L.cdf)[k];
        }
        if constexpr (PP::HAS_SHARED) { pipe.cur = rs_shared; pipe.next = (sb + BATCH < nf) ? rs_shared : rs_nerf[1]; }
        NERFDS_EVAL((eval_shared<G, PL, NT>(ka, rc, pipe, lane, sm, L)));
      }
      ray_sync();              // a NerfMLP batch reads slots that other waves of the ray have written
      for (int sb = 0; sb < n; sb += BATCH) {
        const Samples<NT> sm = samples_at(sb + 32 * NT * q, n);
        pipe.cur = rs_nerf[1];
        pipe.next = (sb + BATCH < n) ? rs_nerf[1] : (PP::HAS_SHARED ? rs_shared : rs_nerf[0]);
        NERFDS_EVAL((eval_nerf<G, PL, NT>(ka, pipe, 1, lane, sm, L)));
      }
      ray_sync();
      if (q == 0)
        // the 'fine' call (models.py:1528-1552): the per-call use_sample_at_infinity override (:1544) and render_opts (:1545)
        NERFDS_RAYPHASE((composite<G>(ka, rc, ray, lane, n, ka.sample_at_infinity_fine != 0, ka.opt_flags, L,
                     (live && ka.ray_fine) ? ka.ray_fine + (size_t)ray * RAY_REC : nullptr,
                     (live && ka.smp_fine) ? ka.smp_fine + (size_t)ray * n * SAMPLE_REC : nullptr)));
      ray_sync();
    }
  }
#ifdef NERFDS_PROF
  {
    const unsigned long long total = prof_now(prof_k_) - prof_start;
    if (lane == 0 && ka.prof != nullptr) {
      atomicAdd(ka.prof + 0, total); atomicAdd(ka.prof + 1, pipe.t_chain); atomicAdd(ka.prof + 2, pipe.t_eval); atomicAdd(ka.prof + 3, 1ull);
      atomicAdd(ka.prof + 4, pipe.t_ray);
      for (int k = 0; k < 4; ++k) atomicAdd(ka.prof + 5 + k, prof_rs[k]);
      atomicAdd(ka.prof + 9, pipe.t_wait); atomicAdd(ka.prof + 10, pipe.t_bar); atomicAdd(ka.prof + 11, pipe.n_bound);
    }
  }
#endif
}

using KernelPlan =
#if defined(NERFDS_MIXED)
    PlanT<NERFDS_MIX_MASK, NERFDS_MIX_WARP, NERFDS_MIX_HYP, NERFDS_MIX_TRUNK, NERFDS_MIX_RGB>;
#elif defined(NERFDS_X3_FINE)
    // NERFDS_PREC_BF16X3_FINE (graphs.h plan_of(5)): split bf16 everywhere except the coarse level's NerfMLP, which runs one f16 MFMA per product
    PlanT<P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_F16, P_F16>;
#else
    PlanT<NERFDS_PREC, NERFDS_PREC, NERFDS_PREC, NERFDS_PREC, NERFDS_PREC>;
#endif
}  // namespace nerfds

template <bool WIDE> static void launch_shape(const nerfds::KArgs& ka, int num_cus, void* stream) {
  using namespace nerfds;
  using SH = Shape<KernelPlan, WIDE>;
  constexpr int lds = BIAS_OFF + bias_bytes<NERFDS_GRAPH>() + SH::RAYS * (int)sizeof(WaveLdsT<SH::MAXS>);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = render_rays_kernel<NERFDS_GRAPH, KernelPlan, WIDE, NERFDS_X3_F16>;
  allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  // one persistent workgroup per CU (LDS-bound), SH::RAYS rays per workgroup iteration
  const long long groups = ((long long)ka.num_rays + SH::RAYS - 1) / SH::RAYS;
  const int grid = (int)(groups < num_cus ? groups : num_cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wg_waves<KernelPlan>()), lds, static_cast<hipStream_t>(stream), ka);
}

extern "C" void NERFDS_CAT(nerfds_launch_, NERFDS_NAME)(const nerfds::KArgs& ka, int num_cus, void* stream) {
  if (ka.nc + ka.nf > nerfds::Shape<nerfds::KernelPlan, false>::MAXS) launch_shape<true>(ka, num_cus, stream);
  else launch_shape<false>(ka, num_cus, stream);
}
