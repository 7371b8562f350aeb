// Training forward of ONE level on the fused field (field.h): see train_forward_kernel below.  Built twice by the Makefile
// (-DNERFDS_TRAIN_HALF=0 -DNERFDS_TRAIN_PIPE=0: fp32 activation stores; =1 / =1: f16 + ReLU bits, pipelined tile epilogue).
#define NERFDS_KERNEL_KIND 1
#include "field.h"
#include "launch.h"

namespace nerfds {

// ------------------------------------------------------------------------------------------------
// Training forward of ONE level (training.py:198-511 -> models.py:867-1417 on the level's samples): the same field evaluation as
// render_rays_kernel - same weight pipe, same register-chained layers - on depths the trainer has already drawn (to.z), with every
// hidden layer's fp32 output and every head's raw output written to the trainer's workspace for the backward pass.  No
// compositing here: the loss kernel composites from sigma / rgb (train_kernels.hip).  The host passes the level's NerfMLP
// stream and biases in slot 1, so the kernel always evaluates "level 0".
// MODE (round 4, the merged step of nerfds_train.cpp run_merged): the mask / warp / hyper-sheet networks see only the observation-space point,
// and the fine level's sorted union repeats the coarse positions - the reference evaluates (and differentiates) them there a second time and
// gets the same numbers (models.py:1528-1546 over 1291-1300).  The trainer runs them ONCE per position:
//   MODE 0  shared networks + NerfMLP on the level's samples (the coarse level; every step that is not a merged one)
//   MODE 1  shared networks only, on the fine level's NEW samples
//   MODE 2  NerfMLP only, on the sorted union: the per-sample state the shared networks produced (warped point, ambient coordinates, screw
//           axis) comes from arrays gathered in union order (to.in_xw / in_wamb / in_wv) and is parked in the ray's LDS block here
// ------------------------------------------------------------------------------------------------
enum { FWD_FULL = 0, FWD_SHARED_ONLY = 1, FWD_NERF_ONLY = 2 };
template <class G, class PL, bool WIDE, int TAG, int MODE>
__global__ __launch_bounds__(64 * wg_waves<PL>(), wg_waves<PL>() / 4) void train_forward_kernel(const KArgs ka, const TrainOut to) {
  using SH = Shape<PL, WIDE>;
  using WaveLds = WaveLdsT<SH::MAXS>;
  constexpr int NT = SH::NT, SPLIT = SH::SPLIT, RAYS_PER_WG = SH::RAYS, WAVES = wg_waves<PL>(), BATCH = 32 * NT * SPLIT;
  using Dm = Dims<G>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slot = wave / SPLIT, q = wave % SPLIT;
  WaveLds& L = *reinterpret_cast<WaveLds*>(g_smem + BIAS_OFF + bias_bytes<G>() + slot * (int)sizeof(WaveLds));
  auto ray_sync = [&]() { if constexpr (SPLIT > 1) __syncthreads(); else WAVE_SYNC(); };
  using PP = Pipe<G, PL>;
  Pipe<G, PL> pipe;
  const rsrc_t rs_nerf = make_rsrc(ka.wstream[1], PP::NERF_PAD * 1024);
  const rsrc_t rs_shared = PP::HAS_SHARED ? make_rsrc(ka.wstream[0], PP::SHARED_PAD * 1024) : rs_nerf;
  pipe.cur = pipe.next = (MODE == FWD_NERF_ONLY) ? rs_nerf : rs_shared;
  pipe.lane16 = lane * 16;
  pipe.wave1k = wave * 1024;
  pipe.prologue((PP::HAS_SHARED && MODE != FWD_NERF_ONLY) ? SEG_SHARED : SEG_NERF);
  {  // biases -> LDS (as render_rays_kernel; only the shared nets and slot 1 are used)
    constexpr int n0 = Dm::SHARED_BIAS_TILES * 32, n1 = Dm::NERF_BIAS_TILES * 32;
    float* dst = reinterpret_cast<float*>(g_smem + BIAS_OFF);
    for (int i0 = 0; i0 < n0 + n1; i0 += 64 * WAVES) {        // uniform trip count, as in render_rays_kernel
      const int ix = i0 + (int)threadIdx.x, i = ix < n0 + n1 ? ix : n0 + n1 - 1;
      const float v = i < n0 ? ka.bias[0][i] : ka.bias[1][i - n0];
      const int t = i >> 5, m = i & 31;
      dst[bias_tile_off(t) / 4 + 128 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3)] = v;
    }
    __syncthreads();
  }
  const int S = ka.nc;
  const int groups = (ka.num_rays + RAYS_PER_WG - 1) / RAYS_PER_WG;
  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const int ray_raw = grp * RAYS_PER_WG + slot;
    const int ray = (ray_raw < ka.num_rays) ? ray_raw : ka.num_rays - 1;     // tail slots redo the last ray: same values to the same rows
    RayConst rc;
    float vdir[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rc.o[c] = ka.origins[3 * (size_t)ray + c];
      rc.d[c] = ka.directions[3 * (size_t)ray + c];
      vdir[c] = (ka.viewdirs ? ka.viewdirs : ka.directions)[3 * (size_t)ray + c];
    }
    rc.gt_mask = (ka.gt_mask != nullptr) ? ka.gt_mask[ray] : 0.f;
    if (q == 0) {
      uint32_t wid = (G::HAS_WARP && ka.warp_id != nullptr) ? ka.warp_id[ray] : 0u;
      wid = wid < (uint32_t)ka.num_embeds ? wid : (uint32_t)(ka.num_embeds - 1);
      if (lane < 8) {
        L.rayc[RC_WEMB + lane] = G::HAS_WARP ? ka.warp_embed[(size_t)wid * 8 + lane] : 0.f;
        L.rayc[RC_MEMB + lane] = G::HAS_MASK ? ka.mask_embed[(size_t)wid * 8 + lane] : 0.f;
      }
      if (lane < 24) {
        const int band = lane / 6, sc = (lane % 6) / 3, ch = lane % 3;
        const float vdc = ch == 0 ? vdir[0] : (ch == 1 ? vdir[1] : vdir[2]);
        L.rayc[RC_VDENC + lane] = sin_cw(fmaf(vdc, (float)(1 << band), sc ? 1.57079637f : 0.0f));
      }
      if (lane < 3) L.rayc[RC_VD + lane] = lane == 0 ? vdir[0] : (lane == 1 ? vdir[1] : vdir[2]);
      for (int i = lane; i < S; i += 64) L.zs[i] = to.z[(size_t)ray * S + i];
    }
    ray_sync();
    for (int sb = 0; sb < S; sb += BATCH) {
      Samples<NT> sm;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int sx = sb + 32 * NT * q + 32 * nt + (lane & 31);
        sm.slot[nt] = sx < S ? sx : S - 1;       // tail lanes repeat sample S - 1: same values, same rows
        sm.z[nt] = L.zs[sm.slot[nt]];
      }
      const size_t row = (size_t)ray * S + (size_t)sm.slot[0];     // this lane's row of the [R * S][width] activation arrays
      if constexpr (MODE != FWD_NERF_ONLY) {
        if constexpr (PP::HAS_SHARED) { pipe.cur = rs_shared; pipe.next = (MODE == FWD_SHARED_ONLY) ? rs_shared : rs_nerf; }
        eval_shared<G, PL, NT, WaveLds, TrainOut>(ka, rc, pipe, lane, sm, L, to, row);
      } else {
        // what eval_shared would have parked for this sample, from the gathered arrays (every lane, unconditionally: the two lane halves and
        // the clamped tail lanes hold the same values); the screw axis exactly as eval_shared derives it from the head outputs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int s = sm.slot[nt];
          const size_t r = (size_t)ray * S + (size_t)s;
          float w[3] = {to.in_wv[6 * r], to.in_wv[6 * r + 1], to.in_wv[6 * r + 2]};
          const float theta = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
          w[0] /= theta; w[1] /= theta; w[2] /= theta;
          float st, ct;
          sincos_cw(theta, st, ct);
#pragma unroll
          for (int c = 0; c < 3; ++c) { L.sv[SV_WP + c][s] = to.in_xw[3 * r + c]; L.sv[SV_AX + c][s] = w[c]; }
          L.sv[SV_WP + 3][s] = to.in_wamb[2 * r]; L.sv[SV_WP + 4][s] = to.in_wamb[2 * r + 1];
          L.sv[SV_SN][s] = st;
          L.sv[SV_OMC][s] = 1.0f - ct;
        }
      }
      if constexpr (MODE != FWD_SHARED_ONLY) {
        pipe.cur = rs_nerf;
        pipe.next = (MODE == FWD_NERF_ONLY) ? rs_nerf : rs_shared;
        eval_nerf<G, PL, NT, WaveLds, TrainOut>(ka, pipe, 0, lane, sm, L, to, row);
      }
    }
    ray_sync();
  }
}

// the trainer's arithmetic: 16-bit split operands everywhere, fp32 products in the warp field
using KernelPlan = PlanT<TRAIN_PLAN.mask, TRAIN_PLAN.warp, TRAIN_PLAN.hyp, TRAIN_PLAN.trunk, TRAIN_PLAN.rgb>;
// MODE 1 (the shared networks alone, 64 - 128 wide): their activations fit 256 registers, so EIGHT waves - two per SIMD, two per ray - share the
// workgroup's weight ring and the stream is walked once per 256 samples instead of once per 128 (the narrow backward chains' change).
// -DNERFDS_FWD_SHARED_WAVES8=0: four waves as the other modes.
#ifndef NERFDS_FWD_SHARED_WAVES8
#define NERFDS_FWD_SHARED_WAVES8 1
#endif
struct SharedOnlyPlan : KernelPlan { static constexpr bool EIGHT_WAVES = NERFDS_FWD_SHARED_WAVES8 != 0; };
template <int MODE> using PlanOfMode = std::conditional_t<MODE == FWD_SHARED_ONLY, SharedOnlyPlan, KernelPlan>;
}  // namespace nerfds

template <bool WIDE, int MODE> static void launch_train(const nerfds::KArgs& ka, const nerfds::TrainOut& to, int num_cus, void* stream) {
  using namespace nerfds;
  using PLM = PlanOfMode<MODE>;
  using SH = Shape<PLM, WIDE>;
  constexpr int lds = BIAS_OFF + bias_bytes<NERFDS_GRAPH>() + SH::RAYS * (int)sizeof(WaveLdsT<SH::MAXS>);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto kern = train_forward_kernel<NERFDS_GRAPH, PLM, WIDE, TRAIN_TAG, MODE>;
  allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds);
  const long long groups = ((long long)ka.num_rays + SH::RAYS - 1) / SH::RAYS;
  const int grid = (int)(groups < num_cus ? groups : num_cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wg_waves<PLM>()), lds, static_cast<hipStream_t>(stream), ka, to);
}
// ka.nc = samples of the level (ka.nf unused); ka.wstream[1] / ka.bias[1] = the level's NerfMLP; to.mode = FWD_* (the partial modes exist in the
// f16-store build only: the merged step is a plain step)
extern "C" void NERFDS_CAT(nerfds_launch_, NERFDS_NAME)(const nerfds::KArgs& ka, const nerfds::TrainOut& to, int num_cus, void* stream) {
  // WIDE: 2 rays per workgroup and twice the waves per ray (built for Nc + Nf > 128) - and ALSO the shape of a SMALL batch: with 4 rays per workgroup
  // a batch of 512 rays (the reference's own, configs/nerf_ds.gin:4) fills 128 of the 256 CUs, each walking the weight stream twice per level; as 256
  // workgroups of 2 rays every CU walks it once (measured at 512 rays: DESIGN 11.5).  The shared-networks-only mode keeps its shape (its 8-wave
  // workgroup covers a ray's 64 new samples in one walk already).  Same arithmetic per sample in either shape.
  const bool few = 2LL * ((ka.num_rays + 3) / 4) <= (long long)num_cus;
  const bool wide_s = ka.nc > nerfds::Shape<nerfds::KernelPlan, false>::MAXS;      // (MAXS does not depend on the wave count)
  const bool wide = wide_s || few;
  if constexpr (nerfds::TRAIN_HALF) {
    if (to.mode == nerfds::FWD_SHARED_ONLY) { if (wide_s) launch_train<true, nerfds::FWD_SHARED_ONLY>(ka, to, num_cus, stream); else launch_train<false, nerfds::FWD_SHARED_ONLY>(ka, to, num_cus, stream); return; }
    if (to.mode == nerfds::FWD_NERF_ONLY) { if (wide) launch_train<true, nerfds::FWD_NERF_ONLY>(ka, to, num_cus, stream); else launch_train<false, nerfds::FWD_NERF_ONLY>(ka, to, num_cus, stream); return; }
  }
  if (wide) launch_train<true, nerfds::FWD_FULL>(ka, to, num_cus, stream);
  else launch_train<false, nerfds::FWD_FULL>(ka, to, num_cus, stream);
}
