// The per-sample field of NeRF-DS on gfx950 (MI355X / CDNA4): [MaskMLP -> SE(3) warp MLP + exp_se3 -> hyper-sheet MLP -> trunk /
// sigma / rgb NerfMLP] on batches of 32-sample tiles, as register-chained transposed MFMA layers fed by one LDS-staged weight stream.
// Shared by the three kinds of fused kernels, one translation unit each:
//   render_kernel.hip      (NERFDS_KERNEL_KIND 0)  the ray kernel: sampling -> field -> compositing -> resample -> field -> compositing
//   train_fwd_kernel.hip   (NERFDS_KERNEL_KIND 1)  the trainer's forward of one level (the same field + activation stores)
//   train_bwd_kernel.hip   (NERFDS_KERNEL_KIND 2)  the trainer's data-gradient chains (the reversed MLPs on the same machinery)
//
// Reference functions restated here (paths under /root/reference/hypernerf/):
//   posenc / posenc_window model_utils.py:398-436                normalize_vector model_utils.py:438-442
//   MLP modules.py:57-83   NerfMLP.query_* modules.py:243-313   HyperSheetMLP modules.py:367-392   MaskMLP 409-434
//   SE3Field.warp warping.py:200-237   exp_se3/exp_so3/skew rigid_body.py:26-101
//   NerfModel.render_samples models.py:867-1417 (the per-sample part)
//
// MFMA mapping (the point of the design).  Every dense layer is computed TRANSPOSED:
//     H^T[out][sample] = W^T[out][k] * X^T[k][sample]
// with the weights as the MFMA A operand (32 output features x 16 k-slots per fragment, streamed from
// L2 in the exact lane order) and the activations as the B operand (16 k-slots x 32 samples).  With
// v_mfma_f32_32x32x16_bf16 the accumulator of lane l holds, for sample (l & 31), the output features
//     row(r, l) = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),   r = 0..15
// and the B operand of lane l must hold, for sample (l & 31), k-slots 8 * (l >> 5) + 0..7.  So the
// 16 accumulator registers of an output tile ARE, after bias/ReLU and a pack to bf16, two B operands
// of the next layer (registers 0-7 and 8-15), provided the next layer's weight fragments were packed
// with the k-slot -> feature permutation   feature = 32*tile + 16*c + (i & 3) + 8 * (i >> 2) + 4 * h
// (c = chunk within the tile, h = lane half, i = element).  The host packer does that once
// (nerfds_host.cpp), so activations never leave registers between layers: no LDS round trip, no
// transposes, no barriers.  The same holds for v_mfma_f32_32x32x2_f32 (fp32-exact mode, one k-slot
// pair per instruction) and for the split-bf16 "bf16x3" mode (hi/lo operands, three MFMAs per product).
//
// A wave carries NT "N-tiles" of 32 samples (NT = 2 in bf16 mode: one weight fragment feeds two MFMAs).
//
// Build-time experiments of rounds 1 - 3 that are no longer in the source (ablation masks, pinned asm epilogue pieces, carried
// tile groups, sign-bit ReLU masks, the dot2 residual, ...) are listed with their measurements in profiles/r3_ab/README.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "graphs.h"
#include "kargs.h"
#include "philox.h"

namespace nerfds {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define DEVI __device__ __forceinline__
#ifndef NERFDS_KERNEL_KIND
#error "include field.h from one of the kernel translation units (NERFDS_KERNEL_KIND 0 render, 1 training forward, 2 training backward)"
#endif
// What differs between the render kernels and the training kernels is decided here, once (each choice was measured):
constexpr bool IS_TRAIN = NERFDS_KERNEL_KIND != 0;
// LDS scratch is private to a wave and a wave's DS ops complete in order: a compiler-level fence is all that is needed.
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// All LDS of the kernel is ONE array (a second __shared__ object de-pipelines LDS-DMA code, cdna guide section 5):
// [0, RING_BYTES) weight ring, then one WaveLds scratch block per wave.
extern __shared__ __attribute__((aligned(16))) char g_smem[];

// ------------------------------------------------------------------------------------------------
// Operand containers
// ------------------------------------------------------------------------------------------------
// NERFDS_X3_F16 (a compile-time fact of the translation unit; Makefile PREC_f16x3): the operands of the SPLIT arithmetic (P_BF16X3: hi + lo, three MFMAs
// per product) are f16 pairs instead of bf16 pairs - 11 + 11 significand bits instead of 8 + 8, the same MFMA count on v_mfma_f32_32x32x16_f16.  Round 4's
// precision budget measured it 20 x more accurate on composited RGB (profiles/r4_precision_budget.md); round 6 built it because split bf16 leaves 1e-4 on a
// few dozen badly conditioned rays of 480 000 in some random-init scenes where the fp32-MFMA kernel holds it (tools/parity_sweep.py).
// Range: an activation beyond 65504 is inf in this arithmetic (bf16 keeps fp32's range) - the reason split bf16 remains the other parity mode.
#ifndef NERFDS_X3_F16
#define NERFDS_X3_F16 0
#endif
#if NERFDS_X3_F16
typedef f16x8 x3x8;
#define NERFDS_X3_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
typedef bf16x8 x3x8;
#define NERFDS_X3_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
template <int P> struct Chunk;   // 16 k-slots x 32 samples of activations (this lane: 8 slots of 1 sample)
template <> struct Chunk<P_BF16> { bf16x8 v; };
template <> struct Chunk<P_BF16X3> { x3x8 hi, lo; };
template <> struct Chunk<P_F32> { float v[8]; };
template <> struct Chunk<P_F16> { f16x8 v; };
template <> struct Chunk<P_BF16X6> { bf16x8 hi, mid, lo; };

template <int P> struct WFrag;   // 32 out rows x 16 k-slots of weights (this lane: 8 slots of 1 row)
template <> struct WFrag<P_BF16> { bf16x8 v; };
template <> struct WFrag<P_BF16X3> { x3x8 hi, lo; };
template <> struct WFrag<P_F32> { f32x4 a, b; };
template <> struct WFrag<P_F16> { f16x8 v; };
template <> struct WFrag<P_BF16X6> { bf16x8 hi, mid, lo; };

// relu on raw float bits: signed-integer max with 0 (one v_max_i32; fmaxf costs a canonicalising v_max on top).
DEVI float relu_f(float x) {
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
template <int P> DEVI void make_chunk(Chunk<P>& c, const float (&x)[8]);
// Accumulator registers -> next layer's B operand, with optional ReLU.
template <int P, bool RELU> DEVI void make_act_chunk(Chunk<P>& c, const float (&x)[8]) {
  if constexpr (RELU && (P == P_BF16 || P == P_F16)) {
    // pair form: v_cvt_pk of two values, then a signed 16-bit max with 0 on the packed halves (a negative half has its sign bit set)
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef std::remove_reference_t<decltype(c.v[0])> E;
    typedef E ex2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x2 f = {x[2 * k], x[2 * k + 1]};
      const ex2 hh = __builtin_convertvector(f, ex2);
      const s16x2 z = {0, 0};
      u[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, hh), z));
    }
    c.v = __builtin_bit_cast(decltype(c.v), u);
    return;
  }
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = RELU ? relu_f(x[i]) : x[i];
  make_chunk<P>(c, y);
}
template <> DEVI void make_chunk<P_BF16>(Chunk<P_BF16>& c, const float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) c.v[i] = (__bf16)x[i];
}
// The wave's share of the weight stream's DMA offset rides in the VECTOR offset (Pipe::issue_stage).  Render kernels: bf16 13.57 -> 13.47 ms per
// 65 536 rays, split bf16 39.95 -> 39.77 (profiles/r3_ab/ab_dma_voffset.txt); the training kernels measured 1 % SLOWER with it (15.84 against
// 15.70 ms per step) and keep the scalar offsets.
constexpr bool DMA_VOFF = !IS_TRAIN;
template <> DEVI void make_chunk<P_BF16X3>(Chunk<P_BF16X3>& c, const float (&x)[8]) {
  // pair by pair, float(hi) taken from the PACKED pair's bits (shift / mask): one v_cvt_pk per pair for hi and one for lo.  Element by element
  // hipcc converts some elements twice (once in the pair, once alone for the subtraction): 10 VALU per pair instead of 6 - 7.  Same bits out.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ uh, ul;
#if NERFDS_X3_F16
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 4; ++k) {        // hi = f16(x) (round to nearest even), lo = f16(x - float(hi)): one v_cvt_pk_f16_f32 per pair each
    const f32x2 r = {x[2 * k], x[2 * k + 1]};
    const f16x2 h = __builtin_convertvector(r, f16x2);
    const f32x2 d = r - __builtin_convertvector(h, f32x2);
    uh[k] = __builtin_bit_cast(unsigned, h);
    ul[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, f16x2));
#if defined(NERFDS_EXP_X3_NOLO)
    ul[k] = 0u;
#endif
  }
#else
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 r = {x[2 * k], x[2 * k + 1]};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
    const f32x2 d = {r[0] - __builtin_bit_cast(float, hb << 16), r[1] - __builtin_bit_cast(float, hb & 0xffff0000u)};
    uh[k] = hb;
    ul[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2));
  }
#endif
  c.hi = __builtin_bit_cast(x3x8, uh);
  c.lo = __builtin_bit_cast(x3x8, ul);
}
template <> DEVI void make_chunk<P_BF16X6>(Chunk<P_BF16X6>& c, const float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 hi = (__bf16)x[i];
    const float r = x[i] - (float)hi;
    const __bf16 mid = (__bf16)r;
    c.hi[i] = hi;
    c.mid[i] = mid;
    c.lo[i] = (__bf16)(r - (float)mid);
  }
}
template <> DEVI void make_chunk<P_F32>(Chunk<P_F32>& c, const float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) c.v[i] = x[i];
}
template <> DEVI void make_chunk<P_F16>(Chunk<P_F16>& c, const float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) c.v[i] = (_Float16)x[i];      // v_cvt_pk_f16_f32: round to nearest even
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
DEVI rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), /*stride*/ 0, (int)bytes, 0x00020000);
}

template <int P> DEVI void mma(f32x16& acc, const WFrag<P>& w, const Chunk<P>& c) {
  if constexpr (P == P_BF16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.v, c.v, acc, 0, 0, 0);
  } else if constexpr (P == P_F16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(w.v, c.v, acc, 0, 0, 0);
  } else if constexpr (P == P_BF16X3) {
    // (w_hi + w_lo)(x_hi + x_lo) ~= w_hi x_lo + w_lo x_hi + w_hi x_hi ; small terms first.
    acc = NERFDS_X3_MFMA(w.hi, c.lo, acc, 0, 0, 0);
    acc = NERFDS_X3_MFMA(w.lo, c.hi, acc, 0, 0, 0);
    acc = NERFDS_X3_MFMA(w.hi, c.hi, acc, 0, 0, 0);
  } else if constexpr (P == P_BF16X6) {
    // every product of total order <= 2 of (hi + mid + lo)(hi + mid + lo), small terms first: fp32-grade (measured 1e-6 vs fp64)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, c.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, c.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, c.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, c.mid, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.mid, c.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, c.hi, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.a[i], c.v[i], acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.b[i], c.v[4 + i], acc, 0, 0, 0);
  }
}

// ------------------------------------------------------------------------------------------------
// Weight pipe.  The fragments of one evaluation form ONE static stream of 1-KiB units: [shared mask|warp|hyper
// nets] followed by [NerfMLP of the level]; a fragment is one unit (bf16 / f16) or two consecutive units (hi | lo of
// split bf16, k 0-3 | k 4-7 of fp32) at any unit position, so networks of different precision can follow each other
// in one stream (graphs.h Plan / walk_seg).  The stream is staged through an LDS ring by LDS-DMA; a register ring keeps the next RD
// units in flight LDS -> VGPR: consuming unit u issues the read of unit u + RD, whatever fragments those units belong
// to.  Every index is a compile-time constant after unrolling.
// ------------------------------------------------------------------------------------------------
// NERFDS_NT = 2 (plans of one-unit networks only): two N-tiles per wave, i.e. every weight fragment read from LDS feeds two
// MFMAs; the activations of the 256-wide trunk then need 256 registers, so one 512-register wave per SIMD.  The Makefile builds the
// nerf_ds / HyperNeRF bf16 and f16 kernels this way, with the compiler-scheduled C++ tile epilogue (history and A/B records: DESIGN 6,
// profiles/r3_ab/ab_nt2_*.txt).  The source default stays one N-tile (the static graph's kernels, every two-unit plan).
#ifndef NERFDS_NT
#define NERFDS_NT 1
#endif
template <int PM, int PW, int PH, int PT, int PR, int PTC = -1, int PRC = -1> struct PlanT {
  static constexpr int MASK = PM, WARP = PW, HYP = PH, TRUNK = PT, RGB = PR;
  // the coarse level's NerfMLP in its own arithmetic (graphs.h Plan::trunk_c): HAS_C, and what eval_nerf<..., CP = true> runs in
  static constexpr bool HAS_C = PTC >= 0;
  static constexpr int TRUNK_C = HAS_C ? PTC : PT, RGB_C = HAS_C ? PRC : PR;
  static constexpr Plan value() { return Plan{PM, PW, PH, PT, PR, PTC, PRC}; }
  static constexpr bool UNIFORM = PM == PW && PW == PH && PH == PT && PT == PR;
  // (plans that mix two-unit networks in keep one N-tile: a 128-wide split-bf16 network with two N-tiles is 256 registers of activations too)
  static constexpr int NT = (is_single(PM) && is_single(PW) && is_single(PH) && is_single(PT) && is_single(PR)) ? NERFDS_NT : 1;
  // 8 waves (two per SIMD, 256 registers each) when the 256-wide trunk runs on one-unit operands with one N-tile; a
  // trunk on two-unit operands (or two N-tiles) needs > 256 registers of activations: one 512-register wave per SIMD.
  static constexpr bool EIGHT_WAVES = is_single(PT) && NT == 1;
};
// STAGE_BYTES (graphs.h): one ring stage = 16 units
// Ring depth: 4 stages.  The training backward's chain kernels own the whole LDS and could go deeper (-DNERFDS_BWD_STAGES=8 builds it; the
// counted wait of boundary() is written for 4 .. 8 stages): measured SLOWER, 11.85 - 11.95 against 11.63 - 11.66 ms per training step
// (profiles/r4_ab/ab_bwd_ring_depth.txt: the 128-wide chains spill 33 - 41 registers with the longer unrolled walk, and a segment padded to a
// multiple of 8 stages has more barrier-only hole stages) - the chains are not waiting for the ring.
#ifndef NERFDS_BWD_STAGES
#define NERFDS_BWD_STAGES 4
#endif
constexpr int NUM_STAGES = NERFDS_KERNEL_KIND == 2 ? NERFDS_BWD_STAGES : 4;
enum { SEG_SHARED = 0, SEG_NERF = 1, SEG_NERF_C = 2 };  // the weight streams (Pipe); SEG_NERF_C: a coarse NerfMLP in its own arithmetic (PlanT::HAS_C)
constexpr int RING_BYTES = NUM_STAGES * STAGE_BYTES;
// Work shape.  NT = N-tiles (32 samples) per wave and evaluation; SPLIT = waves that share one ray's batch of
// 32 * NT * SPLIT samples; RAYS = rays in flight per workgroup (each with its own RayLds block).
//   8 waves: two per SIMD, so one wave's LDS/epilogue latency is covered by the other wave's MFMAs.
//   WIDE (Nc + Nf > 128, e.g. 128 + 128): half as many rays per workgroup, twice the waves per ray and a 256-sample
//         LDS block per ray, so the LDS footprint is unchanged.
template <class PL, bool WIDE> struct Shape {
  static constexpr int NT = PL::NT, SPLIT = (PL::EIGHT_WAVES ? 2 : 1) * (WIDE ? 2 : 1), RAYS = WIDE ? 2 : 4, MAXS = WIDE ? 256 : 128;
};
template <class PL> constexpr int wg_waves() { return Shape<PL, false>::RAYS * Shape<PL, false>::SPLIT; }
// LDS map: [0, RING_BYTES) weight ring | padded fp32 biases (shared, coarse NerfMLP, fine NerfMLP) | RAYS x WaveLds
constexpr int BIAS_OFF = RING_BYTES;
// (the training forward runs one level per launch: the shared nets and ONE NerfMLP)
template <class G> constexpr int bias_tiles() { return Dims<G>::SHARED_BIAS_TILES + (NERFDS_KERNEL_KIND == 1 ? 1 : 2) * Dims<G>::NERF_BIAS_TILES; }
template <class G> constexpr int bias_bytes() { return cdiv(bias_tiles<G>(), 8) * 1024; }

// Uniform split-bf16 plans issue the MFMAs of a tile pair interleaved (accum) and keep 8 units in the register ring (4 otherwise).
// Measured 41.6 -> 40.8 ms per 65 536 rays (same-accumulator MFMA pairs with instructions between them: 2305 -> 590 of 9384).
constexpr int RING_UNITS = 4, RING_UNITS_X3 = 8;
// Render kernels: the LDS-DMA pieces of a stage are issued one by one between the MFMAs of the stage NS - 1 earlier, with a counted
// vmcnt at the boundaries (Pipe::boundary / spread_piece); training kernels: back to back behind the barrier (their own measured scheme).
// Measured +0.9 % on the 8-wave kernels, +-0 on the 4-wave ones.
constexpr bool SPREAD_DMA = !IS_TRAIN;
// the unit of every group of SU / PIECES units behind which a piece goes out
constexpr int SPREAD_AT = 1;
// stream lengths of a graph (graphs.h) or of a reversed network of the training backward (one stream, walked as SEG_NERF)
template <class G, class = void> struct StreamUnits {
  static constexpr int shared(Plan p) { return shared_units<G>(p); }
  static constexpr int nerf(Plan p) { return nerf_units<G>(level_plan(p, 1)); }
  static constexpr int nerf_c(Plan p) { return nerf_units<G>(level_plan(p, 0)); }
};
template <class G> struct StreamUnits<G, std::void_t<decltype(G::BWD_FRAGS)>> {
  static constexpr int shared(Plan) { return 0; }
  static constexpr int nerf(Plan p) { return G::BWD_FRAGS * frag_parts(p.trunk); }
  static constexpr int nerf_c(Plan p) { return nerf(p); }
};
// waves that share the ring: the render / forward shapes (wg_waves), or what a chain graph of the training backward asks for (graphs.h BwdNet)
template <class G, class PL, class = void> struct PipeWaves { static constexpr int value = wg_waves<PL>(); };
template <class G, class PL> struct PipeWaves<G, PL, std::void_t<decltype(G::WG_WAVES)>> { static constexpr int value = G::WG_WAVES; };
template <class G, class PL> struct Pipe {
  static constexpr int SU = STAGE_UNITS;                          // units per stage
  static constexpr int NS = NUM_STAGES;
  // The fragments of the graph are TWO streams, each walked as a "segment": SEG_SHARED = [mask | warp | hyper] nets (the same
  // weights for both levels), SEG_NERF = the NerfMLP of one level.  A segment's stage count is padded to a multiple of the ring
  // depth (hole stages: a barrier, no DMA), so every segment starts in ring slot 0 and any segment can follow any other: which
  // stream the last NS - 1 boundaries of a segment prefetch from is a run-time descriptor (`next`).  Sequence per ray group:
  // shared, nerf(coarse) per coarse batch; then shared on the NEW fine samples only (the coarse samples' warp / hyper / mask
  // results are reused: same networks, same inputs - models.py:1291-1300 evaluates them again and gets the same values) and
  // nerf(fine) on every batch of the sorted union.
  static constexpr int SHARED_UNITS = StreamUnits<G>::shared(PL::value()), NERF_UNITS = StreamUnits<G>::nerf(PL::value());
  static constexpr int NERF_C_UNITS = StreamUnits<G>::nerf_c(PL::value());      // the coarse level's NerfMLP stream (== NERF_UNITS unless PL::HAS_C)
  static constexpr bool HAS_SHARED = SHARED_UNITS > 0;
  // stream positions count the zero padding at the end of each stream (graphs.h pad_units)
  static constexpr int SHARED_PAD = pad_units(SHARED_UNITS), NERF_PAD = pad_units(NERF_UNITS), NERF_C_PAD = pad_units(NERF_C_UNITS);
  static constexpr int seg_units(int seg) { return seg == SEG_SHARED ? SHARED_UNITS : (seg == SEG_NERF_C ? NERF_C_UNITS : NERF_UNITS); }
  static constexpr int seg_used(int seg) { return pad_units(seg_units(seg)) / SU; }                          // stages that hold data
  static constexpr int seg_stages(int seg) { return cdiv(seg_used(seg), NS) * NS; }                          // incl. hole stages
  static_assert((!HAS_SHARED || seg_used(SEG_SHARED) >= NS - 1) && seg_used(SEG_NERF) >= NS - 1 && seg_used(SEG_NERF_C) >= NS - 1,
                "the wrap prefetch needs NS - 1 stages in every segment");
  static constexpr int WAVES = PipeWaves<G, PL>::value;
  static constexpr int PIECES = SU / WAVES;                       // 1 KiB LDS-DMA pieces per wave per stage
  // LDS -> register prefetch distance in units: a ds_read_b128 takes ~100+ cycles to return under load, a bf16 unit is
  // consumed in 32-64 MFMA cycles, so the reads must run several units ahead of the MFMAs.
  static constexpr bool X3_INTERLEAVE = PL::UNIFORM && PL::TRUNK == P_BF16X3 && PL::NT == 1;
  static constexpr int RD = X3_INTERLEAVE ? RING_UNITS_X3 : RING_UNITS;
  // The counted vmcnt of boundary() relies on every piece of stage s + 2 having been issued BEFORE boundary(s) runs.  The pieces of stage
  // s + 2 are triggered by units of stage s - 1, the last of them by unit T = SU - SU / PIECES + SPREAD_AT of that stage; a group of units that
  // is consumed as one step (a fragment of up to 3 units at any position; the 4 units of an interleaved split-bf16 tile pair, always at an even
  // position) runs its boundary first and its triggers after its MFMAs - so no group may hold both unit T of one stage and unit 0 of the next.
  static constexpr bool spread_safe(int group_units, int align) {
    const int every = SU / PIECES, t = SU - every + SPREAD_AT % every;
    for (int u = SU - group_units + 1; u <= t; ++u)
      if (u % align == 0) return false;
    return true;
  }
  static_assert(!SPREAD_DMA || (spread_safe(3, 1) && (!X3_INTERLEAVE || spread_safe(4, 2))),
                "a spread trigger unit may not share a consumption step with the first unit of the next stage");
  static_assert(RD <= SU && RD >= 2, "prefetch reaches at most one stage ahead");
  u32x4 ring[RD];
#ifdef NERFDS_PROF
  unsigned long long t_chain = 0, t_eval = 0, t_ray = 0;      // measurement build: cycles inside the layer chains / the field evaluations / compositing + resampling
  unsigned long long t_wait = 0, t_bar = 0, n_bound = 0;       // -DNERFDS_PROF_BOUND: cycles in the stage boundaries' s_waitcnt / s_barrier (+ one timer read each), boundaries
#endif
  rsrc_t cur;       // stream of the segment being walked
  rsrc_t next;      // stream of the segment walked next (wrap-around prefetch)
  int lane16;
  int wave1k;       // wave index in the workgroup * 1024 (SGPR)

  // This wave's share of stage t of segment `seg` (t >= seg_stages: stage t - seg_stages of the next segment): LDS-DMA
  // (buffer_load ... lds), 1 KiB per instruction, no VGPRs.  Everything but "+ wave * 1024" (one s_add) and the descriptor is a
  // compile-time fact.
  DEVI void issue_stage(int seg, int t, int k0 = 0, int k1 = PIECES) {
    const bool wrap = t >= seg_stages(seg);
    const int tt = wrap ? t - seg_stages(seg) : t, slot = t % NS;
    if (!wrap && tt >= seg_used(seg)) return;                      // hole stage
    const int base = tt * STAGE_BYTES;
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      // readfirstlane makes the uniformity of the scalar operands provable: without it hipcc may keep them in
      // VGPRs under SGPR pressure and wrap every LDS-DMA in a waterfall loop (cdna guide T20).
      const int off = __builtin_amdgcn_readfirstlane(WAVES * k * 1024 + wave1k);
      auto dst = (__attribute__((address_space(3))) void*)(g_smem + slot * STAGE_BYTES + off);
      // DMA_VOFF: the wave's share of the stream offset rides in the VECTOR offset (lane * 16 + wave * 1024, one register for the whole kernel), so the
      // scalar offset of every piece of every stage is a literal: as `constant + wave1k` each of the ~300 (stage, piece) offsets was a loop-invariant scalar
      // that hipcc hoisted out of the persistent loop and then spilled to VGPR lanes (586 SGPR spill slots, 0.3 v_readlane / v_writelane per MFMA)
      if constexpr (DMA_VOFF) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrap ? next : cur, dst, 16, lane16 + wave1k, base + WAVES * k * 1024, 0, 0);
      else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrap ? next : cur, dst, 16, lane16, base + off, 0, 0);
    }
  }
  // Entering stage s: stages s and s + 1 are complete in LDS (so the LDS->register prefetch can run ahead across the
  // next boundary without a cold start), stage s + 2 may still be in flight, every wave is done with stage s - 1,
  // whose slot is refilled with stage s + NS - 1.
  DEVI void boundary(int seg, int s) {
    // What the wait must guarantee: every LDS-DMA of stages s and s + 1 (this wave's share; the barrier extends it to the workgroup) has landed.
    // COUNTED wait, vmcnt(PIECES).  The pieces of stage s + 2 were the last loads this wave issued - during stage s - 1 (spread DMA), or back to
    // back behind the previous boundary - and this boundary does not need them.  Loads complete in order AMONG LOADS, so "at most PIECES
    // operations outstanding" means every load older than the PIECES youngest loads has landed, i.e. stages <= s + 1; stores in flight (ray
    // records, the training kernels' activation stores, spills - they share VM_CNT and retire out of order with respect to loads) or later loads
    // only make the wait longer, never shorter.  (Round 1 saw stale weights, 2e-2 errors on the fine level, with a counted wait whose count
    // assumed stores and loads retire in ONE order; the count here makes no such assumption.)  The argument needs stage s + 2 to have been
    // issued before this point (see the static_assert at RD): when it is a hole (nothing issued) the wait is vmcnt(0).  Measured against
    // vmcnt(0): -0.75 ms per training step, +0.9 % with the spread issue on the 8-wave render kernels.
    // lgkmcnt(0): every LDS read this wave has issued has returned before the barrier.  In program order the reads of the
    // retiring stage s - 1 are all consumed by MFMAs above this point, but hipcc may sink a register-only MFMA - and with it the
    // s_waitcnt of its operand read - BELOW the asm and the barrier; another wave's DMA into that slot would then race the read.
    // (Waiting on vmcnt only was 1.5 % faster and ran clean on the uniform kernels, but the mixed-precision kernel showed
    // run-to-run differences with it: kept safe.)
    // With a ring of NS stages the stages s + 2 .. s + NS - 2 may be in flight (NS = 4: the one stage issued since the previous boundary); those of
    // them that are holes issued nothing and do not count.
    static_assert(NS >= 4 && NS <= 8 && (PIECES == 4 || PIECES == 2), "counted wait: at most 5 stages of PIECES loads in flight");
    int young = 0;
#pragma unroll
    for (int t = s + 2; t <= s + NS - 2; ++t)
      if (t >= seg_stages(seg) || t < seg_used(seg)) young += PIECES;
#ifdef NERFDS_PROF_BOUND
    unsigned long long pb0, pb1, pb2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pb0) :: "memory");      // (the ring reads in flight are drained HERE in this build: t_wait = the vmcnt part + one timer read)
#endif
    // (A counted lgkmcnt in the pinned chains - the 8 youngest ring reads left in flight, safe there because program order is source order -
    // measured +-0 against the full drain: 38.25 against 38.26 ms per 65 536 rays, profiles/r4_ab/ab_x3_pin_variants.txt; not kept.)
#if defined(NERFDS_EXP_LOOSE_VMCNT)
    // measurement build (UNSAFE: weights may be read before they land - timing only, tools/variant.sh): the boundary does not wait for vector-memory
    // operations at all - what the counted vmcnt wait, which on gfx950 also covers every STORE issued since, costs the kernels that store
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (false)
#endif
    if (young == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (young == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else if (young == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else if (young == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else if (young == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (young == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
    else if (young == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (young == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (young == 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    else __builtin_trap();                       // (a count this list does not spell out)
#ifdef NERFDS_PROF_BOUND
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pb1) :: "memory");
#endif
    __builtin_amdgcn_s_barrier();      // raw barrier (no compiler-added fences)
#ifdef NERFDS_PROF_BOUND
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pb2) :: "memory");
    t_wait += pb1 - pb0; t_bar += pb2 - pb1; n_bound += 1;
#endif
    // SPREAD_DMA: the pieces of stage s + NS - 1 are issued one by one between the MFMAs of stage s (spread_piece) instead of
    // back to back behind the barrier, where nothing covers their issue time (a hole stage has no MFMAs: issued here)
    if (!SPREAD_DMA || s >= seg_used(seg)) issue_stage(seg, s + NS - 1);
  }
  // unit u of the segment has been consumed: with SPREAD_DMA, the k-th piece of the stage NS - 1 ahead goes out after the
  // unit SPREAD_AT + k * (SU / PIECES) of the current stage (behind the barrier of this stage: its ring slot is free)
  DEVI void spread_piece(int seg, int u) {
    if (!SPREAD_DMA) return;
    constexpr int EVERY = SU / PIECES;
    if (u % EVERY == SPREAD_AT % EVERY) issue_stage(seg, u / SU + NS - 1, (u % SU) / EVERY, (u % SU) / EVERY + 1);
  }
  // the first NS - 1 stages of the first segment of the kernel
  DEVI void prologue(int seg) {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue_stage(seg, t);
  }
  DEVI u32x4 unit(int u) const {
    const int off = ((u / SU) % NS) * STAGE_BYTES + (u % SU) * 1024;
    return *reinterpret_cast<const u32x4*>(g_smem + off + lane16);
  }
  DEVI void begin_stage(int seg, int u0) {   // u0: first unit of the stage
    boundary(seg, u0 / SU);
    if (u0 == 0) {                           // cold start of a segment: fill the register ring
#pragma unroll
      for (int d = 0; d < RD; ++d) ring[d % RD] = unit(d);
    }
  }
  // unit u has been consumed (or skipped): its ring slot takes unit u + RD (stage (u / SU) + 1 is resident)
  DEVI void refill(int seg, int u) {
#if defined(NERFDS_EXP_LO)
    // measurement builds (wrong or unchecked results, never shipped): what the LDS read of every second unit costs (1: not read at all), and the
    // same unit fetched with a vector load from the stream in L2 instead of from its LDS copy (2)
    if (X3_INTERLEAVE && (u & 1)) {
      if (NERFDS_EXP_LO == 2 && (u + RD) / SU < seg_used(seg))
        ring[u % RD] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(cur, lane16, (u + RD) * 1024, 0));
      return;
    }
#endif
    if ((u + RD) / SU < seg_used(seg)) ring[u % RD] = unit(u + RD);
  }
  template <int P> DEVI WFrag<P> frag(int u) const {
    WFrag<P> w;
    if constexpr (P == P_BF16) {
      w.v = __builtin_bit_cast(bf16x8, ring[u % RD]);
    } else if constexpr (P == P_F16) {
      w.v = __builtin_bit_cast(f16x8, ring[u % RD]);
    } else if constexpr (P == P_BF16X3) {
      w.hi = __builtin_bit_cast(x3x8, ring[u % RD]);
      w.lo = __builtin_bit_cast(x3x8, ring[(u + 1) % RD]);
#if defined(NERFDS_EXP_X3_NOLO)
      if (NERFDS_EXP_X3_NOLO == 2) w.lo = x3x8{};
#endif
    } else if constexpr (P == P_BF16X6) {
      static_assert(RD >= 4 || P != P_BF16X6, "three units of one fragment and one of the next");
      w.hi = __builtin_bit_cast(bf16x8, ring[u % RD]);
      w.mid = __builtin_bit_cast(bf16x8, ring[(u + 1) % RD]);
      w.lo = __builtin_bit_cast(bf16x8, ring[(u + 2) % RD]);
    } else {
      w.a = __builtin_bit_cast(f32x4, ring[u % RD]);
      w.b = __builtin_bit_cast(f32x4, ring[(u + 1) % RD]);
    }
    return w;
  }
  DEVI void finish_segment(int seg) {
    if (SPREAD_DMA) {             // pieces whose trigger unit lies in the zero padding of the last stage
      constexpr int EVERY = SU / PIECES;
      const int last = seg_units(seg) - 1, s = seg_used(seg) - 1;
#pragma unroll
      for (int k = 0; k < PIECES; ++k)
        if (s * SU + k * EVERY + SPREAD_AT % EVERY > last) issue_stage(seg, s + NS - 1, k, k + 1);
    }
    // boundaries of the hole stages keep the barrier count and the ring in step
#pragma unroll
    for (int s = seg_used(seg); s < seg_stages(seg); ++s) boundary(seg, s);
  }
};

struct Cursor {
  int seg;      // SEG_SHARED / SEG_NERF: the stream being walked
  int pos;      // position (unit index) of the next fragment in that stream
  int bt;       // index of the next bias tile (0 = first tile of the shared nets)
};
// Training kernels (train_fwd_kernel.hip, train_bwd_kernel.hip).  The Makefile builds each of them twice; what they store is a compile-time fact:
//   -DNERFDS_TRAIN_HALF=0                        fp32 activations / fp32 g (the steps with a tangent pass; NERFDS_TRAIN_G16=0)
//   -DNERFDS_TRAIN_HALF=1 -DNERFDS_TRAIN_PIPE=1  f16 activations + ReLU bits / scaled f16 g, a tile group's epilogue issued inside the next group's
//                                                MFMA chain (dense(): TRAIN_PIPE branch) - the plain training step
// and the host picks the launcher (TrainOut::half_out / TrainBwd::g_half).  As a run-time test (a uniform branch per tile pair) the mode cut
// every tile group into its own basic blocks, and hipcc schedules within a block; compile-time alone is neutral (15.81 against 15.88 ms per
// step), together with the source-level pipeline and one scheduling region per group it is 15.58 (profiles/r3_ab/ab_tph.txt).
// TRAIN_TAG is a template argument of the training kernels: two translation units that build the SAME template with different
// macros would otherwise emit one mangled kernel name with two bodies, and the runtime binds both launchers to ONE of them (this cost
// round 3 a wrong conclusion: the first two-build attempt silently ran the fp32-store kernel for both modes - a step that barely
// learns, loss 0.17302 instead of 0.1674 - and looked like "compile-time half is slower").
#if NERFDS_KERNEL_KIND != 0
#if !defined(NERFDS_TRAIN_HALF) || !defined(NERFDS_TRAIN_PIPE)
#error "training kernels: compile with -DNERFDS_TRAIN_HALF=<0|1> -DNERFDS_TRAIN_PIPE=<0|1>"
#endif
constexpr bool TRAIN_HALF = NERFDS_TRAIN_HALF != 0, TRAIN_PIPE = NERFDS_TRAIN_PIPE != 0;
#else
constexpr bool TRAIN_HALF = false, TRAIN_PIPE = false;
#endif
constexpr int TRAIN_TAG = 1 + (TRAIN_HALF ? 1 : 0) + 2 * (TRAIN_PIPE ? 1 : 0);
// Training forward: every hidden layer also writes its post-activation output to HBM, row-major [sample][width], for the backward pass:
// fp32 (`row` = this lane's sample row of the layer being computed, + 4 * (lane >> 5) floats) or, TRAIN_HALF, f16 (`row16` = this lane's
// row + ROW16_H * (lane >> 5) halves) plus one "output > 0" bit per feature (`bits` = this lane's u16 run of the layer: one u16 per
// 32-feature tile, bit r = accumulator r).
struct TrainCursor : Cursor {
  float* row;
  uint16_t* row16;
  uint16_t* bits;
};
// Fused backward (train_backward_kernel): the tiles of a hidden layer are masked with the forward's ReLU bits (`mask`: the lane's
// u16 per tile, two tiles per register) and stored as g[sample][width] (fp32, or bf16 with TRAIN_HALF); the input-gradient tiles are
// stored (or added) into a buffer whose row stride `ld_in` is not a multiple of 32 - lanes past it write to `sink`.
struct BwdCursor : TrainCursor {
  unsigned mask[8];
  float* in_row;        // this lane's row of d_in + 4 * (lane >> 5)
  float* sink;
  int ld_in, in_h4;     // in_h4 = 4 * (lane >> 5)
  int in_acc;           // add to what is there (the skip layer's contribution came first)
  float in_scale;       // TrainBwd::g_inv_scale: the chain's values are g_scale * g, the raw-input gradient leaves unscaled
  int live;             // 0: a tail lane that repeats the last row - its read-modify-write of d_in must not touch the row (it goes to the sink)
};
// the same cursor while the tiles being computed are the gradient of the raw input (a TYPE, so that dense() selects the epilogue at
// compile time: a run-time test of the per-lane row pointer is a divergent branch to the compiler, and divergent regions in the
// evaluation are where this hipcc misplaces live-range-split copies - see eval_shared)
struct BwdInCursor : BwdCursor {};
// Accumulator registers 4g .. 4g + 3 of a lane are output features 32 * tile + 8g + 4h + 0..3 of its sample: four 16-byte stores.
// (Tried and measured, DESIGN 8.1: staging the tile through LDS so that every store instruction writes whole 128-byte lines,
// non-temporal stores, a quarter of the bytes per line - none of them changes the cost of the stores, ~3.5 ms per step on top of
// 2.7 ms of arithmetic; only writing every layer into one L2-resident array does.)
template <bool RELU> DEVI void store_tile(float* row_tile, const f32x16& acc) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = RELU ? relu_f(acc[4 * g + i]) : acc[4 * g + i];
    *reinterpret_cast<f32x4*>(row_tile + 8 * g) = v;
  }
}
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#if defined(NERFDS_EXP_NOSTORE)
__device__ uint16_t g_exp_store_sink[32 * 256];      // (measurement builds only: where variant 2 sends every tile store)
#endif
// A tile of 16-bit values (f16 activations, f16 g): pk[2g], pk[2g + 1] = this lane's features 8g + 4h + 0..3 of its sample, as packed pairs.
// The two lanes of a sample (l, l + 32) trade halves with v_permlane32_swap - afterwards lane half h owns features 16h .. 16h + 15 of the
// tile - and each lane writes TWO 16-byte pieces instead of four 8-byte ones (`row16` points at the lane's feature 16h of tile 0).  A training
// kernel runs one wave per SIMD and a global store occupies the wave for its whole issue (address + data transfer of 64 lanes), MFMA pipe
// idle: what the stores cost is their NUMBER, not their bytes.
// NERFDS_STORE_SECTOR (round 6 experiment, default 0 - measured +-0: 11.35 - 11.50 against 11.47 - 11.51 ms per step): WHICH 16 features a lane half ends up with.  0 (shipped): half h owns features 16h .. 16h + 15 - its
// two stores go to bytes [32h, 32h + 16) and [32h + 16, 32h + 32) of the row's 64-byte tile segment, so each store INSTRUCTION leaves two 16-byte pieces 32
// bytes apart in every row.  1: half h owns features 8h .. 8h + 7 and 16 + 8h .. 16 + 8h + 7 - the first instruction writes bytes [0, 32) of the segment (half 0
// the first 16, half 1 the next), the second [32, 64): every instruction writes whole 32-byte sectors.  Same bytes in the same places either way.
#ifndef NERFDS_STORE_SECTOR
#define NERFDS_STORE_SECTOR 0
#endif
constexpr int ROW16_H = NERFDS_STORE_SECTOR ? 8 : 16;      // offset of lane half 1 in a row of 16-bit features
constexpr int ROW16_2ND = NERFDS_STORE_SECTOR ? 16 : 8;   // offset of a lane's second 16-byte piece
DEVI void store_tile_pk16(uint16_t* row_tile, const unsigned (&pk)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned x[4], y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // v_permlane32_swap(a, b): a of lanes 32..63 <-> b of lanes 0..31.  SECTOR 0: (pk[i], pk[4 + i]) - half 0 ends with (own, partner's) features of groups
    // 0 and 1, half 1 with (partner's, own) features of groups 2 and 3.  SECTOR 1: (pk[i], pk[2 + i]) and (pk[4 + i], pk[6 + i]) - half 0 ends with groups 0
    // and 2, half 1 with groups 1 and 3 (a group = 8 consecutive features, its first four in half 0's registers, its last four in half 1's).
    const int ia = NERFDS_STORE_SECTOR ? (i < 2 ? i : 2 + i) : i, ib = NERFDS_STORE_SECTOR ? ia + 2 : 4 + i;
    const u32x2 r = __builtin_amdgcn_permlane32_swap(pk[ia], pk[ib], false, false);
    x[i] = r[0]; y[i] = r[1];
  }
  const u32x4 s0 = {x[0], x[1], y[0], y[1]}, s1 = {x[2], x[3], y[2], y[3]};
#if defined(NERFDS_EXP_NOSTORE)
  // measurement builds (wrong results, never shipped; tools/variant.sh): 1 = the tile stores are not issued at all (what they cost in total: issue slots
  // + what the stage boundaries' vmcnt waits spend on them); 2 = issued, but all to one 2-KiB sink (same instructions, no HBM
  // write stream behind them: what is left is the issue cost)
  if (NERFDS_EXP_NOSTORE == 1) { asm volatile("" :: "v"(s0), "v"(s1)); return; }
  // (3: the sink addressed like the rows of a 256-wide array - lane l writes row l & 31, half l >> 5: 16-byte pieces 512 bytes apart, the shipped
  // scatter pattern without the HBM stream)
  row_tile = NERFDS_EXP_NOSTORE == 3 ? g_exp_store_sink + (threadIdx.x & 31) * 256 + ((threadIdx.x >> 5) & 1) * 16 : g_exp_store_sink + (threadIdx.x & 63) * 16;
#endif
#if defined(NERFDS_EXP_NT_STORE)
  __builtin_nontemporal_store(s0, reinterpret_cast<u32x4*>(row_tile));
  __builtin_nontemporal_store(s1, reinterpret_cast<u32x4*>(row_tile + ROW16_2ND));
#else
  *reinterpret_cast<u32x4*>(row_tile) = s0;
  *reinterpret_cast<u32x4*>(row_tile + ROW16_2ND) = s1;
#endif
#endif
}
// The same tile as f16 (round to nearest even; an activation beyond 65504 becomes inf and shows up as an inf weight gradient - the trainer
// checks the gradient vector for non-finite values after every step, nerfds_train.cpp); returns the tile's 16 ReLU bits, bit r <->
// register r.  Two VALU per bit: the relu'd value has non-negative integer bits, so 0 - bits is negative exactly when the output is > 0,
// and v_alignbit shifts that sign bit in (registers 15 .. 0).
template <bool RELU> DEVI unsigned store_tile_half(uint16_t* row_tile, const f32x16& acc) {
  unsigned bits = 0;
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = RELU ? relu_f(acc[r]) : acc[r];
#pragma unroll
  for (int r = 15; r >= 0; --r) {
    const unsigned neg = 0u - __builtin_bit_cast(unsigned, v[r]);
    bits = __builtin_amdgcn_alignbit(bits, neg, 31);                              // (bits << 1) | (neg >> 31)
  }
  // (as vectors, like make_chunk<P_F16>: one v_cvt_pk_f16_f32 per pair; scalar converts + shift + or took three VALU per pair)
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)v[i]; b[i] = (_Float16)v[8 + i]; }
  const u32x4 pa = __builtin_bit_cast(u32x4, a), pb = __builtin_bit_cast(u32x4, b);
  const unsigned pk[8] = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
  store_tile_pk16(row_tile, pk);
  return bits;
}
// The g arrays of the fused backward as f16 (TRAIN_HALF): the masked accumulators - g_scale * g, see TrainBwd - rounded to nearest even, one
// v_cvt_pk_f16_f32 per pair.  (Through round 3 the stored copy was the hi part of the tile's split-bf16 chunks - free, but bf16: the
// weight-gradient kernels then had to turn the f16 activations into bf16 hi + lo for every tile and spend two MFMAs per product; with both
// operands f16 they read them from the DMA'd stage as they are.)  A value beyond 65504 becomes inf and is reported like an overflowing
// activation (store_tile_half).
DEVI void store_tile_g16(uint16_t* row_tile, const f32x16& acc) {
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)acc[i]; b[i] = (_Float16)acc[8 + i]; }
  const u32x4 pa = __builtin_bit_cast(u32x4, a), pb = __builtin_bit_cast(u32x4, b);
  const unsigned pk[8] = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
  store_tile_pk16(row_tile, pk);
}
// Fused backward: zero the accumulator registers whose ReLU bit is clear.  (Written as a select on purpose: the 2-VALU forms
// "x & sign-extended bit" - through __builtin_amdgcn_sbfe, an opaque v_bfe_i32, or `bits(acc[r]) & m` in C++ - are folded wrongly by
// hipcc 7.2: every register came out anded with register 0's value; found with tools/chain_diag.py.  The fold is in `extractelement +
// bitcast + and`.  The chains are not VALU-bound anyway: 35 % fewer VALU per MFMA in them changed their time by +-0.)
DEVI void apply_mask(f32x16& acc, unsigned bits16) {
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = ((bits16 >> r) & 1u) ? acc[r] : 0.f;
}
// Input-gradient tile: features 32 t + 8 g + 4 h + 0..3 of the row, those below ld_in only (the others go to the sink: no
// divergent store, see eval_shared).
DEVI void store_tile_in(const BwdCursor& cur, int tile, const f32x16& acc) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int n0 = 32 * tile + 8 * g + cur.in_h4;
    float* dst = cur.in_row + (32 * tile + 8 * g);
    const bool ok = (n0 < cur.ld_in) & (cur.live != 0);
    dst = ok ? dst : cur.sink;
    f32x4 v = {acc[4 * g] * cur.in_scale, acc[4 * g + 1] * cur.in_scale, acc[4 * g + 2] * cur.in_scale, acc[4 * g + 3] * cur.in_scale};
    if (cur.in_acc) v += *reinterpret_cast<const f32x4*>(dst);
    *reinterpret_cast<f32x4*>(dst) = v;
  }
}

// LDS image of the biases: tile t, lane half h, accumulator register r <-> row (r & 3) + 8 (r >> 2) + 4 h of the tile at
// byte BIAS_OFF + (t >> 3) * 1024 + (t & 7) * 64 + 512 * h + 4 * r: the 16 values of a lane are one 64-byte run, the
// half is selected by ONE per-lane address bit (lane16 & 512), everything else is an immediate offset.
constexpr int bias_tile_off(int t) { return (t >> 3) * 1024 + (t & 7) * 64; }
constexpr int bias_lds_bytes(int tiles) { return cdiv(tiles, 8) * 1024; }
// Per-lane base of the bias reads: BIAS_OFF + 512 * (lane >> 5).  Recomputed (2 VALU) at the start of every layer and
// opaque to the compiler: as a kernel-lifetime value it is the first thing the register allocator spills, and its
// reload (scratch_load + vmcnt(0)) then also waits for the LDS-DMA in flight.
DEVI int bias_base(int lane16) {
  int hb = lane16;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_and_b32 %0, 0x200, %0\n\tv_add_u32 %0, %1, %0" : "+v"(hb) : "s"(BIAS_OFF));
#else
  hb = (lane16 & 512) + BIAS_OFF;
#endif
  return hb;
}
// Bias of tile t for this lane, as an MFMA C operand.
DEVI f32x16 load_bias(int t, int hb) {
  f32x16 bv;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(g_smem + hb + bias_tile_off(t) + 16 * g);
    bv[4 * g + 0] = b[0]; bv[4 * g + 1] = b[1]; bv[4 * g + 2] = b[2]; bv[4 * g + 3] = b[3];
  }
  return bv;
}

// TP output tiles at once over one input segment (K k16-chunks of precision P): for every chunk, one fragment per tile
// from the stream, each feeding its own accumulator.  TP = 2 is the point: consecutive MFMAs of a wave then never hit
// the same accumulator, and anything issued between two MFMAs on the SAME accumulator (here: the LDS reads of the
// weight ring and their waits) costs ~43 cycles on gfx950 instead of its issue slot (MI355X_MICROARCH.md, cycle
// constants).  The B operand (activation chunk) is shared by the TP MFMAs.
// `slot(j, tp)` runs after the MFMAs of the j-th (chunk, tile) step of the tile group - the place where the previous group's
// epilogue is issued when it is software-pipelined (dense()).
// The split-bf16 render kernel's tile epilogue inside the next group's MFMA chain, in pinned program order (dense(): first branch)
#ifndef NERFDS_X3_PIN
#define NERFDS_X3_PIN 1
#endif
constexpr bool X3_PIN = !IS_TRAIN && NERFDS_X3_PIN != 0;
// `win(w)` (pinned form, interleaved split-bf16 chains only): runs after MFMA w of the tile group; every MFMA is followed by an ordering point -
// an empty volatile asm that "rewrites" both accumulators - so each MFMA sits between two of them in the instruction stream, and what `win`
// issues (volatile asm too) stays in its window: program order is source order, instruction by instruction.  The ring reads of a chunk
// step go out behind its first four MFMAs (loads do not cross volatile asm either).
struct NoWin { static constexpr bool ON = false; DEVI void operator()(int) const {} };
template <class F> struct Win { static constexpr bool ON = true; F& f; DEVI void operator()(int w) const { f(w); } };
template <class G, class PL, int NT, int TP, int P, int K, class SLOT, class WIN = NoWin>
DEVI void accum(f32x16 (&acc)[TP][NT], Pipe<G, PL>& pipe, Cursor& cur, const Chunk<P> (&in)[NT][K], int& j, SLOT&& slot, const WIN& win = WIN()) {
  using PP = Pipe<G, PL>;
  constexpr int NP = frag_parts(P);
  if constexpr (WIN::ON && is_single(P)) {
    // one-unit operands, two N-tiles, tiles one at a time: a fragment feeds the two MFMAs of its tile (one per N-tile, two accumulators)
    static_assert(TP == 1 && NT == 2, "pinned windows: one tile x two N-tiles");
#pragma unroll
    for (int kc = 0; kc < K; ++kc) {
      const int u = cur.pos;
      if (u % PP::SU == 0) pipe.begin_stage(cur.seg, u);
      const WFrag<P> w = pipe.template frag<P>(u);
      mma<P>(acc[0][0], w, in[0][kc]);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
#endif
      pipe.refill(cur.seg, u); pipe.spread_piece(cur.seg, u);
      win(2 * j);
      mma<P>(acc[0][1], w, in[1][kc]);
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" : "+v"(acc[0][0]), "+v"(acc[0][1]));
#endif
      win(2 * j + 1);
      cur.pos += 1;
      j += 1;
    }
    return;
  }
  if constexpr (WIN::ON && !is_single(P)) {
    static_assert(PP::X3_INTERLEAVE && P == P_BF16X3 && TP == 2 && NT == 1, "pinned windows: interleaved split-bf16 tile pairs");
#pragma unroll
    for (int kc = 0; kc < K; ++kc) {
      const int u = cur.pos;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((u + q) % PP::SU == 0) pipe.begin_stage(cur.seg, u + q);
      const WFrag<P> w0 = pipe.template frag<P>(u), w1 = pipe.template frag<P>(u + 2);
      const Chunk<P>& c = in[0][kc];
      // (always_inline: as an ordinary call its argument is a run-time value, the unrolled indices stop being constants and the ring lands in scratch)
      auto after = [&](int m) __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(acc[0][0]), "+v"(acc[1][0]));
#endif
        if (m < 4) { pipe.refill(cur.seg, u + m); pipe.spread_piece(cur.seg, u + m); }
        win(3 * j + m);
      };
      acc[0][0] = NERFDS_X3_MFMA(w0.hi, c.lo, acc[0][0], 0, 0, 0); after(0);
      acc[1][0] = NERFDS_X3_MFMA(w1.hi, c.lo, acc[1][0], 0, 0, 0); after(1);
      acc[0][0] = NERFDS_X3_MFMA(w0.lo, c.hi, acc[0][0], 0, 0, 0); after(2);
      acc[1][0] = NERFDS_X3_MFMA(w1.lo, c.hi, acc[1][0], 0, 0, 0); after(3);
      acc[0][0] = NERFDS_X3_MFMA(w0.hi, c.hi, acc[0][0], 0, 0, 0); after(4);
      acc[1][0] = NERFDS_X3_MFMA(w1.hi, c.hi, acc[1][0], 0, 0, 0); after(5);
      cur.pos += 4;
      j += 2;
    }
    return;
  }
  if constexpr (PP::X3_INTERLEAVE && P == P_BF16X3 && TP == 2) {
    // Split bf16, one wave per SIMD: the three MFMAs of a product go to the same accumulator, and whatever hipcc places
    // between two MFMAs on the SAME accumulator (weight reads, waits, epilogue VALU) costs ~43 cycles instead of its issue
    // slot.  Issued pairwise over the two tiles of the group - hl0 hl1 lh0 lh1 hh0 hh1 - consecutive MFMAs never share an
    // accumulator; per accumulator the order of the terms (hi*lo, lo*hi, hi*hi, chunk by chunk) is unchanged, so the results
    // are the same bits.  Needs the four units of the group in the register ring at once (RD >= 8 keeps the prefetch ahead).
#pragma unroll
    for (int kc = 0; kc < K; ++kc) {
      const int u = cur.pos;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if ((u + q) % PP::SU == 0) pipe.begin_stage(cur.seg, u + q);
      const WFrag<P> w0 = pipe.template frag<P>(u), w1 = pipe.template frag<P>(u + 2);
      const Chunk<P>& c = in[0][kc];
      acc[0][0] = NERFDS_X3_MFMA(w0.hi, c.lo, acc[0][0], 0, 0, 0);
      acc[1][0] = NERFDS_X3_MFMA(w1.hi, c.lo, acc[1][0], 0, 0, 0);
      acc[0][0] = NERFDS_X3_MFMA(w0.lo, c.hi, acc[0][0], 0, 0, 0);
      acc[1][0] = NERFDS_X3_MFMA(w1.lo, c.hi, acc[1][0], 0, 0, 0);
      acc[0][0] = NERFDS_X3_MFMA(w0.hi, c.hi, acc[0][0], 0, 0, 0);
      acc[1][0] = NERFDS_X3_MFMA(w1.hi, c.hi, acc[1][0], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) { pipe.refill(cur.seg, u + q); pipe.spread_piece(cur.seg, u + q); }
      cur.pos += 4;
      slot(j, 0); ++j;
      slot(j, 1); ++j;
    }
    return;
  }
#pragma unroll
  for (int kc = 0; kc < K; ++kc) {
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
      const int u = cur.pos;
      // First fragment that touches a new stage (a multi-unit fragment may straddle: its first units are in the register
      // ring already, and so is every other unit of the stage that is being retired - the ring runs RD units ahead).
#pragma unroll
      for (int q = 0; q < NP; ++q)
        if ((u + q) % PP::SU == 0) pipe.begin_stage(cur.seg, u + q);
      const WFrag<P> w = pipe.template frag<P>(u);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) mma<P>(acc[tp][nt], w, in[nt][kc]);
#pragma unroll
      for (int q = 0; q < NP; ++q) { pipe.refill(cur.seg, u + q); pipe.spread_piece(cur.seg, u + q); }
      cur.pos += NP;
      slot(j, tp);
      ++j;
    }
  }
}

template <int P, int NT, bool RELU, int W>
DEVI void tile_epilogue(Chunk<P> (&out)[NT][W], int ot, const f32x16 (&acc)[NT]) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float x0[8], x1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x0[i] = acc[nt][i]; x1[i] = acc[nt][8 + i]; }
    make_act_chunk<P, RELU>(out[nt][2 * ot], x0);
    make_act_chunk<P, RELU>(out[nt][2 * ot + 1], x1);
  }
}

// ReLU + conversion of one tile for the one-unit operand formats, as ONE asm block: convert pairs first
// (v_cvt_pk_{bf16,f16}_f32), then ReLU on the packed pairs (v_pk_max_i16 with 0: a negative half has its sign bit
// set) - 16 VALU per tile.  Used by the 8-wave kernels (two waves per SIMD, one N-tile: the static graph's bf16 / f16); the
// two-N-tile kernels use the C++ form of make_act_chunk (same instructions, scheduled by hipcc between the next group's MFMAs,
// 1 - 3 % faster there: Makefile NT2FLAGS).  hipcc does not pad the MFMA -> VALU hazard for an asm that
// reads accumulators (checked in the ISA), so the block opens with the 12 wait states itself when `wait` is set (the
// first block of a tile group; dense() pins the last MFMA of every accumulator of the group above it, so the later
// blocks are covered by the first one's instructions).
template <int P> DEVI void tile_epilogue_asm(Chunk<P>& c0, Chunk<P>& c1, const f32x16& a, bool wait) {
  static_assert(is_single(P), "packed-half epilogue");
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned o0, o1, o2, o3, o4, o5, o6, o7;
#define NERFDS_EPI_BODY(CVT)                                                                                              \
  CVT " %0, %8, %9\n\t" CVT " %1, %10, %11\n\t" CVT " %2, %12, %13\n\t" CVT " %3, %14, %15\n\t"                            \
  CVT " %4, %16, %17\n\t" CVT " %5, %18, %19\n\t" CVT " %6, %20, %21\n\t" CVT " %7, %22, %23\n\t"                          \
  "v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tv_pk_max_i16 %2, %2, 0\n\tv_pk_max_i16 %3, %3, 0\n\t"              \
  "v_pk_max_i16 %4, %4, 0\n\tv_pk_max_i16 %5, %5, 0\n\tv_pk_max_i16 %6, %6, 0\n\tv_pk_max_i16 %7, %7, 0"
#define NERFDS_EPI_OPS                                                                                                    \
  : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4), "=&v"(o5), "=&v"(o6), "=&v"(o7)                                \
  : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),          \
    "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15])
  if constexpr (P == P_BF16) {
    if (wait) asm volatile("s_nop 11\n\t" NERFDS_EPI_BODY("v_cvt_pk_bf16_f32") NERFDS_EPI_OPS);
    else asm volatile(NERFDS_EPI_BODY("v_cvt_pk_bf16_f32") NERFDS_EPI_OPS);
  } else {
    if (wait) asm volatile("s_nop 11\n\t" NERFDS_EPI_BODY("v_cvt_pk_f16_f32") NERFDS_EPI_OPS);
    else asm volatile(NERFDS_EPI_BODY("v_cvt_pk_f16_f32") NERFDS_EPI_OPS);
  }
#undef NERFDS_EPI_BODY
#undef NERFDS_EPI_OPS
  const u32x4 r0 = {o0, o1, o2, o3}, r1 = {o4, o5, o6, o7};
  c0.v = __builtin_bit_cast(decltype(c0.v), r0);
  c1.v = __builtin_bit_cast(decltype(c1.v), r1);
#endif
}

// The same for the one-unit two-N-tile kernels (bf16 / f16 on the nerf_ds / HyperNeRF graphs, built with NERFDS_TILE_PAIR = 1): one tile =
// 2 accumulators = 16 value pairs, per pair v_cvt_pk + v_pk_max_i16 with 0 (make_act_chunk's arithmetic) = 32 instructions.
#ifndef NERFDS_X1_PIN
#define NERFDS_X1_PIN 1
#endif
constexpr bool X1_PIN = !IS_TRAIN && NERFDS_X1_PIN != 0 && TILE_PAIR == 1;
#ifndef NERFDS_X1_J0
#define NERFDS_X1_J0 0
#endif
struct X1Epi {
  static constexpr int OPS = 32;
  unsigned u[2][8];
};
template <int P, int W> DEVI void x1_epi_op(int i, const f32x16 (&prev)[2], X1Epi& e, Chunk<P> (&out)[2][W], int t) {
#if defined(__HIP_DEVICE_COMPILE__)
  // op order: cvt p, cvt p + 1, max p, max p + 1 (p even): a conversion and its ReLU are never neighbours
  const int blk = i / 4, r = i % 4, pair = 2 * blk + (r & 1), nt = pair / 8, k = pair % 8;
  if (r < 2) {
    if constexpr (P == P_BF16) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e.u[nt][k]) : "v"(prev[nt][2 * k]), "v"(prev[nt][2 * k + 1]));
    else asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(e.u[nt][k]) : "v"(prev[nt][2 * k]), "v"(prev[nt][2 * k + 1]));
  } else {
    asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(e.u[nt][k]));
    if (k % 4 == 3) {
      const int sub = k / 4;
      const u32x4 rr = {e.u[nt][4 * sub], e.u[nt][4 * sub + 1], e.u[nt][4 * sub + 2], e.u[nt][4 * sub + 3]};
      out[nt][2 * t + sub].v = __builtin_bit_cast(decltype(out[nt][2 * t + sub].v), rr);
    }
  }
#endif
}

// One group's conversion (2 tiles x 16 accumulator registers -> 4 split-bf16 chunks) as 128 single VALU instructions: per value pair
// ReLU, ReLU, hi = cvt_pk, unpack, unpack, subtract, subtract, lo = cvt_pk - make_chunk<P_BF16X3>'s arithmetic on relu_f'd values.  Two pairs
// are in flight at a time (op i: block i / 16 of two pairs, step (i % 16) / 2, pair i % 2), so neighbouring instructions are independent.
struct X3Epi {
  static constexpr int OPS = 128;
  float t0[2], t1[2];
  unsigned hi[2], lo[2], h0[2], h1[2];
  unsigned uh[2][8], ul[2][8];
};
template <int W> DEVI void x3_epi_op(int i, const f32x16 (&prev)[2], X3Epi& e, Chunk<P_BF16X3> (&out)[1][W], int pot) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int w = i % 2, s = (i % 16) / 2, pair = 2 * (i / 16) + w, tp = pair / 8, k = pair % 8;
  if (s == 0) asm volatile("v_max_i32 %0, 0, %1" : "=v"(e.t0[w]) : "v"(prev[tp][2 * k]));
  else if (s == 1) asm volatile("v_max_i32 %0, 0, %1" : "=v"(e.t1[w]) : "v"(prev[tp][2 * k + 1]));
#if NERFDS_X3_F16
  // (split f16: hi = v_cvt_pk_f16_f32, float(hi) by v_cvt_f32_f16 of the packed pair's low half and - SDWA word select - of its high half: the same eight
  // steps per pair as split bf16's shift / mask, so the windows of the pinned chain are unchanged)
  else if (s == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(e.hi[w]) : "v"(e.t0[w]), "v"(e.t1[w]));
  else if (s == 3) asm volatile("v_cvt_f32_f16_e32 %0, %1" : "=v"(e.h0[w]) : "v"(e.hi[w]));
  else if (s == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(e.h1[w]) : "v"(e.hi[w]));
#else
  else if (s == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e.hi[w]) : "v"(e.t0[w]), "v"(e.t1[w]));
  else if (s == 3) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(e.h0[w]) : "v"(e.hi[w]));
  else if (s == 4) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(e.h1[w]) : "v"(e.hi[w]));
#endif
  else if (s == 5) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e.t0[w]) : "v"(e.t0[w]), "v"(e.h0[w]));
  else if (s == 6) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(e.t1[w]) : "v"(e.t1[w]), "v"(e.h1[w]));
  else {
#if NERFDS_X3_F16
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(e.lo[w]) : "v"(e.t0[w]), "v"(e.t1[w]));
#else
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(e.lo[w]) : "v"(e.t0[w]), "v"(e.t1[w]));
#endif
    e.uh[tp][k] = e.hi[w];
    e.ul[tp][k] = e.lo[w];
    if (k % 4 == 3) {                                              // a chunk (8 values) of the pending group is complete
      const int sub = k / 4, t = pot + tp;
      const u32x4 rh = {e.uh[tp][4 * sub], e.uh[tp][4 * sub + 1], e.uh[tp][4 * sub + 2], e.uh[tp][4 * sub + 3]};
      const u32x4 rl = {e.ul[tp][4 * sub], e.ul[tp][4 * sub + 1], e.ul[tp][4 * sub + 2], e.ul[tp][4 * sub + 3]};
      out[0][2 * t + sub].hi = __builtin_bit_cast(x3x8, rh);
      out[0][2 * t + sub].lo = __builtin_bit_cast(x3x8, rl);
    }
  }
#endif
}
#ifdef NERFDS_PROF
// s_memtime at a pinned place of the instruction stream: the asm "rewrites" one register of what the surrounding code produces / consumes
template <int P> DEVI unsigned chunk_word(const Chunk<P>& c) { unsigned w; __builtin_memcpy(&w, &c, 4); return w; }
DEVI unsigned long long prof_now(unsigned& tie) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(tie) :: "memory");
  return t;
}
#define NERFDS_PROF_BEGIN(tievar) unsigned long long prof_t0_ = prof_now(tievar)
#define NERFDS_PROF_END(acc, tievar) do { (acc) += prof_now(tievar) - prof_t0_; } while (0)
#else
#define NERFDS_PROF_BEGIN(tievar) do { } while (0)
#define NERFDS_PROF_END(acc, tievar) do { } while (0)
#endif
template <class T> struct seg_chunks;
template <int P, int NT, int K> struct seg_chunks<Chunk<P>[NT][K]> { static constexpr int value = K; };
template <class... Ins> struct seg_total { static constexpr int value = (seg_chunks<Ins>::value + ... + 0); };

// One dense layer with OT output tiles of 32 features, computed TILE_PAIR tiles at a time (the stream interleaves the
// fragments of the tiles of a pair chunk by chunk, pack.h); inputs are one or more chunk arrays in stream order (each
// in its own precision), the output chunks are produced in the precision PO of the tensor they form.
// The accumulators start from the bias (the first MFMA of a tile reads the bias registers as its C operand; the N-tiles of
// a wave share one copy).
template <class G, class PL, int NT, int OT, bool RELU, class CUR, int PO, class... Ins>
DEVI void dense(Pipe<G, PL>& pipe, CUR& cur, Chunk<PO> (&out)[NT][2 * OT], Ins&... ins) {
  constexpr int TP = TILE_PAIR;
  constexpr bool BWD_IN = std::is_same_v<CUR, BwdInCursor>;
  constexpr bool BWD = std::is_same_v<CUR, BwdCursor> || BWD_IN;
  constexpr bool TRAIN = std::is_same_v<CUR, TrainCursor> || BWD;
  static_assert(!TRAIN || (NT == 1 && (!is_single(PO) || (BWD && PO == P_F16))),
                "the training kernels run two-unit plans (one N-tile, C++ epilogue); the backward chains of the TANGENT pass may run one f16 MFMA per product");
  static_assert(OT % TP == 0, "layers have an even number of 32-feature tiles");
  // (uniform one-unit plans only: in the mixed plan - f16 networks around a split-bf16 warp field - the asm epilogue build gave
  // run-to-run differences on ~1 % of the rays of the fine level; the C++ epilogue build of the same kernel is clean)
  constexpr bool ASM_EPI = PL::NT == 1 && is_single(PO) && RELU && PL::UNIFORM && !PL::HAS_C;
#ifdef NERFDS_PROF
  unsigned prof_tie_ = (unsigned)bias_base(pipe.lane16);
  NERFDS_PROF_BEGIN(prof_tie_);
  const int hb = (int)prof_tie_;          // (every bias read of the layer depends on the timer's asm: the chain starts behind it)
#else
  const int hb = bias_base(pipe.lane16);
#endif
  auto no_slot = [](int, int) {};
  if constexpr (!TRAIN && X1_PIN && is_single(PO) && RELU && NT == 2 && TP == 1 && (OT > 1) && PL::UNIFORM) {
    // One-unit two-N-tile render kernels, tiles one at a time: the conversion of tile t (its two accumulators rest in `prev`) is issued as
    // single-instruction volatile asm statements in the MFMA windows of tile t + 1 - the scheme of the split-bf16 branch below, which has the
    // measurements.  As hipcc scheduled the C++ epilogue of these kernels, a group's 64 - 70 VALU sat in one run behind its last MFMA.
    // J0 = 0: the first 16 instructions read prev[0], whose last MFMA is TWO MFMAs above window 0 (prev[1]'s chain ended in between, the new
    // tile's first MFMA opens the window): the hazard distance holds by construction; prev[1] is first read at instruction 16.
    constexpr int KC = seg_total<Ins...>::value, NW = 2 * KC, J0 = NERFDS_X1_J0;
    static_assert(J0 == 0 || J0 == 2, "window of the first conversion instruction");
    constexpr int PER = cdiv(X1Epi::OPS, NW - J0), IN_CHAIN = (NW - J0) * PER < X1Epi::OPS ? (NW - J0) * PER : X1Epi::OPS;
    f32x16 prev[NT];
    X1Epi e;
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      f32x16 acc[TP][NT];
      {
        const f32x16 bv = load_bias(cur.bt + ot, hb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] = bv;
      }
      int j = 0;
      auto fill = [&](int w) __attribute__((always_inline)) {
        if (ot == 0 || w < J0) return;
#pragma unroll
        for (int q = (w - J0) * PER; q < (w - J0 + 1) * PER; ++q)
          if (q < X1Epi::OPS) x1_epi_op<PO>(q, prev, e, out, ot - 1);
      };
      Win<decltype(fill)> win{fill};
      (accum<G, PL, NT, TP>(acc, pipe, cur, ins, j, no_slot, win), ...);
      if (ot > 0) {
#pragma unroll
        for (int q = IN_CHAIN; q < X1Epi::OPS; ++q) x1_epi_op<PO>(q, prev, e, out, ot - 1);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) prev[nt] = acc[0][nt];
    }
    tile_epilogue<PO, NT, RELU>(out, OT - 1, prev);                         // the layer's last tile: behind its own chain (builtin MFMAs: hipcc pads the hazard)
  } else if constexpr (!TRAIN && X3_PIN && PO == P_BF16X3 && RELU && Pipe<G, PL>::X3_INTERLEAVE && (OT > TP)) {
    // Split bf16 render kernel: one 512-register wave per SIMD, so nothing covers a tile group's epilogue - 128 VALU (ReLU + hi / lo split of 32
    // values) behind 24 - 96 MFMAs.  As hipcc schedules it, every group's conversion is ONE uninterrupted run behind the group's last MFMA
    // (the stage boundaries cut a group into scheduling regions, and nothing moves across them): 15 % of the kernel's time, measured by issuing the
    // conversion twice (38.7 -> 44.8 ms per 65 536 rays, profiles/r4_ab/ab_epi2.txt).  Here the accumulators of group g rest in `prev` and their
    // conversion is issued as single-instruction volatile asm statements in the MFMA windows of group g + 1 (accum: PIN) - program order is the
    // source order, PER VALU per window from window J0 on (the MFMA -> VALU hazard of `prev` is then >= J0 MFMAs old: hipcc does not pad it for
    // asm).  Same arithmetic, same order per accumulator: bit-identical results.
    static_assert(TP == 2 && NT == 1, "pairs of tiles, one N-tile");
    constexpr int KC = seg_total<Ins...>::value, NW = 6 * KC, J0 = 2;       // MFMA windows per group
    constexpr int PER = cdiv(X3Epi::OPS, NW - J0), IN_CHAIN = (NW - J0) * PER < X3Epi::OPS ? (NW - J0) * PER : X3Epi::OPS;
    f32x16 prev[TP];
    X3Epi e;
#pragma unroll
    for (int ot = 0; ot < OT; ot += TP) {
      f32x16 acc[TP][NT];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) acc[tp][0] = load_bias(cur.bt + ot + tp, hb);
      int j = 0;
      auto fill = [&](int w) __attribute__((always_inline)) {
        if (ot == 0 || w < J0) return;
#pragma unroll
        for (int q = (w - J0) * PER; q < (w - J0 + 1) * PER; ++q)
          if (q < X3Epi::OPS) x3_epi_op(q, prev, e, out, ot - TP);
      };
      Win<decltype(fill)> win{fill};
      (accum<G, PL, NT, TP>(acc, pipe, cur, ins, j, no_slot, win), ...);
      if (ot > 0) {
#pragma unroll
        for (int q = IN_CHAIN; q < X3Epi::OPS; ++q) x3_epi_op(q, prev, e, out, ot - TP);      // what the chain had no window for (short inputs)
      }
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) prev[tp] = acc[tp][0];
    }
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {                                     // the layer's last group: behind its own chain (builtin MFMAs: hipcc pads the hazard)
      const f32x16 (&pa)[1] = reinterpret_cast<const f32x16 (&)[1]>(prev[tp]);
      tile_epilogue<PO, NT, RELU>(out, OT - TP + tp, pa);
    }
  } else if constexpr (TRAIN && !BWD_IN && TRAIN_PIPE && (OT > TP) && (BWD ? PO == P_BF16X3 : true)) {
    // Training forward / backward chain, software-pipelined at the source level: the epilogue of tile group g - conversion into the next
    // layer's operand, ReLU bits / mask, 16-bit stores - is cut into pieces that are issued between the MFMA steps of group g + 1 (the
    // accumulators of group g rest in `prev`).  A training kernel runs ONE wave per SIMD: whatever is issued behind a group's last MFMA
    // runs with the matrix pipe idle, whatever is issued between two MFMA steps of the next group runs under them.
    static_assert(TP == 2 && NT == 1, "pairs of tiles, one N-tile");
    constexpr int SLOTS = TP * seg_total<Ins...>::value;       // MFMA steps per group
    constexpr int NPIECE = 3 * TP + 1, J0 = 1;
    constexpr int PPS = cdiv(NPIECE, SLOTS - J0 > 0 ? SLOTS - J0 : 1);
    constexpr int IN_CHAIN = (SLOTS - J0) * PPS < NPIECE ? (SLOTS - J0 > 0 ? (SLOTS - J0) * PPS : 0) : NPIECE;
    f32x16 prev[TP];
    unsigned two = 0;
    // piece q of the group whose first tile is `pot`: q = 3 tp + {0: chunk of registers 0-7, 1: chunk of registers 8-15, 2: store}, q = 3 TP: bits
    auto piece = [&](int q, int pot) {
      if (q < 3 * TP) {
        const int tp = q / 3, sub = q % 3, t = pot + tp;
        if (sub < 2) {
          if constexpr (BWD) { if (sub == 0) apply_mask(prev[tp], (cur.mask[t >> 1] >> (16 * (t & 1))) & 0xffffu); }
          float x[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] = prev[tp][8 * sub + i];
          make_act_chunk<PO, RELU>(out[0][2 * t + sub], x);
        } else if constexpr (BWD) {
          if (TRAIN_HALF) store_tile_g16(cur.row16 + 32 * t, prev[tp]);
          else store_tile<false>(cur.row + 32 * t, prev[tp]);
        } else {
          if (TRAIN_HALF) two |= store_tile_half<RELU>(cur.row16 + 32 * t, prev[tp]) << (16 * tp);
          else store_tile<RELU>(cur.row + 32 * t, prev[tp]);
        }
      } else if constexpr (!BWD) {
        if (TRAIN_HALF) { *reinterpret_cast<unsigned*>(cur.bits + pot) = two; two = 0; }
      }
    };
#pragma unroll
    for (int ot = 0; ot < OT; ot += TP) {
      f32x16 acc[TP][NT];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        if constexpr (BWD) acc[tp][0] = f32x16{};
        else acc[tp][0] = load_bias(cur.bt + ot + tp, hb);
      }
      int j = 0;
      auto slot = [&](int jj, int) {
        if (ot == 0 || jj < J0) return;
#pragma unroll
        for (int q = (jj - J0) * PPS; q < (jj - J0 + 1) * PPS; ++q)
          if (q < NPIECE) piece(q, ot - TP);
      };
      (accum<G, PL, NT, TP>(acc, pipe, cur, ins, j, slot), ...);
      if (ot > 0) {
#pragma unroll
        for (int q = IN_CHAIN; q < NPIECE; ++q) piece(q, ot - TP);      // pieces the chain had no step for (short inputs)
      }
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) prev[tp] = acc[tp][0];
#if defined(__HIP_DEVICE_COMPILE__)
      // nothing else separates the groups: one scheduling region per group (its MFMAs + the previous group's epilogue), so that hipcc
      // interleaves THOSE and does not pull later groups' work - and their registers - forward.  (Asking for the interleave explicitly
      // through sched_group_barrier, 4 or 6 VALU per MFMA, made the forward spill 55 - 67 registers: 15.7 -> 21.6 ms per step.)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) piece(q, OT - TP);                 // the layer's last group: behind its own chain
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
  } else {
#pragma unroll
    for (int ot = 0; ot < OT; ot += TP) {
      f32x16 acc[TP][NT];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        f32x16 bv;
        if constexpr (BWD) bv = f32x16{};                      // the transposed layers have no bias
        else bv = load_bias(cur.bt + ot + tp, hb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[tp][nt] = bv;
      }
      int j = 0;
      (accum<G, PL, NT, TP>(acc, pipe, cur, ins, j, no_slot), ...);
      if constexpr (BWD_IN) {                                  // gradient of the raw input: no mask, no next layer
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) store_tile_in(cur, ot + tp, acc[tp][0]);
      } else if constexpr (BWD) {
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
          apply_mask(acc[tp][0], (cur.mask[(ot + tp) >> 1] >> (16 * ((ot + tp) & 1))) & 0xffffu);
          if (!TRAIN_HALF) store_tile<false>(cur.row + 32 * (ot + tp), acc[tp][0]);
        }
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) tile_epilogue<PO, NT, false>(out, ot + tp, acc[tp]);
        if constexpr (PO == P_BF16X3 || PO == P_F16) {
          if (TRAIN_HALF) {
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) store_tile_g16(cur.row16 + 32 * (ot + tp), acc[tp][0]);
          }
        }
      } else if constexpr (ASM_EPI) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int tp = 0; tp < TP; ++tp)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(acc[tp][nt]));     // every chain of the group ends above the epilogue blocks
#endif
#pragma unroll
        for (int tp = 0; tp < TP; ++tp)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            tile_epilogue_asm<PO>(out[nt][2 * (ot + tp)], out[nt][2 * (ot + tp) + 1], acc[tp][nt], tp == 0 && nt == 0);
      } else {
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) tile_epilogue<PO, NT, RELU>(out, ot + tp, acc[tp]);
        if constexpr (TRAIN) {
          if (TRAIN_HALF) {
            unsigned two = 0;
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) two |= store_tile_half<RELU>(cur.row16 + 32 * (ot + tp), acc[tp][0]) << (16 * tp);
            static_assert(TP == 2, "one u32 of ReLU bits per tile pair");
            *reinterpret_cast<unsigned*>(cur.bits + ot) = two;
          } else {
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) store_tile<RELU>(cur.row + 32 * (ot + tp), acc[tp][0]);
          }
        }
      }
    }
  }
#ifdef NERFDS_PROF
  prof_tie_ = chunk_word(out[0][2 * OT - 1]);      // the layer's last output chunk exists: the chain and its conversion are behind us
#endif
  NERFDS_PROF_END(pipe.t_chain, prof_tie_);
  cur.bt += OT;
}

// Output head (<= 16 logical outputs, duplicated in both lane halves by the packer): logical output j = acc[j].
template <class G, class PL, int NT, int P, int K>
DEVI void head(Pipe<G, PL>& pipe, Cursor& cur, f32x16 (&acc)[1][NT], Chunk<P> (&in)[NT][K]) {
#ifdef NERFDS_PROF
  unsigned prof_tie_ = (unsigned)bias_base(pipe.lane16);
  NERFDS_PROF_BEGIN(prof_tie_);
  const int hb = (int)prof_tie_;
#else
  const int hb = bias_base(pipe.lane16);
#endif
  {
    const f32x16 bv = load_bias(cur.bt, hb);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = bv;
  }
  int j = 0;
  auto no_slot = [](int, int) {};
  accum<G, PL, NT, 1>(acc, pipe, cur, in, j, no_slot);
#ifdef NERFDS_PROF
  prof_tie_ = __builtin_bit_cast(unsigned, acc[0][0][0]);
#endif
  NERFDS_PROF_END(pipe.t_chain, prof_tie_);
  cur.bt += 1;
}

// ------------------------------------------------------------------------------------------------
// Scalar math helpers
// ------------------------------------------------------------------------------------------------
// sin / cos with a 3-term Cody-Waite reduction (fma) and degree-9/8 minimax kernels on [-pi/4, pi/4]: ~1-2 ulp while
// the quadrant count stays exact in fp32 (|a| < ~1e7; posenc arguments are |x| * 2^7 at most).  Branch-free on purpose:
// the field evaluation must not contain divergent regions (see the note in eval_shared), which rules out libm's sinf/cosf.
DEVI void sincos_cw(float a, float& sn_out, float& cs_out) {
  float k = rintf(a * 0.636619772f);
  int q = (int)k;
  float r = fmaf(-k, 1.57079601e+00f, a);
  r = fmaf(-k, 3.13916473e-07f, r);
  r = fmaf(-k, 5.39030253e-15f, r);
  float s = r * r;
  float ps = fmaf(s, 2.86567956e-6f, -1.98559923e-4f);
  ps = fmaf(ps, s, 8.33338592e-3f);
  ps = fmaf(ps, s, -1.66666672e-1f);
  float sn = fmaf(r * s, ps, r);
  float pc = fmaf(s, 2.44677067e-5f, -1.38877297e-3f);
  pc = fmaf(pc, s, 4.16666567e-2f);
  pc = fmaf(pc, s, -0.5f);
  float cs = fmaf(pc, s, 1.0f);
  const float vs = (q & 1) ? cs : sn, vc = (q & 1) ? sn : cs;
  sn_out = (q & 2) ? -vs : vs;
  cs_out = ((q + 1) & 2) ? -vc : vc;
}
DEVI float sin_cw(float a) {
  float sn, cs;
  sincos_cw(a, sn, cs);
  return sn;
}

DEVI float softplus_f(float x) {   // jax.nn.softplus = logaddexp(x, 0)
  return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
}
DEVI float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

DEVI void normalize3(float (&v)[3]) {   // model_utils.py:438-442
  float n2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  float inv = 1.0f / sqrtf(fmaxf(n2, 1.1920929e-07f));
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// wave-wide helpers (64 lanes) on DPP: no LDS round trip (the __shfl forms compile to ds_bpermute_b32, ~300 of them per kernel -
// the per-ray phases run on one wave per SIMD while the matrix pipes idle, so their latency is all exposed).
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes whose row is masked out or whose source lane does not exist keep `old`.
template <int CTRL, int ROW_MASK = 0xf> DEVI float dpp_f(float old, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
#else
  return old + 0.f * v;
#endif
}
enum { DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140, DPP_ROW_BCAST15 = 0x142,
       DPP_ROW_BCAST31 = 0x143, DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118, DPP_WAVE_SHR1 = 0x138 };
DEVI float lane63(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
  return v;
#endif
}
DEVI float wave_sum(float v) {
  v += dpp_f<DPP_QUAD_1032>(0.f, v);
  v += dpp_f<DPP_QUAD_2301>(0.f, v);
  v += dpp_f<DPP_ROW_HALF_MIRROR>(0.f, v);
  v += dpp_f<DPP_ROW_MIRROR>(0.f, v);                       // every lane: the sum of its row of 16
  v += dpp_f<DPP_ROW_BCAST15, 0xA>(0.f, v);                 // rows 1, 3 += rows 0, 2
  v += dpp_f<DPP_ROW_BCAST31, 0xC>(0.f, v);                 // rows 2, 3 += rows 0 + 1
  return lane63(v);
}
DEVI float wave_scan_add(float v, int) {     // inclusive
  v += dpp_f<DPP_ROW_SHR1>(0.f, v);
  v += dpp_f<DPP_ROW_SHR2>(0.f, v);
  v += dpp_f<DPP_ROW_SHR4>(0.f, v);
  v += dpp_f<DPP_ROW_SHR8>(0.f, v);                         // inclusive within each row of 16
  v += dpp_f<DPP_ROW_BCAST15, 0xA>(0.f, v);
  v += dpp_f<DPP_ROW_BCAST31, 0xC>(0.f, v);
  return v;
}
DEVI float wave_scan_mul(float v, int) {     // inclusive
  v *= dpp_f<DPP_ROW_SHR1>(1.f, v);
  v *= dpp_f<DPP_ROW_SHR2>(1.f, v);
  v *= dpp_f<DPP_ROW_SHR4>(1.f, v);
  v *= dpp_f<DPP_ROW_SHR8>(1.f, v);
  v *= dpp_f<DPP_ROW_BCAST15, 0xA>(1.f, v);
  v *= dpp_f<DPP_ROW_BCAST31, 0xC>(1.f, v);
  return v;
}
// value of the lane below (lane 0: `first`)
DEVI float wave_shift_up1(float v, float first) { return dpp_f<DPP_WAVE_SHR1>(first, v); }

// ------------------------------------------------------------------------------------------------
// Input encodings -> B operands.  A feature descriptor says how to produce linear feature f for one sample.
// ------------------------------------------------------------------------------------------------
struct FeatV { int kind; float arg; float win; float val; };   // kind: 0 zero, 1 sin(arg) * win, 2 val

// posenc feature g of a C-channel vector (layout [band][sin, cos][channel], model_utils.py:403-412)
template <int C> DEVI FeatV posenc_feat(int g, const float (&x)[C], const float* win) {
  const int band = g / (2 * C), sc = (g % (2 * C)) / C, ch = g % C;
  FeatV f;
  f.kind = 1;
  // sin(fl(x * 2^band + pi/2)): x * 2^band is exact, so the fma rounds once like the reference's add.
  f.arg = fmaf(x[ch], (float)(1 << band), sc ? 1.57079637f : 0.0f);
  f.win = win ? win[band] : 1.0f;
  f.val = 0.f;
  return f;
}
DEVI FeatV val_feat(float v) { FeatV f; f.kind = 2; f.arg = 0.f; f.win = 0.f; f.val = v; return f; }
DEVI FeatV zero_feat() { FeatV f; f.kind = 0; f.arg = 0.f; f.win = 0.f; f.val = 0.f; return f; }

// sin for the network-input encodings.  One-unit operands (bf16 / f16) round every feature to 8 / 11 significand bits
// anyway, so they use the hardware v_sin_f32 (argument in revolutions, abs error ~1e-6); the others use sin_cw.
// Split-bf16 operands (16 significand bits, 7.6e-6) sit in between: the hardware sine is accurate enough IF the argument reaches it
// reduced exactly - a * (1 / 2 pi) in two floats (product + fma residual + low word), fract of the exact high part, the residual
// added after: 7 issue slots where the 3-term Cody-Waite sin_cw takes 22.  With one wave per SIMD (the 512-register kernels)
// nothing hides the encodings: they were ~9 % of the split-bf16 kernel (80 MFMA-free blocks of 60+ instructions in its ISA).
// The trainer's forward stays on sin_cw (its gradient tests compare with the fp64 oracle at 1e-3-grade bounds).
constexpr bool X3_HW_SIN = !IS_TRAIN;
DEVI float sin_hw_exact(float a) {
  const float p = a * 0.15915494f;                                        // 1 / (2 pi) = 0.15915494 + 6.4206382e-09
  float e = fmaf(a, 0.15915494f, -p);
  e = fmaf(a, 6.4206382e-09f, e);
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(p) + e);
}
template <int P> DEVI float sin_enc(float a) {
  if constexpr (is_single(P)) {
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(a * 0.159154943f));
  } else if constexpr (X3_HW_SIN && P == P_BF16X3) {
    return sin_hw_exact(a);
  } else {
    return sin_cw(a);
  }
}

// Linear-feature chunk c: this lane supplies features 16c + 8h + i, i = 0..7.
template <int P, int KCH, class F> DEVI void build_chunks(Chunk<P> (&out)[KCH], int h, F feat) {
#pragma unroll
  for (int c = 0; c < KCH; ++c) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const FeatV A = feat(16 * c + i), B = feat(16 * c + 8 + i);
      float s = 0.f;
      if (A.kind == 1 || B.kind == 1) s = sin_enc<P>(h ? B.arg : A.arg);
      const float va = (A.kind == 1) ? s * A.win : A.val;
      const float vb = (B.kind == 1) ? s * B.win : B.val;
      x[i] = h ? vb : va;
    }
    make_chunk<P>(out[c], x);
    // Straight-line code (no branches since sin_cw lost its libm fallback): without a fence the scheduler interleaves
    // the sin evaluations of every chunk and spills thousands of registers.
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------------------------------------
// Per-wave LDS scratch
// ------------------------------------------------------------------------------------------------
enum { SV_SIGMA = 0, SV_RGB = 1, SV_MASK = 4, SV_NORM = 5, SV_WP = 8, SV_ROT = 13, SV_TRN = 16,
       SV_AX = 19, SV_SN = 22, SV_OMC = 23, SV_COUNT = 24 };
enum { RC_WEMB = 0, RC_MEMB = 8, RC_VDENC = 16, RC_VD = 40, RC_COUNT = 64 };

template <int MAXS> struct WaveLdsT {
  static constexpr int MAX_S = MAXS;
  float zs[MAXS];      // z of the current level
  float zn[MAXS];      // scratch: unsorted union / bins
  float ws[MAXS];      // compositing weights of the level just rendered
  float cdf[MAXS];
  float sv[SV_COUNT][MAXS];   // per-sample results / parked state (SoA: conflict-free by sample)
  float rayc[RC_COUNT];       // per-ray constants: warp GLO row, mask GLO row, posenc(viewdir)
};

struct RayConst {
  float o[3], d[3];
  float gt_mask;
};

// Rodrigues from the parked (unit axis, sin, 1 - cos): R = I + sin * W + (1 - cos) * W @ W (rigid_body.py:59-74).
DEVI void rodrigues(float (&R)[9], const float (&w)[3], float st, float omc) {
  const float W[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w2 = W[3 * r] * W[c] + W[3 * r + 1] * W[3 + c] + W[3 * r + 2] * W[6 + c];
      R[3 * r + c] = ((r == c) ? 1.f : 0.f) + st * W[3 * r + c] + omc * w2;
    }
}

// ------------------------------------------------------------------------------------------------
// The per-sample field: networks on one batch of 32*NT samples.  Per-sample state that is not needed by
// the next network is parked in the wave's LDS block at once (it is going there for compositing anyway),
// so the 8x256 trunk runs with (almost) only MFMA operands in registers.
// ------------------------------------------------------------------------------------------------
struct NoTrain { static constexpr bool ON = false; };
DEVI void set_row(Cursor&, float*, uint16_t*, uint16_t*) {}
DEVI void set_row(TrainCursor& c, float* p, uint16_t* p16, uint16_t* bits) { c.row = p; c.row16 = p16; c.bits = bits; }

// Which samples the N-tiles of this lane evaluate: depth z and the slot of the ray's LDS block (SoA by sample) that receives /
// holds the sample's per-sample state.  Tail lanes repeat the last sample (same values to the same slot).
template <int NT> struct Samples {
  float z[NT];
  int slot[NT];
};

#define NERFDS_TRAIN_ROW(base, base16, bits, W) do { if constexpr (TO::ON) set_row(cur, (base) + row * (size_t)(W) + 4 * h, \
    (base16) + row * (size_t)(W) + ROW16_H * h, (bits) + (row * 2 + h) * (size_t)((W) / 32)); } while (0)
#define NERFDS_TRAIN_HEAD(base, n) do { if constexpr (TO::ON) { _Pragma("unroll") for (int j_ = 0; j_ < (n); ++j_) (base)[row * (n) + j_] = hacc[0][0][j_]; } } while (0)

// ---- The level-independent networks on one batch of 32 * NT samples: MaskMLP -> SE(3) field + exp_se3 -> hyper sheet.
// Results are parked in the ray's LDS block at the sample's slot: predicted mask, warped point + ambient coordinates (SV_WP),
// rotation / translation fields, and the rotation itself (axis, sin, 1 - cos) for the normal conditioning of eval_nerf.
// `next`-stream bookkeeping is the caller's (pipe.cur / pipe.next).  `row` (training forward): this lane's row of the
// [R * S][width] activation arrays.
template <class G, class PL, int NT, class LT, class TO = NoTrain>
DEVI void eval_shared(const KArgs& ka, const RayConst& rc, Pipe<G, PL>& pipe, int lane, const Samples<NT>& sm, LT& L,
                      const TO& to = TO(), size_t row = 0) {
  using D = Dims<G>;
  const int h = lane >> 5;
  // Per-sample results are stored by EVERY lane, unconditionally, to the sample's slot: the two lane halves of a sample
  // (and the clamped tail lanes, which recompute the last sample) hold bit-identical values, so the duplicate stores are
  // harmless.  They must not be predicated: the evaluation has to stay free of divergent (partial-EXEC) regions, because
  // this hipcc places VGPR->AGPR live-range-split copies at the top of the join block, BEFORE exec is restored; the
  // copy then saves only the active lanes and the later full-EXEC reload returns garbage in the others (seen as
  // run-to-run varying rgb in the split-bf16 kernel).  Same reason for the branch-free sincos_cw above.
  float x[NT][3], xw[NT][3];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x[nt][c] = __fadd_rn(rc.o[c], __fmul_rn(sm.z[nt], rc.d[c]));   // model_utils.py:91-92
      xw[nt][c] = x[nt][c];
    }
  }

  std::conditional_t<TO::ON, TrainCursor, Cursor> cur;
  cur.seg = SEG_SHARED;
  cur.pos = 0;
  cur.bt = 0;

  // ---- MaskMLP (modules.py:409-434; models.py:967-975) ----
  float maskv[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) maskv[nt] = rc.gt_mask;
  if constexpr (G::HAS_MASK) {
    constexpr int W16 = G::MASK_W / 16, W32 = G::MASK_W / 32, P = PL::MASK;
    Chunk<P> in0[NT][D::MASK_KC], a[NT][W16], b[NT][W16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      build_chunks<P, D::MASK_KC>(in0[nt], h, [&](int f) {
        if (f < 6 * G::MASK_BANDS) return posenc_feat<3>(f, x[nt], ka.win_mask);
        if (f < D::MASK_IN) return val_feat(L.rayc[RC_MEMB + ((f - 6 * G::MASK_BANDS) & 7)]);
        return zero_feat();
      });
    static_assert(G::MASK_DEPTH == 8 || !G::HAS_MASK, "mask net is unrolled for depth 8, skip 4");
    NERFDS_TRAIN_ROW(to.mask_h[0], to.mask_h16[0], to.mask_bits[0], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, in0);
    NERFDS_TRAIN_ROW(to.mask_h[1], to.mask_h16[1], to.mask_bits[1], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.mask_h[2], to.mask_h16[2], to.mask_bits[2], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b);
    NERFDS_TRAIN_ROW(to.mask_h[3], to.mask_h16[3], to.mask_bits[3], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.mask_h[4], to.mask_h16[4], to.mask_bits[4], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b, in0);      // skip: [x, inputs] (modules.py:66-67)
    NERFDS_TRAIN_ROW(to.mask_h[5], to.mask_h16[5], to.mask_bits[5], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.mask_h[6], to.mask_h16[6], to.mask_bits[6], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b);
    NERFDS_TRAIN_ROW(to.mask_h[7], to.mask_h16[7], to.mask_bits[7], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, hacc, b);
    NERFDS_TRAIN_HEAD(to.mask_logit, 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float pm = fmaxf(hacc[0][nt][0], 0.f);                              // MaskMLP.output_activation = relu
      maskv[nt] = pm * ka.mask_ratio + rc.gt_mask * (1.0f - ka.mask_ratio);  // models.py:975
      L.sv[SV_MASK][sm.slot[nt]] = pm;
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) L.sv[SV_MASK][sm.slot[nt]] = 0.f;
  }

  // ---- SE3Field (warping.py:200-237) + exp_se3 (rigid_body.py:77-101) ----
  if constexpr (G::HAS_WARP) {
    constexpr int W16 = G::WARP_W / 16, W32 = G::WARP_W / 32, P = PL::WARP;
    Chunk<P> in0[NT][D::WARP_KC], a[NT][W16], b[NT][W16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      build_chunks<P, D::WARP_KC>(in0[nt], h, [&](int f) {
        constexpr int I3 = D::WARP_ID3, PE = I3 + 6 * G::WARP_BANDS;      // [x] | posenc(x) | warp_embed | [mask]
        if (f < I3) return val_feat(x[nt][f < 3 ? f : 0]);                 // identity prefix (model_utils.py:414-417)
        if (f < PE) return posenc_feat<3>(f - I3, x[nt], ka.win_warp);
        if (f < PE + 8) return val_feat(L.rayc[RC_WEMB + ((f - PE) & 7)]);
        if (G::HAS_MASK && f == PE + 8) return val_feat(maskv[nt]);        // models.py:729-730
        return zero_feat();
      });
    static_assert(G::WARP_DEPTH == 6 || !G::HAS_WARP, "warp trunk is unrolled for depth 6, skip 4");
    NERFDS_TRAIN_ROW(to.warp_h[0], to.warp_h16[0], to.warp_bits[0], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, in0);
    NERFDS_TRAIN_ROW(to.warp_h[1], to.warp_h16[1], to.warp_bits[1], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.warp_h[2], to.warp_h16[2], to.warp_bits[2], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b);
    NERFDS_TRAIN_ROW(to.warp_h[3], to.warp_h16[3], to.warp_bits[3], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.warp_h[4], to.warp_h16[4], to.warp_bits[4], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b, in0);
    NERFDS_TRAIN_ROW(to.warp_h[5], to.warp_h16[5], to.warp_bits[5], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, hacc, b);      // logical outputs: w = 0..2, v = 3..5
    NERFDS_TRAIN_HEAD(to.wv, 6);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float w[3] = {hacc[0][nt][0], hacc[0][nt][1], hacc[0][nt][2]};
      float v0 = hacc[0][nt][3], v1 = hacc[0][nt][4], v2 = hacc[0][nt][5];
      const float theta = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);     // warping.py:219 (no epsilon, as the reference)
      w[0] /= theta; w[1] /= theta; w[2] /= theta;
      v0 /= theta; v1 /= theta; v2 /= theta;
      float st, ct;
      sincos_cw(theta, st, ct);
      const float omc = 1.0f - ct, tms = theta - st;
      float Rm[9];
      rodrigues(Rm, w, st, omc);
      // p = (theta I + (1 - cos) W + (theta - sin) W @ W) v   (rigid_body.py:94-95)
      const float W[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
      float pt[3];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float g[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float w2 = W[3 * r] * W[c] + W[3 * r + 1] * W[3 + c] + W[3 * r + 2] * W[6 + c];
          g[c] = ((r == c) ? theta : 0.f) + omc * W[3 * r + c] + tms * w2;
        }
        pt[r] = g[0] * v0 + g[1] * v1 + g[2] * v2;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
        xw[nt][r] = Rm[3 * r] * x[nt][0] + Rm[3 * r + 1] * x[nt][1] + Rm[3 * r + 2] * x[nt][2] + pt[r];
      {
        const int s = sm.slot[nt];
        // rotation field: normalize(R @ normalize(1,1,1)) (models.py:1292-1296); translation field: R @ 0 + p
        float rf[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) rf[r] = (Rm[3 * r] + Rm[3 * r + 1] + Rm[3 * r + 2]) * 0.577350269f;
        normalize3(rf);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          L.sv[SV_WP + c][s] = xw[nt][c];
          L.sv[SV_ROT + c][s] = rf[c];
          L.sv[SV_TRN + c][s] = pt[c];
          L.sv[SV_AX + c][s] = w[c];
        }
        L.sv[SV_SN][s] = st;
        L.sv[SV_OMC][s] = omc;
      }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      {
        const int s = sm.slot[nt];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          L.sv[SV_WP + c][s] = x[nt][c];
          L.sv[SV_ROT + c][s] = 0.577350269f;
          L.sv[SV_TRN + c][s] = 0.f;
        }
      }
  }

  // ---- HyperSheetMLP on the OBSERVATION-space point (modules.py:367-392; models.py:662-666) ----
  float wamb[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) wamb[nt][0] = wamb[nt][1] = 0.f;
  if constexpr (G::HAS_HYPER) {
    constexpr int W16 = G::HYP_W / 16, W32 = G::HYP_W / 32, P = PL::HYP;
    Chunk<P> in0[NT][D::HYP_KC], a[NT][W16], b[NT][W16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      build_chunks<P, D::HYP_KC>(in0[nt], h, [&](int f) {
        if (f < 6 * G::HYP_BANDS) return posenc_feat<3>(f, x[nt], ka.win_hyp);
        if (f < 6 * G::HYP_BANDS + 8) return val_feat(L.rayc[RC_WEMB + ((f - 6 * G::HYP_BANDS) & 7)]);   // hyper_use_warp_embed
        if (G::HAS_MASK && f == 6 * G::HYP_BANDS + 8) return val_feat(maskv[nt]);                         // models.py:731-732
        return zero_feat();
      });
    static_assert(G::HYP_DEPTH == 6 || !G::HAS_HYPER, "hyper sheet is unrolled for depth 6, skip 4");
    NERFDS_TRAIN_ROW(to.hyper_h[0], to.hyper_h16[0], to.hyper_bits[0], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, in0);
    NERFDS_TRAIN_ROW(to.hyper_h[1], to.hyper_h16[1], to.hyper_bits[1], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.hyper_h[2], to.hyper_h16[2], to.hyper_bits[2], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b);
    NERFDS_TRAIN_ROW(to.hyper_h[3], to.hyper_h16[3], to.hyper_bits[3], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.hyper_h[4], to.hyper_h16[4], to.hyper_bits[4], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, a, b, in0);
    NERFDS_TRAIN_ROW(to.hyper_h[5], to.hyper_h16[5], to.hyper_bits[5], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, b, a);
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, hacc, b);
    NERFDS_TRAIN_HEAD(to.wamb, 2);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { wamb[nt][0] = hacc[0][nt][0]; wamb[nt][1] = hacc[0][nt][1]; }
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
    {
      L.sv[SV_WP + 3][sm.slot[nt]] = wamb[nt][0];
      L.sv[SV_WP + 4][sm.slot[nt]] = wamb[nt][1];
    }
  if constexpr (Pipe<G, PL>::HAS_SHARED) pipe.finish_segment(SEG_SHARED);
}

// ---- NerfMLP of one level (modules.py:243-313; models.py:1043-1047, 1268-1270) on one batch of 32 * NT samples whose warped
// point, ambient coordinates and rotation are parked at their slots (eval_shared, possibly of an earlier pass: the coarse
// samples of the fine level).  Parks sigma, rgb and the raw predicted normal at the slots.
template <class G, class PL, int NT, class LT, class TO = NoTrain, bool CP = false>
DEVI void eval_nerf(const KArgs& ka, Pipe<G, PL>& pipe, int level, int lane, const Samples<NT>& sm, LT& L,
                    const TO& to = TO(), size_t row = 0) {
  using D = Dims<G>;
  const int h = lane >> 5;
  constexpr int NSEG = (CP && PL::HAS_C) ? SEG_NERF_C : SEG_NERF;      // CP: the coarse level's own arithmetic and stream (PlanT::HAS_C)
  std::conditional_t<TO::ON, TrainCursor, Cursor> cur;
  cur.seg = NSEG;
  cur.pos = 0;
  cur.bt = D::SHARED_BIAS_TILES + level * D::NERF_BIAS_TILES;
  WAVE_SYNC();                                                // the parked state was written by the twin lane / an earlier pass
  float xw[NT][3], wamb[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
    for (int c = 0; c < 3; ++c) xw[nt][c] = L.sv[SV_WP + c][sm.slot[nt]];
    wamb[nt][0] = L.sv[SV_WP + 3][sm.slot[nt]];
    wamb[nt][1] = L.sv[SV_WP + 4][sm.slot[nt]];
  }
  constexpr int TW16 = G::TRUNK_W / 16, TW32 = G::TRUNK_W / 32;
  {
    constexpr int P = CP ? PL::TRUNK_C : PL::TRUNK, PR = CP ? PL::RGB_C : PL::RGB;
    Chunk<P> in0[NT][D::TRUNK_KC], a[NT][TW16], b[NT][TW16];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      build_chunks<P, D::TRUNK_KC>(in0[nt], h, [&](int f) {
        constexpr int I3 = D::ID3, PE = I3 + 6 * G::SP_BANDS;
        if (f < I3) return val_feat(xw[nt][f < 3 ? f : 0]);                                      // identity prefix
        if (f < PE) return posenc_feat<3>(f - I3, xw[nt], ka.win_sp);                            // models.py:502-507
        if (f < D::TRUNK_IN) return posenc_feat<2>(f - PE, wamb[nt], ka.win_hp);                 // models.py:510-516 (no identity)
        return zero_feat();
      });
    static_assert(G::TRUNK_DEPTH == 8 && G::TRUNK_SKIP == 4, "trunk is unrolled for depth 8, skip 4");
    NERFDS_TRAIN_ROW(to.trunk_h[0], to.trunk_h16[0], to.trunk_bits[0], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, a, in0);
    NERFDS_TRAIN_ROW(to.trunk_h[1], to.trunk_h16[1], to.trunk_bits[1], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.trunk_h[2], to.trunk_h16[2], to.trunk_bits[2], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, a, b);
    NERFDS_TRAIN_ROW(to.trunk_h[3], to.trunk_h16[3], to.trunk_bits[3], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.trunk_h[4], to.trunk_h16[4], to.trunk_bits[4], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, a, b, in0);
    NERFDS_TRAIN_ROW(to.trunk_h[5], to.trunk_h16[5], to.trunk_bits[5], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, b, a);
    NERFDS_TRAIN_ROW(to.trunk_h[6], to.trunk_h16[6], to.trunk_bits[6], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, a, b);
    NERFDS_TRAIN_ROW(to.trunk_h[7], to.trunk_h16[7], to.trunk_bits[7], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, b, a);          // b = trunk_output
    // (the activation-free bottleneck Dense, modules.py:255, is folded into rgb hidden_0 by the packer)
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, hacc, b);                    // alpha_mlp on trunk_output (modules.py:273-274)
    NERFDS_TRAIN_HEAD(to.alphav, Dims<G>::ALPHA_OUT);
    // rgb condition chunks: [posenc(viewdir) | posenc(normal in observation frame)]
    Chunk<PR> cond[NT][D::COND_KC];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      float nin[3] = {0.f, 0.f, 0.f};
      const int sl = sm.slot[nt];
      L.sv[SV_SIGMA][sl] = softplus_f(hacc[0][nt][0]);                    // models.py:577
      if constexpr (G::PREDICT_NORM) {
        float n[3] = {hacc[0][nt][1], hacc[0][nt][2], hacc[0][nt][3]};
        {
#pragma unroll
          for (int c = 0; c < 3; ++c) L.sv[SV_NORM + c][sl] = n[c];
        }
        normalize3(n);                                                      // models.py:1124
        if constexpr (G::HAS_WARP) {
          const float ax[3] = {L.sv[SV_AX][sl], L.sv[SV_AX + 1][sl], L.sv[SV_AX + 2][sl]};
          float Rm[9];
          rodrigues(Rm, ax, L.sv[SV_SN][sl], L.sv[SV_OMC][sl]);
#pragma unroll
          for (int c = 0; c < 3; ++c) nin[c] = Rm[c] * n[0] + Rm[3 + c] * n[1] + Rm[6 + c] * n[2];   // R^T n (models.py:1126)
        } else {
          nin[0] = n[0]; nin[1] = n[1]; nin[2] = n[2];
        }
        normalize3(nin);                                                    // models.py:1138
      } else {
        {
#pragma unroll
          for (int c = 0; c < 3; ++c) L.sv[SV_NORM + c][sl] = 0.f;
        }
      }
      build_chunks<PR, D::COND_KC>(cond[nt], h, [&](int f) {
        constexpr int I3 = D::ID3, VD = D::VD_FEATS;
        if (f < I3) return val_feat(L.rayc[RC_VD + (f < 3 ? f : 0)]);                           // identity prefix of posenc(viewdir)
        if (f < VD) return val_feat(L.rayc[RC_VDENC + ((f - I3) < 24 ? (f - I3) : 0)]);
        if (G::PREDICT_NORM && f < VD + I3) return val_feat(nin[(f - VD) < 3 ? (f - VD) : 0]);                     // identity prefix of posenc(normal)
        if (G::PREDICT_NORM && f < D::COND_IN) return posenc_feat<3>(f - VD - I3, nin, ka.win_nm);   // models.py:1142-1148
        return zero_feat();
      });
    }
    Chunk<PR> c[NT][G::RGB_W / 16];
    NERFDS_TRAIN_ROW(to.rgb_h, to.rgb_h16, to.rgb_bits, G::RGB_W);
    dense<G, PL, NT, G::RGB_W / 32, true>(pipe, cur, c, b, cond);       // K order [trunk_output | cond]
    head<G, PL, NT>(pipe, cur, hacc, c);
    NERFDS_TRAIN_HEAD(to.rgb_logit, 3);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      {
        const int s = sm.slot[nt];
        L.sv[SV_RGB + 0][s] = sigmoid_f(hacc[0][nt][0]);                       // models.py:576
        L.sv[SV_RGB + 1][s] = sigmoid_f(hacc[0][nt][1]);
        L.sv[SV_RGB + 2][s] = sigmoid_f(hacc[0][nt][2]);
      }
  }
  pipe.finish_segment(NSEG);
}
#undef NERFDS_TRAIN_ROW
#undef NERFDS_TRAIN_HEAD

}  // namespace nerfds
