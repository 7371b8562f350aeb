// Fused NeRF-DS ray kernel for gfx950 (MI355X / CDNA4).
//
// One wavefront renders one ray end to end: stratified sampling -> [MaskMLP -> SE(3) warp MLP + exp_se3 ->
// hyper-sheet MLP -> trunk / sigma / rgb NerfMLP] on every sample -> exclusive-cumprod compositing ->
// inverse-CDF resample + sort -> the same networks on the fine samples -> compositing -> one 104-byte record.
// Nothing per-sample ever goes to HBM (unless the optional per-sample record is requested).
//
// Reference functions restated here (paths under /root/reference/hypernerf/):
//   sample_along_rays model_utils.py:55-92      volumetric_rendering model_utils.py:95-159
//   piecewise_constant_pdf / sample_pdf model_utils.py:193-269   compute_depth_* model_utils.py:272-317
//   posenc / posenc_window model_utils.py:398-436                normalize_vector model_utils.py:438-442
//   MLP modules.py:57-83   NerfMLP.query_* modules.py:243-313   HyperSheetMLP modules.py:367-392   MaskMLP 409-434
//   SE3Field.warp warping.py:200-237   exp_se3/exp_so3/skew rigid_body.py:26-101
//   NerfModel.render_samples models.py:867-1417   NerfModel.__call__ models.py:1419-1565
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
// Development only: timing ablations (results are WRONG when any bit is set).  1: no LDS-DMA, 2: no stage barrier,
// 4: no tile epilogue, 8: no sin in the encodings, 16: no compositing / resampling, 32: no LDS->register weight reads
#ifndef NERFDS_ABLATE
#define NERFDS_ABLATE 0
#endif
// LDS scratch is private to a wave and a wave's DS ops complete in order: a compiler-level fence is all that is needed.
#ifndef NERFDS_DBG
#define NERFDS_DBG 0
#endif
#if NERFDS_DBG & 4
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

// All LDS of the kernel is ONE array (a second __shared__ object de-pipelines LDS-DMA code, cdna guide section 5):
// [0, RING_BYTES) weight ring, then one WaveLds scratch block per wave.
extern __shared__ __attribute__((aligned(16))) char g_smem[];

// ------------------------------------------------------------------------------------------------
// Operand containers
// ------------------------------------------------------------------------------------------------
template <int P> struct Chunk;   // 16 k-slots x 32 samples of activations (this lane: 8 slots of 1 sample)
template <> struct Chunk<P_BF16> { bf16x8 v; };
template <> struct Chunk<P_BF16X3> { bf16x8 hi, lo; };
template <> struct Chunk<P_F32> { float v[8]; };
template <> struct Chunk<P_F16> { f16x8 v; };
template <> struct Chunk<P_BF16X6> { bf16x8 hi, mid, lo; };

template <int P> struct WFrag;   // 32 out rows x 16 k-slots of weights (this lane: 8 slots of 1 row)
template <> struct WFrag<P_BF16> { bf16x8 v; };
template <> struct WFrag<P_BF16X3> { bf16x8 hi, lo; };
template <> struct WFrag<P_F32> { f32x4 a, b; };
template <> struct WFrag<P_F16> { f16x8 v; };
template <> struct WFrag<P_BF16X6> { bf16x8 hi, mid, lo; };

// relu on raw float bits: signed-integer max with 0 (one v_max_i32; fmaxf costs a canonicalising v_max on top).
DEVI float relu_f(float x) {
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
template <int P> DEVI void make_chunk(Chunk<P>& c, const float (&x)[8]);
#ifndef NERFDS_CPP_PIPE
#define NERFDS_CPP_PIPE 0
#endif
#ifndef NERFDS_CPP_PIPE_J0
#define NERFDS_CPP_PIPE_J0 1
#endif
#ifndef NERFDS_PK_RELU
#define NERFDS_PK_RELU 1
#endif
// Accumulator registers -> next layer's B operand, with optional ReLU.
template <int P, bool RELU> DEVI void make_act_chunk(Chunk<P>& c, const float (&x)[8]) {
  // (a packed v_pk_max_i16 on the converted pairs would be cheaper still, but hipcc then un-pairs the v_cvt_pk_bf16_f32)
#if NERFDS_PK_RELU
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
#endif
  float y[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = RELU ? relu_f(x[i]) : x[i];
  make_chunk<P>(c, y);
}
template <> DEVI void make_chunk<P_BF16>(Chunk<P_BF16>& c, const float (&x)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) c.v[i] = (__bf16)x[i];
}
#ifndef NERFDS_DMA_VOFF
#if defined(NERFDS_TRAIN_FWD) || defined(NERFDS_TRAIN_BWD)
#define NERFDS_DMA_VOFF 0     // the training kernels measured 1 % SLOWER with it (15.84 against 15.70 ms per step): they keep the scalar offsets
#else
#define NERFDS_DMA_VOFF 1     // render kernels: bf16 13.57 -> 13.47 ms per 65 536 rays, split bf16 39.95 -> 39.77 (profiles/r3_ab/ab_dma_voffset.txt)
#endif
#endif
#ifndef NERFDS_X3_DOT2
#define NERFDS_X3_DOT2 0
#endif
#ifndef NERFDS_X3_PAIRWISE
#define NERFDS_X3_PAIRWISE 1
#endif
template <> DEVI void make_chunk<P_BF16X3>(Chunk<P_BF16X3>& c, const float (&x)[8]) {
#if NERFDS_X3_PAIRWISE
  // pair by pair, float(hi) taken from the PACKED pair's bits (shift / mask): one v_cvt_pk per pair for hi and one for lo.  Element by element
  // hipcc converts some elements twice (once in the pair, once alone for the subtraction): 10 VALU per pair instead of 6 - 7.  Same bits out.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ uh, ul;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x2 r = {x[2 * k], x[2 * k + 1]};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
#if NERFDS_X3_DOT2 && defined(__HIP_DEVICE_COMPILE__)
    // EXPERIMENT (off): r - float(hi) straight from the packed pair, v_dot2c_f32_bf16 d = hi.lo * (-1) + hi.hi * 0 + r0 (and the mirror image for r1): no
    // unpacking of the pair, 2 VALU per pair instead of 4 (VALU per MFMA 2.89 -> 2.39).  Measured 39.34 against 38.75 ms per 65 536 rays (the dot
    // instruction is not a full-rate VALU op) AND a parity test fails with it (its arithmetic is not the exact fp32 subtraction).  Kept for the record.
    const bf16x2 hp = __builtin_bit_cast(bf16x2, hb);
    const bf16x2 m0 = __builtin_bit_cast(bf16x2, 0x0000bf80u), m1 = __builtin_bit_cast(bf16x2, 0xbf800000u);
    const f32x2 d = {__builtin_amdgcn_fdot2_f32_bf16(hp, m0, r[0], false), __builtin_amdgcn_fdot2_f32_bf16(hp, m1, r[1], false)};
#else
    const f32x2 d = {r[0] - __builtin_bit_cast(float, hb << 16), r[1] - __builtin_bit_cast(float, hb & 0xffff0000u)};
#endif
    uh[k] = hb;
    ul[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2));
  }
  c.hi = __builtin_bit_cast(bf16x8, uh);
  c.lo = __builtin_bit_cast(bf16x8, ul);
#else
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    __bf16 hi = (__bf16)x[i];
    c.hi[i] = hi;
    c.lo[i] = (__bf16)(x[i] - (float)hi);
  }
#endif
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
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, c.lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.lo, c.hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.hi, c.hi, acc, 0, 0, 0);
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
// nerf_ds / HyperNeRF bf16 and f16 kernels this way (-DNERFDS_NT=2 -DNERFDS_ASM_EPILOGUE=0: the compiler-scheduled C++ epilogue).
// History: round 2 measured the shape at 16.2 ms (no pipelining) / 16.6 ms (asm epilogue pieces pinned into the next chain, with the
// ordering point that makes the MFMA -> VALU hazard safe by construction) per 65 536 rays against 15.8 ms for 8 waves x one N-tile; an
// earlier build whose first piece could be scheduled right behind the previous group's last MFMA ran 15.2 ms - and gave 1-2 % of the
// rays different values from run to run.  After round 3's changes (level-independent networks once per position, spread DMA, DPP phases)
// the same shape with the plain C++ epilogue is level or ahead on every box tried (Makefile, profiles/r3_ab/ab_nt2_*.txt); with the asm
// pieces it is 1 % behind.  The source default stays one N-tile (the static graph's kernels, every two-unit plan).
#ifndef NERFDS_NT
#define NERFDS_NT 1
#endif
template <int PM, int PW, int PH, int PT, int PR> struct PlanT {
  static constexpr int MASK = PM, WARP = PW, HYP = PH, TRUNK = PT, RGB = PR;
  static constexpr Plan value() { return Plan{PM, PW, PH, PT, PR}; }
  static constexpr bool UNIFORM = PM == PW && PW == PH && PH == PT && PT == PR;
  // (plans that mix two-unit networks in keep one N-tile: a 128-wide split-bf16 network with two N-tiles is 256 registers of activations too)
  static constexpr int NT = (is_single(PM) && is_single(PW) && is_single(PH) && is_single(PT) && is_single(PR)) ? NERFDS_NT : 1;
  // 8 waves (two per SIMD, 256 registers each) when the 256-wide trunk runs on one-unit operands with one N-tile; a
  // trunk on two-unit operands (or two N-tiles) needs > 256 registers of activations: one 512-register wave per SIMD.
  static constexpr bool EIGHT_WAVES = is_single(PT) && NT == 1;
};
// STAGE_BYTES (graphs.h): one ring stage = 16 units
constexpr int NUM_STAGES = 4;          // ring depth
enum { SEG_SHARED = 0, SEG_NERF = 1 };  // the two weight streams (Pipe)
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
#ifdef NERFDS_TRAIN_FWD      // one level per launch: the shared nets and ONE NerfMLP
template <class G> constexpr int bias_tiles() { return Dims<G>::SHARED_BIAS_TILES + Dims<G>::NERF_BIAS_TILES; }
#else
template <class G> constexpr int bias_tiles() { return Dims<G>::SHARED_BIAS_TILES + 2 * Dims<G>::NERF_BIAS_TILES; }
#endif
template <class G> constexpr int bias_bytes() { return cdiv(bias_tiles<G>(), 8) * 1024; }

// 1: uniform split-bf16 plans issue the MFMAs of a tile pair interleaved (accum) and keep 8 units in the register ring.
// Measured 41.6 -> 40.8 ms per 65 536 rays (same-accumulator MFMA pairs with instructions between them: 2305 -> 590 of 9384).
#ifndef NERFDS_X3_INTERLEAVE
#define NERFDS_X3_INTERLEAVE 1
#endif
#ifndef NERFDS_RING_UNITS
#define NERFDS_RING_UNITS 4
#endif
#ifndef NERFDS_RING_UNITS_X3
#define NERFDS_RING_UNITS_X3 8
#endif
// 1: the LDS-DMA pieces of a stage are issued one by one between the MFMAs of the stage NS - 1 earlier, with a counted vmcnt
// at the boundaries (Pipe::boundary / spread_piece); 0: back to back behind the barrier.  Measured +0.9 % on the 8-wave
// kernels, +-0 on the 4-wave ones.  The training forward keeps its own (measured) scheme.
#ifndef NERFDS_SPREAD_DMA
#if defined(NERFDS_TRAIN_FWD) || defined(NERFDS_TRAIN_BWD)
#define NERFDS_SPREAD_DMA 0
#else
#define NERFDS_SPREAD_DMA 1
#endif
#endif
#ifndef NERFDS_SPREAD_AT
#define NERFDS_SPREAD_AT 1
#endif
#ifndef NERFDS_TRAIN_VMCNT
#define NERFDS_TRAIN_VMCNT 1      // 0: the training forward waits with vmcnt(0) at stage boundaries like the render kernels (A/B timing)
#endif
// stream lengths of a graph (graphs.h) or of a reversed network of the training backward (one stream, walked as SEG_NERF)
template <class G, class = void> struct StreamUnits {
  static constexpr int shared(Plan p) { return shared_units<G>(p); }
  static constexpr int nerf(Plan p) { return nerf_units<G>(p); }
};
template <class G> struct StreamUnits<G, std::void_t<decltype(G::BWD_FRAGS)>> {
  static constexpr int shared(Plan) { return 0; }
  static constexpr int nerf(Plan p) { return G::BWD_FRAGS * frag_parts(p.trunk); }
};
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
  static constexpr bool HAS_SHARED = SHARED_UNITS > 0;
  // stream positions count the zero padding at the end of each stream (graphs.h pad_units)
  static constexpr int SHARED_PAD = pad_units(SHARED_UNITS), NERF_PAD = pad_units(NERF_UNITS);
  static constexpr int seg_used(int seg) { return (seg == SEG_SHARED ? SHARED_PAD : NERF_PAD) / SU; }       // stages that hold data
  static constexpr int seg_stages(int seg) { return cdiv(seg_used(seg), NS) * NS; }                          // incl. hole stages
  static_assert((!HAS_SHARED || seg_used(SEG_SHARED) >= NS - 1) && seg_used(SEG_NERF) >= NS - 1, "the wrap prefetch needs NS - 1 stages in every segment");
  static constexpr int WAVES = wg_waves<PL>();
  static constexpr int PIECES = SU / WAVES;                       // 1 KiB LDS-DMA pieces per wave per stage
  // LDS -> register prefetch distance in units: a ds_read_b128 takes ~100+ cycles to return under load, a bf16 unit is
  // consumed in 32-64 MFMA cycles, so the reads must run several units ahead of the MFMAs.
  static constexpr int RD = (NERFDS_X3_INTERLEAVE && PL::UNIFORM && PL::TRUNK == P_BF16X3 && PL::NT == 1) ? NERFDS_RING_UNITS_X3 : NERFDS_RING_UNITS;
  static_assert(RD <= SU && RD >= 2, "prefetch reaches at most one stage ahead");
  u32x4 ring[RD];
  rsrc_t cur;       // stream of the segment being walked
  rsrc_t next;      // stream of the segment walked next (wrap-around prefetch)
  int lane16;
  int wave1k;       // wave index in the workgroup * 1024 (SGPR)

  // This wave's share of stage t of segment `seg` (t >= seg_stages: stage t - seg_stages of the next segment): LDS-DMA
  // (buffer_load ... lds), 1 KiB per instruction, no VGPRs.  Everything but "+ wave * 1024" (one s_add) and the descriptor is a
  // compile-time fact.
  DEVI void issue_stage(int seg, int t, int k0 = 0, int k1 = PIECES) {
    if (NERFDS_ABLATE & 1) return;
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
#if NERFDS_DMA_VOFF
      // the wave's share of the stream offset rides in the VECTOR offset (lane * 16 + wave * 1024, one register for the whole kernel), so the scalar
      // offset of every piece of every stage is a literal: as `constant + wave1k` each of the ~300 (stage, piece) offsets was a loop-invariant scalar
      // that hipcc hoisted out of the persistent loop and then spilled to VGPR lanes (586 SGPR spill slots, 0.3 v_readlane / v_writelane per MFMA)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrap ? next : cur, dst, 16, lane16 + wave1k, base + WAVES * k * 1024, 0, 0);
#else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrap ? next : cur, dst, 16, lane16, base + off, 0, 0);
#endif
    }
  }
  // Entering stage s: stages s and s + 1 are complete in LDS (so the LDS->register prefetch can run ahead across the
  // next boundary without a cold start), stage s + 2 may still be in flight, every wave is done with stage s - 1,
  // whose slot is refilled with stage s + NS - 1.
  DEVI void boundary(int seg, int s) {
    // vmcnt(0): every LDS-DMA this wave has issued (stages <= s + 2, the youngest a full stage ago) has landed.
    // A COUNTED vmcnt(N) is NOT safe here: on gfx9-family VM_CNT, loads and stores complete out of order with respect
    // to each other, so a younger store (ray-record store, register spill) retiring early lets the count drop below N
    // while an older LDS-DMA is still in flight -> stale weights (seen as 2e-2 errors on the fine level).
    // lgkmcnt(0): every LDS read this wave has issued has returned before the barrier.  In program order the reads of the
    // retiring stage s - 1 are all consumed by MFMAs above this point, but hipcc may sink a register-only MFMA - and with it the
    // s_waitcnt of its operand read - BELOW the asm and the barrier; another wave's DMA into that slot would then race the read.
    // (Waiting with vmcnt(0) only was 1.5 % faster and ran clean on the uniform kernels, but the mixed-precision kernel showed
    // run-to-run differences with it: kept safe.)
    static_assert(NS == 4, "protocol is written for a 4-stage ring");
#if ((defined(NERFDS_TRAIN_FWD) || defined(NERFDS_TRAIN_BWD)) && NERFDS_TRAIN_VMCNT) || NERFDS_SPREAD_DMA
    // Counted wait.  Training forward: the activation stores share VM_CNT with the LDS-DMA, and with vmcnt(0) every boundary also
    // waits for the wave's youngest stores to be acknowledged.  Spread DMA: the pieces of stage s + 2 were issued DURING stage
    // s - 1, the last of them a fraction of a stage ago.  vmcnt(PIECES) is enough and safe: loads complete in order AMONG LOADS, so
    // "at most PIECES operations outstanding" means every load older than the PIECES youngest loads has landed - and the PIECES
    // youngest loads are (at least as young as) the pieces of stage s + 2, which this boundary does not need (stages s and s + 1 are
    // older); stores or later loads in flight only make the wait longer, never shorter.  That argument needs stage s + 2 to have been
    // issued: when it is a hole (nothing issued) the wait is vmcnt(0).  Measured: -0.75 ms per training step against vmcnt(0).
    {
      static_assert(PIECES == 4 || PIECES == 2, "vmcnt(PIECES) below");
      const int tprev = s + NS - 2;              // the stage issued since the previous boundary (s == 0: by the previous segment's last stage)
      const bool prev_issued = tprev >= seg_stages(seg) || tprev < seg_used(seg);
#ifdef NERFDS_UNSAFE_VMCNT      // timing experiment only (results may read stale weights): how much of the step is the wait for the stores?
#define NERFDS_STR2(x) #x
#define NERFDS_STR(x) NERFDS_STR2(x)
      if (prev_issued) asm volatile("s_waitcnt vmcnt(" NERFDS_STR(NERFDS_UNSAFE_VMCNT) ") lgkmcnt(0)" ::: "memory");
      else
#endif
      if (prev_issued && PIECES == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      else if (prev_issued) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
#elif defined(NERFDS_BOUNDARY_NO_LGKM)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    if (!(NERFDS_ABLATE & 2)) __builtin_amdgcn_s_barrier();      // raw barrier (no compiler-added fences)
    if (NERFDS_DBG & 2) __builtin_amdgcn_s_sleep(4);
    // NERFDS_SPREAD_DMA: the pieces of stage s + NS - 1 are issued one by one between the MFMAs of stage s (spread_piece) instead of
    // back to back behind the barrier, where nothing covers their issue time (a hole stage has no MFMAs: issued here)
    if (!NERFDS_SPREAD_DMA || s >= seg_used(seg)) issue_stage(seg, s + NS - 1);
  }
  // unit u of the segment has been consumed: with NERFDS_SPREAD_DMA, the k-th piece of the stage NS - 1 ahead goes out after the
  // unit SPREAD_AT + k * (SU / PIECES) of the current stage (behind the barrier of this stage: its ring slot is free)
  DEVI void spread_piece(int seg, int u) {
    if (!NERFDS_SPREAD_DMA) return;
    constexpr int EVERY = SU / PIECES;
    if (u % EVERY == NERFDS_SPREAD_AT % EVERY) issue_stage(seg, u / SU + NS - 1, (u % SU) / EVERY, (u % SU) / EVERY + 1);
  }
  // the first NS - 1 stages of the first segment of the kernel
  DEVI void prologue(int seg) {
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue_stage(seg, t);
  }
  DEVI u32x4 unit(int u) const {
#if (NERFDS_ABLATE & 32) && defined(__HIP_DEVICE_COMPILE__)
    { u32x4 t = {0u, 0u, 0u, (unsigned)u}; asm volatile("" : "+v"(t)); return t; }
#endif
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
    if ((u + RD) / SU < seg_used(seg)) ring[u % RD] = unit(u + RD);
  }
  template <int P> DEVI WFrag<P> frag(int u) const {
    WFrag<P> w;
    if constexpr (P == P_BF16) {
      w.v = __builtin_bit_cast(bf16x8, ring[u % RD]);
    } else if constexpr (P == P_F16) {
      w.v = __builtin_bit_cast(f16x8, ring[u % RD]);
    } else if constexpr (P == P_BF16X3) {
      w.hi = __builtin_bit_cast(bf16x8, ring[u % RD]);
      w.lo = __builtin_bit_cast(bf16x8, ring[(u + 1) % RD]);
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
    if (NERFDS_SPREAD_DMA) {             // pieces whose trigger unit lies in the zero padding of the last stage
      constexpr int EVERY = SU / PIECES;
      const int last = (seg == SEG_SHARED ? SHARED_UNITS : NERF_UNITS) - 1, s = seg_used(seg) - 1;
#pragma unroll
      for (int k = 0; k < PIECES; ++k)
        if (s * SU + k * EVERY + NERFDS_SPREAD_AT % EVERY > last) issue_stage(seg, s + NS - 1, k, k + 1);
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
// Training forward (train_forward_kernel below): every hidden layer also writes its fp32 post-activation output to HBM, row-major
// [sample][width], for the backward pass.  `row` = this lane's sample row of the layer being computed, + 4 * (lane >> 5) floats.
// TrainOut::half_out / TrainBwd::g_half: the Makefile builds every training kernel twice, -DNERFDS_TRAIN_HALF=0 (fp32 stores: the steps with a
// tangent pass, NERFDS_TRAIN_G16=0) and =1 (f16 + ReLU bits / bf16 g: the plain step), and the host picks the launcher.  As a run-time test
// (a uniform branch per tile pair) `half` cuts every tile group into its own basic blocks, and hipcc schedules within a block.  Compile-time
// `half` alone is neutral (15.81 against 15.88 ms per step); together with the source-level pipeline of dense()'s TRAIN branch and one explicit
// scheduling region per group (sched_barrier) it is 15.58 (profiles/r3_ab/ab_tph.txt).  Without the macro (development builds) the run-time
// test remains.
// NERFDS_TRAIN_TAG is a template argument of the training kernels: two translation units that build the SAME template with different
// macros would otherwise emit one mangled kernel name with two bodies, and the runtime binds both launchers to ONE of them (this cost
// round 3 a wrong conclusion: the first two-build attempt silently ran the fp32-store kernel for both modes - a step that barely
// learns, loss 0.17302 instead of 0.1674 - and looked like "compile-time half is slower").
#ifdef NERFDS_TRAIN_HALF
#define NERFDS_TRAIN_TAG (1 + (NERFDS_TRAIN_HALF != 0) + 2 * (NERFDS_TRAIN_PIPE != 0))
#else
#define NERFDS_TRAIN_TAG 0
#endif
#ifdef NERFDS_TRAIN_HALF
#define NERFDS_HALF_TEST(c) (NERFDS_TRAIN_HALF != 0)
#else
#define NERFDS_HALF_TEST(c) ((c).half != 0)
#endif
struct TrainCursor : Cursor {
  float* row;
  // TrainOut::half_out: the layer goes out as f16 [sample][width] (`row16` = this lane's row + 4 * (lane >> 5) halves) plus one
  // "output > 0" bit per feature (`bits` = this lane's u16 run of the layer: one u16 per 32-feature tile, bit r = accumulator r)
  uint16_t* row16;
  uint16_t* bits;
  int half;
};
// Fused backward (train_backward_kernel): the tiles of a hidden layer are masked with the forward's ReLU bits (`mask`: the lane's
// u16 per tile, two tiles per register) and stored as fp32 g[sample][width]; the input-gradient tiles are stored (or added) into a
// buffer whose row stride `ld_in` is not a multiple of 32 - lanes past it write to `sink`.
struct BwdCursor : TrainCursor {
  unsigned mask[8];
  float* in_row;        // this lane's row of d_in + 4 * (lane >> 5)
  float* sink;
  int ld_in, in_h4;     // in_h4 = 4 * (lane >> 5)
  int in_acc;           // add to what is there (the skip layer's contribution came first)
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
// The same tile as f16 (round to nearest even; an activation beyond 65504 becomes inf and shows up as an inf weight gradient): four
// 8-byte stores; returns the tile's 16 ReLU bits, bit r <-> register r.  Two VALU per bit: the relu'd value has non-negative integer
// bits, so 0 - bits is negative exactly when the output is > 0, and v_alignbit shifts that sign bit in (registers 15 .. 0).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
// A tile of 16-bit values (f16 activations, bf16 g): pk[2g], pk[2g + 1] = this lane's features 8g + 4h + 0..3 of its sample, as packed pairs.
// NERFDS_STORE16_WIDE: the two lanes of a sample (l, l + 32) trade halves with v_permlane32_swap - afterwards lane half h owns features
// 16h .. 16h + 15 of the tile - and each lane writes TWO 16-byte pieces instead of four 8-byte ones (`row16` then points at the lane's
// feature 16h of tile 0, not 4h).  A training kernel runs one wave per SIMD and a global store occupies the wave for its whole issue
// (address + data transfer of 64 lanes), MFMA pipe idle: what the stores cost is their NUMBER, not their bytes.
#ifndef NERFDS_STORE16_WIDE
#define NERFDS_STORE16_WIDE 1
#endif
constexpr int ROW16_H = NERFDS_STORE16_WIDE ? 16 : 4;      // offset of lane half 1 in a row of 16-bit features
DEVI void store_tile_pk16(uint16_t* row_tile, const unsigned (&pk)[8]) {
#if NERFDS_STORE16_WIDE && defined(__HIP_DEVICE_COMPILE__)
  unsigned x[4], y[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // swaps pk[i] of lanes 32..63 with pk[4 + i] of lanes 0..31: half 0 ends with (own, partner's) features of groups 0 and 1, half 1 with
    // (partner's, own) features of groups 2 and 3
    const u32x2 r = __builtin_amdgcn_permlane32_swap(pk[i], pk[4 + i], false, false);
    x[i] = r[0]; y[i] = r[1];
  }
  const u32x4 s0 = {x[0], x[1], y[0], y[1]}, s1 = {x[2], x[3], y[2], y[3]};
  *reinterpret_cast<u32x4*>(row_tile) = s0;
  *reinterpret_cast<u32x4*>(row_tile + 8) = s1;
#else
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const u32x2 v = {pk[2 * g], pk[2 * g + 1]};
    *reinterpret_cast<u32x2*>(row_tile + 8 * g) = v;
  }
#endif
}
// NERFDS_SIGN_BITS = 1 (EXPERIMENT, off): the bit of an output is NOT its sign bit - one v_alignbit per register on the raw accumulator, one v_not
// per tile - and the ReLU itself on the converted pairs (v_pk_max_i16 with 0): 33 VALU per tile instead of 56.  Measured 15.72 - 15.78 against
// 15.74 - 15.91 ms per step (within the noise: the forward is not VALU-bound), and NOT the reference's derivative: an accumulator that
// is exactly +0 gets its bit set, and exact zeros are not rare here (a unit whose inputs are all zero - closed windows, zero-initialised
// biases - sits exactly on its bias): the loss after 13 steps moved by 1.3e-4, three times the run-to-run spread.  Kept for the record.
#ifndef NERFDS_SIGN_BITS
#define NERFDS_SIGN_BITS 0
#endif
template <bool RELU> DEVI unsigned store_tile_half(uint16_t* row_tile, const f32x16& acc) {
  unsigned bits = 0;
#if NERFDS_SIGN_BITS
  if constexpr (RELU) {
    // (the accumulator's bits through ONE bit cast of the whole vector: `bit_cast<unsigned>(acc[r])` per register is folded wrongly by hipcc 7.2 -
    //  every v_alignbit then read register 0, the misfold apply_mask documents; seen in the ISA and as a 100 % error of the embedding gradients)
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    const u32x16 ab = __builtin_bit_cast(u32x16, acc);
#pragma unroll
    for (int r = 15; r >= 0; --r) bits = __builtin_amdgcn_alignbit(bits, ab[r], 31);   // (bits << 1) | sign
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    unsigned pk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f32x2 f = {acc[2 * k], acc[2 * k + 1]};
      const s16x2 z = {0, 0};
      pk[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(f, f16x2)), z));
    }
    store_tile_pk16(row_tile, pk);
    return ~bits & 0xffffu;
  }
#endif
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
// The g arrays of the fused backward as bf16 (TrainBwd::g_half).
// The tile's two chunks in split bf16 are what the next (earlier) layer multiplies; their hi parts - bf16(acc[0..7]), bf16(acc[8..15]),
// round to nearest even - ARE the bf16 copy of g: no second conversion, no packing.
DEVI void store_tile_bf16(uint16_t* row_tile, const Chunk<P_BF16X3>& c0, const Chunk<P_BF16X3>& c1) {
  const u32x4 pa = __builtin_bit_cast(u32x4, c0.hi), pb = __builtin_bit_cast(u32x4, c1.hi);
  const unsigned pk[8] = {pa[0], pa[1], pa[2], pa[3], pb[0], pb[1], pb[2], pb[3]};
  store_tile_pk16(row_tile, pk);
}
// Fused backward: zero the accumulator registers whose ReLU bit is clear.  (Written as a select on purpose: the 2-VALU form
// "x & sign-extended bit" through __builtin_amdgcn_sbfe on this (shifted, masked) operand is folded wrongly by hipcc 7.2 - every
// register came out as register 0's value; found with tools/chain_diag.py.)
// (Round 3 tried the two-VALU form again with the bit field as an opaque asm - v_bfe_i32, then `bits(acc[r]) & m` in C++: the SAME misfold,
// every register anded with register 0 - so the fold is in `extractelement + bitcast + and`, not in sbfe.  The chains are not VALU-bound
// anyway: 35 % fewer VALU per MFMA in them changed their time by +-0.)
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
    f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
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
template <class G, class PL, int NT, int TP, int P, int K, class SLOT>
DEVI void accum(f32x16 (&acc)[TP][NT], Pipe<G, PL>& pipe, Cursor& cur, const Chunk<P> (&in)[NT][K], int& j, SLOT&& slot) {
  using PP = Pipe<G, PL>;
  constexpr int NP = frag_parts(P);
  if constexpr (NERFDS_X3_INTERLEAVE && P == P_BF16X3 && TP == 2 && NT == 1 && PL::UNIFORM) {
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
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.hi, c.lo, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1.hi, c.lo, acc[1][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.lo, c.hi, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1.lo, c.hi, acc[1][0], 0, 0, 0);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0.hi, c.hi, acc[0][0], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1.hi, c.hi, acc[1][0], 0, 0, 0);
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
#if (NERFDS_ABLATE & 4) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { f32x16 t = acc[nt]; asm volatile("" ::"v"(t)); }
  return;
#endif
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
// set) - 16 VALU per tile where the source form (v_max on fp32, then convert) needs 24, and hipcc un-pairs the
// conversions when the packed max is written in C++.  hipcc does not pad the MFMA -> VALU hazard for an asm that
// reads accumulators (checked in the ISA), so the block opens with the 12 wait states itself when `wait` is set (the
// first block of a tile group; dense() pins the last MFMA of every accumulator of the group above it, so the later
// blocks are covered by the first one's instructions).
#ifndef NERFDS_ASM_EPILOGUE
#define NERFDS_ASM_EPILOGUE 1
#endif
#ifdef NERFDS_EPI_TAIL_NOP
#define NERFDS_EPI_TAIL "\n\ts_nop 1"
#else
#define NERFDS_EPI_TAIL ""
#endif
template <int P> DEVI void tile_epilogue_asm(Chunk<P>& c0, Chunk<P>& c1, const f32x16& a, bool wait) {
  static_assert(is_single(P), "packed-half epilogue");
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned o0, o1, o2, o3, o4, o5, o6, o7;
#if NERFDS_ABLATE & 64      // timing experiment: conversions only, no ReLU (half the epilogue's VALU, same registers and copies; wrong results)
#define NERFDS_EPI_RELU8 "s_nop 0"
#define NERFDS_EPI_RELU2 "s_nop 0"
#else
#define NERFDS_EPI_RELU8 "v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tv_pk_max_i16 %2, %2, 0\n\tv_pk_max_i16 %3, %3, 0\n\t" \
                         "v_pk_max_i16 %4, %4, 0\n\tv_pk_max_i16 %5, %5, 0\n\tv_pk_max_i16 %6, %6, 0\n\tv_pk_max_i16 %7, %7, 0"
#define NERFDS_EPI_RELU2 "v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0"
#endif
#define NERFDS_EPI_BODY(CVT)                                                                                              \
  CVT " %0, %8, %9\n\t" CVT " %1, %10, %11\n\t" CVT " %2, %12, %13\n\t" CVT " %3, %14, %15\n\t"                            \
  CVT " %4, %16, %17\n\t" CVT " %5, %18, %19\n\t" CVT " %6, %20, %21\n\t" CVT " %7, %22, %23\n\t"                          \
  NERFDS_EPI_RELU8 NERFDS_EPI_TAIL
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

constexpr int mfmas_per_step(int prec) { return prec == P_BF16X3 ? 3 : prec == P_BF16X6 ? 6 : prec == P_F32 ? 8 : 1; }
template <class T> struct seg_mfmas;
template <int P, int NT, int K> struct seg_mfmas<Chunk<P>[NT][K]> { static constexpr int value = K * NT * mfmas_per_step(P); };
template <class... Ins> struct seg_mfma_total { static constexpr int value = (seg_mfmas<Ins>::value + ... + 0); };
template <class T> struct seg_chunks;
template <int P, int NT, int K> struct seg_chunks<Chunk<P>[NT][K]> { static constexpr int value = K; };
template <class... Ins> struct seg_total { static constexpr int value = (seg_chunks<Ins>::value + ... + 0); };

// A quarter of tile_epilogue_asm: accumulator elements [4q, 4q + 4) -> two packed registers.  `order` is the accumulator the
// MFMAs of the current slot have just written: as an in/out operand of this volatile asm it keeps the piece BELOW that MFMA and
// above the next MFMA on the same accumulator, i.e. inside the chain (the wave's other MFMAs float around it freely).
template <int P> DEVI void epilogue_piece_asm(unsigned& o0, unsigned& o1, float a0, float a1, float a2, float a3, f32x16& order) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (P == P_BF16)
    asm volatile("v_cvt_pk_bf16_f32 %0, %3, %4\n\tv_cvt_pk_bf16_f32 %1, %5, %6\n\t" NERFDS_EPI_RELU2
                 : "=&v"(o0), "=&v"(o1), "+v"(order) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
  else
    asm volatile("v_cvt_pk_f16_f32 %0, %3, %4\n\tv_cvt_pk_f16_f32 %1, %5, %6\n\t" NERFDS_EPI_RELU2
                 : "=&v"(o0), "=&v"(o1), "+v"(order) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
#endif
}

// One dense layer with OT output tiles of 32 features, computed TILE_PAIR tiles at a time (the stream interleaves the
// fragments of the tiles of a pair chunk by chunk, pack.h); inputs are one or more chunk arrays in stream order (each
// in its own precision), the output chunks are produced in the precision PO of the tensor they form.
// The accumulators start from the bias (the first MFMA of a tile reads the bias registers as its C operand; the N-tiles of
// a wave share one copy).
// NERFDS_PIPE_EPI (one-unit output formats): the epilogue of tile group g is cut into pieces of 4 VALU and issued inside the
// MFMA chain of group g + 1, from its third slot on, each piece ordered behind the MFMA of its slot through that MFMA's
// accumulator.  With ONE wave per SIMD (two N-tiles per wave) nothing else covers the ~100 instructions between two groups;
// with two waves per SIMD the partner wave does, and it is off.
#ifndef NERFDS_PIPE_EPI
#define NERFDS_PIPE_EPI (NERFDS_NT > 1)
#endif
// Carrying a layer's last group into the next layer's first chain (dense(): Carry) is built and correct, but measured slower:
// hipcc parks the 64 carried accumulators in scratch (857 scratch loads, each with a vmcnt(0) that also drains the LDS-DMA):
// 19.9 ms (128-wide layers only) / 22.2 ms (all layers) against 15.2 ms without.  0 = every layer finishes its last group at once.
// Pipelined epilogue for the split-bf16 kernel (dense(): third branch): built, parity green and deterministic, and measured
// at 46.8 ms against 46.3 ms without per 65 536 rays - that kernel sits at the power limit as well.  Off.
#ifndef NERFDS_PIPE_EPI_X3
#define NERFDS_PIPE_EPI_X3 0
#endif
// 1: the training kernels issue the epilogue of a tile group inside the next group's MFMA chain (dense(): TRAIN branch)
#ifndef NERFDS_TRAIN_PIPE
#define NERFDS_TRAIN_PIPE 0
#endif
#ifndef NERFDS_TRAIN_SGB
// > 0: VALU per MFMA asked of the scheduler in the pipelined training regions (llvm.amdgcn.sched.group.barrier).  Built and measured with 4 and 6:
// 15.7 -> 21.6 ms per step - the forward spills 55 - 67 registers under the requested interleave (3.4 -> 6.8 ms on the fine level), the
// chains lose 3 - 15 %, and the translation unit takes 8 minutes to compile.  Off.
#define NERFDS_TRAIN_SGB 0
#endif
#ifndef NERFDS_PIPE_J0
#define NERFDS_PIPE_J0 2
#endif
#ifndef NERFDS_CARRY_MAX_OT
#define NERFDS_CARRY_MAX_OT 0
#endif
// Split-bf16 piece: accumulator elements (2q, 2q + 1) -> one packed hi pair and one packed lo pair, the arithmetic of
// make_act_chunk<P_BF16X3, true>: r = relu(x); hi = bf16(r); lo = bf16(r - float(hi)).  8 VALU; `order` as in epilogue_piece_asm.
DEVI void epilogue_piece_x3_asm(unsigned& hi, unsigned& lo, float a0, float a1, f32x16& order) {
#if defined(__HIP_DEVICE_COMPILE__)
  float t0, t1;
  unsigned h0, h1;
  asm volatile("v_max_i32 %2, 0, %7\n\tv_max_i32 %3, 0, %8\n\tv_cvt_pk_bf16_f32 %0, %2, %3\n\t"
               "v_lshlrev_b32 %4, 16, %0\n\tv_and_b32 %5, 0xffff0000, %0\n\tv_sub_f32 %2, %2, %4\n\tv_sub_f32 %3, %3, %5\n\t"
               "v_cvt_pk_bf16_f32 %1, %2, %3"
               : "=&v"(hi), "=&v"(lo), "=&v"(t0), "=&v"(t1), "=&v"(h0), "=&v"(h1), "+v"(order) : "v"(a0), "v"(a1));
#endif
}
// issues pieces [q0, q1) of a pending split-bf16 group (8 pieces per accumulator)
template <int NT, int W>
DEVI void pending_pieces_x3(int q0, int q1, const f32x16 (&src)[TILE_PAIR][NT], unsigned (&ph)[TILE_PAIR][NT][8], unsigned (&pl)[TILE_PAIR][NT][8],
                            Chunk<P_BF16X3> (&dst)[NT][W], int first_chunk, f32x16& order) {
#pragma unroll
  for (int q = q0; q < q1; ++q) {
    if (q >= TILE_PAIR * NT * 8) break;
    const int a = q / 8, i8 = q % 8, ptp = a / NT, pnt = a % NT;
    const f32x16& x = src[ptp][pnt];
    epilogue_piece_x3_asm(ph[ptp][pnt][i8], pl[ptp][pnt][i8], x[2 * i8], x[2 * i8 + 1], order);
    if ((i8 & 3) == 3) {          // a chunk (8 values) of the pending group is complete
      const int hf = i8 >> 2;
      const u32x4 rh = {ph[ptp][pnt][4 * hf], ph[ptp][pnt][4 * hf + 1], ph[ptp][pnt][4 * hf + 2], ph[ptp][pnt][4 * hf + 3]};
      const u32x4 rl = {pl[ptp][pnt][4 * hf], pl[ptp][pnt][4 * hf + 1], pl[ptp][pnt][4 * hf + 2], pl[ptp][pnt][4 * hf + 3]};
      dst[pnt][first_chunk + 2 * ptp + hf].hi = __builtin_bit_cast(bf16x8, rh);
      dst[pnt][first_chunk + 2 * ptp + hf].lo = __builtin_bit_cast(bf16x8, rl);
    }
  }
}

// The epilogue of a layer's LAST tile group, carried into the first MFMA chain that follows (the next layer's first group, or
// the head): its accumulators; the chunks it produces are the last 2 * TP chunks of that chain's first input array.
template <int NT> struct Carry {
  f32x16 acc[TILE_PAIR][NT];
  bool live = false;
};
// issues pieces [q0, q1) of a pending group; chunk destination: dst[nt][first_chunk + 2 * tp + half]
template <int P, int NT, int W>
DEVI void pending_pieces(int q0, int q1, const f32x16 (&src)[TILE_PAIR][NT], unsigned (&pk)[TILE_PAIR][NT][8], Chunk<P> (&dst)[NT][W], int first_chunk,
                         f32x16& order) {
#pragma unroll
  for (int q = q0; q < q1; ++q) {
    if (q >= TILE_PAIR * NT * 4) break;
    const int a = q / 4, i4 = q % 4, ptp = a / NT, pnt = a % NT;
    const f32x16& x = src[ptp][pnt];
    epilogue_piece_asm<P>(pk[ptp][pnt][2 * i4], pk[ptp][pnt][2 * i4 + 1], x[4 * i4], x[4 * i4 + 1], x[4 * i4 + 2], x[4 * i4 + 3], order);
    if (i4 & 1) {          // a chunk (8 values) of the pending group is complete
      const int hf = i4 >> 1;
      const u32x4 r = {pk[ptp][pnt][4 * hf], pk[ptp][pnt][4 * hf + 1], pk[ptp][pnt][4 * hf + 2], pk[ptp][pnt][4 * hf + 3]};
      dst[pnt][first_chunk + 2 * ptp + hf].v = __builtin_bit_cast(decltype(dst[0][0].v), r);
    }
  }
}
// Finishes a carried epilogue outside any chain (before code that reads the chunks with VALU, or at the end of a network whose
// consumer is not a dense layer / head over those chunks).
template <int P, int NT, int W> DEVI void flush_carry(Carry<NT>& carry, Chunk<P> (&dst)[NT][W]) {
  if constexpr (is_single(P) && NERFDS_PIPE_EPI && NERFDS_ASM_EPILOGUE && !(NERFDS_ABLATE & 4)) {
    if (carry.live) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
      for (int tp = 0; tp < TILE_PAIR; ++tp)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(carry.acc[tp][nt]));
#endif
#pragma unroll
      for (int tp = 0; tp < TILE_PAIR; ++tp)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          tile_epilogue_asm<P>(dst[nt][W - 2 * TILE_PAIR + 2 * tp], dst[nt][W - 2 * TILE_PAIR + 2 * tp + 1], carry.acc[tp][nt], tp == 0 && nt == 0);
    }
  }
  carry.live = false;
}

template <class A, class... Rest> DEVI A& first_of(A& a, Rest&...) { return a; }
template <class T> struct chunk_prec;
template <int P, int NT, int K> struct chunk_prec<Chunk<P>[NT][K]> { static constexpr int value = P; static constexpr int chunks = K; };

template <class G, class PL, int NT, int OT, bool RELU, class CUR, int PO, class... Ins>
DEVI void dense(Pipe<G, PL>& pipe, CUR& cur, Carry<NT>& carry, Chunk<PO> (&out)[NT][2 * OT], Ins&... ins) {
  constexpr int TP = TILE_PAIR;
  constexpr bool BWD_IN = std::is_same_v<CUR, BwdInCursor>;
  constexpr bool BWD = std::is_same_v<CUR, BwdCursor> || BWD_IN;
  constexpr bool TRAIN = std::is_same_v<CUR, TrainCursor> || BWD;
  static_assert(!TRAIN || (NT == 1 && !is_single(PO)), "the training kernels run two-unit plans: one N-tile, C++ epilogue");
  static_assert(OT % TP == 0, "layers have an even number of 32-feature tiles");
  // (uniform one-unit plans only: in the mixed plan - f16 networks around a split-bf16 warp field - the asm epilogue build gave
  // run-to-run differences on ~1 % of the rays of the fine level; the C++ epilogue build of the same kernel is clean)
  constexpr bool ASM_EPI = NERFDS_ASM_EPILOGUE && !(NERFDS_ABLATE & 4) && is_single(PO) && RELU && PL::UNIFORM;
  const int hb = bias_base(pipe.lane16);
  auto no_slot = [](int, int) {};
  if constexpr (ASM_EPI && NERFDS_PIPE_EPI) {
    constexpr int SLOTS = TP * seg_total<Ins...>::value;       // (chunk, tile) steps per group
    constexpr int NPIECE = TP * NT * 4, J0 = NERFDS_PIPE_J0;
    constexpr int PPS = cdiv(NPIECE, SLOTS - J0 > 0 ? SLOTS - J0 : 1);
    using In0 = std::remove_reference_t<decltype(first_of(ins...))>;
    constexpr int PIN = chunk_prec<In0>::value, KIN = chunk_prec<In0>::chunks;
    f32x16 prev[TP][NT];
    unsigned pk[TP][NT][8];
#pragma unroll
    for (int ot = 0; ot < OT; ot += TP) {
      f32x16 acc[TP][NT];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const f32x16 bv = load_bias(cur.bt + ot + tp, hb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[tp][nt] = bv;
      }
      // Ordering point: the previous group's chains end above it, this group's chains (they read these bias-initialised
      // accumulators as their first C operand) start below it.  The pieces are volatile asm ordered behind the MFMA of slot
      // J0, i.e. at least 2 * J0 + 1 MFMAs below this point: the MFMA -> VALU hazard of the accumulators they read (12 wait
      // states, not padded by hipcc for asm) is covered by construction, not by the scheduler's mood (with the previous
      // group's last MFMA free to sink next to the first piece, 1-2 % of the rays came out different from run to run).
#if defined(__HIP_DEVICE_COMPILE__)
      if (ot > 0) {
        static_assert(TP == 2 && (NT == 1 || NT == 2), "ordering point written for 2 x NT accumulators");
        if constexpr (NT == 2)
          asm volatile("" : "+v"(prev[0][0]), "+v"(prev[0][1]), "+v"(prev[1][0]), "+v"(prev[1][1]),
                            "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]));
        else
          asm volatile("" : "+v"(prev[0][0]), "+v"(prev[1][0]), "+v"(acc[0][0]), "+v"(acc[1][0]));
      }
#endif
      int j = 0;
      auto slot = [&](int jj, int tp_now) {
        if (jj < J0) return;
        if (ot == 0) {
          // the previous layer's last group: its chunks are the tail of this layer's first input (consumed at the end of the chain)
          if constexpr (is_single(PIN) && (KIN - 2 * TP) * TP - J0 > 0) {
            // every piece must be issued before the slot that consumes the first carried chunk, (KIN - 2 TP) * TP
            constexpr int CPPS = cdiv(NPIECE, (KIN - 2 * TP) * TP - J0);
            if (carry.live) pending_pieces<PIN, NT>((jj - J0) * CPPS, (jj - J0 + 1) * CPPS, carry.acc, pk, first_of(ins...), KIN - 2 * TP, acc[tp_now][NT - 1]);
          }
        } else {
          pending_pieces<PO, NT>((jj - J0) * PPS, (jj - J0 + 1) * PPS, prev, pk, out, 2 * (ot - TP), acc[tp_now][NT - 1]);
        }
      };
      (accum<G, PL, NT, TP>(acc, pipe, cur, ins, j, slot), ...);
      if (ot == 0) carry.live = false;
#pragma unroll
      for (int tp = 0; tp < TP; ++tp)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) prev[tp][nt] = acc[tp][nt];
    }
    // the last group of the layer is finished inside whatever chain comes next - if that chain reads other chunks first
    // (layers of more than one tile group); a single-group layer's output is needed by the very first MFMA that follows
#pragma unroll
    for (int tp = 0; tp < TP; ++tp)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) carry.acc[tp][nt] = prev[tp][nt];
    carry.live = true;
    // (NERFDS_CARRY_MAX_OT: layers wider than that finish their last group at once - the carried accumulators are 64 more
    // live registers on top of the 256 that the activations of a 256-wide layer take with two N-tiles)
    if constexpr ((2 * OT - 2 * TP) - 2 <= 0 || OT > NERFDS_CARRY_MAX_OT) flush_carry<PO, NT>(carry, out);
  } else if constexpr (PO == P_BF16X3 && RELU && NERFDS_PIPE_EPI_X3 && PL::UNIFORM && !(NERFDS_ABLATE & 4) && (OT > TP)) {
    // Split bf16 runs one 512-register wave per SIMD: nothing covers a tile group's epilogue (2 x 64 VALU against 48 - 96 MFMAs),
    // so the epilogue of group g is issued as 8-VALU asm pieces inside the MFMA chains of group g + 1 (same ordering rules as the
    // one-unit branch above); the layer's last group is finished by the compiler-scheduled C++ epilogue.
    constexpr int SLOTS = TP * seg_total<Ins...>::value;
    constexpr int NPIECE = TP * NT * 8, J0 = NERFDS_PIPE_J0;
    constexpr int PPS = cdiv(NPIECE, SLOTS - J0 > 0 ? SLOTS - J0 : 1);
    f32x16 prev[TP][NT];
    unsigned ph[TP][NT][8], pl[TP][NT][8];
#pragma unroll
    for (int ot = 0; ot < OT; ot += TP) {
      f32x16 acc[TP][NT];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const f32x16 bv = load_bias(cur.bt + ot + tp, hb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[tp][nt] = bv;
      }
#if defined(__HIP_DEVICE_COMPILE__)
      if (ot > 0) {          // ordering point (see the one-unit branch): previous chains end above, this group's chains start below
        static_assert(TP == 2 && NT == 1, "ordering point written for 2 accumulators");
        asm volatile("" : "+v"(prev[0][0]), "+v"(prev[1][0]), "+v"(acc[0][0]), "+v"(acc[1][0]));
      }
#endif
      int j = 0;
      auto slot = [&](int jj, int tp_now) {
        if (jj < J0 || ot == 0) return;
        pending_pieces_x3<NT>((jj - J0) * PPS, (jj - J0 + 1) * PPS, prev, ph, pl, out, 2 * (ot - TP), acc[tp_now][NT - 1]);
      };
      (accum<G, PL, NT, TP>(acc, pipe, cur, ins, j, slot), ...);
#pragma unroll
      for (int tp = 0; tp < TP; ++tp)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) prev[tp][nt] = acc[tp][nt];
    }
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) tile_epilogue<PO, NT, RELU>(out, OT - TP + tp, prev[tp]);
  } else if constexpr (NERFDS_CPP_PIPE && !TRAIN && is_single(PO) && RELU && (OT > TP)) {
    // One-unit render kernels, two N-tiles (one wave per SIMD), software-pipelined at the source level like the training branch below: the
    // accumulators of group g rest in `prev`; their conversion (one chunk = 4 v_cvt_pk + 4 v_pk_max_i16 per piece) is issued between the
    // MFMA steps of group g + 1, one scheduling region per group.
    constexpr int SLOTS = TP * seg_total<Ins...>::value;
    constexpr int NPIECE = TP * NT * 2, J0 = NERFDS_CPP_PIPE_J0;
    constexpr int PPS = cdiv(NPIECE, SLOTS - J0 > 0 ? SLOTS - J0 : 1);
    constexpr int IN_CHAIN = (SLOTS - J0) * PPS < NPIECE ? (SLOTS - J0 > 0 ? (SLOTS - J0) * PPS : 0) : NPIECE;
    f32x16 prev[TP][NT];
    auto piece = [&](int q, int pot) {
      const int tp = q / (2 * NT), nt = (q / 2) % NT, sub = q % 2, t = pot + tp;
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = prev[tp][nt][8 * sub + i];
      make_act_chunk<PO, RELU>(out[nt][2 * t + sub], x);
    };
#pragma unroll
    for (int ot = 0; ot < OT; ot += TP) {
      f32x16 acc[TP][NT];
#pragma unroll
      for (int tp = 0; tp < TP; ++tp) {
        const f32x16 bv = load_bias(cur.bt + ot + tp, hb);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[tp][nt] = bv;
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
        for (int q = IN_CHAIN; q < NPIECE; ++q) piece(q, ot - TP);
      }
#pragma unroll
      for (int tp = 0; tp < TP; ++tp)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) prev[tp][nt] = acc[tp][nt];
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) piece(q, OT - TP);
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
#endif
  } else if constexpr (TRAIN && !BWD_IN && NERFDS_TRAIN_PIPE && (OT > TP) && (BWD ? PO == P_BF16X3 : true)) {
    // Training forward / backward chain, software-pipelined at the source level: the epilogue of tile group g - conversion into the next
    // layer's operand, ReLU bits / mask, 16-bit stores - is cut into pieces that are issued between the MFMA steps of group g + 1 (the
    // accumulators of group g rest in `prev`).  A training kernel runs ONE wave per SIMD: whatever is issued behind a group's last MFMA
    // runs with the matrix pipe idle, whatever is issued between two MFMA steps of the next group runs under them.  The run-time
    // `half` branch inside the store pieces keeps every piece in its own basic block, i.e. where it was put.
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
          if (NERFDS_HALF_TEST(cur)) store_tile_bf16(cur.row16 + 32 * t, out[0][2 * t], out[0][2 * t + 1]);
          else store_tile<false>(cur.row + 32 * t, prev[tp]);
        } else {
          if (NERFDS_HALF_TEST(cur)) two |= store_tile_half<RELU>(cur.row16 + 32 * t, prev[tp]) << (16 * tp);
          else store_tile<RELU>(cur.row + 32 * t, prev[tp]);
        }
      } else if constexpr (!BWD) {
        if (NERFDS_HALF_TEST(cur)) { *reinterpret_cast<unsigned*>(cur.bits + pot) = two; two = 0; }
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
#if defined(NERFDS_TRAIN_HALF) && defined(__HIP_DEVICE_COMPILE__)
#if NERFDS_TRAIN_SGB
      // ask for the interleave explicitly: after every MFMA of the region a few of the previous group's VALU (and a ring read every other
      // MFMA): llvm.amdgcn.sched.group.barrier - masks 0x8 MFMA, 0x2 VALU, 0x100 DS read, 0x40 VMEM write
      if (ot > 0) {
        constexpr int NM = TP * seg_mfma_total<Ins...>::value;
        constexpr int VPM = (280 + NM - 1) / NM < NERFDS_TRAIN_SGB ? (280 + NM - 1) / NM : NERFDS_TRAIN_SGB;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (i % (NM / 5 > 0 ? NM / 5 : 1) == 1) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
        }
      }
#endif
      // with `half` a compile-time fact nothing else separates the groups: one scheduling region per group (its MFMAs + the previous
      // group's epilogue), so that hipcc interleaves THOSE and does not pull later groups' work - and their registers - forward
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) piece(q, OT - TP);                 // the layer's last group: behind its own chain
#if defined(NERFDS_TRAIN_HALF) && defined(__HIP_DEVICE_COMPILE__)
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
          if (!NERFDS_HALF_TEST(cur)) store_tile<false>(cur.row + 32 * (ot + tp), acc[tp][0]);
        }
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) tile_epilogue<PO, NT, false>(out, ot + tp, acc[tp]);
        if constexpr (PO == P_BF16X3) {
          if (NERFDS_HALF_TEST(cur)) {
#pragma unroll
            for (int tp = 0; tp < TP; ++tp) store_tile_bf16(cur.row16 + 32 * (ot + tp), out[0][2 * (ot + tp)], out[0][2 * (ot + tp) + 1]);
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
          if (NERFDS_HALF_TEST(cur)) {
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
  cur.bt += OT;
}

// Output head (<= 16 logical outputs, duplicated in both lane halves by the packer): logical output j = acc[j].
template <class G, class PL, int NT, int P, int K>
DEVI void head(Pipe<G, PL>& pipe, Cursor& cur, Carry<NT>& carry, f32x16 (&acc)[1][NT], Chunk<P> (&in)[NT][K]) {
  const int hb = bias_base(pipe.lane16);
  {
    const f32x16 bv = load_bias(cur.bt, hb);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[0][nt] = bv;
  }
  int j = 0;
  if constexpr (is_single(P) && NERFDS_PIPE_EPI && NERFDS_ASM_EPILOGUE && !(NERFDS_ABLATE & 4) && (K - 2 * TILE_PAIR - 2 > 0)) {
    // the hidden layer before a head leaves its last tile group pending: its chunks are the tail of `in` (a single-group
    // layer does not carry); every piece before the slot that consumes the first carried chunk, K - 2 * TILE_PAIR
    constexpr int NPIECE = TILE_PAIR * NT * 4, J0 = 2, PPS = cdiv(NPIECE, K - 2 * TILE_PAIR - J0);
    unsigned pk[TILE_PAIR][NT][8];
    auto slot = [&](int jj, int) {
      if (jj < J0 || !carry.live) return;
      pending_pieces<P, NT>((jj - J0) * PPS, (jj - J0 + 1) * PPS, carry.acc, pk, in, K - 2 * TILE_PAIR, acc[0][NT - 1]);
    };
    accum<G, PL, NT, 1>(acc, pipe, cur, in, j, slot);
    carry.live = false;
  } else {
    auto no_slot = [](int, int) {};
    accum<G, PL, NT, 1>(acc, pipe, cur, in, j, no_slot);
  }
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
// NERFDS_X3_HW_SIN=0 keeps sin_cw (A/B).
#ifndef NERFDS_X3_HW_SIN
#if defined(NERFDS_TRAIN_FWD) || defined(NERFDS_TRAIN_BWD)
#define NERFDS_X3_HW_SIN 0        // the trainer's forward stays on sin_cw (its gradient tests compare with the fp64 oracle at 1e-3-grade bounds)
#else
#define NERFDS_X3_HW_SIN 1
#endif
#endif
DEVI float sin_hw_exact(float a) {
  const float p = a * 0.15915494f;                                        // 1 / (2 pi) = 0.15915494 + 6.4206382e-09
  float e = fmaf(a, 0.15915494f, -p);
  e = fmaf(a, 6.4206382e-09f, e);
  return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(p) + e);
}
template <int P> DEVI float sin_enc(float a) {
  if (NERFDS_ABLATE & 8) return a;
  if constexpr (is_single(P)) {
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(a * 0.159154943f));
  } else if constexpr (NERFDS_X3_HW_SIN && P == P_BF16X3) {
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
DEVI void set_row(Cursor&, float*, uint16_t*, uint16_t*, int) {}
DEVI void set_row(TrainCursor& c, float* p, uint16_t* p16, uint16_t* bits, int half) { c.row = p; c.row16 = p16; c.bits = bits; c.half = half; }

// Which samples the N-tiles of this lane evaluate: depth z and the slot of the ray's LDS block (SoA by sample) that receives /
// holds the sample's per-sample state.  Tail lanes repeat the last sample (same values to the same slot).
template <int NT> struct Samples {
  float z[NT];
  int slot[NT];
};

#ifdef NERFDS_EXP_PAGES
// TIMING EXPERIMENT ONLY (wrong results): every 16-bit store of a wave lands in ONE 128-KiB region per wave instead of in ~70 arrays that lie
// gigabytes apart - what a [32-sample block][layer] layout of the activations would do to the address translation of the stores.
#define NERFDS_TRAIN_ROW(base, base16, bits, W) do { if constexpr (TO::ON) { uint16_t* c_ = to.trunk_h16[0] + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 65536; \
    set_row(cur, (base) + row * (size_t)(W) + 4 * h, c_ + (((size_t)(base16) >> 21) & 7) * 8192 + (lane & 31) * (W) + ROW16_H * h, \
            c_ + 61440 + ((lane & 31) * 2 + h) * 8, to.half_out); } } while (0)
#else
#define NERFDS_TRAIN_ROW(base, base16, bits, W) do { if constexpr (TO::ON) set_row(cur, (base) + row * (size_t)(W) + 4 * h, \
    (base16) + row * (size_t)(W) + ROW16_H * h, (bits) + (row * 2 + h) * (size_t)((W) / 32), to.half_out); } while (0)
#endif
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
  Carry<NT> carry;

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
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, in0);
    NERFDS_TRAIN_ROW(to.mask_h[1], to.mask_h16[1], to.mask_bits[1], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.mask_h[2], to.mask_h16[2], to.mask_bits[2], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b);
    NERFDS_TRAIN_ROW(to.mask_h[3], to.mask_h16[3], to.mask_bits[3], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.mask_h[4], to.mask_h16[4], to.mask_bits[4], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b, in0);      // skip: [x, inputs] (modules.py:66-67)
    NERFDS_TRAIN_ROW(to.mask_h[5], to.mask_h16[5], to.mask_bits[5], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.mask_h[6], to.mask_h16[6], to.mask_bits[6], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b);
    NERFDS_TRAIN_ROW(to.mask_h[7], to.mask_h16[7], to.mask_bits[7], G::MASK_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, carry, hacc, b);
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
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, in0);
    NERFDS_TRAIN_ROW(to.warp_h[1], to.warp_h16[1], to.warp_bits[1], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.warp_h[2], to.warp_h16[2], to.warp_bits[2], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b);
    NERFDS_TRAIN_ROW(to.warp_h[3], to.warp_h16[3], to.warp_bits[3], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.warp_h[4], to.warp_h16[4], to.warp_bits[4], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b, in0);
    NERFDS_TRAIN_ROW(to.warp_h[5], to.warp_h16[5], to.warp_bits[5], G::WARP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, carry, hacc, b);      // logical outputs: w = 0..2, v = 3..5
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
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, in0);
    NERFDS_TRAIN_ROW(to.hyper_h[1], to.hyper_h16[1], to.hyper_bits[1], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.hyper_h[2], to.hyper_h16[2], to.hyper_bits[2], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b);
    NERFDS_TRAIN_ROW(to.hyper_h[3], to.hyper_h16[3], to.hyper_bits[3], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.hyper_h[4], to.hyper_h16[4], to.hyper_bits[4], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, a, b, in0);
    NERFDS_TRAIN_ROW(to.hyper_h[5], to.hyper_h16[5], to.hyper_bits[5], G::HYP_W);
    dense<G, PL, NT, W32, true>(pipe, cur, carry, b, a);
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, carry, hacc, b);
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
template <class G, class PL, int NT, class LT, class TO = NoTrain>
DEVI void eval_nerf(const KArgs& ka, Pipe<G, PL>& pipe, int level, int lane, const Samples<NT>& sm, LT& L,
                    const TO& to = TO(), size_t row = 0) {
  using D = Dims<G>;
  const int h = lane >> 5;
  std::conditional_t<TO::ON, TrainCursor, Cursor> cur;
  cur.seg = SEG_NERF;
  cur.pos = 0;
  cur.bt = D::SHARED_BIAS_TILES + level * D::NERF_BIAS_TILES;
  Carry<NT> carry;
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
    constexpr int P = PL::TRUNK, PR = PL::RGB;
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
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, a, in0);
    NERFDS_TRAIN_ROW(to.trunk_h[1], to.trunk_h16[1], to.trunk_bits[1], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.trunk_h[2], to.trunk_h16[2], to.trunk_bits[2], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, a, b);
    NERFDS_TRAIN_ROW(to.trunk_h[3], to.trunk_h16[3], to.trunk_bits[3], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.trunk_h[4], to.trunk_h16[4], to.trunk_bits[4], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, a, b, in0);
    NERFDS_TRAIN_ROW(to.trunk_h[5], to.trunk_h16[5], to.trunk_bits[5], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, b, a);
    NERFDS_TRAIN_ROW(to.trunk_h[6], to.trunk_h16[6], to.trunk_bits[6], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, a, b);
    NERFDS_TRAIN_ROW(to.trunk_h[7], to.trunk_h16[7], to.trunk_bits[7], G::TRUNK_W);
    dense<G, PL, NT, TW32, true>(pipe, cur, carry, b, a);          // b = trunk_output
    // (the activation-free bottleneck Dense, modules.py:255, is folded into rgb hidden_0 by the packer)
    f32x16 hacc[1][NT];
    head<G, PL, NT>(pipe, cur, carry, hacc, b);                    // alpha_mlp on trunk_output (modules.py:273-274)
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
    dense<G, PL, NT, G::RGB_W / 32, true>(pipe, cur, carry, c, b, cond);       // K order [trunk_output | cond]
    head<G, PL, NT>(pipe, cur, carry, hacc, c);
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
  pipe.finish_segment(SEG_NERF);
}
#undef NERFDS_TRAIN_ROW
#undef NERFDS_TRAIN_HEAD

// ------------------------------------------------------------------------------------------------
// Compositing of one level (model_utils.py:95-159, 272-317; models.py:1346-1415) -> ray record.
// ------------------------------------------------------------------------------------------------
template <class G, class LT>
DEVI void composite(const KArgs& ka, const RayConst& rc, int ray, int lane, int S, bool at_infinity, LT& L,
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
    float alpha = valid ? (1.0f - expf(-sigma * dist)) : 0.0f;             // model_utils.py:129
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
    float xo[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xo[c] = __fadd_rn(rc.o[c], __fmul_rn(z, rc.d[c]));
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
template <class LT> DEVI void resample(const KArgs& ka, int ray, int lane, int nc, int nf, LT& L) {
  constexpr int MAX_SAMPLES = LT::MAX_S;
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
  constexpr int NJ = MAX_SAMPLES / 64, NSTATE = 1 + (SV_COUNT - SV_WP);
  int rk[NJ];
  float st[NJ][NSTATE];
  {
    // rank of union element i = #{q : z_q < z_i or (z_q == z_i and q < i)}.  One pass over the union for ALL of the lane's elements,
    // four values per (broadcast) LDS read and four reads in flight: as one dependent 4-byte read per comparison this loop was the
    // largest single piece of the per-ray phases (~20 k of their ~34 k cycles per ray group).
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
}

// ------------------------------------------------------------------------------------------------
// Kernel: persistent waves, one ray per wave per iteration.
// ------------------------------------------------------------------------------------------------
template <class G, class PL, bool WIDE>
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

#ifdef NERFDS_SETPRIO_HALF      // static priority for the second-dispatched half of an 8-wave workgroup (cdna guide T5, static form)
  if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(NERFDS_SETPRIO_HALF);
#endif
  if (NERFDS_DBG & 1) {
    constexpr int total = BIAS_OFF + bias_bytes<G>() + RAYS_PER_WG * (int)sizeof(WaveLds);
    for (int i = threadIdx.x; i < total / 4; i += 64 * WAVES) reinterpret_cast<float*>(g_smem)[i] = 0.f;
    __syncthreads();
  }
  using PP = Pipe<G, PL>;
  Pipe<G, PL> pipe;
  const rsrc_t rs_nerf[2] = {make_rsrc(ka.wstream[1], PP::NERF_PAD * 1024), make_rsrc(ka.wstream[2], PP::NERF_PAD * 1024)};
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
        L.rayc[RC_WEMB + lane] = G::HAS_WARP ? ka.warp_embed[(size_t)wid * 8 + lane] : 0.f;
        L.rayc[RC_MEMB + lane] = G::HAS_MASK ? ka.mask_embed[(size_t)wid * 8 + lane] : 0.f;
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
    for (int sb = 0; sb < nc; sb += BATCH) {
      const Samples<NT> sm = samples_at(sb + 32 * NT * q, nc);
      if constexpr (PP::HAS_SHARED) { pipe.cur = rs_shared; pipe.next = rs_nerf[0]; }
      eval_shared<G, PL, NT>(ka, rc, pipe, lane, sm, L);
      pipe.cur = rs_nerf[0];
      if constexpr (PP::HAS_SHARED) pipe.next = rs_shared;      // another coarse batch, the fine level's new samples, or the next ray group
      else pipe.next = (sb + BATCH < nc) ? rs_nerf[0] : (nf > 0 ? rs_nerf[1] : rs_nerf[0]);
      eval_nerf<G, PL, NT>(ka, pipe, 0, lane, sm, L);
    }
    ray_sync();
    if (q == 0 && !(NERFDS_ABLATE & 16)) {
      float* rec = live ? ((nf > 0) ? ka.ray_coarse : ka.ray_fine) : nullptr;
      float* smp = live ? ((nf > 0) ? ka.smp_coarse : ka.smp_fine) : nullptr;
      composite<G>(ka, rc, ray, lane, nc, ka.sample_at_infinity != 0, L,
                   rec ? rec + (size_t)ray * RAY_REC : nullptr,
                   smp ? smp + (size_t)ray * nc * SAMPLE_REC : nullptr);
      WAVE_SYNC();
      if (nf > 0) resample(ka, ray, lane, nc, nf, L);
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
        eval_shared<G, PL, NT>(ka, rc, pipe, lane, sm, L);
      }
      ray_sync();              // a NerfMLP batch reads slots that other waves of the ray have written
      for (int sb = 0; sb < n; sb += BATCH) {
        const Samples<NT> sm = samples_at(sb + 32 * NT * q, n);
        pipe.cur = rs_nerf[1];
        pipe.next = (sb + BATCH < n) ? rs_nerf[1] : (PP::HAS_SHARED ? rs_shared : rs_nerf[0]);
        eval_nerf<G, PL, NT>(ka, pipe, 1, lane, sm, L);
      }
      ray_sync();
      if (q == 0 && !(NERFDS_ABLATE & 16))
        composite<G>(ka, rc, ray, lane, n, ka.sample_at_infinity != 0, L,
                     (live && ka.ray_fine) ? ka.ray_fine + (size_t)ray * RAY_REC : nullptr,
                     (live && ka.smp_fine) ? ka.smp_fine + (size_t)ray * n * SAMPLE_REC : nullptr);
      ray_sync();
    }
  }
}

#ifdef NERFDS_TRAIN_FWD
// ------------------------------------------------------------------------------------------------
// Training forward of ONE level (training.py:198-511 -> models.py:867-1417 on the level's samples): the same field evaluation as
// render_rays_kernel - same weight pipe, same register-chained layers - on depths the trainer has already drawn (to.z), with every
// hidden layer's fp32 output and every head's raw output written to the trainer's workspace for the backward pass.  No
// compositing here: the loss kernel composites from sigma / rgb (train_kernels.hip).  The host passes the level's NerfMLP
// stream and biases in slot 1, so the kernel always evaluates "level 0".
// ------------------------------------------------------------------------------------------------
template <class G, class PL, bool WIDE, int TAG>
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
  pipe.cur = pipe.next = rs_shared;
  pipe.lane16 = lane * 16;
  pipe.wave1k = wave * 1024;
  pipe.prologue(PP::HAS_SHARED ? SEG_SHARED : SEG_NERF);
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
      if constexpr (PP::HAS_SHARED) { pipe.cur = rs_shared; pipe.next = rs_nerf; }
      eval_shared<G, PL, NT, WaveLds, TrainOut>(ka, rc, pipe, lane, sm, L, to, row);
      pipe.cur = rs_nerf;
      pipe.next = rs_shared;
      eval_nerf<G, PL, NT, WaveLds, TrainOut>(ka, pipe, 0, lane, sm, L, to, row);
    }
    ray_sync();
  }
}
#endif  // NERFDS_TRAIN_FWD

#ifdef NERFDS_TRAIN_BWD
// ------------------------------------------------------------------------------------------------
// Training backward of ONE network (training.py:494 differentiates the whole model.apply): the data-gradient chain of the reversed
// MLP on the machinery of the forward - transposed weight fragments streamed through the LDS ring, every layer's gradient kept in
// registers as the next (earlier) layer's B operand (the accumulator of a transposed tile IS g^T[feature][sample]), split-bf16
// operands with fp32 accumulation.  Per 32-sample tile a wave reads the head gradients and the ReLU bits of every layer, and
// writes g_l = d loss / d (pre-activation of layer l) for every hidden layer (the dY of the weight-gradient kernels) and the
// gradient of the raw input.  dX of a hidden layer never goes to HBM as an operand of the next data-gradient kernel, nor do the
// fp32 activations come back as masks: 1 array pass per layer where the layer-by-layer backward made 3.
// ------------------------------------------------------------------------------------------------
template <int W> DEVI void load_bits(unsigned (&m)[W / 64 > 0 ? W / 64 : 1], const uint16_t* bits, long long r, int h) {
  const unsigned* p = reinterpret_cast<const unsigned*>(bits + ((size_t)r * 2 + h) * (W / 32));
#pragma unroll
  for (int j = 0; j < W / 64; ++j) m[j] = p[j];
}
template <class BG, class PL, int OT, class OUT, class... Ins>
DEVI void bwd_hidden(Pipe<BG, PL>& pipe, BwdCursor& cur, Carry<1>& carry, const unsigned (&m)[OT / 2], float* g_base, size_t g_off, OUT& out, Ins&... ins) {
#pragma unroll
  for (int j = 0; j < OT / 2; ++j) cur.mask[j] = m[j];
  cur.row = g_base + g_off;                                        // fp32 [M][width] ...
  cur.row16 = reinterpret_cast<uint16_t*>(g_base) + g_off + (ROW16_H / 4 - 1) * cur.in_h4;   // ... or bf16 [M][width] in the same buffer (cur.half)
  dense<BG, PL, 1, OT, false>(pipe, cur, carry, out, ins...);
}
template <class BG, class PL, int P, int K>
DEVI void bwd_input(Pipe<BG, PL>& pipe, BwdCursor& cur, Carry<1>& carry, float* in_row, int acc, Chunk<P> (&in)[1][K]) {
  Chunk<P> none[1][4];
  BwdInCursor ic;
  static_cast<BwdCursor&>(ic) = cur;
  ic.in_row = in_row;
  ic.in_acc = acc;
  dense<BG, PL, 1, 2, false>(pipe, ic, carry, none, in);
  cur.pos = ic.pos;
  cur.bt = ic.bt;
}

template <class BG, class PL>
DEVI void bwd_chain(const TrainBwd& tb, Pipe<BG, PL>& pipe, int lane, long long r, int live) {
  constexpr int W = BG::W, D = BG::DEPTH, W16 = W / 16, W32 = W / 32, P = P_BF16X3, MW = W / 64;
  static_assert(BG::SKIP == 4 && (D == 8 || D == 6) && W % 64 == 0, "chains are written out for depth 8 / 6, skip 4");
  const int h = lane >> 5;
  BwdCursor cur;
  cur.seg = SEG_NERF; cur.pos = 0; cur.bt = 0;
  cur.row = nullptr; cur.row16 = nullptr; cur.bits = nullptr; cur.half = tb.g_half;
  cur.sink = tb.sink; cur.ld_in = tb.ld_in; cur.in_h4 = 4 * h; cur.in_row = nullptr; cur.in_acc = 0; cur.live = live;
  Carry<1> carry;
  // every load of the tile up front (one wait): ReLU bits of all layers, head gradients
  unsigned mk[D][MW];
#pragma unroll
  for (int l = 0; l < D; ++l) load_bits<W>(mk[l], tb.bits[l], r, h);
  const size_t g_off = (size_t)r * W + 4 * h;
  float* const in_row = tb.d_in + (size_t)r * tb.ld_in + 4 * h;
  Chunk<P> a[1][W16], b[1][W16];
  if constexpr (BG::IS_NERF) {
    constexpr int RW = BG::RGB_W, R16 = RW / 16;
    unsigned mr[RW / 64];
    load_bits<RW>(mr, tb.bits[8], r, h);
    Chunk<P> drgb[1][1], dalpha[1][1], c[1][R16];
    build_chunks<P, 1>(drgb[0], h, [&](int f) { return f < 3 ? val_feat(tb.d_head[(size_t)r * tb.ld_head + f]) : zero_feat(); });
    build_chunks<P, 1>(dalpha[0], h, [&](int f) { return f < 4 ? val_feat(tb.d_head2[(size_t)r * 4 + f]) : zero_feat(); });
    bwd_hidden<BG, PL, RW / 32>(pipe, cur, carry, mr, tb.g[8], (size_t)r * RW + 4 * h, c, drgb);     // g_rgb = mask(W_rgb d rgb_logit)
    bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[7], tb.g[7], g_off, a, c, dalpha);                          // g_7 = mask(F^T g_rgb + W_alpha d alpha)
  } else {
    Chunk<P> dh[1][1];
    build_chunks<P, 1>(dh[0], h, [&](int f) { return f < BG::NHEAD ? val_feat(tb.d_head[(size_t)r * tb.ld_head + f]) : zero_feat(); });
    bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[D - 1], tb.g[D - 1], g_off, a, dh);                          // g_{D-1} = mask(W_head d head)
  }
  if constexpr (D == 8) {
    bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[6], tb.g[6], g_off, b, a);
    bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[5], tb.g[5], g_off, a, b);
    bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[4], tb.g[4], g_off, b, a);
  } else {
    bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[4], tb.g[4], g_off, b, a);
  }
  // the skip layer [h_3 | raw input] (modules.py:66-67): its hidden rows give g_3, its raw-input rows the first part of d input
  bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[3], tb.g[3], g_off, a, b);
  bwd_input<BG, PL>(pipe, cur, carry, in_row, 0, b);
  bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[2], tb.g[2], g_off, b, a);
  bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[1], tb.g[1], g_off, a, b);
  bwd_hidden<BG, PL, W32>(pipe, cur, carry, mk[0], tb.g[0], g_off, b, a);
  bwd_input<BG, PL>(pipe, cur, carry, in_row, 1, b);
  pipe.finish_segment(SEG_NERF);
}

template <class BG, class PL, int TAG>
__global__ __launch_bounds__(64 * wg_waves<PL>(), wg_waves<PL>() / 4) void train_backward_kernel(const TrainBwd tb) {
  using PP = Pipe<BG, PL>;
  static_assert(wg_waves<PL>() == 4 && !PP::HAS_SHARED, "one 512-register wave per SIMD, one stream");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Pipe<BG, PL> pipe;
  pipe.cur = pipe.next = make_rsrc(tb.wstream, PP::NERF_PAD * 1024);
  pipe.lane16 = lane * 16;
  pipe.wave1k = wave * 1024;
  pipe.prologue(SEG_NERF);
  const long long groups = (tb.M + 127) / 128;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long rr = grp * 128 + wave * 32 + (lane & 31);
    bwd_chain<BG, PL>(tb, pipe, lane, rr < tb.M ? rr : tb.M - 1, rr < tb.M ? 1 : 0);     // tail lanes redo the last row: same g values to the same places
  }
}
#endif  // NERFDS_TRAIN_BWD

}  // namespace nerfds

// One translation unit per (graph, precision plan):
//   -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_PREC=P_BF16 -DNERFDS_NAME=nerfds_bf16        uniform plan
//   -DNERFDS_GRAPH=GraphNerfDS -DNERFDS_MIXED -DNERFDS_NAME=nerfds_mixed            graphs.h NERFDS_MIX_* plan
#ifndef NERFDS_GRAPH
#error "compile with -DNERFDS_GRAPH=<GraphNerfDS|GraphStatic|GraphHyperNeRF> -DNERFDS_PREC=<P_BF16|P_BF16X3|P_F32|P_F16> (or -DNERFDS_MIXED) -DNERFDS_NAME=<suffix>"
#endif
#define NERFDS_CAT2(a, b) a##b
#define NERFDS_CAT(a, b) NERFDS_CAT2(a, b)
namespace nerfds {
#if defined(NERFDS_TRAIN_BWD)
using KernelPlan = PlanT<P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3>;      // the data-gradient chains: split bf16 throughout
#elif defined(NERFDS_TRAIN_FWD)
// the trainer's arithmetic: 16-bit split operands everywhere, fp32 products in the warp field
using KernelPlan = PlanT<TRAIN_PLAN.mask, TRAIN_PLAN.warp, TRAIN_PLAN.hyp, TRAIN_PLAN.trunk, TRAIN_PLAN.rgb>;
#elif defined(NERFDS_MIXED)
using KernelPlan = PlanT<NERFDS_MIX_MASK, NERFDS_MIX_WARP, NERFDS_MIX_HYP, NERFDS_MIX_TRUNK, NERFDS_MIX_RGB>;
#else
using KernelPlan = PlanT<NERFDS_PREC, NERFDS_PREC, NERFDS_PREC, NERFDS_PREC, NERFDS_PREC>;
#endif
}  // namespace nerfds

#ifdef NERFDS_TRAIN_BWD
template <class BG> static void launch_bwd(const nerfds::TrainBwd& tb, int num_cus, void* stream) {
  using namespace nerfds;
  using PLX = PlanT<P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3>;
  static bool attr_set = false;
  auto kern = train_backward_kernel<BG, PLX, NERFDS_TRAIN_TAG>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, RING_BYTES);
    attr_set = true;
  }
  const long long groups = (tb.M + 127) / 128;
  // the 64 / 128-wide chains need <= 256 registers: two workgroups per CU (two waves per SIMD cover each other's waits); the trunk's
  // takes the whole register file
  const long long want = (long long)num_cus * (BG::W <= 128 ? 2 : 1);
  const int grid = (int)(groups < want ? groups : want);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wg_waves<PLX>()), RING_BYTES, static_cast<hipStream_t>(stream), tb);
}
// net: 0 NerfMLP (trunk + rgb branch + alpha head), 1 hyper sheet, 2 warp field, 3 mask net
extern "C" void NERFDS_CAT(nerfds_launch_, NERFDS_NAME)(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream) {
  using G = nerfds::NERFDS_GRAPH;
  if (net == 0) launch_bwd<nerfds::BwdNerf<G>>(tb, num_cus, stream);
  else if (net == 1) launch_bwd<nerfds::BwdHyper<G>>(tb, num_cus, stream);
  else if (net == 2) launch_bwd<nerfds::BwdWarp<G>>(tb, num_cus, stream);
  else launch_bwd<nerfds::BwdMask<G>>(tb, num_cus, stream);
}
#elif defined(NERFDS_TRAIN_FWD)
template <bool WIDE> static void launch_train(const nerfds::KArgs& ka, const nerfds::TrainOut& to, int num_cus, void* stream) {
  using namespace nerfds;
  using SH = Shape<KernelPlan, WIDE>;
  constexpr int lds = BIAS_OFF + bias_bytes<NERFDS_GRAPH>() + SH::RAYS * (int)sizeof(WaveLdsT<SH::MAXS>);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = train_forward_kernel<NERFDS_GRAPH, KernelPlan, WIDE, NERFDS_TRAIN_TAG>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  const long long groups = ((long long)ka.num_rays + SH::RAYS - 1) / SH::RAYS;
  const int grid = (int)(groups < num_cus ? groups : num_cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wg_waves<KernelPlan>()), lds, static_cast<hipStream_t>(stream), ka, to);
}
// ka.nc = samples of the level (ka.nf unused); ka.wstream[1] / ka.bias[1] = the level's NerfMLP
extern "C" void NERFDS_CAT(nerfds_launch_, NERFDS_NAME)(const nerfds::KArgs& ka, const nerfds::TrainOut& to, int num_cus, void* stream) {
  if (ka.nc > nerfds::Shape<nerfds::KernelPlan, false>::MAXS) launch_train<true>(ka, to, num_cus, stream);
  else launch_train<false>(ka, to, num_cus, stream);
}
#else
template <bool WIDE> static void launch_shape(const nerfds::KArgs& ka, int num_cus, void* stream) {
  using namespace nerfds;
  using SH = Shape<KernelPlan, WIDE>;
  constexpr int lds = BIAS_OFF + bias_bytes<NERFDS_GRAPH>() + SH::RAYS * (int)sizeof(WaveLdsT<SH::MAXS>);
  static_assert(lds <= 160 * 1024, "LDS budget");
  static bool attr_set = false;
  auto kern = render_rays_kernel<NERFDS_GRAPH, KernelPlan, WIDE>;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  // one persistent workgroup per CU (LDS-bound), SH::RAYS rays per workgroup iteration
  const long long groups = ((long long)ka.num_rays + SH::RAYS - 1) / SH::RAYS;
  const int grid = (int)(groups < num_cus ? groups : num_cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * wg_waves<KernelPlan>()), lds, static_cast<hipStream_t>(stream), ka);
}

extern "C" void NERFDS_CAT(nerfds_launch_, NERFDS_NAME)(const nerfds::KArgs& ka, int num_cus, void* stream) {
  if (ka.nc + ka.nf > nerfds::Shape<nerfds::KernelPlan, false>::MAXS) launch_shape<true>(ka, num_cus, stream);
  else launch_shape<false>(ka, num_cus, stream);
}
#endif
