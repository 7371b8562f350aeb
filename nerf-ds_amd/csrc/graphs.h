// Compile-time descriptions of the render graphs that have HIP kernels, shared by the host-side
// weight-stream packer (nerfds_host.cpp) and the device code (field.h and its three kernel files) so that both walk
// the layers in the same order.
//
// Dimensions follow SURVEY.md section 8 "Configuration resolved" (configs/nerf_ds.gin over
// configs/defaults.gin of the reference) and BASELINE.json configs[0] for the static graph.
#pragma once

namespace nerfds {

// Arithmetic of one activation tensor (and of the weights it is multiplied with):
//   bf16 / f16 : one v_mfma_f32_32x32x16_{bf16,f16} per product (8 / 11 significand bits)
//   bf16x3     : split bf16 hi + lo operands, three MFMAs per product (16 bits, fp32-grade results)
//   f32        : v_mfma_f32_32x32x2_f32, exact fp32 fma chains
//   bf16x6     : split bf16 hi + mid + lo operands, six MFMAs per product (fp32-grade products at 6 x 32 cycles per fragment where the
//                fp32 MFMA takes 8 x 64): the forward of the warp field in the training step
enum Prec : int { P_BF16 = 0, P_BF16X3 = 1, P_F32 = 2, P_F16 = 3, P_BF16X6 = 4 };
constexpr bool is_single(int prec) { return prec == P_BF16 || prec == P_F16; }

// One MFMA weight fragment = a [32 out rows] x [16 k-slots] block of a layer, laid out exactly as the
// 64 lanes of a wave consume it (16 bytes per lane per part, lane-linear, so one coalesced 1 KiB load).
// A part is one 1-KiB *unit* of the weight stream:
//   bf16 / f16 : 1 unit  (8 halves / lane)
//   bf16x3     : 2 units (hi bf16x8, then lo bf16x8)
//   f32        : 2 units (k-slots 0-3, then k-slots 4-7 as float4)
//   bf16x6     : 3 units (hi, mid, lo bf16x8)
constexpr int frag_parts(int prec) { return is_single(prec) ? 1 : (prec == P_BF16X6 ? 3 : 2); }
constexpr int frag_bytes(int prec) { return 1024 * frag_parts(prec); }

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
// The kernel stages the stream through LDS in 16 KiB stages; each of the two streams is zero-padded to a
// whole number of stages so that a stage never straddles the shared -> NerfMLP seam.
constexpr int STAGE_BYTES = 16384;
// Output tiles of a hidden layer are computed TILE_PAIR at a time: within a pair the stream holds, chunk by chunk, one
// fragment of each tile, so that consecutive MFMAs of a wave go to different accumulators (field.h accum).
#ifndef NERFDS_TILE_PAIR
#define NERFDS_TILE_PAIR 2
#endif
constexpr int TILE_PAIR = NERFDS_TILE_PAIR;
// EXPERIMENT (off: NERFDS_NT2_TILE_PAIR = 2): the two-N-tile render kernels (nerf_ds / HyperNeRF graph, bf16 / f16: Makefile NT2FLAGS) taking their tiles
// ONE at a time - with two N-tiles one tile already gives a wave two independent accumulators, and the 32 registers the second tile takes could hold the
// previous group's results while their conversion is issued inside the next chain (a source-level pipelined epilogue, since removed: profiles/r3_ab/README.md).  Needs the stream in that order
// (the host packs it per kernel: StreamWriter::tile_pair) and -DNERFDS_TILE_PAIR=1 on those kernels (tools/variant_tp1.sh builds such a library).
// Measured, parity green: bf16 13.65 (13.69 with the pipelined epilogue) against 13.54 ms per 65 536 rays, f16 14.01 (14.02) against 13.92
// (profiles/r3_ab/ab_tp1.txt) - the exposed epilogue is not what these kernels wait for.
#ifndef NERFDS_NT2_TILE_PAIR
#define NERFDS_NT2_TILE_PAIR 2
#endif
constexpr int stream_tile_pair(bool two_n_tile_kernel) { return two_n_tile_kernel ? NERFDS_NT2_TILE_PAIR : 2; }
constexpr int STAGE_UNITS = STAGE_BYTES / 1024;
constexpr int pad_units(int n) { return cdiv(n, STAGE_UNITS) * STAGE_UNITS; }
constexpr int chunks(int feats) { return cdiv(feats, 16); }   // k16 chunks needed for `feats` linear features

// Precision plan: which arithmetic each network of the graph runs in.  `trunk` covers the NerfMLP trunk, its raw input
// and trunk_output (so also the alpha head and the trunk_output segment of rgb hidden_0); `rgb` covers the rgb
// condition inputs, rgb hidden_0's output and the rgb head.  The uniform plans are the round-1 kernels; the mixed
// plan (NERFDS_PREC_MIXED) keeps the networks whose error is amplified downstream (the warp field moves the points
// that the 2^7-frequency encoding of the template reads) in split bf16 and runs the bulk of the FLOPs in one MFMA.
// trunk_c / rgb_c (>= 0): the COARSE level's NerfMLP in its own arithmetic.  The fine level sees of the coarse NerfMLP only the compositing weights its
// depths are drawn from (model_utils.py:193-269, behind a stop_gradient), and render_fn returns the fine level alone (evaluation.py:121-124):
// measured over 131 072 rays of the bench frame (tools/precision_study_gpu.py --levels, profiles/r5_precision_levels.md), a coarse NerfMLP in ONE
// f16 MFMA per product leaves the fine level's composited RGB where the all-split-bf16 run has it (3.8e-5 against 3.7e-5) while the coarse level's
// own RGB is f16-grade (4.7e-4); one bf16 MFMA is not enough (fine level 2.7e-4).
struct Plan { int mask, warp, hyp, trunk, rgb; int trunk_c = -1, rgb_c = -1; };
constexpr Plan uniform_plan(int p) { return Plan{p, p, p, p, p}; }
// the plan the NerfMLP of `level` (0 coarse, 1 fine) runs in
constexpr Plan level_plan(Plan p, int level) { return (level == 0 && p.trunk_c >= 0) ? Plan{p.mask, p.warp, p.hyp, p.trunk_c, p.rgb_c} : Plan{p.mask, p.warp, p.hyp, p.trunk, p.rgb}; }
#ifndef NERFDS_MIX_MASK
#define NERFDS_MIX_MASK P_F16
#endif
#ifndef NERFDS_MIX_WARP
#define NERFDS_MIX_WARP P_BF16X3
#endif
#ifndef NERFDS_MIX_HYP
#define NERFDS_MIX_HYP P_F16
#endif
#ifndef NERFDS_MIX_TRUNK
#define NERFDS_MIX_TRUNK P_F16
#endif
#ifndef NERFDS_MIX_RGB
#define NERFDS_MIX_RGB P_F16
#endif
constexpr int NUM_PLANS = 7;     // == number of NERFDS_PREC_* values of include/nerfds.h
constexpr Plan plan_of(int prec_index) {
  return prec_index == 4 ? Plan{NERFDS_MIX_MASK, NERFDS_MIX_WARP, NERFDS_MIX_HYP, NERFDS_MIX_TRUNK, NERFDS_MIX_RGB}
         : prec_index == 5 ? Plan{P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_F16, P_F16}      // NERFDS_PREC_BF16X3_FINE
         : prec_index == 6 ? uniform_plan(P_BF16X3)      // NERFDS_PREC_F16X3: the split plan; WHICH 16-bit format its hi / lo parts are is a fact of the kernel build
                           : uniform_plan(prec_index);      //   (field.h NERFDS_X3_F16) and of the packer (pack.h StreamWriter::x3_f16)
}
// The plan of the fused training forward (train_fwd_kernel.hip): split bf16 operands as the trainer's layer
// kernels (train_gemm.hip), exact fp32 products in the warp field (its 16-bit rounding is amplified to ~1 % on the warp-field
// gradients by the 2^7-frequency encoding of the warped point).  Two units per fragment (three in a P_BF16X6 warp field).
// (NERFDS_TRAIN_WARP_X6: the warp field in P_BF16X6 instead - 6 x 32 MFMA cycles per fragment where fp32 takes 8 x 64.  Measured 21.2 ->
// 20.8 ms per step, but two warp-side leaves of the multi-tile gradient test move from just under to just over their bounds
// (6.4e-3 against 6e-3, 2.0e-2 against 1.5e-2): the exact fp32 products stay the default.  Round 5, re-measured at 11.4 ms per step: 11.3 ms with it, the
// multi-tile rgb test's warp leaves at 2.1e-2 (fp32: 7.9e-3) - and 1.9e-2 with ALL NINE products of the three-way split, so it is not the dropped
// terms but the bf16 MFMA's internal accumulation that is coarser than the fp32 MFMA's.)
#ifdef NERFDS_TRAIN_WARP_X6
constexpr Plan TRAIN_PLAN = Plan{P_BF16X3, P_BF16X6, P_BF16X3, P_BF16X3, P_BF16X3};
#else
constexpr Plan TRAIN_PLAN = Plan{P_BF16X3, P_F32, P_BF16X3, P_BF16X3, P_BF16X3};
#endif
// Stream position after a fragment segment of `kc` k16-chunks in precision p, starting at unit `pos`.  There is no
// alignment: a two-unit fragment may start on an odd unit and even straddle two LDS stages (the kernel fetches
// units, not fragments).
constexpr int walk_seg(int pos, int kc, int p) { return pos + kc * frag_parts(p); }
// A plain hidden stack: depth layers of `width`, input `in_chunks` k16-chunks, skip re-concatenation of the raw input
// before layer `skip` (modules.py:66-67), plus an optional 32-row head tile; everything in precision p.
constexpr int walk_mlp(int pos, int depth, int width, int in_chunks, int skip, bool head, int p) {
  for (int l = 0; l < depth; ++l)
    for (int t = 0; t < width / 32; ++t) {
      pos = walk_seg(pos, l == 0 ? in_chunks : width / 16, p);
      if (l == skip && l > 0) pos = walk_seg(pos, in_chunks, p);
    }
  if (head) pos = walk_seg(pos, width / 16, p);
  return pos;
}

constexpr int mlp_bias_tiles(int depth, int width, bool head) { return depth * (width / 32) + (head ? 1 : 0); }

// configs/nerf_ds.gin
struct GraphNerfDS {
  static constexpr int ID = 0;
  static constexpr bool HAS_MASK = true, HAS_WARP = true, HAS_HYPER = true, PREDICT_NORM = true, X_IN_RGB = true;
  static constexpr bool IDENT = false, WARP_IDENT = false;      // use_posenc_identity (defaults.gin:111,138)
  static constexpr int GLO = 8;
  static constexpr int MASK_BANDS = 6, MASK_DEPTH = 8, MASK_W = 128, MASK_SKIP = 4;
  static constexpr int WARP_BANDS = 4, WARP_DEPTH = 6, WARP_W = 128, WARP_SKIP = 4;
  static constexpr int HYP_BANDS = 6, HYP_DEPTH = 6, HYP_W = 64, HYP_SKIP = 4, HYP_DIMS = 2;
  static constexpr int SP_BANDS = 8, HP_BANDS = 1, VD_BANDS = 4, NM_BANDS = 4;
  static constexpr int TRUNK_DEPTH = 8, TRUNK_W = 256, TRUNK_SKIP = 4, RGB_W = 128;
};

// BASELINE.json configs[0]: static scene, no warp / hyper / mask / normal.
struct GraphStatic {
  static constexpr int ID = 1;
  static constexpr bool HAS_MASK = false, HAS_WARP = false, HAS_HYPER = false, PREDICT_NORM = false, X_IN_RGB = false;
  static constexpr bool IDENT = false, WARP_IDENT = false;
  static constexpr int GLO = 8;
  static constexpr int MASK_BANDS = 0, MASK_DEPTH = 0, MASK_W = 128, MASK_SKIP = 4;
  static constexpr int WARP_BANDS = 0, WARP_DEPTH = 0, WARP_W = 128, WARP_SKIP = 4;
  static constexpr int HYP_BANDS = 0, HYP_DEPTH = 0, HYP_W = 64, HYP_SKIP = 4, HYP_DIMS = 0;
  static constexpr int SP_BANDS = 8, HP_BANDS = 0, VD_BANDS = 4, NM_BANDS = 0;
  static constexpr int TRUNK_DEPTH = 8, TRUNK_W = 256, TRUNK_SKIP = 4, RGB_W = 128;
};

// configs/base.gin (the HyperNeRF graph the reference inherits; BASELINE config 5 per SURVEY 8d): SE(3) warp with 6 bands
// + identity, bendy-sheet hyper slicing, posenc identity on x' and viewdirs, no mask net, no predicted normal.
struct GraphHyperNeRF {
  static constexpr int ID = 2;
  static constexpr bool HAS_MASK = false, HAS_WARP = true, HAS_HYPER = true, PREDICT_NORM = false, X_IN_RGB = false;
  static constexpr bool IDENT = true, WARP_IDENT = true;        // base.gin:21-22
  static constexpr int GLO = 8;
  static constexpr int MASK_BANDS = 0, MASK_DEPTH = 0, MASK_W = 128, MASK_SKIP = 4;
  static constexpr int WARP_BANDS = 6, WARP_DEPTH = 6, WARP_W = 128, WARP_SKIP = 4;
  static constexpr int HYP_BANDS = 6, HYP_DEPTH = 6, HYP_W = 64, HYP_SKIP = 4, HYP_DIMS = 2;
  static constexpr int SP_BANDS = 8, HP_BANDS = 1, VD_BANDS = 4, NM_BANDS = 0;
  static constexpr int TRUNK_DEPTH = 8, TRUNK_W = 256, TRUNK_SKIP = 4, RGB_W = 128;
};

template <class G> struct Dims {
  // linear input widths (reference concatenation order) and their k16-chunk counts
  static constexpr int ID3 = G::IDENT ? 3 : 0, WARP_ID3 = G::WARP_IDENT ? 3 : 0;   // identity prefix of posenc (model_utils.py:414-417)
  static constexpr int MASK_IN = 6 * G::MASK_BANDS + G::GLO;                       // posenc(x) | mask_embed
  static constexpr int WARP_IN = WARP_ID3 + 6 * G::WARP_BANDS + G::GLO + (G::HAS_MASK ? 1 : 0);   // [x] posenc(x) | warp_embed | [mask]
  static constexpr int HYP_IN = 6 * G::HYP_BANDS + G::GLO + (G::HAS_MASK ? 1 : 0);  // posenc(x) | warp_embed | [mask]
  static constexpr int TRUNK_IN = ID3 + 6 * G::SP_BANDS + 2 * G::HYP_DIMS * G::HP_BANDS;   // [x'] posenc(x') | posenc(w)
  static constexpr int VD_FEATS = ID3 + 6 * G::VD_BANDS;                           // [viewdir] posenc(viewdir)
  static constexpr int NM_FEATS = G::PREDICT_NORM ? ID3 + 6 * G::NM_BANDS : 0;     // [n] posenc(n)
  static constexpr int COND_IN = VD_FEATS + NM_FEATS;
  static constexpr int MASK_KC = chunks(MASK_IN), WARP_KC = chunks(WARP_IN), HYP_KC = chunks(HYP_IN);
  static constexpr int TRUNK_KC = chunks(TRUNK_IN), COND_KC = chunks(COND_IN);
  static constexpr int ALPHA_OUT = 1 + (G::PREDICT_NORM ? 3 : 0);
  static constexpr int RGB_IN = G::TRUNK_W + VD_FEATS + (G::X_IN_RGB ? G::TRUNK_W : 0) + NM_FEATS;

  // bias-tile counts of the two weight streams (precision independent; stream lengths: shared_units / nerf_units below)
  static constexpr int SHARED_BIAS_TILES = (G::HAS_MASK ? mlp_bias_tiles(G::MASK_DEPTH, G::MASK_W, true) : 0) +
                                           (G::HAS_WARP ? mlp_bias_tiles(G::WARP_DEPTH, G::WARP_W, true) : 0) +
                                           (G::HAS_HYPER ? mlp_bias_tiles(G::HYP_DEPTH, G::HYP_W, true) : 0);
  static constexpr int TW16 = G::TRUNK_W / 16, TW32 = G::TRUNK_W / 32;
  // The bottleneck Dense has no activation (modules.py:255), so it is folded into rgb hidden_0 by the packer:
  //   rgb_pre = trunk_out @ (W_bn @ W_rgb[bottleneck rows] + W_rgb[trunk_out rows]) + cond @ W_rgb[cond rows] + b'
  // i.e. no bottleneck layer in the stream and one 256-wide segment (instead of two) in rgb hidden_0.
  static constexpr int NERF_BIAS_TILES = G::TRUNK_DEPTH * TW32 + 1 + G::RGB_W / 32 + 1;
};

// Lengths, in 1-KiB units, of the two weight streams under a precision plan.  The order of the walk is the order in
// which pack.h emits and field.h consumes the fragments.
template <class G> constexpr int shared_units(Plan pl) {
  using D = Dims<G>;
  int pos = 0;
  if (G::HAS_MASK) pos = walk_mlp(pos, G::MASK_DEPTH, G::MASK_W, D::MASK_KC, G::MASK_SKIP, true, pl.mask);
  if (G::HAS_WARP) pos = walk_mlp(pos, G::WARP_DEPTH, G::WARP_W, D::WARP_KC, G::WARP_SKIP, true, pl.warp);
  if (G::HAS_HYPER) pos = walk_mlp(pos, G::HYP_DEPTH, G::HYP_W, D::HYP_KC, G::HYP_SKIP, true, pl.hyp);
  return pos;
}
template <class G> constexpr int nerf_units(Plan pl) {
  using D = Dims<G>;
  int pos = walk_mlp(0, G::TRUNK_DEPTH, G::TRUNK_W, D::TRUNK_KC, G::TRUNK_SKIP, false, pl.trunk);   // trunk
  pos = walk_seg(pos, D::TW16, pl.trunk);                                                            // alpha head
  for (int t = 0; t < G::RGB_W / 32; ++t) {                                                          // rgb hidden_0 (bottleneck folded in)
    pos = walk_seg(pos, D::TW16, pl.trunk);
    pos = walk_seg(pos, D::COND_KC, pl.rgb);
  }
  return walk_seg(pos, G::RGB_W / 16, pl.rgb);                                                       // rgb head
}

// ---- Reversed networks of the fused training backward (train_bwd_kernel.hip, nerfds_train.cpp) ----
// One stream per network: the TRANSPOSED layers in the order the data-gradient chain walks them, every fragment two units
// (split bf16).  A plain MLP with head (mask, warp, hyper sheet):
//   head^T (one linear chunk of head gradients -> width), layers depth-1 .. skip+1 transposed, the skip layer's [hidden rows]
//   (-> width) and [raw-input rows] (-> 2 tiles of input gradient), layers skip-1 .. 1 transposed, layer 0 transposed (-> 2 tiles).
// The NerfMLP: rgb head^T (-> rgb width), then [rgb hidden with the bottleneck folded in | alpha head]^T (-> trunk width), then the
// trunk like a plain MLP without head.
template <int W_, int DEPTH_, int NHEAD_, bool NERF_ = false, int RGB_W_ = 0> struct BwdNet {
  static constexpr int W = W_, DEPTH = DEPTH_, NHEAD = NHEAD_, SKIP = 4, RGB_W = RGB_W_;
  static constexpr bool IS_NERF = NERF_;
  // Waves per workgroup of the chain kernel.  The 64 / 128-wide chains need <= 256 registers per wave: EIGHT waves (two per SIMD) share one
  // weight ring, so the transposed weight set is streamed L2 -> LDS once per 256 rows (through round 3: two 4-wave workgroups per CU, each
  // with its own ring - the same occupancy, twice the stream per row; the chains run at the rate of that stream).  -DNERFDS_BWD_WAVES8=0: 4.
#ifndef NERFDS_BWD_WAVES8
#define NERFDS_BWD_WAVES8 1
#endif
  static constexpr int WG_WAVES = (NERFDS_BWD_WAVES8 && W_ <= 128) ? 8 : 4;
  static constexpr int BWD_FRAGS = (NERF_ ? (RGB_W_ / 32) * 1 + (W_ / 32) * (RGB_W_ / 16 + 1) : (W_ / 32) * 1) +
                                   (DEPTH_ - 1) * (W_ / 32) * (W_ / 16) + 2 * 2 * (W_ / 16);
};
template <class G> using BwdMask = BwdNet<G::MASK_W, G::MASK_DEPTH, 1>;
template <class G> using BwdWarp = BwdNet<G::WARP_W, G::WARP_DEPTH, 6>;
template <class G> using BwdHyper = BwdNet<G::HYP_W, G::HYP_DEPTH, G::HYP_DIMS>;
template <class G> using BwdNerf = BwdNet<G::TRUNK_W, G::TRUNK_DEPTH, 3, true, G::RGB_W>;
// the trunk behind its alpha head alone (no rgb branch): the backward of the TANGENT pass of the second-order terms, whose cotangent enters at
// the alpha head's tangent (d sigma_raw / d x, models.py:1035-1077)
template <class G> using BwdTrunkAlpha = BwdNet<G::TRUNK_W, G::TRUNK_DEPTH, 4>;

// ---- Tangent chains of the second-order terms (norm loss, elastic regulariser; train_bwd_kernel.hip train_tangent_kernel) ----
// Forward-mode tangents of a ReLU MLP are a masked LINEAR chain: t_l = 1[h_l > 0] * (W_l^T t_{l-1}), no bias, the mask the primal layer's.
// One stream per network in FORWARD orientation, split bf16 like the data-gradient chains: layer 0 (raw tangent input -> width), layers
// 1 .. skip-1, the skip layer over [hidden | raw input] (modules.py:66-67), layers skip+1 .. depth-1, then the head as one 32-row tile.
template <int W_, int DEPTH_, int IN_DIM_, int NHEAD_> struct TanNet {
  static constexpr int W = W_, DEPTH = DEPTH_, IN_DIM = IN_DIM_, IN_KC = chunks(IN_DIM_), NHEAD = NHEAD_, SKIP = 4;
  static constexpr int WG_WAVES = (NERFDS_BWD_WAVES8 && W_ <= 128) ? 8 : 4;
  // (field.h's Pipe reads the length of a one-stream graph under this name)
  static constexpr int BWD_FRAGS = (W_ / 32) * (2 * chunks(IN_DIM_) + (DEPTH_ - 1) * (W_ / 16)) + W_ / 16;
};
template <class G> using TanWarp = TanNet<G::WARP_W, G::WARP_DEPTH, Dims<G>::WARP_IN, 6>;
template <class G> using TanHyper = TanNet<G::HYP_W, G::HYP_DEPTH, Dims<G>::HYP_IN, G::HYP_DIMS>;
template <class G> using TanTrunk = TanNet<G::TRUNK_W, G::TRUNK_DEPTH, Dims<G>::TRUNK_IN, 4>;

}  // namespace nerfds
