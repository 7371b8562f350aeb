// Training backward of ONE network on the machinery of the fused field (field.h): see train_backward_kernel below.  Built twice by the
// Makefile (-DNERFDS_TRAIN_HALF=0 -DNERFDS_TRAIN_PIPE=0: fp32 g; =1 / =1: scaled f16 g, pipelined tile epilogue).
#define NERFDS_KERNEL_KIND 2
#include "field.h"
#include "launch.h"

namespace nerfds {

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
DEVI void bwd_hidden(Pipe<BG, PL>& pipe, BwdCursor& cur, const unsigned (&m)[OT / 2], float* g_base, size_t g_off, OUT& out, Ins&... ins) {
#pragma unroll
  for (int j = 0; j < OT / 2; ++j) cur.mask[j] = m[j];
  cur.row = g_base + g_off;                                        // fp32 [M][width] ...
  cur.row16 = reinterpret_cast<uint16_t*>(g_base) + g_off + (ROW16_H / 4 - 1) * cur.in_h4;   // ... or f16 [M][width] in the same buffer (TRAIN_HALF)
  dense<BG, PL, 1, OT, false>(pipe, cur, out, ins...);
}
template <class BG, class PL, int P, int K>
DEVI void bwd_input(Pipe<BG, PL>& pipe, BwdCursor& cur, float* in_row, int acc, Chunk<P> (&in)[1][K]) {
  Chunk<P> none[1][4];
  BwdInCursor ic;
  static_cast<BwdCursor&>(ic) = cur;
  ic.in_row = in_row;
  ic.in_acc = acc;
  dense<BG, PL, 1, 2, false>(pipe, ic, none, in);
  cur.pos = ic.pos;
  cur.bt = ic.bt;
}

template <class BG, class PL>
DEVI void bwd_chain(const TrainBwd& tb, Pipe<BG, PL>& pipe, int lane, long long r, int live) {
  // P: split bf16 - or, for the backward of the TANGENT pass (second-order terms only; launch_bwd<BG, true>), one f16 MFMA per product
  constexpr int W = BG::W, D = BG::DEPTH, W16 = W / 16, W32 = W / 32, P = PL::TRUNK, MW = W / 64;
  static_assert(P == P_BF16X3 || P == P_F16, "chain arithmetic");
  static_assert(BG::SKIP == 4 && (D == 8 || D == 6) && W % 64 == 0, "chains are written out for depth 8 / 6, skip 4");
  const int h = lane >> 5;
  BwdCursor cur;
  cur.seg = SEG_NERF; cur.pos = 0; cur.bt = 0;
  cur.row = nullptr; cur.row16 = nullptr; cur.bits = nullptr;
  cur.sink = tb.sink; cur.ld_in = tb.ld_in; cur.in_h4 = 4 * h; cur.in_row = nullptr; cur.in_acc = 0; cur.live = live;
  // the chain runs on g_scale * g (TrainBwd): a host value, or (scale_dev) picked on the device from this launch's largest cotangent
  const float gs = tb.scale_dev != nullptr ? tb.scale_dev[0] : tb.g_scale;
  cur.in_scale = tb.scale_dev != nullptr ? tb.scale_dev[1] : tb.g_inv_scale;
  // every load of the tile up front (one wait): ReLU bits of all layers, head gradients
  const long long rb = tb.mask_div == 3 ? (long long)((unsigned)r / 3u) : r;      // tangent rows read their SAMPLE's mask (rows < 2^31: nerfds_train.cpp)
  unsigned mk[D][MW];
#pragma unroll
  for (int l = 0; l < D; ++l) load_bits<W>(mk[l], tb.bits[l], rb, h);
  const size_t g_off = (size_t)r * W + 4 * h;
  float* const in_row = tb.d_in + (size_t)r * tb.ld_in + 4 * h;
  Chunk<P> a[1][W16], b[1][W16];
  if constexpr (BG::IS_NERF) {
    constexpr int RW = BG::RGB_W, R16 = RW / 16;
    unsigned mr[RW / 64];
    load_bits<RW>(mr, tb.bits[8], rb, h);
    Chunk<P> drgb[1][1], dalpha[1][1], c[1][R16];
    build_chunks<P, 1>(drgb[0], h, [&](int f) { return f < 3 ? val_feat(gs * tb.d_head[(size_t)r * tb.ld_head + f]) : zero_feat(); });
    build_chunks<P, 1>(dalpha[0], h, [&](int f) { return f < 4 ? val_feat(gs * tb.d_head2[(size_t)r * 4 + f]) : zero_feat(); });
    bwd_hidden<BG, PL, RW / 32>(pipe, cur, mr, tb.g[8], (size_t)r * RW + 4 * h, c, drgb);     // g_rgb = mask(W_rgb d rgb_logit)
    bwd_hidden<BG, PL, W32>(pipe, cur, mk[7], tb.g[7], g_off, a, c, dalpha);                          // g_7 = mask(F^T g_rgb + W_alpha d alpha)
  } else {
    Chunk<P> dh[1][1];
    build_chunks<P, 1>(dh[0], h, [&](int f) { return f < BG::NHEAD ? val_feat(gs * tb.d_head[(size_t)r * tb.ld_head + f]) : zero_feat(); });
    bwd_hidden<BG, PL, W32>(pipe, cur, mk[D - 1], tb.g[D - 1], g_off, a, dh);                          // g_{D-1} = mask(W_head d head)
  }
  if constexpr (D == 8) {
    bwd_hidden<BG, PL, W32>(pipe, cur, mk[6], tb.g[6], g_off, b, a);
    bwd_hidden<BG, PL, W32>(pipe, cur, mk[5], tb.g[5], g_off, a, b);
    bwd_hidden<BG, PL, W32>(pipe, cur, mk[4], tb.g[4], g_off, b, a);
  } else {
    bwd_hidden<BG, PL, W32>(pipe, cur, mk[4], tb.g[4], g_off, b, a);
  }
  // the skip layer [h_3 | raw input] (modules.py:66-67): its hidden rows give g_3, its raw-input rows the first part of d input
  bwd_hidden<BG, PL, W32>(pipe, cur, mk[3], tb.g[3], g_off, a, b);
  bwd_input<BG, PL>(pipe, cur, in_row, 0, b);
  bwd_hidden<BG, PL, W32>(pipe, cur, mk[2], tb.g[2], g_off, b, a);
  bwd_hidden<BG, PL, W32>(pipe, cur, mk[1], tb.g[1], g_off, a, b);
  bwd_hidden<BG, PL, W32>(pipe, cur, mk[0], tb.g[0], g_off, b, a);
  bwd_input<BG, PL>(pipe, cur, in_row, 1, b);
  pipe.finish_segment(SEG_NERF);
}

// ------------------------------------------------------------------------------------------------
// Tangent FORWARD chain of one network (the second-order terms of the objective: the norm loss differentiates d sigma_raw / d x, models.py:
// 1035-1077 with training.py:323-332; the elastic regulariser the warp field's Jacobian, training.py:112-156).  Forward-mode tangents of a
// ReLU MLP are a masked linear chain - t_l = 1[h_l > 0] * (W_l^T t_{l-1}), no bias - so they run on the machinery of the data-gradient
// chains above, in forward orientation (graphs.h TanNet): rows = tangents (three per sample, row 3 m + k), masks = the PRIMAL layers' ReLU bits
// of sample r / 3, every hidden tangent written as scaled f16 [rows][width] (the X operand of the tangent pass's weight gradients, k_wgrad_tr)
// and kept in registers as the next layer's B operand, the head's tangent written as fp32 [rows][ld_in].  TrainBwd is read as:
//   d_head / ld_head  the raw tangent input [rows][ld_head] (t_warp_in / t_hyper_in / t_tin)       g[l]      the f16 tangent of hidden layer l
//   d_in / ld_in      the head tangent out  [rows][ld_in]   (t_wv / t_wamb / t_alpha)              g_scale   the power of two the chain runs at
// Until round 5 the tangent pass ran layer by layer on fp32 rows through HBM (42 + 30 launches per level, 12 GB): DESIGN 10.
// ------------------------------------------------------------------------------------------------
#if NERFDS_TRAIN_HALF
template <class TG, class PL>
DEVI void tan_chain(const TrainBwd& tb, Pipe<TG, PL>& pipe, int lane, long long r) {
  // P: split bf16 - or (launch_tan<TG, true>: experiment NERFDS_TRAIN_TAN_FWD_F16, nerfds_train.cpp) one f16 MFMA per product
  constexpr int W = TG::W, D = TG::DEPTH, W16 = W / 16, W32 = W / 32, P = PL::TRUNK, MW = W / 64, KC = TG::IN_KC;
  static_assert(P == P_BF16X3 || P == P_F16, "chain arithmetic");
  static_assert(TG::SKIP == 4 && (D == 8 || D == 6) && W % 64 == 0 && TG::NHEAD <= 16, "chains are written out for depth 8 / 6, skip 4");
  const int h = lane >> 5;
  BwdCursor cur;
  cur.seg = SEG_NERF; cur.pos = 0; cur.bt = 0;
  cur.row = nullptr; cur.row16 = nullptr; cur.bits = nullptr;
  cur.sink = tb.sink; cur.ld_in = 0; cur.in_h4 = 4 * h; cur.in_row = nullptr; cur.in_acc = 0; cur.live = 0; cur.in_scale = 1.f;
  const float ts = tb.g_scale;
  const long long rb = tb.mask_div == 3 ? (long long)((unsigned)r / 3u) : r;
  unsigned mk[D][MW];
#pragma unroll
  for (int l = 0; l < D; ++l) load_bits<W>(mk[l], tb.bits[l], rb, h);
  const size_t g_off = (size_t)r * W + 4 * h;
  const float* const tin_row = tb.d_head + (size_t)r * tb.ld_head;
  Chunk<P> tin[1][KC], a[1][W16], b[1][W16];
  build_chunks<P, KC>(tin[0], h, [&](int f) { return f < TG::IN_DIM ? val_feat(ts * tin_row[f < TG::IN_DIM ? f : 0]) : zero_feat(); });
  bwd_hidden<TG, PL, W32>(pipe, cur, mk[0], tb.g[0], g_off, a, tin);
  bwd_hidden<TG, PL, W32>(pipe, cur, mk[1], tb.g[1], g_off, b, a);
  bwd_hidden<TG, PL, W32>(pipe, cur, mk[2], tb.g[2], g_off, a, b);
  bwd_hidden<TG, PL, W32>(pipe, cur, mk[3], tb.g[3], g_off, b, a);
  bwd_hidden<TG, PL, W32>(pipe, cur, mk[4], tb.g[4], g_off, a, b, tin);                // the skip layer: [h_3 | raw input] (modules.py:66-67)
  bwd_hidden<TG, PL, W32>(pipe, cur, mk[5], tb.g[5], g_off, b, a);
  f32x16 acc[1][1];
  acc[0][0] = f32x16{};
  int j = 0;
  auto no_slot = [](int, int) {};
  if constexpr (D == 8) {
    bwd_hidden<TG, PL, W32>(pipe, cur, mk[6], tb.g[6], g_off, a, b);
    bwd_hidden<TG, PL, W32>(pipe, cur, mk[7], tb.g[7], g_off, b, a);
  }
  accum<TG, PL, 1, 1>(acc, pipe, cur, b, j, no_slot);                                    // the head: logical output j = acc[j] (packer: is_head)
  float* const out = tb.d_in + (size_t)r * tb.ld_in;
  const float inv = tb.g_inv_scale;
#pragma unroll
  for (int o = 0; o < TG::NHEAD; ++o) out[o] = acc[0][0][o] * inv;                        // (both lane halves and the tail lanes hold the row's own values)
  pipe.finish_segment(SEG_NERF);
}

template <class TG, class PL, int TAG>
__global__ __launch_bounds__(64 * TG::WG_WAVES) void train_tangent_kernel(const TrainBwd tb) {
  using PP = Pipe<TG, PL>;
  static_assert((PP::WAVES == 4 || PP::WAVES == 8) && !PP::HAS_SHARED, "one 512-register wave or two 256-register waves per SIMD, one stream");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Pipe<TG, PL> pipe;
  pipe.cur = pipe.next = make_rsrc(tb.wstream, PP::NERF_PAD * 1024);
  pipe.lane16 = lane * 16;
  pipe.wave1k = wave * 1024;
  pipe.prologue(SEG_NERF);
  constexpr int ROWS = 32 * PP::WAVES;
  const long long groups = (tb.M + ROWS - 1) / ROWS;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long rr = grp * ROWS + wave * 32 + (lane & 31);
    tan_chain<TG, PL>(tb, pipe, lane, rr < tb.M ? rr : tb.M - 1);                        // tail lanes redo the last row: same values to the same places
  }
}
#endif

template <class BG, class PL, int TAG>
__global__ __launch_bounds__(64 * BG::WG_WAVES) void train_backward_kernel(const TrainBwd tb) {
  using PP = Pipe<BG, PL>;
  static_assert((PP::WAVES == 4 || PP::WAVES == 8) && !PP::HAS_SHARED, "one 512-register wave or two 256-register waves per SIMD, one stream");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  Pipe<BG, PL> pipe;
  pipe.cur = pipe.next = make_rsrc(tb.wstream, PP::NERF_PAD * 1024);
  pipe.lane16 = lane * 16;
  pipe.wave1k = wave * 1024;
  pipe.prologue(SEG_NERF);
  constexpr int ROWS = 32 * PP::WAVES;
  const long long groups = (tb.M + ROWS - 1) / ROWS;
  for (long long grp = blockIdx.x; grp < groups; grp += gridDim.x) {
    const long long rr = grp * ROWS + wave * 32 + (lane & 31);
    bwd_chain<BG, PL>(tb, pipe, lane, rr < tb.M ? rr : tb.M - 1, rr < tb.M ? 1 : 0);     // tail lanes redo the last row: same g values to the same places
  }
}

}  // namespace nerfds

template <class BG, bool F16 = false> static void launch_bwd(const nerfds::TrainBwd& tb, int num_cus, void* stream) {
  using namespace nerfds;
  // the data-gradient chains: split bf16 throughout; F16: one f16 MFMA per product (the tangent pass's backward, nerfds_train.cpp)
  using PLX = std::conditional_t<F16, PlanT<P_F16, P_F16, P_F16, P_F16, P_F16>, PlanT<P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3>>;
  auto kern = train_backward_kernel<BG, PLX, TRAIN_TAG>;
  allow_dynamic_lds(reinterpret_cast<const void*>(kern), RING_BYTES);
  constexpr int WAVES = BG::WG_WAVES, ROWS = 32 * WAVES;
  const long long groups = (tb.M + ROWS - 1) / ROWS;
  // two waves per SIMD for the 64 / 128-wide chains (<= 256 registers; they cover each other's waits): one 8-wave workgroup per CU on one
  // ring, or (NERFDS_BWD_WAVES8=0) two 4-wave workgroups; the trunk's chain takes the whole register file with four waves
  const long long want = (long long)num_cus * (BG::W <= 128 && WAVES == 4 ? 2 : 1);
  const int grid = (int)(groups < want ? groups : want);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), RING_BYTES, static_cast<hipStream_t>(stream), tb);
}
// net: 0 NerfMLP (trunk + rgb branch + alpha head), 1 hyper sheet, 2 warp field, 3 mask net, 4 trunk behind its alpha head alone (tangent pass)
extern "C" void NERFDS_CAT(nerfds_launch_, NERFDS_NAME)(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream) {
  using G = nerfds::NERFDS_GRAPH;
  if (net == 0) launch_bwd<nerfds::BwdNerf<G>>(tb, num_cus, stream);
  else if (net == 1) launch_bwd<nerfds::BwdHyper<G>>(tb, num_cus, stream);
  else if (net == 2) launch_bwd<nerfds::BwdWarp<G>>(tb, num_cus, stream);
  else if (net == 3) launch_bwd<nerfds::BwdMask<G>>(tb, num_cus, stream);
  else launch_bwd<nerfds::BwdTrunkAlpha<G>>(tb, num_cus, stream);
}

#if NERFDS_TRAIN_HALF
// the data-gradient chains in one f16 MFMA per product (the primal step's - NERFDS_TRAIN_BWD_F16 - and the tangent pass's): net 0 NerfMLP, 1 hyper sheet,
// 2 warp field, 3 mask net, 4 trunk + alpha head (f16-store build only)
extern "C" void nerfds_launch_train_bwd16f_nerfds(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream) {
  using G = nerfds::NERFDS_GRAPH;
  if (net == 1) launch_bwd<nerfds::BwdHyper<G>, true>(tb, num_cus, stream);
  else if (net == 2) launch_bwd<nerfds::BwdWarp<G>, true>(tb, num_cus, stream);
  else if (net == 0) launch_bwd<nerfds::BwdNerf<G>, true>(tb, num_cus, stream);
  else if (net == 3) launch_bwd<nerfds::BwdMask<G>, true>(tb, num_cus, stream);
  else launch_bwd<nerfds::BwdTrunkAlpha<G>, true>(tb, num_cus, stream);
}
template <class TG, bool F16 = false> static void launch_tan(const nerfds::TrainBwd& tb, int num_cus, void* stream) {
  using namespace nerfds;
  using PLX = std::conditional_t<F16, PlanT<P_F16, P_F16, P_F16, P_F16, P_F16>, PlanT<P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3, P_BF16X3>>;
  auto kern = train_tangent_kernel<TG, PLX, TRAIN_TAG>;
  allow_dynamic_lds(reinterpret_cast<const void*>(kern), RING_BYTES);
  constexpr int WAVES = TG::WG_WAVES, ROWS = 32 * WAVES;
  const long long groups = (tb.M + ROWS - 1) / ROWS;
  const int grid = (int)(groups < num_cus ? groups : num_cus);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), RING_BYTES, static_cast<hipStream_t>(stream), tb);
}
// the tangent forward chain of net 1 hyper sheet, 2 warp field, 4 trunk + alpha head (f16-store build only)
extern "C" void nerfds_launch_train_tan16_nerfds(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream) {
  using G = nerfds::NERFDS_GRAPH;
  if (net == 1) launch_tan<nerfds::TanHyper<G>>(tb, num_cus, stream);
  else if (net == 2) launch_tan<nerfds::TanWarp<G>>(tb, num_cus, stream);
  else launch_tan<nerfds::TanTrunk<G>>(tb, num_cus, stream);
}
// the trunk's tangent forward chain in one f16 MFMA per product (a training step that differentiates the tangent pass; never the rendered target_norm)
extern "C" void nerfds_launch_train_tan16f_nerfds(const nerfds::TrainBwd& tb, int num_cus, void* stream) {
  launch_tan<nerfds::TanTrunk<nerfds::NERFDS_GRAPH>, true>(tb, num_cus, stream);
}
#endif
