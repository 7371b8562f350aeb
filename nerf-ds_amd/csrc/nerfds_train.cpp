// Training step of BASELINE config 4: loss = MSE(rgb_fine, gt) + MSE(rgb_coarse, gt)
// (training.py:265-274, 459-466, 481), gradients w.r.t. every parameter of the nerf_ds graph (training.py:494) and the
// Adam update (training.py:508).  Activations of every layer stay resident in HBM (needed for dW); the dense layers
// (forward X W, backward dX = dZ W^T and dW = X^T dZ) run on the hand-written MFMA kernels of train_gemm.hip and on the
// fused forward / backward chains of train_fwd_kernel.hip / train_bwd_kernel.hip (over field.h) (no BLAS library is linked), everything else in train_kernels.hip.  The two levels are processed one
// after the other through the same workspace: their losses are independent sums and the fine z samples carry a
// stop_gradient (model_utils.py:241), so no gradient crosses from the fine level into the coarse one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "nerfds.h"
#include "train_kernels.h"
#include "train_gemm.h"
#include "graphs.h"
#include "kargs.h"
#include "pack.h"

// train_fwd_kernel.hip: the fused forward of one level (TRAIN_PLAN arithmetic)
extern "C" void nerfds_launch_train_fwd_nerfds(const nerfds::KArgs& ka, const nerfds::TrainOut& to, int num_cus, void* stream);     // fp32 activations
extern "C" void nerfds_launch_train_fwd16_nerfds(const nerfds::KArgs& ka, const nerfds::TrainOut& to, int num_cus, void* stream);   // f16 + ReLU bits
// train_bwd_kernel.hip: the data-gradient chain of one network (0 NerfMLP, 1 hyper sheet, 2 warp, 3 mask)
extern "C" void nerfds_launch_train_bwd_nerfds(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream);      // g as fp32
extern "C" void nerfds_launch_train_bwd16_nerfds(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream);    // g as scaled f16
extern "C" void nerfds_launch_train_bwd16f_nerfds(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream);   // the same chains in one f16 MFMA per product (tangent pass only)
extern "C" void nerfds_launch_train_tan16_nerfds(const nerfds::TrainBwd& tb, int net, int num_cus, void* stream);    // tangent forward chains (train_bwd_kernel.hip)
extern "C" void nerfds_launch_train_tan16f_nerfds(const nerfds::TrainBwd& tb, int num_cus, void* stream);            // the trunk's, one f16 MFMA per product

using namespace nerfds_train;

namespace {

thread_local std::string g_train_error;

struct Leaf { std::string name; int64_t off; int rows, cols; };
struct LayerP { int64_t w, b; int K, N; };
struct MlpP { std::vector<LayerP> hidden; int in_dim = 0, in_ld = 0, width = 0, depth = 0, skip = -1; };      // in_ld: row stride of the input buffer
struct Seg {
  const float* x; int ld; int K; float* dx; int dld; bool acc;
  const float* dx_relu_y = nullptr;   // dx is the dY of a ReLU layer whose output is dx_relu_y: mask it and add its column sums to dx_bias_grad
  float* dx_bias_grad = nullptr;
};
constexpr int GRAD_REPS = 16;
constexpr size_t ARENA_BYTES = 96u << 20;     // fragment packs of every layer, both orientations
constexpr size_t WPACK_BYTES = 1 << 20;      // packed fragments of the largest layer (560 x 128 or 320 x 256 as hi / lo bf16) fit twice

void window(float* out, int bands, float alpha) {   // model_utils.py:420-436
  for (int b = 0; b < 8; ++b) {
    const double x = std::min(std::max((double)alpha - b, 0.0), 1.0);
    out[b] = b < bands ? (float)(0.5 * (1.0 + std::cos(M_PI * x + M_PI))) : 0.f;
  }
}

}  // namespace

struct nerfds_trainer {
  int device = 0;
  nerfds_model_cfg cfg{};
  Dims D{};
  int64_t max_rays = 0;
  std::vector<Leaf> leaves;
  int64_t P = 0;
  MlpP mask, warp, hyper, trunk[2];
  LayerP mask_out, warp_w, warp_v, hyper_out, bott[2], alpha[2], rgb_h[2], rgb_out[2];
  int64_t warp_tbl = -1, mask_tbl = -1;
  float *theta = nullptr, *grad = nullptr, *m1 = nullptr, *m2 = nullptr;
  // the optimizer's step count + the bias corrections of the update being applied, on the DEVICE (train_kernels.hip k_adam_prepare): 8 + 2 x 4 bytes
  unsigned char* adam_dev = nullptr;
  long long* adam_step() const { return reinterpret_cast<long long*>(adam_dev); }
  float* adam_corr() const { return reinterpret_cast<float*>(adam_dev + 8); }
  // a caller-defined loss (nerfds_trainer_forward / nerfds_render_rays_bwd): per level (0 coarse, 1 fine) the cotangents that replace the squared error's
  // gradient, where the level's rgb / depth / acc go, and "stop after the forward of each level" (set for the duration of one call)
  LevelCot cot[2] = {{0, nullptr, nullptr, nullptr}, {0, nullptr, nullptr, nullptr}};
  LevelOut lout[2] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  bool fwd_only = false;
  // dynamic loss scaling of the stored f16 g (include/nerfds.h nerfds_trainer_set_loss_scale_adjust): log2 offset on the heuristic exponent
  int g_scale_adjust = 0;
  // Numeric policy of a step (include/nerfds.h nerfds_train_numerics; the host's overflow ladder, nerfds_amd/training.py Trainer.step, sets it):
  //   tan_scale_adjust  log2 offset on BOTH powers of two of the second-order terms' f16 storage: the stored tangents' (2^(-6 + adjust)) and the target the
  //                     device-picked cotangent scale aims at (2^(5 + adjust)) - lower = more headroom for what the chains amplify
  //   chain_arith       0: the chains the build runs in one f16 MFMA per product (tan_bwd_f16 / rev_fwd_f16 / tan_fwd_f16 / bwd_f16) do; 1: every chain in split
  //                     bf16 (operands with fp32's exponent range: nothing overflows BETWEEN the layers of a chain)
  //   fp32_step         the step keeps fp32 activations and fp32 g and runs layer by layer (no f16 storage anywhere: the arithmetic range of the reference's step)
  int tan_scale_adjust = 0, chain_arith = 0;
  bool fp32_step = false;
  bool diag = false;         // nerfds_train_numerics.diagnose: scan the coarse NerfMLP's arrays before the fine level overwrites them (diag_scan_nerf)
  bool use_tan_bwd_f16() const { return tan_bwd_f16 && chain_arith == 0; }
  bool use_rev_fwd_f16() const { return rev_fwd_f16 && chain_arith == 0; }
  bool use_tan_fwd_f16() const { return tan_fwd_f16 && chain_arith == 0; }
  float scale_target() const { return 5.f + (float)tan_scale_adjust; }      // the largest cotangent of a tangent chain lands at 2^5 by default: 2^11 of headroom
  // what the last step ran on (nerfds_trainer_overflow_sources scans these rows): rays, rows of the tangent arrays, second-order terms on / off
  int64_t last_R = 0, last_tan_rows = 0;
  bool last_half = false, last_keep_tangents = false, last_tan16 = false;
  float* ws = nullptr;      // one workspace allocation
  size_t ws_floats = 0;
  float* loss_dev = nullptr;
  float* tws = nullptr;     // tangent workspace of the sigma gradient (allocated on first use)
  float *t_warp_in = nullptr, *t_hyper_in = nullptr, *tA = nullptr, *tB = nullptr, *t_wv = nullptr, *t_xw = nullptr, *t_wamb = nullptr, *t_tin = nullptr, *t_alpha = nullptr;
  float* tws32 = nullptr;   // tA / tB and the fp32 hidden tangents of the layer-by-layer tangent pass when the fused one is built too (fp32_step / NERFDS_TRAIN_HALF_TANGENTS=0)
  float* tn[2] = {nullptr, nullptr};      // target_norm of the coarse / fine level of the last step
  float* nws = nullptr;     // norm-loss workspace: stored tangents of every layer + tangent gradients (allocated on first use)
  std::vector<float*> tw_h, th_h, tt_h;
  float *d_t_alpha = nullptr, *d_t_tin = nullptr, *d_t_xw = nullptr, *d_t_wamb = nullptr, *d_t_wv = nullptr, *du = nullptr, *ghat = nullptr, *dwamb_extra = nullptr, *dwv_extra = nullptr;
  bool tn_valid = false;
  bool tangents_warp_only = false;   // this step's tangent pass stops at the warped point (elastic regulariser without norm loss / sigma-gradient flag)
  bool keep_tangents = false;
  void* wpack = nullptr;    // MFMA fragments of the layer being run (train_gemm.hip)
  // Gradient replicas: the MFMA kernels end with float atomics from every workgroup at once; on one copy of a small leaf they queue
  // up per address (40-60 us per kernel).  Workgroup b adds into replica b % GRAD_REPS; the replicas are summed into grad once per step.
  float* grad_rep = nullptr;
  // Side streams of the fused backward: a weight-gradient launch ends with ~50 us of float atomics during which HBM idles (one workgroup
  // per CU, 256 K adds each); the launches of a level are independent of each other, so they alternate over the caller's stream and
  // these two - the tail of one runs under the streaming phase of the next.  Forked / joined with events inside every level.
  static constexpr int SIDE = 6;     // at most; nside = the streams in use (NERFDS_TRAIN_SIDE_STREAMS, default 3; 0 = none)
  int nside = 0;
  int nside_eff = 0;                 // the side streams THIS step uses (a small batch: one - step_impl)
  int wgrad_cu_cap = 0;              // CUs the side streams' persistent weight-gradient workgroups may take THIS step (0: all - step_impl)
  hipStream_t side[SIDE] = {};
  hipEvent_t fork_ev = nullptr, join_ev[SIDE] = {};
  // Fragment packs of the layers (train_gemm.h): the first step packs each (weight block, orientation, split) when it is first used and
  // records it; from then on ONE kernel at the start of a step packs them all into the arena (150 small launches less per step).
  std::vector<PackEntry> packs;
  std::unordered_map<std::string, size_t> pack_index;
  PackEntry* packs_dev = nullptr;
  size_t packs_dev_n = 0, arena_used = 0;
  int max_frag_lanes = 0;
  char* arena = nullptr;
  uint64_t pack_epoch = 0;
  std::vector<uint64_t> pack_fresh;
  int num_cus = 256;
  bool fuse_bwd = true;     // NERFDS_TRAIN_FUSE_BWD=0: the narrow layers' backward as two kernels (A/B timing)
  // Fused forward (train_fwd_kernel.hip): the whole field evaluation of a level in ONE launch, activations written once
  // for the backward pass.  Its weight streams [shared, coarse, fine] are re-packed on the device from theta at the start of every step
  // through index maps built once (build_fused_forward).  NERFDS_TRAIN_FUSED_FWD=0 runs the forward layer by layer (A/B timing).
  bool fused_fwd = false;
  int* fmap[3] = {nullptr, nullptr, nullptr};       // per float slot of the fp32-layout stream: 1 + parameter index (past P: fold), 0 = zero
  int* fbmap[3] = {nullptr, nullptr, nullptr};
  void* fstream[3] = {nullptr, nullptr, nullptr};
  float* fbias[3] = {nullptr, nullptr, nullptr};
  float* fold = nullptr;    // [2 levels][(TW + 1) x RGB_W]: rgb hidden_0 with the bottleneck folded in, then its bias
  int fstream_frags[3] = {0, 0, 0}, fbias_n[3] = {0, 0, 0}, f32_lo = 0, f32_hi = 0;
  // Fused backward (train_bwd_kernel.hip), the plain step's default when the fused forward is on: the forward
  // writes every hidden layer as f16 + ReLU bits, ONE launch per network walks its data-gradient chain (dX stays in registers
  // between layers) and leaves g_l = d loss / d pre-activation of every hidden layer in the fp32 arrays the forward did not use,
  // and one weight-gradient launch per layer reads X (f16) and g_l.  Streams [NerfMLP coarse, NerfMLP fine, hyper, warp, mask] are
  // re-packed from theta every step through index maps, like the forward's.  NERFDS_TRAIN_FUSED_BWD=0: layer-by-layer backward.
  bool fused_bwd = false;
  int* bmap[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  void* bstream[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int bfrags[5] = {0, 0, 0, 0, 0};
  // Fused tangent pass of the second-order terms (round 5; train_bwd_kernel.hip train_tangent_kernel + the data-gradient chains on tangent rows):
  // index maps / streams of the tangent FORWARD chains [trunk + alpha head coarse, fine, hyper sheet, warp field] and of the reversed
  // trunk-behind-its-alpha-head chain [coarse, fine] (graphs.h TanNet / BwdTrunkAlpha), re-packed from theta in the steps that run tangents;
  // the tangents of every hidden layer (tw16 / th16 / tt16: [3 M][width] f16, scaled by tan_x_scale) and their cotangents (gw16 / gh16 / gt16,
  // scaled on the device: tan_slot) are the operands of the tangent pass's weight gradients.  NERFDS_TRAIN_FUSED_TAN=0: layer by layer on fp32 rows.
  bool fused_tan = false;
  int* tmap[4] = {nullptr, nullptr, nullptr, nullptr};
  void* tstream[4] = {nullptr, nullptr, nullptr, nullptr};
  int tfrags[4] = {0, 0, 0, 0};
  int* amap[2] = {nullptr, nullptr};
  void* astream[2] = {nullptr, nullptr};
  int afrags[2] = {0, 0};
  // the reversed streams of the tangent pass's backward as ONE f16 unit per fragment: [trunk + alpha coarse, fine, hyper, warp] (tan_bwd_f16: the
  // second-order terms' data-gradient chains in one f16 MFMA per product; NERFDS_TRAIN_TAN_BWD_F16=0: split bf16 like the primal chains)
  bool tan_bwd_f16 = false;
  void* bstream16[4] = {nullptr, nullptr, nullptr, nullptr};
  // OPTION (NERFDS_TRAIN_BWD_F16=1, read at every step; off): the PRIMAL data-gradient chains in one f16 MFMA per product too - the five reversed
  // streams as f16 units.  Every g these chains produce is handed to the weight gradients as loss-scaled f16 already; on f16 operands they also
  // PROPAGATE it at 11 bits per layer (fp32 accumulate).  rgb step 11.7 -> 10.3 ms, nerf_ds.gin 22.6 -> 21.3.  Against the fp64 oracle: 64 rays 2.97e-3
  // (2.90e-3 in split bf16), 33 rays 3.2e-3 (2.8e-3), multi-tile 8.8e-3 on the noisy leaf (7.9e-3), but the 6-ray case 1.4e-3 (3.5e-4) and the 16-ray
  // golden digests 2.2e-3 / 1.4e-3 (6.9e-4 / 1.1e-4), and the head cotangents times the loss scale must now fit f16 (split bf16 has fp32's range:
  // the scale sweep of test_gradient_scale_of_the_f16_g_arrays overflows at 2^22) - so it is the caller's choice, not the default.
  bool bwd_f16 = false;
  void* pstream16[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // the trunk's tangent FORWARD stream of each level as one f16 unit per fragment (tan_fwd_f16: NERFDS_TRAIN_TAN_FWD_F16=1, training steps only)
  bool tan_fwd_f16 = false;
  bool rev_fwd_f16 = false;      // the reverse-mode path's directional tangent pass of the trunk in f16 (NERFDS_TRAIN_REV_FWD_F16=0: split bf16)
  void* tstream16[2] = {nullptr, nullptr};
  uint16_t* tws16 = nullptr;      // the tangents (allocated on first use)
  uint16_t* gws16 = nullptr;      // their cotangents (allocated on first use by a step that differentiates the tangent pass)
  std::vector<uint16_t*> tw16, th16, tt16, gw16, gh16, gt16;
  float* tan_slot = nullptr;      // device [3][4]: {amax bits, scale, 1 / scale, 1 / (scale * tan_x_scale)} of the trunk / hyper / warp cotangents (pick_scale)
  // the power of two the stored tangents carry: 2^-6 puts the largest raw tangent input (d posenc / d x at the 2^7 band... 2^6 * window) at 1
  float tan_x_scale = 0.015625f;
  uint16_t* hws = nullptr;   // f16 activations + ReLU bits
  std::vector<uint16_t*> mask_h16, warp_h16, hyper_h16, trunk_h16, mask_bits, warp_bits, hyper_bits, trunk_bits;
  uint16_t *rgb_h16 = nullptr, *rgb_bits = nullptr;
  float* sink = nullptr;
  // the fused backward's chains leave g_l as f16, scaled by g_scale (the weight-gradient kernels' dY operand: with the forward's f16
  // activations one MFMA per product and no conversion pass; the chains themselves keep g in registers at full precision).
  // NERFDS_TRAIN_G16=0: fp32 g arrays.
  bool g16 = true;
  // Loss scaling of the stored g: a power of two, 64 x the ray count rounded up to a power of two, so that a head gradient of the mean squared
  // error - at most 2 / (3 R) per unit of colour error - lands near 2^5 with 2^11 of headroom below f16's largest value for what the layers
  // amplify and 2^-19 .. 2^-29 of it (f16's normal .. denormal range) still represented; a g beyond the range gives an inf weight gradient
  // and the step reports NERFDS_ENONFINITE.  NERFDS_TRAIN_G_SCALE_LOG2 overrides the exponent.
  float g_scale = 1.f;
  bool half_step = false;    // this step's forward wrote f16 + bits
  std::string err;
  // workspace views (set by carve())
  float *zc, *zf, *wc, *rs_scratch, *x, *mask_in, *mask_logit, *warp_in, *wv, *xw, *hyper_in, *wamb, *trunk_in, *bottv, *alphav, *sigma,
      *cond, *rgb_hv, *rgb_logit, *weights, *rgb_ray, *g0, *g1, *g2, *d_trunk_in, *d_rgb_logit, *d_alpha, *dxw, *dwamb, *dwv, *d_warp_in,
      *d_hyper_in, *d_mask_in, *d_mask_logit, *dxw_reg, *d_pm, *dwamb_reg;
  float* terms_dev = nullptr;   // [2 levels][4]: weighted warp_reg, back_facing, mask terms of the last step
  std::vector<float*> mask_h, warp_h, hyper_h, trunk_h;

  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
  ~nerfds_trainer() {
    for (float* p : {theta, grad, m1, m2, ws, loss_dev, tws, tws32, terms_dev, nws, tan_slot}) if (p) (void)hipFree(p);
    for (uint16_t* p : {tws16, gws16}) if (p) (void)hipFree(p);
    for (int i = 0; i < 4; ++i) { if (tmap[i]) (void)hipFree(tmap[i]); if (tstream[i]) (void)hipFree(tstream[i]); }
    for (int i = 0; i < 2; ++i) { if (amap[i]) (void)hipFree(amap[i]); if (astream[i]) (void)hipFree(astream[i]); }
    for (int i = 0; i < 4; ++i) if (bstream16[i]) (void)hipFree(bstream16[i]);
    for (int i = 0; i < 5; ++i) if (pstream16[i]) (void)hipFree(pstream16[i]);
    for (int i = 0; i < 2; ++i) if (tstream16[i]) (void)hipFree(tstream16[i]);
    if (adam_dev) (void)hipFree(adam_dev);
    if (wpack) (void)hipFree(wpack);
    if (grad_rep) (void)hipFree(grad_rep);
    for (int i = 0; i < SIDE; ++i) { if (side[i]) (void)hipStreamDestroy(side[i]); if (join_ev[i]) (void)hipEventDestroy(join_ev[i]); }
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    if (arena) (void)hipFree(arena);
    if (packs_dev) (void)hipFree(packs_dev);
    for (int i = 0; i < 3; ++i) {
      if (fmap[i]) (void)hipFree(fmap[i]);
      if (fbmap[i]) (void)hipFree(fbmap[i]);
      if (fstream[i]) (void)hipFree(fstream[i]);
      if (fbias[i]) (void)hipFree(fbias[i]);
    }
    if (fold) (void)hipFree(fold);
    for (int i = 0; i < 5; ++i) {
      if (bmap[i]) (void)hipFree(bmap[i]);
      if (bstream[i]) (void)hipFree(bstream[i]);
    }
    if (hws) (void)hipFree(hws);
    if (sink) (void)hipFree(sink);
  }
};

namespace {

int64_t add_leaf(nerfds_trainer& t, const std::string& name, int rows, int cols) {
  const int64_t off = t.P;
  t.leaves.push_back({name, off, rows, cols});
  t.P += (int64_t)rows * cols;
  return off;
}
LayerP add_dense(nerfds_trainer& t, const std::string& name, int K, int N) {
  LayerP l;
  l.K = K; l.N = N;
  l.w = add_leaf(t, name + "/kernel", K, N);
  l.b = add_leaf(t, name + "/bias", 1, N);
  return l;
}
MlpP add_mlp(nerfds_trainer& t, const std::string& name, int in_dim, int width, int depth, int skip) {
  MlpP m;
  m.in_dim = in_dim; m.in_ld = in_dim; m.width = width; m.depth = depth; m.skip = skip;
  for (int l = 0; l < depth; ++l) {
    const int K = (l == 0 ? in_dim : width) + ((l == skip && l > 0) ? in_dim : 0);     // modules.py:66-67
    m.hidden.push_back(add_dense(t, name + "/hidden_" + std::to_string(l), K, width));
  }
  return m;
}

struct Run {
  nerfds_trainer& t;
  hipStream_t st;
  int64_t M;
  bool ok = true;
  // Set around the FORWARD of the warp field: its output moves the points that the 2^7-frequency posenc of the template
  // reads, which amplifies the 16-bit operand rounding of the two-way split to ~1 % on the warp-field gradients (measured
  // against the fp64 oracle).  Those layers run the three-way split (fp32-level products).
  bool precise_layers = false;
  // Weight gradients of the TANGENT pass (fused tangent chains): rows are tangents - no bias gradients -, the f16 X operand carries
  // t.tan_x_scale and the f16 g a scale picked on the device (tan_slot: {amax, scale, 1 / scale, 1 / (scale * tan_x_scale)})
  bool tan = false;
  const float* tan_slot = nullptr;
  // Every dense layer of the step runs on the hand-written MFMA kernels of train_gemm.hip.  A shape they do not cover is an
  // error (NERFDS_ENOTSUP from the step), never a detour through a library GEMM.
  std::string unsupported_what;
  void unsupported(const char* what, int K, int N, int64_t rows) {
    if (ok) {
      char buf[160];
      snprintf(buf, sizeof buf, "%s: shape K=%d N=%d rows=%lld is not covered by the MFMA layer kernels", what, K, N, (long long)rows);
      unsupported_what = buf;
    }
    ok = false;
  }

  // y[M x N] (ldy) = act(sum_s x_s W[rows of s] + b)
  // where the MFMA kernels add a gradient that lives at g in t.grad: replica 0 of the same offset (kernels add b % nrep replicas on)
  float* rep(float* g) const { return (g && t.grad_rep) ? t.grad_rep + (g - t.grad) : g; }
  int nrep() const { return t.grad_rep ? GRAD_REPS : 1; }
  // fragment pack of W[row0.. , :] for (in_dim, out_dim, orientation, split): from the arena (packed at the start of the step) or,
  // the first time this pack is asked for, packed now and recorded for the following steps; nullptr = arena full (caller packs into wpack)
  const void* packed(const float* W, int ldw, int row0, int in_dim, int out_dim, int transpose, int parts) {
    if (!t.arena) return nullptr;
    char key[96];
    snprintf(key, sizeof key, "%lld/%d/%d/%d/%d/%d/%d", (long long)(W - t.theta), ldw, row0, in_dim, out_dim, transpose, parts);
    auto it = t.pack_index.find(key);
    if (it != t.pack_index.end() && t.pack_fresh[it->second] == t.pack_epoch) return t.arena + t.packs[it->second].dst;
    size_t idx;
    if (it == t.pack_index.end()) {
      const size_t bytes = (frag_bytes(in_dim, out_dim, parts) + 255) & ~(size_t)255;
      if (t.arena_used + bytes > ARENA_BYTES) return nullptr;
      idx = t.packs.size();
      t.packs.push_back({(long long)(W - t.theta), (long long)t.arena_used, ldw, row0, in_dim, out_dim, transpose, parts});
      t.pack_fresh.push_back(0);
      t.pack_index[key] = idx;
      t.arena_used += bytes;
      t.max_frag_lanes = std::max(t.max_frag_lanes, ((out_dim + 31) / 32) * ((in_dim + 15) / 16) * 64);
    } else idx = it->second;
    pack_frags(st, W, ldw, row0, in_dim, out_dim, transpose, t.arena + t.packs[idx].dst, parts);
    t.pack_fresh[idx] = t.pack_epoch;
    return t.arena + t.packs[idx].dst;
  }
  // the hand-written weight-stationary layer (train_gemm.hip); false = shape not covered
  bool ws_layer(const std::vector<Seg>& segs, const float* W, int ldw, int row0, int in_dim, int out_dim, int transpose, const float* bias, float* y,
                int ldy, int64_t rows, bool relu, const float* mask_y, int ld_mask, int mask_div, bool accumulate, float* colsum = nullptr) {
    const int parts = precise_layers ? 3 : 2;
    if (segs.size() > 4 || frag_bytes(in_dim, out_dim, parts) > WPACK_BYTES) return false;
    DenseArgs A{};
    A.nseg = (int)segs.size();
    for (int i = 0; i < A.nseg; ++i) A.seg[i] = {segs[i].x, segs[i].ld, segs[i].K};
    A.k_total = in_dim; A.wfrag = t.wpack; A.bias = bias; A.y = y; A.ldy = ldy; A.n_out = out_dim; A.M = rows; A.relu = relu ? 1 : 0;
    A.mask_y = mask_y; A.ld_mask = ld_mask; A.mask_div = mask_div; A.accumulate = accumulate ? 1 : 0;
    A.zeros = static_cast<const char*>(t.wpack) + WPACK_BYTES; A.colsum = rep(colsum); A.rep_stride = t.P; A.nrep = nrep(); A.precise = precise_layers ? 1 : 0;
    if (!dense_ws_supported(A)) return false;
    if (const void* f = packed(W, ldw, row0, in_dim, out_dim, transpose, parts)) A.wfrag = f;
    else pack_frags(st, W, ldw, row0, in_dim, out_dim, transpose, t.wpack, parts);
    return dense_ws(st, A, t.num_cus);
  }
  void dense_fwd(const LayerP& L, const std::vector<Seg>& segs, float* y, int ldy, bool relu) {
    if (!ws_layer(segs, t.theta + L.w, L.N, 0, L.K, L.N, 0, t.theta + L.b, y, ldy, M, relu, nullptr, 0, 1, false))
      unsupported("dense forward", L.K, L.N, M);
  }
  // dy[M x N] (ldy) is d loss / d (post-activation output y); relu_y != nullptr -> mask with y > 0 first (needs ldy == N)
  // dW[K x N] += X[M x K]^T dY[M x N] (contraction over the sample axis): train_gemm.hip k_wgrad, one partial per workgroup
  // accumulated in registers and added to a gradient replica with float atomics.
  void weight_grad(const float* X, int ldx, int K, const float* dy, int ldy, int N, float* dW, int64_t rows = -1, bool x_half = false,
                   float* bias_grad = nullptr, bool dy_half = false) {
    const int64_t M = rows < 0 ? this->M : rows;
    WgradArgs A{X, ldx, K, dy, ldy, N, M, nullptr, rep(dW), static_cast<const char*>(t.wpack) + WPACK_BYTES, 0, 0, t.P, nrep()};
    A.x_half = x_half ? 1 : 0;
    A.dy_half = dy_half ? 1 : 0;
    A.out_scale = dy_half ? 1.f / t.g_scale : 1.f;
    A.colsum = rep(bias_grad);
    if (tan) {
      A.colsum = nullptr;
      // g scaled on the device; X carries tan_x_scale: a stored f16 tangent does, a raw fp32 tangent input is multiplied on its way into f16 (x_scale)
      if (dy_half) { A.out_scale = 1.f; A.out_scale_dev = tan_slot + 3; if (!x_half) A.x_scale = t.tan_x_scale; }
      else A.out_scale = x_half ? 1.f / t.tan_x_scale : 1.f;                                      // a head: fp32 cotangent, f16 tangent
    }
    if (!(wgrad_supported(A) && wgrad(wgrad_stream(), A, wgrad_grid(A, wgrad_cus())))) unsupported("weight gradient", K, N, M);
  }
  // fork(): the side streams wait for everything issued to st so far, and the weight-gradient launches that follow rotate over
  // st and the side streams; join(): st waits for the side streams.  Without side streams both are no-ops.
  int wg_turn = -1;
  bool wg_main = true;     // the caller's stream takes a turn too (false while it still has chains to launch)
  int ns = 0, ns_join = 0; // side streams in the rotation since the last fork / touched since the last join
  // CUs a weight-gradient launch on a SIDE stream may take: its workgroups are persistent, one per CU, and fill the CU's registers (k_wgrad_tr<8, 8>: two
  // 236-register waves per SIMD), so a kernel of the caller's stream - the step's critical path - waits for one of them to END (step_impl: wgrad_cu_cap)
  int wgrad_cus() const { return (wg_turn >= 0 && t.wgrad_cu_cap > 0 && t.wgrad_cu_cap < t.num_cus) ? t.wgrad_cu_cap : t.num_cus; }
  hipStream_t wgrad_stream() {
    if (wg_turn < 0) return st;
    const int k = wg_turn++ % (ns + (wg_main ? 1 : 0));
    return wg_main ? (k == 0 ? st : t.side[k - 1]) : t.side[k];
  }
  // (widening the LAST fork of a small-batch step to every side stream - its tail is a run of short launches - measured 2 % SLOWER at 512 rays, 2.64
  // against 2.59 ms, the full objective 5.50 against 5.29: fork / join events and contention cost more than the serial run)
  void fork(bool with_main = true) {
    if (!t.side[0]) return;
    ns = t.nside_eff;
    ns_join = ns > ns_join ? ns : ns_join;
    (void)hipEventRecord(t.fork_ev, st);
    for (int i = 0; i < ns; ++i) (void)hipStreamWaitEvent(t.side[i], t.fork_ev, 0);
    if (wg_turn < 0) wg_turn = 0;
    wg_main = with_main;
  }
  void join() {
    if (wg_turn < 0) return;
    for (int i = 0; i < ns_join; ++i) { (void)hipEventRecord(t.join_ev[i], t.side[i]); (void)hipStreamWaitEvent(st, t.join_ev[i], 0); }
    wg_turn = -1; ns_join = 0;
  }
  // Weight and bias gradients of an MLP whose data-gradient chain has run (fused backward): g[l] = d loss / d pre-activation of
  // layer l (fp32 [M x width]), h16[l] = its f16 output.  One launch per input segment; the bias gradient rides on the first.
  void mlp_wgrads(const MlpP& m, const float* in0, const std::vector<uint16_t*>& h16, const std::vector<float*>& g) {
    // the width x width parts of hidden layers 1 .. depth - 1 in ONE launch (WgradArgs::nl): their operands all exist once the chain has run, and
    // seven 128 x 128 products over 524 288 rows are 0.05 ms each - a launch apiece spends more on filling and draining the machine than on the rows
    static const bool multi_on = !(getenv("NERFDS_WGRAD_MULTI") && std::string(getenv("NERFDS_WGRAD_MULTI")) == "0");
    bool batched = false;
    if (multi_on && t.g16 && m.depth >= 3 && m.depth <= 9) {
      WgradArgs A{reinterpret_cast<const float*>(h16[0]), m.width, m.width, g[1], m.width, m.width, M, nullptr, rep(t.grad + m.hidden[1].w),
                  static_cast<const char*>(t.wpack) + WPACK_BYTES, 0, 0, t.P, nrep()};
      A.x_half = 1; A.dy_half = 1; A.out_scale = 1.f / t.g_scale; A.colsum = rep(t.grad + m.hidden[1].b);
      if (tan) { A.out_scale = 1.f; A.out_scale_dev = tan_slot + 3; A.colsum = nullptr; }
      A.nl = m.depth - 1;
      for (int l = 1; l < m.depth; ++l) {
        A.mx[l - 1] = h16[l - 1]; A.mdy[l - 1] = g[l]; A.mdw[l - 1] = rep(t.grad + m.hidden[l].w); A.mcs[l - 1] = tan ? nullptr : rep(t.grad + m.hidden[l].b);
      }
      if (wgrad_supported(A) && wgrad_multi_supported(A)) {
        if (!wgrad(wgrad_stream(), A, wgrad_grid(A, wgrad_cus()))) unsupported("weight gradient (layers batched)", m.width, m.width, M);
        batched = true;
      }
    }
    for (int l = 0; l < m.depth; ++l) {
      const LayerP& L = m.hidden[l];
      int k0 = 0;
      bool first = true;
      if (l > 0) {
        if (!batched)
          weight_grad(reinterpret_cast<const float*>(h16[l - 1]), m.width, m.width, g[l], m.width, m.width, t.grad + L.w, -1, true, t.grad + L.b, t.g16);
        k0 = m.width;
        first = false;
      }
      if (l == 0 || l == m.skip)
        weight_grad(in0, m.in_ld, m.in_dim, g[l], m.width, m.width, t.grad + L.w + (int64_t)k0 * L.N, -1, false, first ? t.grad + L.b : nullptr, t.g16);
    }
  }
  // a head on the last hidden layer: dW = h^T d_head, db = column sums of d_head (N <= 6: the scalar-DMA shapes of k_wgrad)
  void head_wgrads(const LayerP& L, const uint16_t* h16, int width, const float* d_head, int ld) {
    weight_grad(reinterpret_cast<const float*>(h16), width, width, d_head, ld, L.N, t.grad + L.w, -1, true, t.grad + L.b);
  }
  // premasked: the kernel that produced dy already applied this layer's ReLU mask and added the bias gradient.
  // Returns true when the dx of the segment that asked for it (Seg::dx_relu_y) was masked / column-summed by the MFMA layer.
  bool dense_bwd(const LayerP& L, const std::vector<Seg>& segs, float* dy, int ldy, const float* relu_y, bool premasked = false) {
    if (!premasked) {
      if (relu_y && L.N <= 256) relu_bwd_colsum(st, dy, relu_y, M, L.N, t.grad + L.b);      // mask with y > 0, db[N] += dY^T 1
      else {
        if (relu_y) relu_bwd(st, dy, relu_y, M * L.N);
        colsum_add(st, dy, M, L.N, ldy, t.grad + L.b);
      }
    }
    bool fused = false;
    int k0 = 0;
    for (const Seg& s : segs) {
      // narrow hidden layer whose input is the ReLU output that also masks its gradient: dW, dX, mask and column sums in ONE pass
      if (t.fuse_bwd && s.dx && !s.acc && s.dx_relu_y == s.x && s.dld == s.K && s.ld == s.K) {
        BwdFusedArgs F{s.x, s.ld, s.K, dy, ldy, L.N, t.wpack, s.dx, s.dld, rep(s.dx_bias_grad), rep(t.grad + L.w + (int64_t)k0 * L.N), M,
                       static_cast<const char*>(t.wpack) + WPACK_BYTES, t.P, nrep()};
        if (bwd_fused_supported(F)) {
          if (const void* f = packed(t.theta + L.w, L.N, k0, L.N, s.K, 1, 2)) F.wfrag = f;
          else pack_frags(st, t.theta + L.w, L.N, k0, L.N, s.K, 1, t.wpack);
          if (bwd_fused(st, F, t.num_cus)) { fused = true; k0 += s.K; continue; }
        }
      }
      weight_grad(s.x, s.ld, s.K, dy, ldy, L.N, t.grad + L.w + (int64_t)k0 * L.N);
      if (s.dx) {
        const bool want = s.dx_relu_y != nullptr;          // (with acc: this is the last contribution to dx, the mask covers the total)
        if (ws_layer({{dy, ldy, L.N, nullptr, 0, false}}, t.theta + L.w, L.N, k0, L.N, s.K, 1, nullptr, s.dx, s.dld, M, false, want ? s.dx_relu_y : nullptr,
                     s.dld, 1, s.acc, want ? s.dx_bias_grad : nullptr)) fused = fused || want;
        else unsupported("data gradient", L.N, s.K, M);
      }
      k0 += s.K;
    }
    return fused;
  }
  // tangents (3 rows per sample, no bias): y[3M x N] = sum_s t_s W[rows of s]
  // relu_y != nullptr: the rows are masked with the PRIMAL activation of their sample (y_t = 0 where relu_y[row / 3] <= 0)
  void dense_jvp(const LayerP& L, const std::vector<Seg>& segs, float* y, int ldy, const float* relu_y = nullptr) {
    if (!ws_layer(segs, t.theta + L.w, L.N, 0, L.K, L.N, 0, nullptr, y, ldy, 3 * M, false, relu_y, L.N, 3, false))
      unsupported("tangent forward", L.K, L.N, 3 * M);
  }
  // returns the tangent of the last hidden layer (in cur or other)
  float* mlp_jvp(const MlpP& m, const float* t_in0, const std::vector<float*>& h, float* cur, float* other,
                 const std::vector<float*>* store = nullptr) {
    for (int l = 0; l < m.depth; ++l) {
      std::vector<Seg> segs;
      if (l > 0) segs.push_back({cur, m.width, m.width, nullptr, 0, false});
      if (l == 0 || l == m.skip) segs.push_back({t_in0, m.in_ld, m.in_dim, nullptr, 0, false});
      float* dst = store ? (*store)[l] : other;
      dense_jvp(m.hidden[l], segs, dst, m.width, h[l]);
      if (store) cur = dst; else std::swap(cur, other);
    }
    return cur;
  }
  // backward of a tangent layer: dy[3M x N] is d loss / d tangent output (masked with the PRIMAL activation when relu_y != nullptr)
  // premasked / return value: as dense_bwd (no bias gradient here: tangent layers have no bias)
  bool dense_jvp_bwd(const LayerP& L, const std::vector<Seg>& segs, float* dy, int ldy, const float* relu_y, bool premasked = false) {
    if (relu_y && !premasked) relu_mask3(st, dy, relu_y, M, L.N);
    bool fused = false;
    int k0 = 0;
    for (const Seg& s : segs) {
      weight_grad(s.x, s.ld, s.K, dy, ldy, L.N, t.grad + L.w + (int64_t)k0 * L.N, 3 * M);
      if (s.dx) {
        const bool want = s.dx_relu_y != nullptr;
        if (ws_layer({{dy, ldy, L.N, nullptr, 0, false}}, t.theta + L.w, L.N, k0, L.N, s.K, 1, nullptr, s.dx, s.dld, 3 * M, false,
                     want ? s.dx_relu_y : nullptr, s.dld, 3, s.acc)) fused = fused || want;
        else unsupported("tangent data gradient", L.N, s.K, 3 * M);
      }
      k0 += s.K;
    }
    return fused;
  }
  // cur holds d loss / d tangent of h[depth-1]; th = the stored tangents of every layer
  void mlp_jvp_bwd(const MlpP& m, const float* t_in0, const std::vector<float*>& th, const std::vector<float*>& h, float* cur, float* other,
                   float* d_t_in0, bool premasked = false) {
    bool in0_written = false;
    for (int l = m.depth - 1; l >= 0; --l) {
      std::vector<Seg> segs;
      if (l > 0) segs.push_back({th[l - 1], m.width, m.width, other, m.width, false, h[l - 1], nullptr});
      if (l == 0 || l == m.skip) { segs.push_back({t_in0, m.in_ld, m.in_dim, d_t_in0, m.in_ld, in0_written}); in0_written = d_t_in0 != nullptr; }
      premasked = dense_jvp_bwd(m.hidden[l], segs, cur, m.width, h[l], premasked);
      std::swap(cur, other);
    }
  }
  void mlp_fwd(const MlpP& m, const float* in0, const std::vector<float*>& h) {
    for (int l = 0; l < m.depth; ++l) {
      std::vector<Seg> segs;
      if (l > 0) segs.push_back({h[l - 1], m.width, m.width, nullptr, 0, false});
      if (l == 0 || l == m.skip) segs.push_back({in0, m.in_ld, m.in_dim, nullptr, 0, false});
      dense_fwd(m.hidden[l], segs, h[l], m.width, true);
    }
  }
  // cur holds d loss / d h[depth-1] on entry; other is a second [M x width] buffer; d_in0 receives d loss / d in0
  void mlp_bwd(const MlpP& m, const float* in0, const std::vector<float*>& h, float* cur, float* other, float* d_in0, bool premasked = false) {
    bool in0_written = false;
    for (int l = m.depth - 1; l >= 0; --l) {
      std::vector<Seg> segs;
      // d h[l-1] is the dY of layer l-1: its ReLU mask and bias gradient ride on the epilogue of the kernel that writes it
      if (l > 0) segs.push_back({h[l - 1], m.width, m.width, other, m.width, false, h[l - 1], t.grad + m.hidden[l - 1].b});
      if (l == 0 || l == m.skip) { segs.push_back({in0, m.in_ld, m.in_dim, d_in0, m.in_ld, in0_written}); in0_written = true; }
      premasked = dense_bwd(m.hidden[l], segs, cur, m.width, h[l], premasked);
      std::swap(cur, other);
    }
  }
};

void carve(nerfds_trainer& t) {
  const int64_t R = t.max_rays, Nc = t.cfg.num_coarse_samples, Nf = t.cfg.num_fine_samples, S = Nc + Nf, M = R * S;
  size_t need = 0;
  std::vector<std::pair<float**, size_t>> views;
  auto take = [&](float** p, size_t n) { views.push_back({p, n}); need += (n + 63) & ~(size_t)63; };
  const Dims& D = t.D;
  take(&t.zc, R * Nc); take(&t.zf, R * S); take(&t.wc, R * Nc); take(&t.rs_scratch, R * (2 * Nc + Nf));
  take(&t.x, M * 3); take(&t.mask_in, M * D.mask_in); take(&t.mask_logit, M);
  t.mask_h.assign(t.mask.depth, nullptr); for (auto& p : t.mask_h) take(&p, M * t.mask.width);
  take(&t.warp_in, M * D.warp_ld); t.warp_h.assign(t.warp.depth, nullptr); for (auto& p : t.warp_h) take(&p, M * t.warp.width);
  take(&t.wv, M * 6); take(&t.xw, M * 3);
  take(&t.hyper_in, M * D.hyper_ld); t.hyper_h.assign(t.hyper.depth, nullptr); for (auto& p : t.hyper_h) take(&p, M * t.hyper.width);
  take(&t.wamb, M * 2); take(&t.trunk_in, M * D.trunk_in);
  t.trunk_h.assign(t.trunk[0].depth, nullptr); for (auto& p : t.trunk_h) take(&p, M * t.trunk[0].width);
  take(&t.bottv, M * t.trunk[0].width); take(&t.alphav, M * 4); take(&t.sigma, M); take(&t.cond, M * (6 * D.vd_bands + 6 * D.nm_bands));
  take(&t.rgb_hv, M * t.rgb_h[0].N); take(&t.rgb_logit, M * 3); take(&t.weights, M); take(&t.rgb_ray, R * 3);
  take(&t.g0, M * t.trunk[0].width); take(&t.g1, M * t.trunk[0].width); take(&t.g2, M * t.trunk[0].width);
  take(&t.d_trunk_in, M * D.trunk_in); take(&t.d_rgb_logit, M * 3); take(&t.d_alpha, M * 4); take(&t.dxw, M * 3); take(&t.dwamb, M * 2);
  take(&t.dwv, M * 6); take(&t.d_warp_in, M * D.warp_ld); take(&t.d_hyper_in, M * D.hyper_ld); take(&t.d_mask_in, M * D.mask_in);
  take(&t.d_mask_logit, M); take(&t.dxw_reg, M * 3); take(&t.d_pm, M); take(&t.dwamb_reg, M * 2);
  t.ws_floats = need;
  // (pointers into vectors: the vectors are not resized after this point)
  float* base = t.ws;
  if (!base) return;
  for (auto& v : views) { *v.first = base; base += (v.second + 63) & ~(size_t)63; }
}

bool ensure_tangent_ws(nerfds_trainer& t) {
  if (t.tws) return true;
  const int64_t R = t.max_rays, Nc = t.cfg.num_coarse_samples, S = Nc + t.cfg.num_fine_samples, M3 = 3 * R * S;
  const Dims& D = t.D;
  const int TW = t.trunk[0].width;
  size_t need = 0;
  std::vector<std::pair<float**, size_t>> views;
  auto take = [&](float** p, size_t n) { views.push_back({p, n}); need += (n + 63) & ~(size_t)63; };
  take(&t.t_warp_in, M3 * D.warp_ld); take(&t.t_hyper_in, M3 * D.hyper_ld);
  if (!t.fused_tan) { take(&t.tA, M3 * TW); take(&t.tB, M3 * TW); }      // ping-pong rows of the layer-by-layer tangent pass
  take(&t.t_wv, M3 * 6); take(&t.t_xw, M3 * 3); take(&t.t_wamb, M3 * 2); take(&t.t_tin, M3 * D.trunk_in); take(&t.t_alpha, M3 * 4);
  take(&t.tn[0], R * Nc * 3); take(&t.tn[1], R * S * 3);
  if (hipMalloc(&t.tws, need * sizeof(float)) != hipSuccess) return false;
  float* base = t.tws;
  for (auto& v : views) { *v.first = base; base += (v.second + 63) & ~(size_t)63; }
  return true;
}

bool ensure_norm_ws(nerfds_trainer& t) {
  if (t.nws) return true;
  const int64_t R = t.max_rays, S = t.cfg.num_coarse_samples + t.cfg.num_fine_samples, M = R * S, M3 = 3 * M;
  const Dims& D = t.D;
  size_t need = 0;
  std::vector<std::pair<float**, size_t>> views;
  auto take = [&](float** p, size_t n) { views.push_back({p, n}); need += (n + 63) & ~(size_t)63; };
  if (!t.fused_tan) {      // (the fused tangent pass keeps the hidden tangents as f16: ensure_tan16)
    t.tw_h.assign(t.warp.depth, nullptr); for (auto& p : t.tw_h) take(&p, M3 * t.warp.width);
    t.th_h.assign(t.hyper.depth, nullptr); for (auto& p : t.th_h) take(&p, M3 * t.hyper.width);
    t.tt_h.assign(t.trunk[0].depth, nullptr); for (auto& p : t.tt_h) take(&p, M3 * t.trunk[0].width);
  }
  take(&t.d_t_alpha, M3 * 4); take(&t.d_t_tin, M3 * D.trunk_in); take(&t.d_t_xw, M3 * 3); take(&t.d_t_wamb, M3 * 2); take(&t.d_t_wv, M3 * 6);
  take(&t.du, M * 3); take(&t.ghat, M * 3); take(&t.dwamb_extra, M * 2); take(&t.dwv_extra, M * 6);
  if (hipMalloc(&t.nws, need * sizeof(float)) != hipSuccess) return false;
  float* base = t.nws;
  for (auto& v : views) { *v.first = base; base += (v.second + 63) & ~(size_t)63; }
  return true;
}

// The layer-by-layer tangent pass on fp32 rows (sigma_gradient / run_level below, !half_step) when the trainer was built with the fused tangent chains:
// ensure_tangent_ws / ensure_norm_ws then leave out its ping-pong rows tA / tB and the fp32 hidden tangents tw_h / th_h / tt_h - a step that keeps fp32
// activations (nerfds_train_numerics.fp32_step, NERFDS_TRAIN_HALF_TANGENTS=0) allocates them here, on first use (`keep`: the hidden tangents too).
bool ensure_tangent_ws32(nerfds_trainer& t, bool keep) {
  if (!t.fused_tan) return true;                       // the two functions above allocated them
  if (t.tws32 && (!keep || !t.tt_h.empty())) return true;
  if (t.tws32) { (void)hipDeviceSynchronize(); (void)hipFree(t.tws32); t.tws32 = nullptr; t.tA = t.tB = nullptr; t.tw_h.clear(); t.th_h.clear(); t.tt_h.clear(); }
  const int64_t M3 = 3 * t.max_rays * (t.cfg.num_coarse_samples + t.cfg.num_fine_samples);
  const int TW = t.trunk[0].width;
  size_t need = 0;
  std::vector<std::pair<float**, size_t>> views;
  auto take = [&](float** p, size_t n) { views.push_back({p, n}); need += (n + 63) & ~(size_t)63; };
  take(&t.tA, M3 * TW); take(&t.tB, M3 * TW);
  if (keep) {
    t.tw_h.assign(t.warp.depth, nullptr); for (auto& p : t.tw_h) take(&p, M3 * t.warp.width);
    t.th_h.assign(t.hyper.depth, nullptr); for (auto& p : t.th_h) take(&p, M3 * t.hyper.width);
    t.tt_h.assign(t.trunk[0].depth, nullptr); for (auto& p : t.tt_h) take(&p, M3 * t.trunk[0].width);
  }
  if (hipMalloc(&t.tws32, need * sizeof(float)) != hipSuccess) { t.tA = t.tB = nullptr; t.tw_h.clear(); t.th_h.clear(); t.tt_h.clear(); return false; }
  float* base = t.tws32;
  for (auto& v : views) { *v.first = base; base += (v.second + 63) & ~(size_t)63; }
  return true;
}

void fused_tangent(nerfds_trainer& t, hipStream_t st, int net, int level, int64_t M3, const float* t_in, int ld_in, float* t_head, int ld_head, int mask_div = 3,
                   bool f16_ok = false);
void fused_reverse(nerfds_trainer& t, hipStream_t st, int net, int level, int64_t M, const float* d_head, int ld_head, float* d_in, int ld_in);

// d sigma_raw / d x by forward-mode tangents through warp MLP -> exp_se3, hyper sheet, posenc, trunk, alpha head
// (the mask is a constant input, models.py:1035-1069), then target_norm (models.py:1077, 1273-1277, 1328).  Uses the
// activations of the forward pass that has just run for this level.
void sigma_gradient(nerfds_trainer& t, Run& r, int level, const Windows& W) {
  const Dims& D = t.D;
  hipStream_t st = r.st;
  const int64_t M = r.M;
  static const bool rev_on = !(getenv("NERFDS_TRAIN_REVERSE_SIGMA") && std::string(getenv("NERFDS_TRAIN_REVERSE_SIGMA")) == "0");
  if (t.fused_tan && t.half_step && rev_on && !t.keep_tangents && !t.tangents_warp_only && t.g1) {
    // target_norm alone (NERFDS_TRAIN_SIGMA_GRAD without a term that differentiates it: NerfModel.apply(use_sigma_gradient=True)): d sigma_raw / d x by
    // ONE reverse pass per sample - the networks' data-gradient chains with the cotangent e_sigma, as jax.grad does it (models.py:1035-1069) - instead
    // of three forward-mode rows (run_merged_full `rev` (A)); scratch in the layer-by-layer backward's buffers, the chains' f16 g into the tangent arrays
    float* p = t.g1;
    auto take = [&](int64_t n) { float* q = p; p += (n + 63) & ~(int64_t)63; return q; };
    float* cot = take(4 * M); float* sa = take(3 * M); float* sb = take(2 * M); float* sc = take(6 * M); float* gx = take(3 * M);
    fill_head4(st, M, nullptr, 1.f, cot);
    fused_reverse(t, st, 4, level, M, cot, 4, t.t_tin, D.trunk_in);
    trunk_in_bwd(st, D, M, t.t_tin, t.xw, t.wamb, W, nullptr, nullptr, sa, sb);
    se3_bwd(st, M, t.wv, t.x, sa, nullptr, sc);
    fused_reverse(t, st, 1, 0, M, sb, 2, t.d_hyper_in, D.hyper_ld);
    fused_reverse(t, st, 2, 0, M, sc, 6, t.d_warp_in, D.warp_ld);
    posenc_rev_x(st, D, M, t.x, t.d_warp_in, t.d_hyper_in, W, gx);
    sigma_grad_assemble(st, M, t.wv, sa, gx, t.t_alpha);
    target_norm(st, M, t.t_alpha, t.wv, t.tn[level]);
    return;
  }
  encode_tangents(st, D, M, t.x, W, t.t_warp_in, t.t_hyper_in);
  if (t.fused_tan && t.half_step) {
    // ONE launch per network (train_bwd_kernel.hip train_tangent_kernel): the three tangents of a sample as three rows of a masked linear chain in
    // split bf16, masks = the primal layers' stored ReLU bits, hidden tangents to HBM once as scaled f16 (what the tangent pass's weight gradients
    // read), between the networks the element-wise tangent kernels as before.  (Through round 4: 42 + 30 layer launches per level on fp32 rows.)
    fused_tangent(t, st, 2, level, 3 * M, t.t_warp_in, D.warp_ld, t.t_wv, 6);
    se3_jvp(st, M, t.wv, t.x, t.t_wv, t.t_xw);
    if (t.tangents_warp_only) return;
    fused_tangent(t, st, 1, level, 3 * M, t.t_hyper_in, D.hyper_ld, t.t_wamb, 2);
    trunk_in_jvp(st, D, M, t.xw, t.wamb, t.t_xw, t.t_wamb, W, t.t_tin);
    fused_tangent(t, st, 4, level, 3 * M, t.t_tin, D.trunk_in, t.t_alpha, 4);
    target_norm(st, M, t.t_alpha, t.wv, t.tn[level]);
    return;
  }
  const bool keep = t.nws != nullptr && t.keep_tangents;
  r.precise_layers = true;       // as the primal warp field: its tangent goes through the 2^7-frequency posenc
  float* tw = r.mlp_jvp(t.warp, t.t_warp_in, t.warp_h, t.tA, t.tB, keep ? &t.tw_h : nullptr);
  r.dense_jvp(t.warp_w, {{tw, t.warp.width, t.warp.width, nullptr, 0, false}}, t.t_wv, 6);
  r.dense_jvp(t.warp_v, {{tw, t.warp.width, t.warp.width, nullptr, 0, false}}, t.t_wv + 3, 6);
  r.precise_layers = false;
  se3_jvp(st, M, t.wv, t.x, t.t_wv, t.t_xw);
  if (t.tangents_warp_only) return;        // the elastic regulariser reads d x' / d x and nothing behind it
  float* th = r.mlp_jvp(t.hyper, t.t_hyper_in, t.hyper_h, t.tA, t.tB, keep ? &t.th_h : nullptr);
  r.dense_jvp(t.hyper_out, {{th, t.hyper.width, t.hyper.width, nullptr, 0, false}}, t.t_wamb, 2);
  trunk_in_jvp(st, D, M, t.xw, t.wamb, t.t_xw, t.t_wamb, W, t.t_tin);
  const MlpP& trunk = t.trunk[level];
  float* tt = r.mlp_jvp(trunk, t.t_tin, t.trunk_h, t.tA, t.tB, keep ? &t.tt_h : nullptr);
  r.dense_jvp(t.alpha[level], {{tt, trunk.width, trunk.width, nullptr, 0, false}}, t.t_alpha, 4);
  target_norm(st, M, t.t_alpha, t.wv, t.tn[level]);
}

// ---- fused forward: index maps, built once ----
// The render packer (pack.h) is run over a fake parameter vector whose values are their own indices + 1, in the uniform fp32
// plan (raw floats in the stream, same fragment positions as TRAIN_PLAN: both are two units per fragment): what comes out is,
// for every float slot of the stream, WHICH parameter belongs there.  k_pack_stream then gathers the current values every step.
bool build_fused_forward(nerfds_trainer& t) {
  using G = nerfds::GraphNerfDS;
  using Dm = nerfds::Dims<G>;
  const nerfds_model_cfg& c = t.cfg;
  const bool same_graph = c.mask_max_deg == G::MASK_BANDS && c.mask_depth == G::MASK_DEPTH && c.mask_width == G::MASK_W && c.mask_skip == G::MASK_SKIP &&
                          c.warp_max_deg == G::WARP_BANDS && c.warp_trunk_depth == G::WARP_DEPTH && c.warp_trunk_width == G::WARP_W && c.warp_skip == G::WARP_SKIP &&
                          c.hyper_sheet_max_deg == G::HYP_BANDS && c.hyper_sheet_depth == G::HYP_DEPTH && c.hyper_sheet_width == G::HYP_W &&
                          c.hyper_sheet_skip == G::HYP_SKIP && c.spatial_point_max_deg == G::SP_BANDS && c.hyper_point_max_deg == G::HP_BANDS &&
                          c.viewdir_max_deg == G::VD_BANDS && c.norm_input_max_deg == G::NM_BANDS && c.nerf_trunk_depth == G::TRUNK_DEPTH &&
                          c.nerf_trunk_width == G::TRUNK_W && c.nerf_skip == G::TRUNK_SKIP && c.nerf_rgb_branch_width == G::RGB_W &&
                          c.num_coarse_samples + c.num_fine_samples <= nerfds::MAX_SAMPLES;
  if (!same_graph) return false;       // other widths: the layer-by-layer forward
  constexpr nerfds::Plan F32 = nerfds::uniform_plan(nerfds::P_F32);
  static_assert(nerfds::nerf_units<G>(F32) == nerfds::nerf_units<G>(nerfds::TRAIN_PLAN) && nerfds::TRAIN_PLAN.mask == nerfds::P_BF16X3 &&
                (nerfds::TRAIN_PLAN.warp == nerfds::P_BF16X6 || nerfds::TRAIN_PLAN.warp == nerfds::P_F32) && nerfds::TRAIN_PLAN.hyp == nerfds::P_BF16X3,
                "index maps are packed in the two-unit fp32 layout; k_pack_stream keeps the warp field's fragments fp32 or widens them to three units");
  constexpr int TW = G::TRUNK_W, RW = G::RGB_W, VD = Dm::VD_FEATS, NM = Dm::NM_FEATS, FL = (TW + 1) * RW;
  const int levels = c.num_fine_samples > 0 ? 2 : 1;
  if (t.P + 2 * FL + 1 >= (1 << 24)) return false;     // indices travel through the packer as floats
  std::vector<float> idx((size_t)t.P);
  for (int64_t i = 0; i < t.P; ++i) idx[(size_t)i] = (float)(i + 1);
  auto view = [&](const LayerP& L) { nerfds::DenseView d; d.kernel = idx.data() + L.w; d.bias = idx.data() + L.b; d.in_dim = L.K; d.out_dim = L.N; return d; };
  nerfds::SharedNets sn;
  for (int l = 0; l < t.mask.depth; ++l) sn.mask_hidden[l] = view(t.mask.hidden[l]);
  for (int l = 0; l < t.warp.depth; ++l) sn.warp_hidden[l] = view(t.warp.hidden[l]);
  for (int l = 0; l < t.hyper.depth; ++l) sn.hyper_hidden[l] = view(t.hyper.hidden[l]);
  sn.mask_out = view(t.mask_out); sn.warp_w = view(t.warp_w); sn.warp_v = view(t.warp_v); sn.hyper_out = view(t.hyper_out);
  // fragments of the warp field keep fp32 (TRAIN_PLAN.warp): they follow the mask net in the shared stream
  const int mask_units = nerfds::walk_mlp(0, G::MASK_DEPTH, G::MASK_W, Dm::MASK_KC, G::MASK_SKIP, true, nerfds::P_F32);
  const int warp_units = nerfds::walk_mlp(mask_units, G::WARP_DEPTH, G::WARP_W, Dm::WARP_KC, G::WARP_SKIP, true, nerfds::P_F32);
  t.f32_lo = mask_units / 2; t.f32_hi = warp_units / 2;
  if (hipMalloc(&t.fold, (size_t)2 * FL * sizeof(float)) != hipSuccess) return false;
  for (int which = 0; which < 1 + levels; ++which) {
    const int64_t wb = (int64_t)nerfds::pad_units(which == 0 ? nerfds::shared_units<G>(F32) : nerfds::nerf_units<G>(F32)) * 1024;
    const int64_t bf = (int64_t)(which == 0 ? Dm::SHARED_BIAS_TILES : Dm::NERF_BIAS_TILES) * 32;
    std::vector<uint8_t> w((size_t)wb, 0);
    std::vector<float> b((size_t)bf, 0.f);
    nerfds::StreamWriter sw{w.data(), b.data()};
    std::vector<float> pf, pfb;
    if (which == 0) {
      nerfds::pack_shared<G>(sw, sn, F32);
    } else {
      const int lv = which - 1;
      nerfds::NerfNet nn;
      for (int l = 0; l < t.trunk[lv].depth; ++l) nn.trunk[l] = view(t.trunk[lv].hidden[l]);
      nn.bottleneck = view(t.bott[lv]); nn.alpha = view(t.alpha[lv]); nn.rgb_hidden[0] = view(t.rgb_h[lv]); nn.rgb = view(t.rgb_out[lv]);
      // rgb hidden_0 in the kernel's K order [trunk_output (folded rows, in `fold`) | viewdir | normal (rows of the parameter itself)]
      const LayerP& K = t.rgb_h[lv];
      const int row_vd = TW, row_nm = TW + VD + TW;
      pf.resize((size_t)(TW + VD + NM) * RW);
      pfb.resize(RW);
      for (int cc = 0; cc < RW; ++cc) {
        for (int r = 0; r < TW; ++r) pf[(size_t)r * RW + cc] = (float)(t.P + (int64_t)lv * FL + (int64_t)r * RW + cc + 1);
        for (int q = 0; q < VD; ++q) pf[(size_t)(TW + q) * RW + cc] = (float)(K.w + (int64_t)(row_vd + q) * RW + cc + 1);
        for (int q = 0; q < NM; ++q) pf[(size_t)(TW + VD + q) * RW + cc] = (float)(K.w + (int64_t)(row_nm + q) * RW + cc + 1);
        pfb[cc] = (float)(t.P + (int64_t)lv * FL + (int64_t)TW * RW + cc + 1);
      }
      nn.prefolded = pf.data(); nn.prefolded_bias = pfb.data();
      nerfds::pack_nerf<G>(sw, nn, F32);
    }
    if ((int64_t)sw.wbytes > wb || (int64_t)sw.bfloats != bf) return false;
    std::vector<int> map((size_t)wb / 4), bmap((size_t)bf);
    const float* wf = reinterpret_cast<const float*>(w.data());
    for (size_t i = 0; i < map.size(); ++i) map[i] = (int)wf[i];
    for (size_t i = 0; i < bmap.size(); ++i) bmap[i] = (int)b[i];
    // the stream itself is in TRAIN_PLAN's layout (the warp field's fragments take three units)
    const int64_t sb = (int64_t)nerfds::pad_units(which == 0 ? nerfds::shared_units<G>(nerfds::TRAIN_PLAN) : nerfds::nerf_units<G>(nerfds::TRAIN_PLAN)) * 1024;
    if (hipMalloc(&t.fmap[which], map.size() * 4) != hipSuccess || hipMalloc(&t.fbmap[which], bmap.size() * 4) != hipSuccess ||
        hipMalloc(&t.fstream[which], (size_t)sb) != hipSuccess || hipMemset(t.fstream[which], 0, (size_t)sb) != hipSuccess ||
        hipMalloc(&t.fbias[which], bmap.size() * 4) != hipSuccess ||
        hipMemcpy(t.fmap[which], map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(t.fbmap[which], bmap.data(), bmap.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return false;
    t.fstream_frags[which] = (which == 0 ? nerfds::shared_units<G>(F32) : nerfds::nerf_units<G>(F32)) / 2;      // real fragments (the padding of the stream stays zero)
    t.fbias_n[which] = (int)bf;
  }
  return true;
}

// every step: fold (launched), then the three streams and their biases from the current theta (collected in pb: ONE launch for the step's whole packing, step_impl)
void pack_fused_forward(nerfds_trainer& t, hipStream_t st, PackBatch& pb) {
  using G = nerfds::GraphNerfDS;
  using Dm = nerfds::Dims<G>;
  constexpr int TW = G::TRUNK_W, RW = G::RGB_W, FL = (TW + 1) * RW;
  const int levels = t.cfg.num_fine_samples > 0 ? 2 : 1;
  for (int lv = 0; lv < levels; ++lv)
    fold_rgb(st, t.theta + t.bott[lv].w, t.theta + t.bott[lv].b, t.theta + t.rgb_h[lv].w, t.theta + t.rgb_h[lv].b, TW, RW, TW + Dm::VD_FEATS,
             t.fold + (size_t)lv * FL);
  for (int which = 0; which < 1 + levels; ++which) {
    pb.stream(t.fmap[which], t.fstream[which], t.fstream_frags[which], which == 0 ? t.f32_lo : 0, which == 0 ? t.f32_hi : 0,
              nerfds::TRAIN_PLAN.warp == nerfds::P_F32 ? 1 : 0);
    pb.bias(t.fbmap[which], t.fbias[which], t.fbias_n[which]);
  }
}

// ---- fused backward: index maps of the TRANSPOSED layers in chain order (graphs.h BwdNet), built once like the forward's ----
// which: 0 / 1 NerfMLP coarse / fine, 2 hyper sheet, 3 warp field, 4 mask net.  Values are parameter indices + 1 (fold rows past P).
bool build_fused_backward(nerfds_trainer& t) {
  using G = nerfds::GraphNerfDS;
  using Dm = nerfds::Dims<G>;
  constexpr int TW = G::TRUNK_W, RW = G::RGB_W, FL = (TW + 1) * RW;
  const int levels = t.cfg.num_fine_samples > 0 ? 2 : 1;
  auto idx = [](int64_t i) { return (float)(i + 1); };
  auto layer = [](std::function<float(int, int)> W, std::vector<nerfds::Seg> segs, int n_tiles, int n_out) {
    nerfds::Layer L;
    L.segs = std::move(segs); L.n_tiles = n_tiles; L.n_out = n_out; L.is_head = false; L.W = std::move(W); L.B = [](int) { return 0.f; };
    return L;
  };
  auto tile_seg = [](int width) { std::vector<int> r; nerfds::rows_tile(r, width, 0); return nerfds::Seg{std::move(r), nerfds::P_F32}; };
  auto lin_seg = [](int row0, int n) { std::vector<int> r; nerfds::rows_linear(r, 1, [&](int sl) { return sl < n ? row0 + sl : -1; }); return nerfds::Seg{std::move(r), nerfds::P_F32}; };
  // hidden layers D-1 .. 0 of an MLP, transposed (the head or whatever produces g_{D-1} has been emitted by the caller)
  auto emit_trunk = [&](nerfds::StreamWriter& sw, const MlpP& m) {
    const int W = m.width;
    for (int l = m.depth - 1; l >= 1; --l) {
      const int64_t w = m.hidden[l].w;
      sw.emit(layer([=](int n, int k) { return idx(w + (int64_t)k * W + n); }, {tile_seg(W)}, W / 32, W));                    // d h_{l-1} = W_l[:W] g_l
      if (l == m.skip) sw.emit(layer([=](int n, int c) { return idx(w + (int64_t)(W + c) * W + n); }, {tile_seg(W)}, 2, m.in_dim));   // raw-input rows of the skip layer
    }
    const int64_t w0 = m.hidden[0].w;
    sw.emit(layer([=](int n, int c) { return idx(w0 + (int64_t)c * W + n); }, {tile_seg(W)}, 2, m.in_dim));
  };
  for (int which = 0; which < 5; ++which) {
    if (which == 1 && levels < 2) continue;
    int frags = 0;
    const MlpP* m = nullptr;
    if (which <= 1) { frags = nerfds::BwdNerf<G>::BWD_FRAGS; m = &t.trunk[which]; }
    else if (which == 2) { frags = nerfds::BwdHyper<G>::BWD_FRAGS; m = &t.hyper; }
    else if (which == 3) { frags = nerfds::BwdWarp<G>::BWD_FRAGS; m = &t.warp; }
    else { frags = nerfds::BwdMask<G>::BWD_FRAGS; m = &t.mask; }
    if (m->skip != 4 || (m->depth != 8 && m->depth != 6) || m->in_dim > 64) return false;
    const int64_t wb = (int64_t)nerfds::pad_units(2 * frags) * 1024;
    std::vector<uint8_t> w((size_t)wb, 0);
    nerfds::StreamWriter sw{w.data(), nullptr};
    const int W = m->width;
    if (which <= 1) {
      const int lv = which;
      const int64_t ro = t.rgb_out[lv].w, al = t.alpha[lv].w, fo = t.P + (int64_t)lv * FL;
      sw.emit(layer([=](int j, int k) { return idx(ro + (int64_t)k * 3 + j); }, {lin_seg(0, 3)}, RW / 32, RW));                 // g_rgb <- W_rgb d rgb_logit
      sw.emit(layer([=](int r, int k) { return r < RW ? idx(fo + (int64_t)k * RW + r) : idx(al + (int64_t)k * 4 + (r - RW)); },
                    {tile_seg(RW), lin_seg(RW, 4)}, TW / 32, TW));                                                               // g_7 <- F^T g_rgb + W_alpha d alpha
    } else if (which == 2) {
      const int64_t ho = t.hyper_out.w;
      sw.emit(layer([=](int j, int k) { return idx(ho + (int64_t)k * 2 + j); }, {lin_seg(0, 2)}, W / 32, W));
    } else if (which == 3) {
      const int64_t ww = t.warp_w.w, wv = t.warp_v.w;
      sw.emit(layer([=](int j, int k) { return j < 3 ? idx(ww + (int64_t)k * 3 + j) : idx(wv + (int64_t)k * 3 + (j - 3)); }, {lin_seg(0, 6)}, W / 32, W));
    } else {
      const int64_t mo = t.mask_out.w;
      sw.emit(layer([=](int j, int k) { return idx(mo + (int64_t)k + j); }, {lin_seg(0, 1)}, W / 32, W));
    }
    emit_trunk(sw, *m);
    if ((int64_t)sw.wbytes != (int64_t)frags * 2048) return false;
    std::vector<int> map((size_t)wb / 4);
    const float* wf = reinterpret_cast<const float*>(w.data());
    for (size_t i = 0; i < map.size(); ++i) map[i] = (int)wf[i];
    if (hipMalloc(&t.bmap[which], map.size() * 4) != hipSuccess || hipMalloc(&t.bstream[which], (size_t)wb) != hipSuccess ||
        hipMemcpy(t.bmap[which], map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
      return false;
    t.bfrags[which] = (int)(wb / 2048);
  }
  // ---- the tangent pass of the second-order terms (graphs.h TanNet / BwdTrunkAlpha): forward-orientation maps of trunk + alpha head (per level),
  // hyper sheet, warp field; the reversed trunk behind its alpha head alone (per level) ----
  static const bool tan_on = !(getenv("NERFDS_TRAIN_FUSED_TAN") && std::string(getenv("NERFDS_TRAIN_FUSED_TAN")) == "0");
  if (tan_on) {
    auto upload = [&](const std::vector<uint8_t>& w, int** map_dev, void** stream_dev, int* frags_out) {
      std::vector<int> map(w.size() / 4);
      const float* wf = reinterpret_cast<const float*>(w.data());
      for (size_t i = 0; i < map.size(); ++i) map[i] = (int)wf[i];
      if (hipMalloc(map_dev, map.size() * 4) != hipSuccess || hipMalloc(stream_dev, w.size()) != hipSuccess ||
          hipMemcpy(*map_dev, map.data(), map.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
        return false;
      *frags_out = (int)(w.size() / 2048);
      return true;
    };
    auto head_layer = [&](std::function<float(int, int)> Wf, int width, int n_out) {
      nerfds::Layer L = layer(std::move(Wf), {tile_seg(width)}, 1, n_out);
      L.is_head = true;
      return L;
    };
    // a modules.MLP in forward orientation (pack.h emit_mlp with index values): W(r, c) = kernel[r][c]
    auto emit_fwd = [&](nerfds::StreamWriter& sw, const MlpP& m) {
      const int W = m.width, kc = nerfds::chunks(m.in_dim);
      auto raw_seg = [&](int row0) { std::vector<int> r; nerfds::rows_linear(r, kc, [&](int sl) { return sl < m.in_dim ? row0 + sl : -1; }); return nerfds::Seg{std::move(r), nerfds::P_F32}; };
      for (int l = 0; l < m.depth; ++l) {
        const int64_t w = m.hidden[l].w;
        std::vector<nerfds::Seg> segs;
        if (l == 0) segs.push_back(raw_seg(0));
        else {
          segs.push_back(tile_seg(W));
          if (l == m.skip) segs.push_back(raw_seg(W));
        }
        sw.emit(layer([=](int r, int c) { return idx(w + (int64_t)r * W + c); }, std::move(segs), W / 32, W));
      }
    };
    for (int which = 0; which < 4; ++which) {
      if (which == 1 && levels < 2) continue;
      const MlpP* m = which <= 1 ? &t.trunk[which] : (which == 2 ? &t.hyper : &t.warp);
      const int frags = which <= 1 ? nerfds::TanTrunk<G>::BWD_FRAGS : (which == 2 ? nerfds::TanHyper<G>::BWD_FRAGS : nerfds::TanWarp<G>::BWD_FRAGS);
      const int in_dim_k = which <= 1 ? nerfds::TanTrunk<G>::IN_DIM : (which == 2 ? nerfds::TanHyper<G>::IN_DIM : nerfds::TanWarp<G>::IN_DIM);
      if (m->skip != 4 || (m->depth != 8 && m->depth != 6) || m->in_dim != in_dim_k) return false;
      std::vector<uint8_t> w((size_t)nerfds::pad_units(2 * frags) * 1024, 0);
      nerfds::StreamWriter sw{w.data(), nullptr};
      emit_fwd(sw, *m);
      const int W = m->width;
      if (which <= 1) {
        const int64_t al = t.alpha[which].w;
        sw.emit(head_layer([=](int r, int c) { return idx(al + (int64_t)r * 4 + c); }, W, 4));                                  // t_alpha = t_h7 W_alpha
      } else if (which == 2) {
        const int64_t ho = t.hyper_out.w;
        sw.emit(head_layer([=](int r, int c) { return idx(ho + (int64_t)r * 2 + c); }, W, 2));
      } else {
        const int64_t ww = t.warp_w.w, wv = t.warp_v.w;
        sw.emit(head_layer([=](int r, int c) { return c < 3 ? idx(ww + (int64_t)r * 3 + c) : idx(wv + (int64_t)r * 3 + (c - 3)); }, W, 6));   // [w | v] (warping.py:217-218)
      }
      if ((int64_t)sw.wbytes != (int64_t)frags * 2048) return false;
      if (!upload(w, &t.tmap[which], &t.tstream[which], &t.tfrags[which])) return false;
    }
    for (int lv = 0; lv < levels; ++lv) {
      const MlpP& m = t.trunk[lv];
      const int frags = nerfds::BwdTrunkAlpha<G>::BWD_FRAGS;
      std::vector<uint8_t> w((size_t)nerfds::pad_units(2 * frags) * 1024, 0);
      nerfds::StreamWriter sw{w.data(), nullptr};
      const int64_t al = t.alpha[lv].w;
      sw.emit(layer([=](int j, int k) { return idx(al + (int64_t)k * 4 + j); }, {lin_seg(0, 4)}, m.width / 32, m.width));          // g_7 <- W_alpha d t_alpha
      emit_trunk(sw, m);
      if ((int64_t)sw.wbytes != (int64_t)frags * 2048) return false;
      if (!upload(w, &t.amap[lv], &t.astream[lv], &t.afrags[lv])) return false;
    }
    if (hipMalloc(&t.tan_slot, 16 * sizeof(float)) != hipSuccess) return false;
    // a one-unit-per-fragment stream of `frags` fragments, zero-padded to whole stages (the kernel's descriptor covers pad_units(frags) units)
    auto alloc16 = [](void** p, int frags) {
      const size_t bytes = (size_t)nerfds::pad_units(frags) * 1024;
      return hipMalloc(p, bytes) == hipSuccess && hipMemset(*p, 0, bytes) == hipSuccess;
    };
    static const bool f16_on = !(getenv("NERFDS_TRAIN_TAN_BWD_F16") && std::string(getenv("NERFDS_TRAIN_TAN_BWD_F16")) == "0");
    if (f16_on) {
      const int fr[4] = {t.afrags[0], t.afrags[1], t.bfrags[2], t.bfrags[3]};
      for (int i = 0; i < 4; ++i)
        if (fr[i] > 0 && !alloc16(&t.bstream16[i], fr[i])) return false;
      t.tan_bwd_f16 = true;
    }
    static const bool f16_fwd = getenv("NERFDS_TRAIN_TAN_FWD_F16") && std::string(getenv("NERFDS_TRAIN_TAN_FWD_F16")) == "1";
    static const bool rev_f16 = !(getenv("NERFDS_TRAIN_REV_FWD_F16") && std::string(getenv("NERFDS_TRAIN_REV_FWD_F16")) == "0");
    for (int i = 0; i < levels; ++i)
      if (!alloc16(&t.tstream16[i], t.tfrags[i])) return false;
    t.tan_fwd_f16 = f16_fwd;
    t.rev_fwd_f16 = rev_f16;
    t.fused_tan = true;
  }
  // f16 activations + ReLU bits of every hidden layer (one allocation), the sink of the input-gradient stores
  const int64_t M = t.max_rays * (t.cfg.num_coarse_samples + t.cfg.num_fine_samples);
  size_t need = 0;
  std::vector<std::pair<uint16_t**, size_t>> views;
  auto take = [&](uint16_t** p, size_t n) { views.push_back({p, n}); need += (n + 127) & ~(size_t)127; };
  auto take_net = [&](std::vector<uint16_t*>& h, std::vector<uint16_t*>& b, int depth, int width) {
    h.assign(depth, nullptr); b.assign(depth, nullptr);
    for (auto& p : h) take(&p, (size_t)M * width);
    for (auto& p : b) take(&p, (size_t)M * width / 16);
  };
  take_net(t.mask_h16, t.mask_bits, t.mask.depth, t.mask.width);
  take_net(t.warp_h16, t.warp_bits, t.warp.depth, t.warp.width);
  take_net(t.hyper_h16, t.hyper_bits, t.hyper.depth, t.hyper.width);
  take_net(t.trunk_h16, t.trunk_bits, t.trunk[0].depth, t.trunk[0].width);
  take(&t.rgb_h16, (size_t)M * RW); take(&t.rgb_bits, (size_t)M * RW / 16);
  if (hipMalloc(&t.hws, need * sizeof(uint16_t)) != hipSuccess || hipMalloc(&t.sink, 256) != hipSuccess) return false;
  uint16_t* base = t.hws;
  for (auto& v : views) { *v.first = base; base += (v.second + 127) & ~(size_t)127; }
  (void)Dm::TRUNK_KC;
  return true;
}

void pack_fused_backward(nerfds_trainer& t, PackBatch& pb) {
  const bool f16_on = getenv("NERFDS_TRAIN_BWD_F16") && std::string(getenv("NERFDS_TRAIN_BWD_F16")) == "1" && t.chain_arith == 0;
  t.bwd_f16 = false;
  for (int which = 0; which < 5; ++which)
    if (t.bmap[which]) pb.stream(t.bmap[which], t.bstream[which], t.bfrags[which]);
  if (f16_on && t.g16) {
    for (int which = 0; which < 5; ++which) {
      if (!t.bmap[which]) continue;
      if (!t.pstream16[which]) {
        const size_t bytes = (size_t)nerfds::pad_units(t.bfrags[which]) * 1024;
        if (hipMalloc(&t.pstream16[which], bytes) != hipSuccess || hipMemset(t.pstream16[which], 0, bytes) != hipSuccess) return;
      }
      pb.stream(t.bmap[which], t.pstream16[which], t.bfrags[which], 0, 0, 2);
    }
    t.bwd_f16 = true;
  }
}

// the streams of the tangent pass from the current parameters (steps that run tangents only)
void pack_fused_tangents(nerfds_trainer& t, hipStream_t st) {
  PackBatch pb;
  for (int which = 0; which < 4; ++which)
    if (t.tmap[which]) pb.stream(t.tmap[which], t.tstream[which], t.tfrags[which]);
  for (int lv = 0; lv < 2; ++lv)
    if (t.amap[lv]) pb.stream(t.amap[lv], t.astream[lv], t.afrags[lv]);
  if (t.use_tan_fwd_f16() || t.use_rev_fwd_f16())
    for (int lv = 0; lv < 2; ++lv)
      if (t.tstream16[lv]) pb.stream(t.tmap[lv], t.tstream16[lv], t.tfrags[lv], 0, 0, 2);
  if (t.use_tan_bwd_f16()) {      // the same maps, one f16 unit per fragment (k_pack_stream mode 2)
    for (int lv = 0; lv < 2; ++lv)
      if (t.amap[lv]) pb.stream(t.amap[lv], t.bstream16[lv], t.afrags[lv], 0, 0, 2);
    pb.stream(t.bmap[2], t.bstream16[2], t.bfrags[2], 0, 0, 2);
    pb.stream(t.bmap[3], t.bstream16[3], t.bfrags[3], 0, 0, 2);
  }
  pack_batch(st, t.theta, t.fold, t.P, pb);
}
// f16 [3 M][width] per hidden layer of warp field, hyper sheet, trunk: the tangents (ensure_tan16: `all` = one array per layer - a step that
// differentiates the tangent pass reads them as X of its weight gradients; otherwise ONE array per network that every layer overwrites: the
// chain stores unconditionally, and nobody reads) and their cotangents (ensure_tan16_g)
static bool carve16(nerfds_trainer& t, uint16_t** base_out, std::vector<uint16_t*>& w, std::vector<uint16_t*>& h, std::vector<uint16_t*>& tr, bool all) {
  const int64_t M3 = 3 * t.max_rays * (t.cfg.num_coarse_samples + t.cfg.num_fine_samples);
  size_t need = 0;
  std::vector<std::pair<uint16_t**, size_t>> views;
  auto take = [&](uint16_t** p, size_t n) { views.push_back({p, n}); need += (n + 127) & ~(size_t)127; };
  auto net = [&](std::vector<uint16_t*>& v, int depth, int width) {
    v.assign(depth, nullptr);
    for (int l = 0; l < (all ? depth : 1); ++l) take(&v[l], (size_t)M3 * width);
  };
  net(w, t.warp.depth, t.warp.width); net(h, t.hyper.depth, t.hyper.width); net(tr, t.trunk[0].depth, t.trunk[0].width);
  if (hipMalloc(base_out, need * sizeof(uint16_t)) != hipSuccess) return false;
  uint16_t* base = *base_out;
  for (auto& v : views) { *v.first = base; base += (v.second + 127) & ~(size_t)127; }
  if (!all) for (auto* v : {&w, &h, &tr}) for (size_t l = 1; l < v->size(); ++l) (*v)[l] = (*v)[0];
  return true;
}
bool ensure_tan16(nerfds_trainer& t, bool all) {
  if (t.tws16 && (!all || t.tw16.size() < 2 || t.tw16[1] != t.tw16[0])) return true;
  if (t.tws16) { (void)hipDeviceSynchronize(); (void)hipFree(t.tws16); t.tws16 = nullptr; }      // grown from the aliased form to one array per layer
  return carve16(t, &t.tws16, t.tw16, t.th16, t.tt16, all);
}
bool ensure_tan16_g(nerfds_trainer& t) {
  if (t.gws16) return true;
  return carve16(t, &t.gws16, t.gw16, t.gh16, t.gt16, true);
}
// tangent FORWARD chain of net 1 hyper sheet, 2 warp field, 4 trunk + alpha head of `level`: t_in [3 M][ld_in] -> t_head [3 M][ld_head], hidden tangents -> store16
// (mask_div 3: three tangent rows per sample, row r reads the masks of sample r / 3; 1: one row per sample - the reverse-mode second-order path)
// f16_ok: the caller's tangents feed weight gradients only (the reverse-mode path: target_norm comes from the reverse pass), so the trunk's chain may
// run in one f16 MFMA per product like the tangent pass's backward chains
void fused_tangent(nerfds_trainer& t, hipStream_t st, int net, int level, int64_t M3, const float* t_in, int ld_in, float* t_head, int ld_head, int mask_div,
                   bool f16_ok) {
  nerfds::TrainBwd tb{};
  t.last_tan_rows = std::max(t.last_tan_rows, M3);
  tb.M = M3; tb.d_head = t_in; tb.ld_head = ld_in; tb.d_in = t_head; tb.ld_in = ld_head; tb.sink = t.sink;
  tb.g_half = 1; tb.g_scale = t.tan_x_scale; tb.g_inv_scale = 1.f / t.tan_x_scale; tb.mask_div = mask_div;
  const std::vector<uint16_t*>* bits = nullptr;
  const std::vector<uint16_t*>* store = nullptr;
  if (net == 1) { tb.wstream = t.tstream[2]; bits = &t.hyper_bits; store = &t.th16; }
  else if (net == 2) { tb.wstream = t.tstream[3]; bits = &t.warp_bits; store = &t.tw16; }
  else { tb.wstream = t.tstream[level]; bits = &t.trunk_bits; store = &t.tt16; }
  for (size_t l = 0; l < bits->size(); ++l) { tb.bits[l] = (*bits)[l]; tb.g[l] = reinterpret_cast<float*>((*store)[l]); }
  // (experiment, off: the trunk's chain of a step that differentiates the tangent pass in one f16 MFMA per product - see profiles/r5_ab/README.md)
  if (net == 4 && t.tstream16[level] && ((f16_ok && t.use_rev_fwd_f16()) || (t.use_tan_fwd_f16() && t.keep_tangents))) { tb.wstream = t.tstream16[level]; nerfds_launch_train_tan16f_nerfds(tb, t.num_cus, st); }
  else nerfds_launch_train_tan16_nerfds(tb, net, t.num_cus, st);
}
// data-gradient chain of the TANGENT pass: cotangent of the head's tangent [3 M][ld_head] (its scale picked on the device: slot) -> g of every hidden
// tangent (f16, store16) and, if d_in, the cotangent of the raw tangent input [3 M][ld_in]
void fused_tangent_backward(nerfds_trainer& t, hipStream_t st, int net, int level, int64_t M3, const float* d_head, int ld_head, float* d_in, int ld_in,
                            const float* slot, int mask_div = 3) {
  nerfds::TrainBwd tb{};
  t.last_tan_rows = std::max(t.last_tan_rows, M3);
  tb.M = M3; tb.d_head = d_head; tb.ld_head = ld_head; tb.sink = t.sink;
  tb.d_in = d_in ? d_in : t.sink; tb.ld_in = d_in ? ld_in : 0;            // ld_in 0: every input-gradient store goes to the sink (no parameters behind t_in)
  tb.g_half = 1; tb.g_scale = 1.f; tb.g_inv_scale = 1.f; tb.scale_dev = slot + 1; tb.mask_div = mask_div;
  const std::vector<uint16_t*>* bits = nullptr;
  const std::vector<uint16_t*>* store = nullptr;
  const bool f16 = t.use_tan_bwd_f16();
  if (net == 1) { tb.wstream = f16 ? t.bstream16[2] : t.bstream[2]; bits = &t.hyper_bits; store = &t.gh16; }
  else if (net == 2) { tb.wstream = f16 ? t.bstream16[3] : t.bstream[3]; bits = &t.warp_bits; store = &t.gw16; }
  else { tb.wstream = f16 ? t.bstream16[level] : t.astream[level]; bits = &t.trunk_bits; store = &t.gt16; }
  for (size_t l = 0; l < bits->size(); ++l) { tb.bits[l] = (*bits)[l]; tb.g[l] = reinterpret_cast<float*>((*store)[l]); }
  // The backward of the tangent pass carries the SECOND-ORDER terms' gradients only (norm loss, elastic regulariser).  Its chains hand every g to the
  // weight gradients as f16 anyway; run in one f16 MFMA per product they also PROPAGATE it at 11 bits per layer - measured on the gradient tests of
  // those terms (bounds unchanged) - at a third of the MFMAs.  The primal chains and the tangent FORWARD (target_norm) stay split bf16.
  if (f16) nerfds_launch_train_bwd16f_nerfds(tb, net, t.num_cus, st);
  else nerfds_launch_train_bwd16_nerfds(tb, net, t.num_cus, st);
}

// d (one scalar per row) / d input of a network by its data-gradient chain in split bf16, at scale 1, on the primal ReLU bits: net 4 trunk behind its
// alpha head (of `level`), 1 hyper sheet, 2 warp field.  The f16 g the chain stores on its way goes to the tangent pass's g arrays (nobody reads it).
void fused_reverse(nerfds_trainer& t, hipStream_t st, int net, int level, int64_t M, const float* d_head, int ld_head, float* d_in, int ld_in) {
  nerfds::TrainBwd tb{};
  tb.M = M; tb.d_head = d_head; tb.ld_head = ld_head; tb.d_in = d_in; tb.ld_in = ld_in; tb.sink = t.sink;
  tb.g_half = 1; tb.g_scale = 1.f; tb.g_inv_scale = 1.f; tb.mask_div = 0;
  const std::vector<uint16_t*>* bits = nullptr;
  const std::vector<uint16_t*>* store = nullptr;
  // (a step that differentiates no tangent pass has no g arrays: the tangent arrays - one per network then, every layer over the same rows - take the stores)
  if (net == 1) { tb.wstream = t.bstream[2]; bits = &t.hyper_bits; store = t.gh16.empty() ? &t.th16 : &t.gh16; }
  else if (net == 2) { tb.wstream = t.bstream[3]; bits = &t.warp_bits; store = t.gw16.empty() ? &t.tw16 : &t.gw16; }
  else { tb.wstream = t.astream[level]; bits = &t.trunk_bits; store = t.gt16.empty() ? &t.tt16 : &t.gt16; }
  for (size_t l = 0; l < bits->size(); ++l) { tb.bits[l] = (*bits)[l]; tb.g[l] = reinterpret_cast<float*>((*store)[l]); }
  nerfds_launch_train_bwd16_nerfds(tb, net, t.num_cus, st);
}

// the data-gradient chain of one network: net 0 NerfMLP of `level`, 1 hyper sheet, 2 warp field, 3 mask net
void fused_backward(nerfds_trainer& t, hipStream_t st, int net, int level, int64_t M, const float* d_head, int ld_head, const float* d_head2,
                    float* d_in, int ld_in) {
  nerfds::TrainBwd tb{};
  tb.M = M; tb.d_head = d_head; tb.ld_head = ld_head; tb.d_head2 = d_head2; tb.d_in = d_in; tb.ld_in = ld_in; tb.sink = t.sink;
  tb.g_half = t.g16 ? 1 : 0;
  tb.g_scale = t.g16 ? t.g_scale : 1.f; tb.g_inv_scale = 1.f / tb.g_scale;
  const std::vector<uint16_t*>* bits = nullptr;
  const std::vector<float*>* g = nullptr;
  if (net == 0) { tb.wstream = t.bstream[level]; bits = &t.trunk_bits; g = &t.trunk_h; tb.bits[8] = t.rgb_bits; tb.g[8] = t.rgb_hv; }
  else if (net == 1) { tb.wstream = t.bstream[2]; bits = &t.hyper_bits; g = &t.hyper_h; }
  else if (net == 2) { tb.wstream = t.bstream[3]; bits = &t.warp_bits; g = &t.warp_h; }
  else { tb.wstream = t.bstream[4]; bits = &t.mask_bits; g = &t.mask_h; }
  for (size_t l = 0; l < bits->size(); ++l) { tb.bits[l] = (*bits)[l]; tb.g[l] = (*g)[l]; }
  if (tb.g_half && t.bwd_f16) {
    tb.wstream = t.pstream16[net == 0 ? level : (net == 1 ? 2 : (net == 2 ? 3 : 4))];
    nerfds_launch_train_bwd16f_nerfds(tb, net, t.num_cus, st);
  } else if (tb.g_half) nerfds_launch_train_bwd16_nerfds(tb, net, t.num_cus, st);
  else nerfds_launch_train_bwd_nerfds(tb, net, t.num_cus, st);
}

// mode / row_off / in_*: the merged step (run_merged) - mode 1 runs the shared networks only and writes their arrays `row_off` rows further down
// (the block of the fine level's new samples); mode 2 runs the NerfMLP only on shared-network results gathered into the level's row order
void fused_forward(nerfds_trainer& t, hipStream_t st, int level, int R, int S, const float* z, const nerfds_rays* rays, const nerfds_extra* ex, const Windows& W,
                   int mode = 0, int64_t row_off = 0, const float* in_xw = nullptr, const float* in_wamb = nullptr, const float* in_wv = nullptr) {
  nerfds::KArgs ka{};
  ka.origins = rays->origins; ka.directions = rays->directions; ka.viewdirs = rays->viewdirs;
  ka.warp_id = rays->warp_id; ka.gt_mask = rays->gt_mask;
  ka.wstream[0] = t.fstream[0]; ka.wstream[1] = ka.wstream[2] = t.fstream[1 + level];
  ka.bias[0] = t.fbias[0]; ka.bias[1] = ka.bias[2] = t.fbias[1 + level];
  ka.warp_embed = t.theta + t.warp_tbl; ka.mask_embed = t.theta + t.mask_tbl;
  ka.num_rays = R; ka.num_embeds = t.cfg.num_warp_embeds; ka.nc = S; ka.nf = 0;
  ka.mask_ratio = ex->mask_ratio; ka.near_ = ex->near; ka.far_ = ex->far;
  std::memcpy(ka.win_mask, W.mask, sizeof W.mask); std::memcpy(ka.win_warp, W.warp, sizeof W.warp); std::memcpy(ka.win_hyp, W.hyp, sizeof W.hyp);
  std::memcpy(ka.win_sp, W.sp, sizeof W.sp); std::memcpy(ka.win_hp, W.hp, sizeof W.hp); std::memcpy(ka.win_nm, W.nm, sizeof W.nm);
  nerfds::TrainOut to{};
  for (int l = 0; l < 8; ++l) { to.mask_h[l] = t.mask_h[l]; to.trunk_h[l] = t.trunk_h[l]; }
  for (int l = 0; l < 6; ++l) { to.warp_h[l] = t.warp_h[l]; to.hyper_h[l] = t.hyper_h[l]; }
  to.rgb_h = t.rgb_hv; to.mask_logit = t.mask_logit; to.wv = t.wv; to.wamb = t.wamb; to.alphav = t.alphav; to.rgb_logit = t.rgb_logit;
  to.z = z; to.level = level;
  to.mode = mode; to.in_xw = in_xw; to.in_wamb = in_wamb; to.in_wv = in_wv;
  to.half_out = t.half_step ? 1 : 0;
  if (t.half_step) {
    for (int l = 0; l < 8; ++l) { to.mask_h16[l] = t.mask_h16[l]; to.mask_bits[l] = t.mask_bits[l]; to.trunk_h16[l] = t.trunk_h16[l]; to.trunk_bits[l] = t.trunk_bits[l]; }
    for (int l = 0; l < 6; ++l) { to.warp_h16[l] = t.warp_h16[l]; to.warp_bits[l] = t.warp_bits[l]; to.hyper_h16[l] = t.hyper_h16[l]; to.hyper_bits[l] = t.hyper_bits[l]; }
    to.rgb_h16 = t.rgb_h16; to.rgb_bits = t.rgb_bits;
  }
  if (row_off) {        // the shared networks' arrays, `row_off` rows down ([M][width] each; ReLU bits: one u16 per row, lane half and 32-feature tile)
    const int MW = t.mask.width, WW = t.warp.width, HW = t.hyper.width;
    for (int l = 0; l < 8; ++l) { to.mask_h[l] += row_off * MW; if (to.mask_h16[l]) { to.mask_h16[l] += row_off * MW; to.mask_bits[l] += row_off * 2 * (MW / 32); } }
    for (int l = 0; l < 6; ++l) {
      to.warp_h[l] += row_off * WW; to.hyper_h[l] += row_off * HW;
      if (to.warp_h16[l]) { to.warp_h16[l] += row_off * WW; to.warp_bits[l] += row_off * 2 * (WW / 32); to.hyper_h16[l] += row_off * HW; to.hyper_bits[l] += row_off * 2 * (HW / 32); }
    }
    to.mask_logit += row_off; to.wv += row_off * 6; to.wamb += row_off * 2;
  }
  if (to.half_out) nerfds_launch_train_fwd16_nerfds(ka, to, t.num_cus, st);
  else nerfds_launch_train_fwd_nerfds(ka, to, t.num_cus, st);
}

// nerfds_train_numerics.diagnose: the NerfMLP arrays of `level` (M rows) scanned for inf / NaN into the overflow flags, at the point of the step where that
// level's chains have run and the next level has not yet overwritten them (the coarse level's arrays are invisible to nerfds_trainer_overflow_sources
// afterwards).  `second`: the level ran a tangent pass whose backward was differentiated (tan_rows rows of tt16 / gt16).
void diag_scan_nerf(nerfds_trainer& t, hipStream_t st, int64_t M, bool second, int64_t tan_rows) {
  if (!t.diag) return;
  unsigned* flags = reinterpret_cast<unsigned*>(t.terms_dev + 15);
  auto r8 = [](int64_t n) { return n & ~(int64_t)7; };
  const int TW = t.trunk[0].width, RW = t.rgb_h[0].N;
  if (t.half_step) {
    for (auto* p : t.trunk_h16) if (p) scan_half(st, p, r8(M * TW), flags, NERFDS_OVF_ACTIVATION);
    if (t.rgb_h16) scan_half(st, t.rgb_h16, r8(M * RW), flags, NERFDS_OVF_ACTIVATION);
    for (auto* p : t.trunk_h) {
      if (t.g16) scan_half(st, reinterpret_cast<const uint16_t*>(p), r8(M * TW), flags, NERFDS_OVF_PRIMAL_G);
      else scan_float(st, p, M * TW, flags, NERFDS_OVF_PRIMAL_G);
    }
    if (t.g16) scan_half(st, reinterpret_cast<const uint16_t*>(t.rgb_hv), r8(M * RW), flags, NERFDS_OVF_PRIMAL_G);
    else scan_float(st, t.rgb_hv, M * RW, flags, NERFDS_OVF_PRIMAL_G);
  }
  scan_float(st, t.alphav, M * 4, flags, NERFDS_OVF_FP32 | (1u << (NERFDS_OVF_DETAIL_SHIFT + 2)));
  scan_float(st, t.rgb_logit, M * 3, flags, NERFDS_OVF_FP32 | (1u << (NERFDS_OVF_DETAIL_SHIFT + 3)));
  scan_float(st, t.d_alpha, M * 4, flags, NERFDS_OVF_FP32_BACKWARD | (1u << (NERFDS_OVF_DETAIL_SHIFT + 6)));
  scan_float(st, t.d_rgb_logit, M * 3, flags, NERFDS_OVF_FP32_BACKWARD | (1u << (NERFDS_OVF_DETAIL_SHIFT + 7)));
  scan_float(st, t.d_trunk_in, M * t.D.trunk_in, flags, NERFDS_OVF_FP32_BACKWARD | (1u << (NERFDS_OVF_DETAIL_SHIFT + 11)));
  if (second && t.fused_tan && t.half_step && tan_rows > 0) {
    const uint16_t* prev = nullptr;
    for (auto* p : t.tt16) { if (p && p != prev) scan_half(st, p, r8(tan_rows * TW), flags, NERFDS_OVF_TANGENT); prev = p; }
    prev = nullptr;
    for (auto* p : t.gt16) { if (p && p != prev) scan_half(st, p, r8(tan_rows * TW), flags, NERFDS_OVF_COTANGENT); prev = p; }
    // the raw tangent input of the trunk enters its weight gradients as f16 at the stored tangents' scale (WgradArgs::x_scale)
    scan_float(st, t.t_tin, tan_rows * t.D.trunk_in, flags, NERFDS_OVF_TANGENT, 65504.f / t.tan_x_scale);
    scan_float(st, t.t_alpha, tan_rows * 4, flags, NERFDS_OVF_FP32_SECOND_ORDER | (1u << (NERFDS_OVF_DETAIL_SHIFT + 12)));
    scan_float(st, t.d_t_alpha, tan_rows * 4, flags, NERFDS_OVF_FP32_SECOND_ORDER | (1u << (NERFDS_OVF_DETAIL_SHIFT + 13)));
    scan_float(st, t.d_t_tin, tan_rows * t.D.trunk_in, flags, NERFDS_OVF_FP32_SECOND_ORDER | (1u << (NERFDS_OVF_DETAIL_SHIFT + 14)));
  }
}

// The plain two-level step with the level-independent networks evaluated - and differentiated - ONCE per sample position (round 4).
// The mask network, the SE(3) field and the hyper sheet see only the observation-space point, the GLO rows and the mask; the fine level's
// sorted union repeats the Nc coarse positions, and the reference runs the three networks there a second time (models.py:1528-1546 over
// 1291-1300) on identical inputs.  The gradient of a loss through two identical evaluations is the gradient through one evaluation with the two
// upstream gradients added.  So: position rows = [R x Nc coarse samples | R x Nf new samples]; the shared networks run forward once on each
// block; the fine NerfMLP runs on the union with their results gathered into its row order; its gradient w.r.t. the warped points / ambient
// coordinates is scattered back to the position rows (added, for the coarse block, to what the coarse NerfMLP left there); and ONE backward
// pass of the three networks (chains + weight gradients) covers all R x (Nc + Nf) positions instead of R x Nc + R x (Nc + Nf) rows: a third
// less shared-network work per step.  Same loss, same gradients up to the order of the sums (tests/test_training.py compares with the oracle).
int run_merged(nerfds_trainer& t, hipStream_t st, int R, const float* zc, const nerfds_rays* rays, const float* target, const nerfds_extra* ex,
               const Windows& W, const nerfds_rand* rnd) {
  const Dims& D = t.D;
  const int Nc = t.cfg.num_coarse_samples, Nf = t.cfg.num_fine_samples, S = Nc + Nf, strat = ex->use_stratified_sampling;
  const int64_t Mc = (int64_t)R * Nc, Mn = (int64_t)R * Nf, Mf = Mc + Mn, Ms = Mc + Mn;
  const float* viewdirs = rays->viewdirs ? rays->viewdirs : rays->directions;
  const int VD = 6 * D.vd_bands, NM = 6 * D.nm_bands, CW = VD + NM;
  // scratch in buffers only the layer-by-layer backward uses (g0 / g1 / g2: M x 256 floats each)
  float* xw_f = t.g0; float* wamb_f = xw_f + 3 * Mf; float* wv_f = wamb_f + 2 * Mf;
  float* dxw_f = t.g1; float* dwamb_f = dxw_f + 3 * Mf;
  float* z_new = t.g2; int* src = reinterpret_cast<int*>(z_new + Mn);
  auto wg_nerf = [&](Run& r, int level) {
    const MlpP& trunk = t.trunk[level];
    const LayerP& K = t.rgb_h[level];
    const int TW = trunk.width, RW = K.N;
    r.head_wgrads(t.rgb_out[level], t.rgb_h16, RW, t.d_rgb_logit, 3);
    r.head_wgrads(t.alpha[level], t.trunk_h16.back(), TW, t.d_alpha, 4);
    r.weight_grad(reinterpret_cast<const float*>(t.trunk_h16.back()), TW, TW, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)(TW + VD) * RW, -1, true, t.grad + K.b, t.g16);
    r.weight_grad(t.cond, CW, VD, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)TW * RW, -1, false, nullptr, t.g16);
    r.weight_grad(t.cond + VD, CW, NM, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)(2 * TW + VD) * RW, -1, false, nullptr, t.g16);
    r.mlp_wgrads(trunk, t.trunk_in, t.trunk_h16, t.trunk_h);
  };
  // ---------------- coarse level: shared networks (block A) + coarse NerfMLP, its loss and its NerfMLP backward ----------------
  Run rc{t, st, Mc};
  encode_inputs(st, D, R, Nc, rays->origins, rays->directions, zc, rays->warp_id, t.cfg.num_warp_embeds, t.theta + t.warp_tbl, t.theta + t.mask_tbl, W,
                t.x, t.mask_in, t.warp_in, t.hyper_in);
  fused_forward(t, st, 0, R, Nc, zc, rays, ex, W);
  mask_post(st, D, R, Nc, t.mask_logit, rays->gt_mask, ex->mask_ratio, t.warp_in, t.hyper_in);
  se3_fwd(st, Mc, t.wv, t.x, t.xw);
  trunk_in(st, D, Mc, t.xw, t.wamb, W, t.trunk_in);
  alpha_post(st, D, R, Nc, t.alphav, t.wv, viewdirs, W, t.sigma, t.cond);
  composite_loss(st, R, Nc, zc, rays->directions, t.sigma, t.rgb_logit, target, t.cfg.use_sample_at_infinity, t.cfg.use_white_background, t.rgb_ray, t.wc,
                 t.loss_dev + 0, t.d_rgb_logit, t.d_alpha, t.cot[0], t.lout[0]);
  fused_backward(t, st, 0, 0, Mc, t.d_rgb_logit, 3, t.d_alpha, t.d_trunk_in, D.trunk_in);
  rc.fork(false); wg_nerf(rc, 0);
  trunk_in_bwd(st, D, Mc, t.d_trunk_in, t.xw, t.wamb, W, nullptr, nullptr, t.dxw, t.dwamb);          // position rows of block A
  diag_scan_nerf(t, st, Mc, false, 0);
  // ---------------- the fine level's new samples: shared networks only (block B), under the coarse weight gradients ----------------
  resample(st, R, Nc, Nf, zc, t.wc, strat, rnd ? rnd->u_rand : nullptr, rnd ? rnd->seed : 0, rnd ? rnd->first_ray : 0, t.zf, t.rs_scratch, z_new, src);
  encode_inputs(st, D, R, Nf, rays->origins, rays->directions, z_new, rays->warp_id, t.cfg.num_warp_embeds, t.theta + t.warp_tbl, t.theta + t.mask_tbl, W,
                t.x + 3 * Mc, t.mask_in + Mc * D.mask_in, t.warp_in + Mc * D.warp_ld, t.hyper_in + Mc * D.hyper_ld);
  fused_forward(t, st, 1, R, Nf, z_new, rays, ex, W, 1, Mc);
  mask_post(st, D, R, Nf, t.mask_logit + Mc, rays->gt_mask, ex->mask_ratio, t.warp_in + Mc * D.warp_ld, t.hyper_in + Mc * D.hyper_ld);
  se3_fwd(st, Mn, t.wv + 6 * Mc, t.x + 3 * Mc, t.xw + 3 * Mc);
  gather_rows(st, Mf, src, t.xw, t.wamb, t.wv, xw_f, wamb_f, wv_f);
  // ---------------- fine level: NerfMLP on the sorted union (the coarse NerfMLP's activations are free once its weight gradients ran) ----------------
  rc.join();
  if (!rc.ok) return t.fail(NERFDS_ENOTSUP, "%s", rc.unsupported_what.c_str());
  Run rf{t, st, Mf};
  fused_forward(t, st, 1, R, S, t.zf, rays, ex, W, 2, 0, xw_f, wamb_f, wv_f);
  trunk_in(st, D, Mf, xw_f, wamb_f, W, t.trunk_in);
  alpha_post(st, D, R, S, t.alphav, wv_f, viewdirs, W, t.sigma, t.cond);
  composite_loss(st, R, S, t.zf, rays->directions, t.sigma, t.rgb_logit, target, t.cfg.use_sample_at_infinity, t.cfg.use_white_background, t.rgb_ray, t.weights,
                 t.loss_dev + 1, t.d_rgb_logit, t.d_alpha, t.cot[1], t.lout[1]);
  fused_backward(t, st, 0, 1, Mf, t.d_rgb_logit, 3, t.d_alpha, t.d_trunk_in, D.trunk_in);
  rf.fork(false); wg_nerf(rf, 1);
  trunk_in_bwd(st, D, Mf, t.d_trunk_in, xw_f, wamb_f, W, nullptr, nullptr, dxw_f, dwamb_f);
  scatter_rows(st, Mf, src, Mc, dxw_f, dwamb_f, t.dxw, t.dwamb);
  // ---------------- the shared networks' backward, once over every position row ----------------
  rf.M = Ms;
  fused_backward(t, st, 1, 1, Ms, t.dwamb, 2, nullptr, t.d_hyper_in, D.hyper_ld);
  rf.fork(false);
  rf.head_wgrads(t.hyper_out, t.hyper_h16.back(), t.hyper.width, t.dwamb, 2);
  rf.mlp_wgrads(t.hyper, t.hyper_in, t.hyper_h16, t.hyper_h);
  se3_bwd(st, Ms, t.wv, t.x, t.dxw, nullptr, t.dwv);
  fused_backward(t, st, 2, 1, Ms, t.dwv, 6, nullptr, t.d_warp_in, D.warp_ld);
  rf.fork(false);
  rf.head_wgrads(t.warp_w, t.warp_h16.back(), t.warp.width, t.dwv, 6);
  rf.head_wgrads(t.warp_v, t.warp_h16.back(), t.warp.width, t.dwv + 3, 6);
  rf.mlp_wgrads(t.warp, t.warp_in, t.warp_h16, t.warp_h);
  shared_in_bwd(st, D, R, Nc, t.d_warp_in, t.d_hyper_in, t.mask_logit, ex->mask_ratio, nullptr, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.warp_tbl, t.d_mask_logit);
  shared_in_bwd(st, D, R, Nf, t.d_warp_in + Mc * D.warp_ld, t.d_hyper_in + Mc * D.hyper_ld, t.mask_logit + Mc, ex->mask_ratio, nullptr, rays->warp_id,
                t.cfg.num_warp_embeds, t.grad + t.warp_tbl, t.d_mask_logit + Mc);
  fused_backward(t, st, 3, 1, Ms, t.d_mask_logit, 1, nullptr, t.d_mask_in, D.mask_in);
  mask_in_bwd(st, D, R, Nc, t.d_mask_in, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.mask_tbl);
  mask_in_bwd(st, D, R, Nf, t.d_mask_in + Mc * D.mask_in, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.mask_tbl);
  rf.fork(true);
  rf.head_wgrads(t.mask_out, t.mask_h16.back(), t.mask.width, t.d_mask_logit, 1);
  rf.mlp_wgrads(t.mask, t.mask_in, t.mask_h16, t.mask_h);
  rf.join();
  if (!rf.ok) return t.fail(NERFDS_ENOTSUP, "%s", rf.unsupported_what.c_str());
  return NERFDS_OK;
}

// run_merged with the whole objective (round 5): the per-ray auxiliary losses and the second-order terms (norm loss at both levels, elastic regulariser
// at the coarse one) on the same position rows.  What the shared networks contribute to those terms is shared like their primal pass: the tangent
// chains of the warp field and the hyper sheet (three rows per position) run ONCE over [coarse | new] positions, the fine level reads them gathered into
// its row order, and every gradient the fine level leaves on them - cotangents of the tangents of x' and of the ambient coordinates, the exp_se3
// rotation term of target_norm, the predicted mask's gradient - is scattered back and added to the coarse level's before ONE backward of the
// tangent chains and ONE primal backward.  Per level remain: the NerfMLP, its tangent chain and their backwards.  A third less shared-network work,
// primal and tangent, than the level-by-level flow (run_level twice), same sums in a different order (tests/test_training.py, every term and leaf).
int run_merged_full(nerfds_trainer& t, hipStream_t st, int R, const float* zc, const nerfds_rays* rays, const float* target, const nerfds_extra* ex,
                    const Windows& W, const nerfds_rand* rnd, const Objective& ob, float norm_weight) {
  const Dims& D = t.D;
  const int Nc = t.cfg.num_coarse_samples, Nf = t.cfg.num_fine_samples, S = Nc + Nf, strat = ex->use_stratified_sampling;
  const int64_t Mc = (int64_t)R * Nc, Mn = (int64_t)R * Nf, Mf = Mc + Mn, Ms = Mc + Mn;
  const float* viewdirs = rays->viewdirs ? rays->viewdirs : rays->directions;
  const int VD = 6 * D.vd_bands, NM = 6 * D.nm_bands, CW = VD + NM;
  const bool nl = norm_weight != 0.f, el = ob.elastic_weight != 0.f, so = nl || el, hreg = ob.hyper_reg_weight != 0.f;
  const int64_t Mt = nl ? Ms : Mc;            // positions with tangents (the elastic regulariser alone: the coarse block's warp tangents)
  // scratch in buffers only the layer-by-layer backward uses (g0 / g1 / g2: M x 256 floats each): the fine level's gathered inputs and what it leaves
  float* p0 = t.g0; float* p1 = t.g1;
  auto take = [](float*& p, int64_t n) { float* r = p; p += (n + 63) & ~(int64_t)63; return r; };
  float* xw_f = take(p0, 3 * Mf); float* wamb_f = take(p0, 2 * Mf); float* wv_f = take(p0, 6 * Mf); float* x_f = take(p0, 3 * Mf);
  float* ml_f = take(p0, Mf); float* t_xw_f = take(p0, 9 * Mf); float* t_wamb_f = take(p0, 6 * Mf);
  float* dxw_f = take(p1, 3 * Mf); float* dwamb_f = take(p1, 2 * Mf); float* d_t_xw_f = take(p1, 9 * Mf); float* d_t_wamb_f = take(p1, 6 * Mf);
  float* dxw_reg_f = take(p1, 3 * Mf); float* dwamb_extra_f = take(p1, 2 * Mf); float* d_pm_f = take(p1, Mf); float* du_f = take(p1, 3 * Mf);
  float* ghat_f = take(p1, 3 * Mf); float* rot_f = take(p1, 6 * Mf); float* rot = take(p1, 6 * Ms);
  // the reverse-mode second-order path (below): per-level scratch in the level's row order (_l) and in position order (_p)
  float* sa_l = take(p1, 3 * Mf); float* sb_l = take(p1, 2 * Mf); float* sc_l = take(p1, 6 * Mf); float* sb_p = take(p1, 2 * Ms); float* sc_p = take(p1, 6 * Ms);
  float* gx_p = take(p1, 3 * Ms); float* gx_l = take(p1, 3 * Mf); float* dir_l = take(p1, 3 * Mf); float* dir_p = take(p1, 3 * Ms); float* cot_l = take(p1, 4 * Mf);
  float* dtx_p = take(p1, 3 * Ms); float* dtw_p = take(p1, 2 * Ms); float* rot_p = take(p1, 6 * Ms);
  float* z_new = t.g2; int* src = reinterpret_cast<int*>(z_new + Mn);
  {   // what was just carved must fit the scratch buffers (each max_rays * (Nc + Nf) * trunk width floats): checked on the ACTUAL footprint, not on a
      // hand-kept constant - a narrower trunk (or more scratch views) fails here instead of writing past g0 / g1 / g2
    const int64_t cap = t.max_rays * (int64_t)(Nc + Nf) * t.trunk[0].width;
    if (p0 - t.g0 > cap || p1 - t.g1 > cap || Mn + Mf > cap)
      return t.fail(NERFDS_ENOTSUP, "run_merged_full: scratch of %lld / %lld floats does not fit the %lld-float layer buffers (trunk width %d)",
                    (long long)(p0 - t.g0), (long long)(p1 - t.g1), (long long)cap, t.trunk[0].width);
  }
  // NERFDS_TRAIN_REVERSE_SIGMA=0: the three-unit-direction tangent pass for the norm loss, as before (A/B, fallback; the elastic regulariser needs the
  // whole warp Jacobian and keeps it)
  static const bool rev_on = !(getenv("NERFDS_TRAIN_REVERSE_SIGMA") && std::string(getenv("NERFDS_TRAIN_REVERSE_SIGMA")) == "0");
  const bool rev = nl && !el && rev_on;
  auto as_f = [](const std::vector<uint16_t*>& v) { std::vector<float*> o; for (auto* p : v) o.push_back(reinterpret_cast<float*>(p)); return o; };
  const std::vector<float*> gt = as_f(t.gt16), gh = as_f(t.gh16), gw = as_f(t.gw16);
  const float scale_target = t.scale_target();   // (run_level: the largest cotangent lands at 2^5; nerfds_train_numerics moves it)
  bool ok = true; std::string what;
  auto wg_nerf = [&](Run& r, int level) {
    const MlpP& trunk = t.trunk[level];
    const LayerP& K = t.rgb_h[level];
    const int TW = trunk.width, RW = K.N;
    r.head_wgrads(t.rgb_out[level], t.rgb_h16, RW, t.d_rgb_logit, 3);
    r.head_wgrads(t.alpha[level], t.trunk_h16.back(), TW, t.d_alpha, 4);
    r.weight_grad(reinterpret_cast<const float*>(t.trunk_h16.back()), TW, TW, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)(TW + VD) * RW, -1, true, t.grad + K.b, t.g16);
    r.weight_grad(t.cond, CW, VD, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)TW * RW, -1, false, nullptr, t.g16);
    r.weight_grad(t.cond + VD, CW, NM, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)(2 * TW + VD) * RW, -1, false, nullptr, t.g16);
    r.mlp_wgrads(trunk, t.trunk_in, t.trunk_h16, t.trunk_h);
  };
  // the level's own part of the objective beyond the rgb loss, BEFORE its primal NerfMLP backward (it adds to d_alpha): the trunk's tangent chain ->
  // target_norm, the per-ray auxiliary terms, the norm loss, the backward of the trunk's tangent chain down to the tangents of x' / the ambient
  // coordinates (in the level's row order), the rotation term of target_norm, the elastic regulariser.  Arrays *_l: the level's rows.
  auto second_order = [&](int level, int Sl, const float* z, const float* weights, const float* x_l, const float* xw_l, const float* wamb_l, const float* wv_l,
                          const float* ml_l, const float* t_xw_l, const float* t_wamb_l, float* d_t_xw_l, float* d_t_wamb_l, float* dxw_reg_l,
                          float* dwamb_extra_l, float* d_pm_l, float* du_l, float* ghat_l, float* rot_l) {
    const int64_t M = (int64_t)R * Sl;
    const MlpP& trunk = t.trunk[level];
    Objective obl = ob;
    if (level != 0) { obl.hyper_reg_weight = 0.f; obl.elastic_weight = 0.f; }
    if (rev) {
      // ---- reverse mode for grad_x sigma, forward over reverse for its gradient (train_kernels.hip k_fill_head4 ...).  Shared networks: in POSITION
      // order (their ReLU bits live there) - the coarse level's rows ARE positions [0, Mc); the fine level's cotangents / directions are scattered to the
      // positions (each exactly once) and the results gathered back. ----
      const bool fine = level != 0;
      const int64_t Mp = fine ? Ms : Mc;
      const float* x_p = t.x; const float* wv_p = t.wv;
      // (A) d sigma_raw / d x: trunk + alpha head, encodings, hyper sheet, exp_se3, warp field - one row per sample, split bf16
      fill_head4(st, M, nullptr, 1.f, cot_l);
      fused_reverse(t, st, 4, level, M, cot_l, 4, t.d_t_tin, D.trunk_in);
      trunk_in_bwd(st, D, M, t.d_t_tin, xw_l, wamb_l, W, nullptr, nullptr, sa_l, sb_l);          // a = d sigma / d x', b = d sigma / d ambient
      se3_bwd(st, M, wv_l, x_l, sa_l, nullptr, sc_l);                                             // c = d sigma / d (w, v)
      const float* b_pp = sb_l; const float* c_pp = sc_l;
      if (fine) { scatter_cols(st, M, src, 0, 1, 2, sb_l, sb_p); scatter_cols(st, M, src, 0, 1, 6, sc_l, sc_p); b_pp = sb_p; c_pp = sc_p; }
      fused_reverse(t, st, 1, 0, Mp, b_pp, 2, t.d_hyper_in, D.hyper_ld);
      fused_reverse(t, st, 2, 0, Mp, c_pp, 6, t.d_warp_in, D.warp_ld);
      posenc_rev_x(st, D, Mp, x_p, t.d_warp_in, t.d_hyper_in, W, gx_p);
      const float* gx_ll = gx_p;
      if (fine) { gather_cols(st, M, src, 1, 3, gx_p, gx_l); gx_ll = gx_l; }
      sigma_grad_assemble(st, M, wv_l, sa_l, gx_ll, t.t_alpha);
      target_norm(st, M, t.t_alpha, wv_l, t.tn[level]);
      aux_losses(st, R, Sl, obl, z, weights, x_l, xw_l, t.alphav, viewdirs, ml_l, rays->gt_mask, t.terms_dev + 4 * level, dxw_reg_l, t.d_alpha, d_pm_l, wamb_l,
                 t.terms_dev + 9 + level, t.dwamb_reg, t.terms_dev + 13 + level);
      norm_loss(st, R, Sl, norm_weight, weights, t.alphav, t.t_alpha, wv_l, t.tn[level], t.terms_dev + 4 * level + 3, t.d_alpha, t.d_t_alpha, du_l, ghat_l);
      // (B) the tangent pass along c = d L / d (grad_x sigma), ONE row per sample, scaled to O(1) by a power of two K (slot 12 .. 15)
      pick_scale(st, t.d_t_alpha, 3 * M * 4, 0.f, 1.f, t.tan_slot + 12);
      make_dir(st, M, t.d_t_alpha, t.tan_slot + 12, dir_l, cot_l);
      const float* dir_pp = dir_l;
      if (fine) { scatter_cols(st, M, src, 0, 1, 3, dir_l, dir_p); dir_pp = dir_p; }
      encode_tangent_dir(st, D, Mp, x_p, dir_pp, W, t.t_warp_in, t.t_hyper_in);
      fused_tangent(t, st, 2, 0, Mp, t.t_warp_in, D.warp_ld, t.t_wv, 6, 1);
      se3_jvp_dir(st, Mp, wv_p, x_p, dir_pp, t.t_wv, t.t_xw);
      fused_tangent(t, st, 1, 0, Mp, t.t_hyper_in, D.hyper_ld, t.t_wamb, 2, 1);
      const float* txw_l = t.t_xw; const float* twa_l = t.t_wamb;
      if (fine) { gather_cols(st, M, src, 1, 3, t.t_xw, const_cast<float*>(t_xw_l)); gather_cols(st, M, src, 1, 2, t.t_wamb, const_cast<float*>(t_wamb_l)); txw_l = t_xw_l; twa_l = t_wamb_l; }
      trunk_in_jvp(st, D, M, xw_l, wamb_l, txw_l, twa_l, W, t.t_tin, 1);
      fused_tangent(t, st, 4, level, M, t.t_tin, D.trunk_in, t.t_alpha, 4, 1, true);
      // (C) its backward from the head cotangent 1 / K, and the weight gradients: products of (B)'s stored tangents and (C)'s g, one row per sample
      {
        Run rt{t, st, M};
        rt.tan = true;
        pick_scale(st, cot_l, M * 4, scale_target, t.tan_x_scale, t.tan_slot);
        rt.tan_slot = t.tan_slot;
        fused_tangent_backward(t, st, 4, level, M, cot_l, 4, t.d_t_tin, D.trunk_in, t.tan_slot, 1);
        rt.fork(false);
        rt.head_wgrads(t.alpha[level], t.tt16.back(), trunk.width, cot_l, 4);
        rt.mlp_wgrads(trunk, t.t_tin, t.tt16, gt);
        rt.wg_turn = -1;
        if (!rt.ok) { ok = false; what = rt.unsupported_what; }
      }
      trunk_in_jvp_bwd(st, D, M, t.d_t_tin, xw_l, wamb_l, txw_l, twa_l, W, d_t_xw_l, d_t_wamb_l, dxw_reg_l, dwamb_extra_l, 1);
      if (hreg && level == 0) add_inplace(st, dwamb_extra_l, t.dwamb_reg, 2 * M);
      se3_rot_bwd(st, M, wv_l, du_l, ghat_l, rot_l);
      const float* dtx_pp = d_t_xw_l; const float* dtw_pp = d_t_wamb_l; const float* rot_pp = rot_l;
      if (fine) {
        scatter_cols(st, M, src, 0, 1, 3, d_t_xw_l, dtx_p); scatter_cols(st, M, src, 0, 1, 2, d_t_wamb_l, dtw_p); scatter_cols(st, M, src, 0, 1, 6, rot_l, rot_p);
        dtx_pp = dtx_p; dtw_pp = dtw_p; rot_pp = rot_p;
      }
      {
        Run rt{t, st, Mp};
        rt.tan = true;
        pick_scale(st, dtw_pp, Mp * 2, scale_target, t.tan_x_scale, t.tan_slot + 4);
        rt.tan_slot = t.tan_slot + 4;
        fused_tangent_backward(t, st, 1, 0, Mp, dtw_pp, 2, nullptr, 0, t.tan_slot + 4, 1);
        rt.fork(false);
        rt.head_wgrads(t.hyper_out, t.th16.back(), t.hyper.width, dtw_pp, 2);
        rt.mlp_wgrads(t.hyper, t.t_hyper_in, t.th16, gh);
        // exp_se3's second derivatives: this level's share of dwv_extra, per position (the coarse level writes its block and clears the rest, the
        // fine level adds to all of it), on top of the level's rotation term
        if (!fine) {
          se3_jvp_bwd(st, Mp, wv_p, x_p, t.t_wv, dtx_pp, nullptr, nullptr, t.d_t_wv, t.dwv_extra, rot_pp, dir_pp);
          (void)hipMemsetAsync(t.dwv_extra + 6 * Mc, 0, (size_t)6 * (Ms - Mc) * sizeof(float), st);
        } else {
          add_inplace(st, t.dwv_extra, rot_pp, 6 * Mp);
          se3_jvp_bwd(st, Mp, wv_p, x_p, t.t_wv, dtx_pp, nullptr, nullptr, t.d_t_wv, t.dwv_extra, t.dwv_extra, dir_pp);
        }
        pick_scale(st, t.d_t_wv, Mp * 6, scale_target, t.tan_x_scale, t.tan_slot + 8);
        rt.tan_slot = t.tan_slot + 8;
        fused_tangent_backward(t, st, 2, 0, Mp, t.d_t_wv, 6, nullptr, 0, t.tan_slot + 8, 1);
        rt.fork(false);
        rt.head_wgrads(t.warp_w, t.tw16.back(), t.warp.width, t.d_t_wv, 6);
        rt.head_wgrads(t.warp_v, t.tw16.back(), t.warp.width, t.d_t_wv + 3, 6);
        rt.mlp_wgrads(t.warp, t.t_warp_in, t.tw16, gw);
        rt.wg_turn = -1;
        if (!rt.ok) { ok = false; what = rt.unsupported_what; }
      }
      return;
    }
    if (nl) {
      trunk_in_jvp(st, D, M, xw_l, wamb_l, t_xw_l, t_wamb_l, W, t.t_tin);
      fused_tangent(t, st, 4, level, 3 * M, t.t_tin, D.trunk_in, t.t_alpha, 4);
      target_norm(st, M, t.t_alpha, wv_l, t.tn[level]);
    }
    aux_losses(st, R, Sl, obl, z, weights, x_l, xw_l, t.alphav, viewdirs, ml_l, rays->gt_mask, t.terms_dev + 4 * level, dxw_reg_l, t.d_alpha, d_pm_l, wamb_l,
               t.terms_dev + 9 + level, t.dwamb_reg, t.terms_dev + 13 + level);
    const bool el_l = el && level == 0;
    if (nl) {
      norm_loss(st, R, Sl, norm_weight, weights, t.alphav, t.t_alpha, wv_l, t.tn[level], t.terms_dev + 4 * level + 3, t.d_alpha, t.d_t_alpha, du_l, ghat_l);
      Run rt{t, st, 3 * M};
      rt.tan = true;
      pick_scale(st, t.d_t_alpha, 3 * M * 4, scale_target, t.tan_x_scale, t.tan_slot);
      rt.tan_slot = t.tan_slot;
      fused_tangent_backward(t, st, 4, level, 3 * M, t.d_t_alpha, 4, t.d_t_tin, D.trunk_in, t.tan_slot);
      rt.fork(false);
      rt.head_wgrads(t.alpha[level], t.tt16.back(), trunk.width, t.d_t_alpha, 4);
      rt.mlp_wgrads(trunk, t.t_tin, t.tt16, gt);
      trunk_in_jvp_bwd(st, D, M, t.d_t_tin, xw_l, wamb_l, t_xw_l, t_wamb_l, W, d_t_xw_l, d_t_wamb_l, dxw_reg_l, dwamb_extra_l);
      if (hreg && level == 0) add_inplace(st, dwamb_extra_l, t.dwamb_reg, 2 * M);
      se3_rot_bwd(st, M, wv_l, du_l, ghat_l, rot_l);
      rt.wg_turn = -1;                        // (joined with the level's primal weight gradients)
      if (!rt.ok) { ok = false; what = rt.unsupported_what; }
    } else if (el_l) {
      (void)hipMemsetAsync(d_t_xw_l, 0, (size_t)9 * M * sizeof(float), st);
    }
    if (el_l) elastic_loss(st, R, Sl, ob.elastic_weight, ob.elastic_by_weight, weights, t_xw_l, t.terms_dev + 12, d_t_xw_l);
  };
  // ---------------- coarse level forward (shared networks on block A + coarse NerfMLP) and its loss ----------------
  Run rc{t, st, Mc};
  encode_inputs(st, D, R, Nc, rays->origins, rays->directions, zc, rays->warp_id, t.cfg.num_warp_embeds, t.theta + t.warp_tbl, t.theta + t.mask_tbl, W,
                t.x, t.mask_in, t.warp_in, t.hyper_in);
  fused_forward(t, st, 0, R, Nc, zc, rays, ex, W);
  mask_post(st, D, R, Nc, t.mask_logit, rays->gt_mask, ex->mask_ratio, t.warp_in, t.hyper_in);
  se3_fwd(st, Mc, t.wv, t.x, t.xw);
  trunk_in(st, D, Mc, t.xw, t.wamb, W, t.trunk_in);
  alpha_post(st, D, R, Nc, t.alphav, t.wv, viewdirs, W, t.sigma, t.cond);
  composite_loss(st, R, Nc, zc, rays->directions, t.sigma, t.rgb_logit, target, t.cfg.use_sample_at_infinity, t.cfg.use_white_background, t.rgb_ray, t.wc,
                 t.loss_dev + 0, t.d_rgb_logit, t.d_alpha, t.cot[0], t.lout[0]);
  // ---------------- the fine level's new samples: shared networks only (block B) ----------------
  resample(st, R, Nc, Nf, zc, t.wc, strat, rnd ? rnd->u_rand : nullptr, rnd ? rnd->seed : 0, rnd ? rnd->first_ray : 0, t.zf, t.rs_scratch, z_new, src);
  encode_inputs(st, D, R, Nf, rays->origins, rays->directions, z_new, rays->warp_id, t.cfg.num_warp_embeds, t.theta + t.warp_tbl, t.theta + t.mask_tbl, W,
                t.x + 3 * Mc, t.mask_in + Mc * D.mask_in, t.warp_in + Mc * D.warp_ld, t.hyper_in + Mc * D.hyper_ld);
  fused_forward(t, st, 1, R, Nf, z_new, rays, ex, W, 1, Mc);
  mask_post(st, D, R, Nf, t.mask_logit + Mc, rays->gt_mask, ex->mask_ratio, t.warp_in + Mc * D.warp_ld, t.hyper_in + Mc * D.hyper_ld);
  se3_fwd(st, Mn, t.wv + 6 * Mc, t.x + 3 * Mc, t.xw + 3 * Mc);
  // ---------------- the shared networks' tangent chains, once over the positions ----------------
  if (so && !rev) {
    encode_tangents(st, D, Mt, t.x, W, t.t_warp_in, t.t_hyper_in);
    fused_tangent(t, st, 2, 0, 3 * Mt, t.t_warp_in, D.warp_ld, t.t_wv, 6);
    se3_jvp(st, Mt, t.wv, t.x, t.t_wv, t.t_xw);
    if (nl) fused_tangent(t, st, 1, 0, 3 * Mt, t.t_hyper_in, D.hyper_ld, t.t_wamb, 2);
  }
  // ---------------- coarse level: second-order and auxiliary terms, NerfMLP backward ----------------
  second_order(0, Nc, zc, t.wc, t.x, t.xw, t.wamb, t.wv, t.mask_logit, t.t_xw, t.t_wamb, t.d_t_xw, t.d_t_wamb, t.dxw_reg, t.dwamb_extra, t.d_pm, t.du, t.ghat, rot);
  fused_backward(t, st, 0, 0, Mc, t.d_rgb_logit, 3, t.d_alpha, t.d_trunk_in, D.trunk_in);
  rc.fork(false); wg_nerf(rc, 0);
  trunk_in_bwd(st, D, Mc, t.d_trunk_in, t.xw, t.wamb, W, t.dxw_reg, nl ? t.dwamb_extra : (hreg ? t.dwamb_reg : nullptr), t.dxw, t.dwamb);
  diag_scan_nerf(t, st, Mc, nl, rev ? Mc : 3 * Mc);
  // ---------------- fine level: NerfMLP on the sorted union of the positions ----------------
  gather_rows(st, Mf, src, t.xw, t.wamb, t.wv, xw_f, wamb_f, wv_f);
  gather_cols(st, Mf, src, 1, 3, t.x, x_f);
  gather_cols(st, Mf, src, 1, 1, t.mask_logit, ml_f);
  if (nl && !rev) { gather_cols(st, Mf, src, 3, 3, t.t_xw, t_xw_f); gather_cols(st, Mf, src, 3, 2, t.t_wamb, t_wamb_f); }
  rc.join();
  if (!rc.ok) return t.fail(NERFDS_ENOTSUP, "%s", rc.unsupported_what.c_str());
  Run rf{t, st, Mf};
  fused_forward(t, st, 1, R, S, t.zf, rays, ex, W, 2, 0, xw_f, wamb_f, wv_f);
  trunk_in(st, D, Mf, xw_f, wamb_f, W, t.trunk_in);
  alpha_post(st, D, R, S, t.alphav, wv_f, viewdirs, W, t.sigma, t.cond);
  composite_loss(st, R, S, t.zf, rays->directions, t.sigma, t.rgb_logit, target, t.cfg.use_sample_at_infinity, t.cfg.use_white_background, t.rgb_ray, t.weights,
                 t.loss_dev + 1, t.d_rgb_logit, t.d_alpha, t.cot[1], t.lout[1]);
  second_order(1, S, t.zf, t.weights, x_f, xw_f, wamb_f, wv_f, ml_f, t_xw_f, t_wamb_f, d_t_xw_f, d_t_wamb_f, dxw_reg_f, dwamb_extra_f, d_pm_f, du_f, ghat_f, rot_f);
  fused_backward(t, st, 0, 1, Mf, t.d_rgb_logit, 3, t.d_alpha, t.d_trunk_in, D.trunk_in);
  rf.fork(false); wg_nerf(rf, 1);
  trunk_in_bwd(st, D, Mf, t.d_trunk_in, xw_f, wamb_f, W, dxw_reg_f, nl ? dwamb_extra_f : nullptr, dxw_f, dwamb_f);
  scatter_rows(st, Mf, src, Mc, dxw_f, dwamb_f, t.dxw, t.dwamb);
  scatter_cols(st, Mf, src, Mc, 1, 1, d_pm_f, t.d_pm);
  if (nl && !rev) {
    scatter_cols(st, Mf, src, Mc, 3, 3, d_t_xw_f, t.d_t_xw);
    scatter_cols(st, Mf, src, Mc, 3, 2, d_t_wamb_f, t.d_t_wamb);
    scatter_cols(st, Mf, src, Mc, 1, 6, rot_f, rot);
  }
  if (!ok) return t.fail(NERFDS_ENOTSUP, "%s", what.c_str());
  // ---------------- the backward of the shared networks' tangent chains, once over the positions ----------------
  if (so && !rev) {
    Run rt{t, st, 3 * Mt};
    rt.tan = true;
    if (nl) {
      pick_scale(st, t.d_t_wamb, 3 * Mt * 2, scale_target, t.tan_x_scale, t.tan_slot + 4);
      rt.tan_slot = t.tan_slot + 4;
      fused_tangent_backward(t, st, 1, 0, 3 * Mt, t.d_t_wamb, 2, nullptr, 0, t.tan_slot + 4);
      rt.fork(false);
      rt.head_wgrads(t.hyper_out, t.th16.back(), t.hyper.width, t.d_t_wamb, 2);
      rt.mlp_wgrads(t.hyper, t.t_hyper_in, t.th16, gh);
    }
    se3_jvp_bwd(st, Mt, t.wv, t.x, t.t_wv, t.d_t_xw, nullptr, nullptr, t.d_t_wv, t.dwv_extra, nl ? rot : nullptr);
    if (Mt < Ms) (void)hipMemsetAsync(t.dwv_extra + 6 * Mt, 0, (size_t)6 * (Ms - Mt) * sizeof(float), st);
    pick_scale(st, t.d_t_wv, 3 * Mt * 6, scale_target, t.tan_x_scale, t.tan_slot + 8);
    rt.tan_slot = t.tan_slot + 8;
    fused_tangent_backward(t, st, 2, 0, 3 * Mt, t.d_t_wv, 6, nullptr, 0, t.tan_slot + 8);
    rt.fork(false);
    rt.head_wgrads(t.warp_w, t.tw16.back(), t.warp.width, t.d_t_wv, 6);
    rt.head_wgrads(t.warp_v, t.tw16.back(), t.warp.width, t.d_t_wv + 3, 6);
    rt.mlp_wgrads(t.warp, t.t_warp_in, t.tw16, gw);
    rt.wg_turn = -1;
    if (!rt.ok) return t.fail(NERFDS_ENOTSUP, "%s", rt.unsupported_what.c_str());
  }
  // ---------------- the shared networks' primal backward, once over every position row ----------------
  rf.M = Ms;
  fused_backward(t, st, 1, 1, Ms, t.dwamb, 2, nullptr, t.d_hyper_in, D.hyper_ld);
  rf.fork(false);
  rf.head_wgrads(t.hyper_out, t.hyper_h16.back(), t.hyper.width, t.dwamb, 2);
  rf.mlp_wgrads(t.hyper, t.hyper_in, t.hyper_h16, t.hyper_h);
  se3_bwd(st, Ms, t.wv, t.x, t.dxw, so ? t.dwv_extra : nullptr, t.dwv);
  fused_backward(t, st, 2, 1, Ms, t.dwv, 6, nullptr, t.d_warp_in, D.warp_ld);
  rf.fork(false);
  rf.head_wgrads(t.warp_w, t.warp_h16.back(), t.warp.width, t.dwv, 6);
  rf.head_wgrads(t.warp_v, t.warp_h16.back(), t.warp.width, t.dwv + 3, 6);
  rf.mlp_wgrads(t.warp, t.warp_in, t.warp_h16, t.warp_h);
  shared_in_bwd(st, D, R, Nc, t.d_warp_in, t.d_hyper_in, t.mask_logit, ex->mask_ratio, t.d_pm, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.warp_tbl, t.d_mask_logit);
  shared_in_bwd(st, D, R, Nf, t.d_warp_in + Mc * D.warp_ld, t.d_hyper_in + Mc * D.hyper_ld, t.mask_logit + Mc, ex->mask_ratio, t.d_pm + Mc, rays->warp_id,
                t.cfg.num_warp_embeds, t.grad + t.warp_tbl, t.d_mask_logit + Mc);
  fused_backward(t, st, 3, 1, Ms, t.d_mask_logit, 1, nullptr, t.d_mask_in, D.mask_in);
  mask_in_bwd(st, D, R, Nc, t.d_mask_in, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.mask_tbl);
  mask_in_bwd(st, D, R, Nf, t.d_mask_in + Mc * D.mask_in, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.mask_tbl);
  rf.fork(true);
  rf.head_wgrads(t.mask_out, t.mask_h16.back(), t.mask.width, t.d_mask_logit, 1);
  rf.mlp_wgrads(t.mask, t.mask_in, t.mask_h16, t.mask_h);
  rf.join();
  if (!rf.ok) return t.fail(NERFDS_ENOTSUP, "%s", rf.unsupported_what.c_str());
  return NERFDS_OK;
}

// Background regulariser (training.py:159-183, 468-479): the SE(3) field alone on a batch of points that should not move, each with the GLO row of
// its id and mask 0 (models.py:766-773 apply_warp), loss = weight * mean general_loss(|warp(x) - x|^2).  Runs after the levels on their (now free)
// buffers, layer by layer on the MFMA layer kernels (the three-way split forward of the warp field, as every step's): points as "rays" of one
// sample at depth 0, so the encodings, the GLO gather and the GLO table's gradient are the step's own kernels.
int run_background(nerfds_trainer& t, hipStream_t st, const nerfds_train_objective& o, const Windows& W) {
  const Dims& D = t.D;
  const int64_t B = o.num_background_points;
  const int WW = t.warp.width;
  Run r{t, st, B};
  (void)hipMemsetAsync(t.sigma, 0, (size_t)B * sizeof(float), st);                    // depth 0 and mask logit 0 for every point
  (void)hipMemsetAsync(t.d_hyper_in, 0, (size_t)B * D.hyper_ld * sizeof(float), st);  // the hyper sheet is not part of apply_warp
  encode_inputs(st, D, (int)B, 1, o.background_points, o.background_points, t.sigma, o.background_ids, t.cfg.num_warp_embeds, t.theta + t.warp_tbl,
                t.theta + t.mask_tbl, W, t.x, t.mask_in, t.warp_in, t.hyper_in);
  mask_post(st, D, (int)B, 1, t.sigma, nullptr, 1.0f, t.warp_in, t.hyper_in);         // "assume background has 0 mask"
  r.precise_layers = true;
  r.mlp_fwd(t.warp, t.warp_in, t.warp_h);
  r.dense_fwd(t.warp_w, {{t.warp_h.back(), WW, WW, nullptr, 0, false}}, t.wv, 6, false);
  r.dense_fwd(t.warp_v, {{t.warp_h.back(), WW, WW, nullptr, 0, false}}, t.wv + 3, 6, false);
  r.precise_layers = false;
  se3_fwd(st, B, t.wv, t.x, t.xw);
  background_loss(st, B, t.x, t.xw, o.background_loss_weight, o.background_loss_alpha, o.background_loss_scale, t.terms_dev + 11, t.dxw);
  se3_bwd(st, B, t.wv, t.x, t.dxw, nullptr, t.dwv);
  r.dense_bwd(t.warp_w, {{t.warp_h.back(), WW, WW, t.g0, WW, false}}, t.dwv, 6, nullptr);
  const bool pm = r.dense_bwd(t.warp_v, {{t.warp_h.back(), WW, WW, t.g0, WW, true, t.warp_h.back(), t.grad + t.warp.hidden.back().b}}, t.dwv + 3, 6, nullptr);
  r.mlp_bwd(t.warp, t.warp_in, t.warp_h, t.g0, t.g1, t.d_warp_in, pm);
  shared_in_bwd(st, D, (int)B, 1, t.d_warp_in, t.d_hyper_in, t.sigma, 1.0f, nullptr, o.background_ids, t.cfg.num_warp_embeds, t.grad + t.warp_tbl, t.d_mask_logit);
  if (!r.ok) return t.fail(NERFDS_ENOTSUP, "%s", r.unsupported_what.c_str());
  return NERFDS_OK;
}

int run_level(nerfds_trainer& t, hipStream_t st, int level, int R, int S, const float* z, const nerfds_rays* rays, const float* target,
              const nerfds_extra* ex, const Windows& W, float* weights_out, bool want_sigma_gradient, const Objective* ob,
              float norm_weight) {
  const Dims& D = t.D;
  Run r{t, st, (int64_t)R * S};
  const int64_t M = r.M;
  const float* viewdirs = rays->viewdirs ? rays->viewdirs : rays->directions;
  // ---------------- forward ----------------
  encode_inputs(st, D, R, S, rays->origins, rays->directions, z, rays->warp_id, t.cfg.num_warp_embeds, t.theta + t.warp_tbl, t.theta + t.mask_tbl,
                W, t.x, t.mask_in, t.warp_in, t.hyper_in);
  const MlpP& trunk = t.trunk[level];
  const int TW = trunk.width, VD = 6 * D.vd_bands, NM = 6 * D.nm_bands, CW = VD + NM;
  float* tout = t.trunk_h.back();
  if (t.fused_fwd) {
    // ONE launch evaluates the five networks on the level's samples and writes every hidden layer + head output (train_fwd_kernel.hip
    // train_forward_kernel); what follows only materialises the layer INPUTS the backward pass differentiates through.
    fused_forward(t, st, level, R, S, z, rays, ex, W);
    mask_post(st, D, R, S, t.mask_logit, rays->gt_mask, ex->mask_ratio, t.warp_in, t.hyper_in);
    se3_fwd(st, M, t.wv, t.x, t.xw);
    trunk_in(st, D, M, t.xw, t.wamb, W, t.trunk_in);
    // (layer-by-layer backward: dW of rgb hidden_0 reads the bottleneck output, which the kernel folds away; the fused backward gets
    // the bottleneck's gradients from S = trunk_out^T g_rgb instead, bott_grads)
    if (!t.half_step) r.dense_fwd(t.bott[level], {{tout, TW, TW, nullptr, 0, false}}, t.bottv, TW, false);
    alpha_post(st, D, R, S, t.alphav, t.wv, viewdirs, W, t.sigma, t.cond);
  } else {
  r.mlp_fwd(t.mask, t.mask_in, t.mask_h);
  r.dense_fwd(t.mask_out, {{t.mask_h.back(), t.mask.width, t.mask.width, nullptr, 0, false}}, t.mask_logit, 1, false);
  mask_post(st, D, R, S, t.mask_logit, rays->gt_mask, ex->mask_ratio, t.warp_in, t.hyper_in);
  r.precise_layers = true;
  r.mlp_fwd(t.warp, t.warp_in, t.warp_h);
  r.dense_fwd(t.warp_w, {{t.warp_h.back(), t.warp.width, t.warp.width, nullptr, 0, false}}, t.wv, 6, false);
  r.dense_fwd(t.warp_v, {{t.warp_h.back(), t.warp.width, t.warp.width, nullptr, 0, false}}, t.wv + 3, 6, false);
  r.precise_layers = false;
  se3_fwd(st, M, t.wv, t.x, t.xw);
  r.mlp_fwd(t.hyper, t.hyper_in, t.hyper_h);
  r.dense_fwd(t.hyper_out, {{t.hyper_h.back(), t.hyper.width, t.hyper.width, nullptr, 0, false}}, t.wamb, 2, false);
  trunk_in(st, D, M, t.xw, t.wamb, W, t.trunk_in);
  r.mlp_fwd(trunk, t.trunk_in, t.trunk_h);
  r.dense_fwd(t.bott[level], {{tout, TW, TW, nullptr, 0, false}}, t.bottv, TW, false);          // modules.py:255 (no activation)
  r.dense_fwd(t.alpha[level], {{tout, TW, TW, nullptr, 0, false}}, t.alphav, 4, false);          // modules.py:273-274
  alpha_post(st, D, R, S, t.alphav, t.wv, viewdirs, W, t.sigma, t.cond);
  // query_rgb input order [bottleneck | viewdir enc | trunk_output | normal enc] (modules.py:300-310)
  r.dense_fwd(t.rgb_h[level], {{t.bottv, TW, TW, nullptr, 0, false}, {t.cond, CW, VD, nullptr, 0, false}, {tout, TW, TW, nullptr, 0, false},
                               {t.cond + VD, CW, NM, nullptr, 0, false}}, t.rgb_hv, t.rgb_h[level].N, true);
  r.dense_fwd(t.rgb_out[level], {{t.rgb_hv, t.rgb_h[level].N, t.rgb_h[level].N, nullptr, 0, false}}, t.rgb_logit, 3, false);
  }
  composite_loss(st, R, S, z, rays->directions, t.sigma, t.rgb_logit, target, t.cfg.use_sample_at_infinity, t.cfg.use_white_background, t.rgb_ray,
                 weights_out, t.loss_dev + level, t.d_rgb_logit, t.d_alpha, t.cot[level], t.lout[level]);
  if (t.fwd_only) {      // nerfds_trainer_forward: the level's outputs are written (and the weights the resample reads); no backward
    if (!r.ok) return t.fail(NERFDS_ENOTSUP, "%s", r.unsupported_what.c_str());
    return NERFDS_OK;
  }
  if (want_sigma_gradient) {
    if (t.half_step && !t.fused_tan) {      // the layer-by-layer tangent pass reads the primal layers' ReLU masks as fp32 arrays: the f16 activations, widened (their sign is all it uses)
      for (int l = 0; l < t.warp.depth; ++l) expand_half(st, t.warp_h16[l], t.warp_h[l], M * t.warp.width);
      if (!t.tangents_warp_only) {
        for (int l = 0; l < t.hyper.depth; ++l) expand_half(st, t.hyper_h16[l], t.hyper_h[l], M * t.hyper.width);
        for (int l = 0; l < trunk.depth; ++l) expand_half(st, t.trunk_h16[l], t.trunk_h[l], M * trunk.width);
      }
    }
    sigma_gradient(t, r, level, W);
  }
  Objective ob_level;
  if (ob) { ob_level = *ob; if (level != 0) { ob_level.hyper_reg_weight = 0.f; ob_level.elastic_weight = 0.f; } ob = &ob_level; }
  if (ob)     // auxiliary first-order losses: extra upstream gradients for x', the raw normal and the predicted mask
    aux_losses(st, R, S, *ob, z, weights_out, t.x, t.xw, t.alphav, viewdirs, t.mask_logit, rays->gt_mask, t.terms_dev + 4 * level, t.dxw_reg,
               t.d_alpha, t.d_pm, t.wamb, t.terms_dev + 9 + level, t.dwamb_reg, t.terms_dev + 13 + level);
  const bool nl = norm_weight != 0.f;
  // hyper-point regulariser: one more upstream gradient of the ambient coordinates - of the COARSE level only (training.py:461-466 passes
  // use_hyper_reg_loss to the coarse level's _compute_loss_and_stats; the fine level runs with its default, False)
  const bool hreg = ob && ob->hyper_reg_weight != 0.f && level == 0;
  if (nl) norm_loss(st, R, S, norm_weight, weights_out, t.alphav, t.t_alpha, t.wv, t.tn[level], t.terms_dev + 4 * level + 3, t.d_alpha, t.d_t_alpha,
                    t.du, t.ghat);
  // elastic regulariser on the warp Jacobian (the tangent pass's t_xw), coarse level only; like the norm loss it is second order in the warp
  // field: its gradient enters the backward of the tangent pass at d_t_xw (part 3 below)
  const bool el = ob && ob->elastic_weight != 0.f && level == 0;
  const bool so = nl || el;
  if (el && !nl) {            // no norm loss: nothing else writes the tangent gradients of x' or exp_se3's direct terms
    (void)hipMemsetAsync(t.d_t_xw, 0, (size_t)9 * M * sizeof(float), st);
    (void)hipMemsetAsync(t.du, 0, (size_t)3 * M * sizeof(float), st);
    (void)hipMemsetAsync(t.ghat, 0, (size_t)3 * M * sizeof(float), st);
  }
  // Backward of the tangent pass (second order: norm loss, elastic regulariser), BEFORE the primal backward: it reads the primal layers' ReLU masks in
  // the fp32 activation arrays - which the half step's chains are about to overwrite with g - and leaves what the primal backward adds to its
  // own upstream gradients: dxw_reg / dwamb_extra (second-derivative terms of the encodings) and dwv_extra (of exp_se3).
  if (so && t.fused_tan && t.half_step) {
    // Fused: the backward of a tangent chain is the network's data-gradient chain on the tangent rows (train_bwd_kernel.hip: masks of row r / 3, the
    // cotangent's loss scale picked on the device), and its weight gradients are k_wgrad_tr products of the stored f16 tangents with the chain's f16 g -
    // the primal step's kernels, three rows per sample, no bias terms.
    Run rt{t, st, 3 * M};
    rt.tan = true;
    auto as_f = [](const std::vector<uint16_t*>& v) { std::vector<float*> o; for (auto* p : v) o.push_back(reinterpret_cast<float*>(p)); return o; };
    const std::vector<float*> gt = as_f(t.gt16), gh = as_f(t.gh16), gw = as_f(t.gw16);
    const float target = t.scale_target();    // the largest cotangent lands at 2^5: 2^11 of headroom below f16's largest value for what the layers amplify (as g_scale)
    if (nl) {   // part 1: alpha head, trunk, trunk input (adds second-derivative terms to d x', d w)
      pick_scale(st, t.d_t_alpha, 3 * M * 4, target, t.tan_x_scale, t.tan_slot);
      rt.tan_slot = t.tan_slot;
      fused_tangent_backward(t, st, 4, level, 3 * M, t.d_t_alpha, 4, t.d_t_tin, D.trunk_in, t.tan_slot);
      rt.fork(false);
      rt.head_wgrads(t.alpha[level], t.tt16.back(), trunk.width, t.d_t_alpha, 4);
      rt.mlp_wgrads(trunk, t.t_tin, t.tt16, gt);
      trunk_in_jvp_bwd(st, D, M, t.d_t_tin, t.xw, t.wamb, t.t_xw, t.t_wamb, W, t.d_t_xw, t.d_t_wamb, t.dxw_reg, t.dwamb_extra);
      if (hreg) add_inplace(st, t.dwamb_extra, t.dwamb_reg, 2 * M);
    }
    if (el) elastic_loss(st, R, S, ob->elastic_weight, ob->elastic_by_weight, weights_out, t.t_xw, t.terms_dev + 12, t.d_t_xw);
    if (nl) {   // part 2: hyper sheet tangents
      pick_scale(st, t.d_t_wamb, 3 * M * 2, target, t.tan_x_scale, t.tan_slot + 4);
      rt.tan_slot = t.tan_slot + 4;
      fused_tangent_backward(t, st, 1, level, 3 * M, t.d_t_wamb, 2, nullptr, 0, t.tan_slot + 4);
      rt.fork(false);
      rt.head_wgrads(t.hyper_out, t.th16.back(), t.hyper.width, t.d_t_wamb, 2);
      rt.mlp_wgrads(t.hyper, t.t_hyper_in, t.th16, gh);
    }
    // part 3: exp_se3 tangents (second derivatives) and the warp net's tangents
    se3_jvp_bwd(st, M, t.wv, t.x, t.t_wv, t.d_t_xw, t.du, t.ghat, t.d_t_wv, t.dwv_extra);
    pick_scale(st, t.d_t_wv, 3 * M * 6, target, t.tan_x_scale, t.tan_slot + 8);
    rt.tan_slot = t.tan_slot + 8;
    fused_tangent_backward(t, st, 2, level, 3 * M, t.d_t_wv, 6, nullptr, 0, t.tan_slot + 8);
    rt.fork(false);
    rt.head_wgrads(t.warp_w, t.tw16.back(), t.warp.width, t.d_t_wv, 6);
    rt.head_wgrads(t.warp_v, t.tw16.back(), t.warp.width, t.d_t_wv + 3, 6);
    rt.mlp_wgrads(t.warp, t.t_warp_in, t.tw16, gw);
    rt.wg_turn = -1;                          // (the side streams are joined with the primal backward's, below)
    if (!rt.ok) return t.fail(NERFDS_ENOTSUP, "%s", rt.unsupported_what.c_str());
  } else if (so) {
    const float* tout_m = t.trunk_h.back();
    bool pm2;
    if (nl) {   // part 1: alpha head, trunk, trunk input (adds second-derivative terms to d x', d w)
      pm2 = r.dense_jvp_bwd(t.alpha[level], {{t.tt_h.back(), trunk.width, trunk.width, t.tA, trunk.width, false, tout_m, nullptr}}, t.d_t_alpha, 4, nullptr);
      r.mlp_jvp_bwd(trunk, t.t_tin, t.tt_h, t.trunk_h, t.tA, t.tB, t.d_t_tin, pm2);
      trunk_in_jvp_bwd(st, D, M, t.d_t_tin, t.xw, t.wamb, t.t_xw, t.t_wamb, W, t.d_t_xw, t.d_t_wamb, t.dxw_reg, t.dwamb_extra);
      if (hreg) add_inplace(st, t.dwamb_extra, t.dwamb_reg, 2 * M);      // both extra gradients of the ambient coordinates in one array
    }
    if (el) elastic_loss(st, R, S, ob->elastic_weight, ob->elastic_by_weight, weights_out, t.t_xw, t.terms_dev + 12, t.d_t_xw);   // (+= : behind trunk_in_jvp_bwd's write)
    if (nl) {   // part 2: hyper sheet tangents
      pm2 = r.dense_jvp_bwd(t.hyper_out, {{t.th_h.back(), t.hyper.width, t.hyper.width, t.tA, t.hyper.width, false, t.hyper_h.back(), nullptr}}, t.d_t_wamb, 2, nullptr);
      r.mlp_jvp_bwd(t.hyper, t.t_hyper_in, t.th_h, t.hyper_h, t.tA, t.tB, nullptr, pm2);
    }
    // part 3: exp_se3 tangents (second derivatives) and the warp net's tangents
    se3_jvp_bwd(st, M, t.wv, t.x, t.t_wv, t.d_t_xw, t.du, t.ghat, t.d_t_wv, t.dwv_extra);
    r.dense_jvp_bwd(t.warp_w, {{t.tw_h.back(), t.warp.width, t.warp.width, t.tA, t.warp.width, false}}, t.d_t_wv, 6, nullptr);
    pm2 = r.dense_jvp_bwd(t.warp_v, {{t.tw_h.back(), t.warp.width, t.warp.width, t.tA, t.warp.width, true, t.warp_h.back(), nullptr}}, t.d_t_wv + 3, 6, nullptr);
    r.mlp_jvp_bwd(t.warp, t.t_warp_in, t.tw_h, t.warp_h, t.tA, t.tB, nullptr, pm2);
  }
  // ---------------- backward ----------------
  const int RW = t.rgb_h[level].N;
  if (t.half_step) {
    // Fused backward: ONE launch per network walks its data-gradient chain and leaves g_l of every hidden layer (in the fp32
    // activation arrays, unused in this mode) and the gradient of the raw input; one weight-gradient launch per layer segment then
    // reads X (f16) and g_l once.  Between the chains: the element-wise backward of the encodings, exp_se3 and the mask blend.
    // Each network's weight gradients are forked off as soon as its chain has run (env NERFDS_TRAIN_EARLY_FORK=0: all after the last chain).
    static const bool early = !(getenv("NERFDS_TRAIN_EARLY_FORK") && std::string(getenv("NERFDS_TRAIN_EARLY_FORK")) == "0");
    const LayerP& K = t.rgb_h[level];
    auto wg_nerf = [&] {   // rgb branch: heads on rgb hidden / trunk_out, rgb hidden_0 = [bottleneck | viewdir | trunk_out | normal] rows
      r.head_wgrads(t.rgb_out[level], t.rgb_h16, RW, t.d_rgb_logit, 3);
      r.head_wgrads(t.alpha[level], t.trunk_h16.back(), TW, t.d_alpha, 4);
      r.weight_grad(reinterpret_cast<const float*>(t.trunk_h16.back()), TW, TW, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)(TW + VD) * RW, -1, true, t.grad + K.b, t.g16);   // S
      r.weight_grad(t.cond, CW, VD, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)TW * RW, -1, false, nullptr, t.g16);
      r.weight_grad(t.cond + VD, CW, NM, t.rgb_hv, RW, RW, t.grad + K.w + (int64_t)(2 * TW + VD) * RW, -1, false, nullptr, t.g16);
      r.mlp_wgrads(trunk, t.trunk_in, t.trunk_h16, t.trunk_h);
    };
    auto wg_hyper = [&] {
      r.head_wgrads(t.hyper_out, t.hyper_h16.back(), t.hyper.width, t.dwamb, 2);
      r.mlp_wgrads(t.hyper, t.hyper_in, t.hyper_h16, t.hyper_h);
    };
    auto wg_warp = [&] {
      r.head_wgrads(t.warp_w, t.warp_h16.back(), t.warp.width, t.dwv, 6);
      r.head_wgrads(t.warp_v, t.warp_h16.back(), t.warp.width, t.dwv + 3, 6);
      r.mlp_wgrads(t.warp, t.warp_in, t.warp_h16, t.warp_h);
    };
    auto wg_mask = [&] {
      r.head_wgrads(t.mask_out, t.mask_h16.back(), t.mask.width, t.d_mask_logit, 1);
      r.mlp_wgrads(t.mask, t.mask_in, t.mask_h16, t.mask_h);
    };
    fused_backward(t, st, 0, level, M, t.d_rgb_logit, 3, t.d_alpha, t.d_trunk_in, D.trunk_in);
    if (early) { r.fork(false); wg_nerf(); }
    trunk_in_bwd(st, D, M, t.d_trunk_in, t.xw, t.wamb, W, ob ? t.dxw_reg : nullptr, nl ? t.dwamb_extra : (hreg ? t.dwamb_reg : nullptr), t.dxw, t.dwamb);
    fused_backward(t, st, 1, level, M, t.dwamb, 2, nullptr, t.d_hyper_in, D.hyper_ld);
    if (early) { r.fork(false); wg_hyper(); }
    se3_bwd(st, M, t.wv, t.x, t.dxw, so ? t.dwv_extra : nullptr, t.dwv);
    fused_backward(t, st, 2, level, M, t.dwv, 6, nullptr, t.d_warp_in, D.warp_ld);
    if (early) { r.fork(false); wg_warp(); }
    shared_in_bwd(st, D, R, S, t.d_warp_in, t.d_hyper_in, t.mask_logit, ex->mask_ratio, ob ? t.d_pm : nullptr, rays->warp_id, t.cfg.num_warp_embeds,
                  t.grad + t.warp_tbl, t.d_mask_logit);
    fused_backward(t, st, 3, level, M, t.d_mask_logit, 1, nullptr, t.d_mask_in, D.mask_in);
    mask_in_bwd(st, D, R, S, t.d_mask_in, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.mask_tbl);
    r.fork(true);
    if (!early) { wg_nerf(); wg_hyper(); wg_warp(); }
    wg_mask();
    r.join();
    if (!r.ok) return t.fail(NERFDS_ENOTSUP, "%s", r.unsupported_what.c_str());
    return NERFDS_OK;
  }
  // (the last writer of a ReLU layer's output gradient masks it and adds its bias gradient: Seg::dx_relu_y / dx_bias_grad)
  bool pm = r.dense_bwd(t.rgb_out[level], {{t.rgb_hv, RW, RW, t.g0, RW, false, t.rgb_hv, t.grad + t.rgb_h[level].b}}, t.d_rgb_logit, 3, nullptr);
  r.dense_bwd(t.rgb_h[level], {{t.bottv, TW, TW, t.g1, TW, false}, {t.cond, CW, VD, nullptr, 0, false}, {tout, TW, TW, t.g2, TW, false},
                               {t.cond + VD, CW, NM, nullptr, 0, false}}, t.g0, RW, t.rgb_hv, pm);
  r.dense_bwd(t.bott[level], {{tout, TW, TW, t.g2, TW, true}}, t.g1, TW, nullptr);
  pm = r.dense_bwd(t.alpha[level], {{tout, TW, TW, t.g2, TW, true, tout, t.grad + trunk.hidden.back().b}}, t.d_alpha, 4, nullptr);
  r.mlp_bwd(trunk, t.trunk_in, t.trunk_h, t.g2, t.g0, t.d_trunk_in, pm);
  trunk_in_bwd(st, D, M, t.d_trunk_in, t.xw, t.wamb, W, ob ? t.dxw_reg : nullptr, nl ? t.dwamb_extra : (hreg ? t.dwamb_reg : nullptr), t.dxw, t.dwamb);
  pm = r.dense_bwd(t.hyper_out, {{t.hyper_h.back(), t.hyper.width, t.hyper.width, t.g0, t.hyper.width, false, t.hyper_h.back(), t.grad + t.hyper.hidden.back().b}}, t.dwamb, 2, nullptr);
  r.mlp_bwd(t.hyper, t.hyper_in, t.hyper_h, t.g0, t.g1, t.d_hyper_in, pm);
  se3_bwd(st, M, t.wv, t.x, t.dxw, so ? t.dwv_extra : nullptr, t.dwv);
  r.dense_bwd(t.warp_w, {{t.warp_h.back(), t.warp.width, t.warp.width, t.g0, t.warp.width, false}}, t.dwv, 6, nullptr);
  pm = r.dense_bwd(t.warp_v, {{t.warp_h.back(), t.warp.width, t.warp.width, t.g0, t.warp.width, true, t.warp_h.back(), t.grad + t.warp.hidden.back().b}}, t.dwv + 3, 6, nullptr);
  r.mlp_bwd(t.warp, t.warp_in, t.warp_h, t.g0, t.g1, t.d_warp_in, pm);
  shared_in_bwd(st, D, R, S, t.d_warp_in, t.d_hyper_in, t.mask_logit, ex->mask_ratio, ob ? t.d_pm : nullptr, rays->warp_id, t.cfg.num_warp_embeds,
                t.grad + t.warp_tbl, t.d_mask_logit);
  pm = r.dense_bwd(t.mask_out, {{t.mask_h.back(), t.mask.width, t.mask.width, t.g0, t.mask.width, false, t.mask_h.back(), t.grad + t.mask.hidden.back().b}}, t.d_mask_logit, 1, nullptr);
  r.mlp_bwd(t.mask, t.mask_in, t.mask_h, t.g0, t.g1, t.d_mask_in, pm);
  mask_in_bwd(st, D, R, S, t.d_mask_in, rays->warp_id, t.cfg.num_warp_embeds, t.grad + t.mask_tbl);
  if (!r.ok) return t.fail(NERFDS_ENOTSUP, "%s", r.unsupported_what.c_str());
  return NERFDS_OK;
}

}  // namespace

extern "C" {

const char* nerfds_trainer_last_error(const nerfds_trainer* t) { return t ? t->err.c_str() : g_train_error.c_str(); }

int nerfds_trainer_create(nerfds_trainer** out, int device, const nerfds_model_cfg* c, int64_t max_rays) {
  if (!out || !c || max_rays <= 0) { g_train_error = "null / invalid argument"; return NERFDS_EINVAL; }
  *out = nullptr;
  if (c->abi_version != NERFDS_ABI_VERSION) { g_train_error = "abi_version mismatch"; return NERFDS_EINVAL; }
  // the graph of configs/nerf_ds.gin (the one BASELINE config 4 names); anything else has no training path yet
  if (!(c->use_warp && c->use_hyper_sheet && c->use_predicted_mask && c->predict_norm && c->use_x_in_rgb_condition && c->use_mask_in_warp &&
        c->use_mask_in_hyper && c->use_viewdirs && c->mask_output_relu && c->nerf_rgb_branch_depth == 1 && c->glo_num_dims == 8 &&
        c->hyper_num_dims == 2 && c->num_warp_embeds > 0 && c->num_coarse_samples >= 4 && c->num_fine_samples >= 0)) {
    g_train_error = "training step is built for the configs/nerf_ds.gin graph only";
    return NERFDS_ENOTSUP;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { g_train_error = "no such HIP device (no CPU fallback)"; return NERFDS_EDEVICE; }
  if (hipSetDevice(device) != hipSuccess) { g_train_error = "hipSetDevice failed"; return NERFDS_EDEVICE; }
  std::unique_ptr<nerfds_trainer> t(new nerfds_trainer);
  t->device = device; t->cfg = *c; t->max_rays = max_rays;
  Dims& D = t->D;
  D.mask_bands = c->mask_max_deg; D.warp_bands = c->warp_max_deg; D.hyp_bands = c->hyper_sheet_max_deg; D.sp_bands = c->spatial_point_max_deg;
  D.hp_bands = c->hyper_point_max_deg; D.vd_bands = c->viewdir_max_deg; D.nm_bands = c->norm_input_max_deg;
  if (std::max({D.mask_bands, D.warp_bands, D.hyp_bands, D.sp_bands, D.hp_bands, D.vd_bands, D.nm_bands}) > 8) { g_train_error = "more than 8 posenc bands"; return NERFDS_ENOTSUP; }
  D.mask_in = 6 * D.mask_bands + 8; D.warp_in = 6 * D.warp_bands + 8 + 1; D.hyper_in = 6 * D.hyp_bands + 8 + 1; D.trunk_in = 6 * D.sp_bands + 4 * D.hp_bands;
  D.warp_ld = (D.warp_in + 3) & ~3; D.hyper_ld = (D.hyper_in + 3) & ~3;
  // flat parameter vector: leaves in this order, names = the Flax paths (params.py)
  t->warp_tbl = add_leaf(*t, "warp_embed/embed/embedding", c->num_warp_embeds, 8);
  t->warp = add_mlp(*t, "warp_field/trunk", D.warp_in, c->warp_trunk_width, c->warp_trunk_depth, c->warp_skip);
  t->warp.in_ld = D.warp_ld;
  t->warp_w = add_dense(*t, "warp_field/branches_w/logit", c->warp_trunk_width, 3);
  t->warp_v = add_dense(*t, "warp_field/branches_v/logit", c->warp_trunk_width, 3);
  t->mask_tbl = add_leaf(*t, "mask_embed/embed/embedding", c->num_warp_embeds, 8);
  t->mask = add_mlp(*t, "mask_mlp/MLP_0", D.mask_in, c->mask_width, c->mask_depth, c->mask_skip);
  t->mask_out = add_dense(*t, "mask_mlp/MLP_0/logit", c->mask_width, 1);
  t->hyper = add_mlp(*t, "hyper_sheet_mlp/MLP_0", D.hyper_in, c->hyper_sheet_width, c->hyper_sheet_depth, c->hyper_sheet_skip);
  t->hyper.in_ld = D.hyper_ld;
  t->hyper_out = add_dense(*t, "hyper_sheet_mlp/MLP_0/logit", c->hyper_sheet_width, 2);
  const int levels = c->num_fine_samples > 0 ? 2 : 1;
  const int TW = c->nerf_trunk_width, cond = 6 * D.vd_bands + 6 * D.nm_bands;
  for (int lv = 0; lv < levels; ++lv) {
    const std::string pre = lv ? "nerf_mlps_fine" : "nerf_mlps_coarse";
    t->trunk[lv] = add_mlp(*t, pre + "/trunk_mlp", D.trunk_in, TW, c->nerf_trunk_depth, c->nerf_skip);
    t->bott[lv] = add_dense(*t, pre + "/bottleneck", TW, TW);
    t->alpha[lv] = add_dense(*t, pre + "/alpha_mlp/logit", TW, 4);
    t->rgb_h[lv] = add_dense(*t, pre + "/rgb_mlp/hidden_0", 2 * TW + cond, c->nerf_rgb_branch_width);
    t->rgb_out[lv] = add_dense(*t, pre + "/rgb_mlp/logit", c->nerf_rgb_branch_width, 3);
  }
  carve(*t);      // sizes only
  const size_t pbytes = (size_t)t->P * sizeof(float);
  if (hipMalloc(&t->theta, pbytes) != hipSuccess || hipMalloc(&t->grad, pbytes) != hipSuccess || hipMalloc(&t->m1, pbytes) != hipSuccess ||
      hipMalloc(&t->m2, pbytes) != hipSuccess || hipMalloc(&t->loss_dev, 4 * sizeof(float)) != hipSuccess || hipMalloc(&t->adam_dev, 16) != hipSuccess || hipMalloc(&t->terms_dev, 16 * sizeof(float)) != hipSuccess ||
      hipMalloc(&t->ws, t->ws_floats * sizeof(float)) != hipSuccess) {
    g_train_error = "hipMalloc failed (workspace of " + std::to_string(t->ws_floats * 4 >> 20) + " MiB)";
    return NERFDS_ENOMEM;
  }
  (void)hipMemset(t->terms_dev, 0, 16 * sizeof(float));      // [8]: the non-finite-gradient flag (adam_update)
  (void)hipMemset(t->adam_dev, 0, 16);
  (void)hipMemset(t->theta, 0, pbytes); (void)hipMemset(t->m1, 0, pbytes); (void)hipMemset(t->m2, 0, pbytes); (void)hipMemset(t->grad, 0, pbytes);
  carve(*t);
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, t->device) == hipSuccess && prop.multiProcessorCount > 0) t->num_cus = prop.multiProcessorCount;
    const char* fb = getenv("NERFDS_TRAIN_FUSE_BWD");
    t->fuse_bwd = !(fb && std::string(fb) == "0");
    const char* ff = getenv("NERFDS_TRAIN_FUSED_FWD");
    t->fused_fwd = !(ff && std::string(ff) == "0") && build_fused_forward(*t);
    const char* fbw = getenv("NERFDS_TRAIN_FUSED_BWD");
    t->fused_bwd = t->fused_fwd && !(fbw && std::string(fbw) == "0") && build_fused_backward(*t);
    const char* g16 = getenv("NERFDS_TRAIN_G16");
    t->g16 = !(g16 && std::string(g16) == "0");
    if (hipMalloc(&t->arena, ARENA_BYTES) != hipSuccess) { g_train_error = "hipMalloc failed (fragment arena)"; return NERFDS_ENOMEM; }
    if (hipMalloc(&t->grad_rep, (size_t)GRAD_REPS * t->P * sizeof(float)) != hipSuccess) { g_train_error = "hipMalloc failed (gradient replicas)"; return NERFDS_ENOMEM; }
    const char* ss = getenv("NERFDS_TRAIN_SIDE_STREAMS");
    int want_side = ss ? atoi(ss) : 3;      // (2 / 3 / 4 / 6 side streams: 13.3 / 13.0 / 13.3 / 13.2 ms per step - the weight gradients are bound by their CU time, not by how many run at once)
    want_side = want_side < 0 ? 0 : (want_side > nerfds_trainer::SIDE ? nerfds_trainer::SIDE : want_side);
    if (t->fused_bwd && want_side > 0) {
      bool ok = hipEventCreateWithFlags(&t->fork_ev, hipEventDisableTiming) == hipSuccess;
      t->nside = t->nside_eff = want_side;
      // The side streams at the device's LEAST priority (NERFDS_TRAIN_SIDE_PRIO=0: default priority; A/B), so that the workgroups of the chains / element-wise kernels
      // on the caller's stream - the step's critical path - are dispatched ahead of queued weight-gradient workgroups
      const char* sp = getenv("NERFDS_TRAIN_SIDE_PRIO");
      const bool side_prio = !(sp && std::string(sp) == "0");
      int prio_least = 0, prio_greatest = 0;
      if (side_prio) (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
      for (int i = 0; i < t->nside && ok; ++i)
      {
        // (a runtime that refuses the priority gets a plain stream: the priority is scheduling advice, not something the step's correctness rests on)
        if (!(side_prio && hipStreamCreateWithPriority(&t->side[i], hipStreamNonBlocking, prio_least) == hipSuccess)) {
          (void)hipGetLastError();
          ok = hipStreamCreateWithFlags(&t->side[i], hipStreamNonBlocking) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&t->join_ev[i], hipEventDisableTiming) == hipSuccess;
      }
      if (!ok) { g_train_error = "hipStreamCreate failed (weight-gradient side streams)"; return NERFDS_EDEVICE; }
    }
    if (hipMalloc(&t->wpack, WPACK_BYTES + 256) != hipSuccess || hipMemset(t->wpack, 0, WPACK_BYTES + 256) != hipSuccess) { g_train_error = "hipMalloc failed (weight fragments)"; return NERFDS_ENOMEM; }
  }
  *out = t.release();
  return NERFDS_OK;
}

int nerfds_trainer_destroy(nerfds_trainer* t) {
  if (!t) return NERFDS_OK;
  (void)hipSetDevice(t->device);
  delete t;
  return NERFDS_OK;
}

int64_t nerfds_trainer_param_count(const nerfds_trainer* t) { return t ? t->P : -1; }
int nerfds_trainer_num_leaves(const nerfds_trainer* t) { return t ? (int)t->leaves.size() : -1; }
int nerfds_trainer_leaf(const nerfds_trainer* t, int index, char* name, int name_cap, int64_t* offset, int32_t* rows, int32_t* cols) {
  if (!t || index < 0 || index >= (int)t->leaves.size()) return NERFDS_EINVAL;
  const Leaf& l = t->leaves[index];
  if (name && name_cap > 0) { std::strncpy(name, l.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = l.off;
  if (rows) *rows = l.rows;
  if (cols) *cols = l.cols;
  return NERFDS_OK;
}
float* nerfds_trainer_params(nerfds_trainer* t) { return t ? t->theta : nullptr; }
float* nerfds_trainer_grads(nerfds_trainer* t) { return t ? t->grad : nullptr; }

// which: 0 parameters, 1 gradients, 2 / 3 Adam first / second moment
static float* vec_of(nerfds_trainer* t, int which) { return which == 0 ? t->theta : which == 1 ? t->grad : which == 2 ? t->m1 : which == 3 ? t->m2 : nullptr; }
int nerfds_trainer_download(nerfds_trainer* t, int which, float* host) {
  if (!t || !host || !vec_of(t, which)) return NERFDS_EINVAL;
  (void)hipSetDevice(t->device);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, vec_of(t, which), (size_t)t->P * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "download failed");
  return NERFDS_OK;
}
int nerfds_trainer_upload(nerfds_trainer* t, int which, const float* host) {
  if (!t || !host || !vec_of(t, which)) return NERFDS_EINVAL;
  (void)hipSetDevice(t->device);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(vec_of(t, which), host, (size_t)t->P * 4, hipMemcpyHostToDevice) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "upload failed");
  return NERFDS_OK;
}

int nerfds_trainer_target_norm(nerfds_trainer* t, int level, int64_t num_rays, float* host) {
  if (!t || !host || level < 0 || level > 1 || num_rays <= 0 || num_rays > t->max_rays) return NERFDS_EINVAL;
  if (!t->tn_valid || !t->tws) return t->fail(NERFDS_EINVAL, "the last step did not run with NERFDS_TRAIN_SIGMA_GRAD");
  if (level == 1 && t->cfg.num_fine_samples == 0) return t->fail(NERFDS_EINVAL, "no fine level");
  const int64_t S = t->cfg.num_coarse_samples + (level ? t->cfg.num_fine_samples : 0);
  (void)hipSetDevice(t->device);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, t->tn[level], (size_t)num_rays * S * 3 * 4, hipMemcpyDeviceToHost) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "download failed");
  return NERFDS_OK;
}

// Development / tests: copies an internal buffer of the last step to the host.  name: "<net>_h16_<l>", "<net>_bits_<l>", "<net>_g_<l>"
// (net = mask | warp | hyper | trunk, l = layer), "rgb_h16", "rgb_bits", "rgb_g", "d_rgb_logit", "d_alpha", "d_trunk_in", "d_hyper_in",
// "d_warp_in", "d_mask_in", "dwamb", "dwv", "d_mask_logit".  Returns the number of bytes copied (<= max_bytes) or a negative error.
long long nerfds_trainer_debug_read(nerfds_trainer* t, const char* name, void* host, long long max_bytes) {
  if (!t || !name || !host || max_bytes <= 0) return NERFDS_EINVAL;
  const std::string n(name);
  if (n == "g_scale") {                  // the power of two the last step's stored g carries (host value)
    if (max_bytes < 4) return NERFDS_EINVAL;
    std::memcpy(host, &t->g_scale, 4);
    return 4;
  }
  const void* p = nullptr;
  long long cap = 0;                     // bytes the named view holds for the largest level (rows = max_rays * (Nc + Nf))
  bool half_only = false;                // views only a half step (f16 activations + ReLU bits, fused backward) fills
  const long long M = (long long)t->max_rays * (t->cfg.num_coarse_samples + t->cfg.num_fine_samples);
  const long long gsz = t->g16 ? 2 : 4;  // the g arrays are (scaled) f16 in the default mode, fp32 under NERFDS_TRAIN_G16=0
  auto layer_of = [&](const std::string& pre, int width, const std::vector<uint16_t*>& h16, const std::vector<uint16_t*>& bits, const std::vector<float*>& g) {
    for (size_t l = 0; l < g.size(); ++l) {
      if (n == pre + "_h16_" + std::to_string(l) && l < h16.size()) { p = h16[l]; cap = M * width * 2; half_only = true; }
      if (n == pre + "_bits_" + std::to_string(l) && l < bits.size()) { p = bits[l]; cap = M * 2 * (width / 32) * 2; half_only = true; }
      if (n == pre + "_g_" + std::to_string(l)) { p = g[l]; cap = M * width * gsz; half_only = true; }
    }
  };
  layer_of("mask", t->mask.width, t->mask_h16, t->mask_bits, t->mask_h); layer_of("warp", t->warp.width, t->warp_h16, t->warp_bits, t->warp_h);
  layer_of("hyper", t->hyper.width, t->hyper_h16, t->hyper_bits, t->hyper_h); layer_of("trunk", t->trunk[0].width, t->trunk_h16, t->trunk_bits, t->trunk_h);
  const int RW = t->rgb_h[0].N;
  if (n == "rgb_h16") { p = t->rgb_h16; cap = M * RW * 2; half_only = true; }
  else if (n == "rgb_bits") { p = t->rgb_bits; cap = M * 2 * (RW / 32) * 2; half_only = true; }
  else if (n == "rgb_g") { p = t->rgb_hv; cap = M * RW * gsz; half_only = true; }
  else if (n == "d_rgb_logit") { p = t->d_rgb_logit; cap = M * 3 * 4; }
  else if (n == "d_alpha") { p = t->d_alpha; cap = M * 4 * 4; }
  else if (n == "d_trunk_in") { p = t->d_trunk_in; cap = M * t->D.trunk_in * 4; }
  else if (n == "d_hyper_in") { p = t->d_hyper_in; cap = M * t->D.hyper_ld * 4; }
  else if (n == "d_warp_in") { p = t->d_warp_in; cap = M * t->D.warp_ld * 4; }
  else if (n == "d_mask_in") { p = t->d_mask_in; cap = M * t->D.mask_in * 4; }
  else if (n == "dwamb") { p = t->dwamb; cap = M * 2 * 4; }
  else if (n == "dwv") { p = t->dwv; cap = M * 6 * 4; }
  else if (n == "d_mask_logit") { p = t->d_mask_logit; cap = M * 4; }
  else if (n == "wv") { p = t->wv; cap = M * 6 * 4; }                    // forward state per POSITION row (merged step: [coarse | new samples])
  else if (n == "wamb") { p = t->wamb; cap = M * 2 * 4; }
  else if (n == "xw") { p = t->xw; cap = M * 3 * 4; }
  else if (n == "mask_logit") { p = t->mask_logit; cap = M * 4; }
  else if (n == "sigma") { p = t->sigma; cap = M * 4; }
  else if (n == "rgb_logit") { p = t->rgb_logit; cap = M * 3 * 4; }
  if (!p) return t->fail(NERFDS_EINVAL, "debug_read: no buffer named %s", name);
  if (half_only && !t->half_step)
    return t->fail(NERFDS_EINVAL, "debug_read: %s is written by a half step only (the last step kept fp32 activations: tangent pass or NERFDS_TRAIN_FUSED_*=0)", name);
  if (max_bytes > cap) return t->fail(NERFDS_EINVAL, "debug_read: %s holds %lld bytes (%s elements), %lld asked for", name, cap,
                                      half_only && n.find("_g") != std::string::npos ? (t->g16 ? "f16" : "fp32") : "its own", max_bytes);
  (void)hipSetDevice(t->device);
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, p, (size_t)max_bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "debug_read failed");
  return max_bytes;
}

int nerfds_trainer_set_step(nerfds_trainer* t, int64_t step) {
  if (!t || step < 0) return NERFDS_EINVAL;
  (void)hipSetDevice(t->device);
  const long long v = step;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(t->adam_step(), &v, sizeof v, hipMemcpyHostToDevice) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "step upload failed");
  return NERFDS_OK;
}

int nerfds_trainer_get_step(nerfds_trainer* t, int64_t* step_out) {
  if (!t || !step_out) return NERFDS_EINVAL;
  (void)hipSetDevice(t->device);
  long long v = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, t->adam_step(), sizeof v, hipMemcpyDeviceToHost) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "step read-back failed");
  *step_out = v;
  return NERFDS_OK;
}

int nerfds_trainer_set_loss_scale_adjust(nerfds_trainer* t, int32_t log2_adjust) {
  if (!t || log2_adjust < -40 || log2_adjust > 16) return NERFDS_EINVAL;
  t->g_scale_adjust = log2_adjust;
  return NERFDS_OK;
}

int nerfds_trainer_set_numerics(nerfds_trainer* t, const nerfds_train_numerics* n) {
  if (!t || !n) return NERFDS_EINVAL;
  if (n->loss_scale_log2_adjust < -40 || n->loss_scale_log2_adjust > 16 || n->tangent_scale_log2_adjust < -24 || n->tangent_scale_log2_adjust > 12 ||
      (n->chain_arith != NERFDS_CHAINS_DEFAULT && n->chain_arith != NERFDS_CHAINS_SPLIT_BF16) || (n->fp32_step != 0 && n->fp32_step != 1) || (n->diagnose != 0 && n->diagnose != 1))
    return t->fail(NERFDS_EINVAL, "nerfds_train_numerics: loss_scale_log2_adjust in [-40, 16], tangent_scale_log2_adjust in [-24, 12], chain_arith 0 / 1, fp32_step 0 / 1, diagnose 0 / 1");
  t->g_scale_adjust = n->loss_scale_log2_adjust;
  t->tan_scale_adjust = n->tangent_scale_log2_adjust;
  t->chain_arith = n->chain_arith;
  t->fp32_step = n->fp32_step != 0;
  t->diag = n->diagnose != 0;
  return NERFDS_OK;
}

int nerfds_trainer_get_numerics(const nerfds_trainer* t, nerfds_train_numerics* n) {
  if (!t || !n) return NERFDS_EINVAL;
  n->loss_scale_log2_adjust = t->g_scale_adjust; n->tangent_scale_log2_adjust = t->tan_scale_adjust; n->chain_arith = t->chain_arith; n->fp32_step = t->fp32_step ? 1 : 0; n->diagnose = t->diag ? 1 : 0;
  return NERFDS_OK;
}

// Which stored array of the LAST step holds an inf / NaN (called by the host after NERFDS_ENONFINITE / nerfds_trainer_nonfinite, never inside a step: the
// scans cost a pass over the workspace).  Rows: what the last step wrote (last_R rays; the tangent arrays up to the rows its chains ran on).
int nerfds_trainer_overflow_sources(nerfds_trainer* t, uint32_t* mask_out) {
  if (!t || !mask_out) return NERFDS_EINVAL;
  *mask_out = 0;
  if (hipSetDevice(t->device) != hipSuccess) return t->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  if (hipDeviceSynchronize() != hipSuccess) return t->fail(NERFDS_EDEVICE, "device synchronisation failed");
  if (t->last_R <= 0) return NERFDS_OK;
  unsigned* flags = reinterpret_cast<unsigned*>(t->terms_dev + 15);      // (cleared at the start of the step; a diagnose step has OR-ed the coarse level's in)
  hipStream_t st = nullptr;
  const int Nc = t->cfg.num_coarse_samples, Nf = t->cfg.num_fine_samples;
  const int64_t M = t->last_R * (int64_t)(Nc + Nf);
  const int RW = t->rgb_h[0].N, TW = t->trunk[0].width;
  auto r8 = [](int64_t n) { return n & ~(int64_t)7; };
  if (t->last_half) {
    auto net16 = [&](const std::vector<uint16_t*>& h, int width) { for (auto* p : h) if (p) scan_half(st, p, r8(M * width), flags, NERFDS_OVF_ACTIVATION); };
    net16(t->mask_h16, t->mask.width); net16(t->warp_h16, t->warp.width); net16(t->hyper_h16, t->hyper.width); net16(t->trunk_h16, TW);
    if (t->rgb_h16) scan_half(st, t->rgb_h16, r8(M * RW), flags, NERFDS_OVF_ACTIVATION);
    // the stored g of the primal chains: (loss-scaled) f16 in the first half of the fp32 activation arrays, or fp32 (NERFDS_TRAIN_G16=0)
    auto netg = [&](const std::vector<float*>& g, int width) {
      for (auto* p : g) {
        if (!p) continue;
        if (t->g16) scan_half(st, reinterpret_cast<const uint16_t*>(p), r8(M * width), flags, NERFDS_OVF_PRIMAL_G);
        else scan_float(st, p, M * width, flags, NERFDS_OVF_PRIMAL_G);
      }
    };
    netg(t->mask_h, t->mask.width); netg(t->warp_h, t->warp.width); netg(t->hyper_h, t->hyper.width); netg(t->trunk_h, TW);
    if (t->g16) scan_half(st, reinterpret_cast<const uint16_t*>(t->rgb_hv), r8(M * RW), flags, NERFDS_OVF_PRIMAL_G);
    else scan_float(st, t->rgb_hv, M * RW, flags, NERFDS_OVF_PRIMAL_G);
  }
  if (t->last_tan16 && t->last_tan_rows > 0) {
    const int64_t T = std::min<int64_t>(t->last_tan_rows, 3 * t->max_rays * (int64_t)(Nc + Nf));
    auto tan16 = [&](const std::vector<uint16_t*>& v, int width, unsigned bit) {
      const uint16_t* prev = nullptr;
      for (auto* p : v) { if (p && p != prev) scan_half(st, p, r8(T * width), flags, bit); prev = p; }
    };
    tan16(t->tw16, t->warp.width, NERFDS_OVF_TANGENT); tan16(t->th16, t->hyper.width, NERFDS_OVF_TANGENT); tan16(t->tt16, TW, NERFDS_OVF_TANGENT);
    if (t->last_keep_tangents && t->gws16) {
      tan16(t->gw16, t->warp.width, NERFDS_OVF_COTANGENT); tan16(t->gh16, t->hyper.width, NERFDS_OVF_COTANGENT); tan16(t->gt16, TW, NERFDS_OVF_COTANGENT);
    }
  }
  // fp32 values no power of two rescues: the losses, the heads' outputs and cotangents, the second-order terms' fp32 cotangents.  Beside the class bit
  // (NERFDS_OVF_FP32 / _FP32_SECOND_ORDER) each array sets its own DETAIL bit (1 << (NERFDS_OVF_DETAIL_SHIFT + k), k in the order below): diagnosis only.
  int k = 0;
  auto f32 = [&](const float* p, int64_t n, unsigned cls) { if (p && n > 0) scan_float(st, p, n, flags, cls | (1u << (NERFDS_OVF_DETAIL_SHIFT + k))); ++k; };
  f32(t->loss_dev, 2, NERFDS_OVF_FP32);                 // 0 rgb losses
  f32(t->terms_dev, 8, NERFDS_OVF_FP32);                // 1 auxiliary terms ...
  f32(t->terms_dev + 9, 6, NERFDS_OVF_FP32); --k;       //   ... (same detail bit)
  f32(t->alphav, M * 4, NERFDS_OVF_FP32);               // 2 alpha head output
  f32(t->rgb_logit, M * 3, NERFDS_OVF_FP32);            // 3 rgb head output
  f32(t->wv, M * 6, NERFDS_OVF_FP32);                   // 4 screw axis
  f32(t->xw, M * 3, NERFDS_OVF_FP32);                   // 5 warped points
  f32(t->d_alpha, M * 4, NERFDS_OVF_FP32_BACKWARD);              // 6 alpha head cotangent
  f32(t->d_rgb_logit, M * 3, NERFDS_OVF_FP32_BACKWARD);          // 7 rgb head cotangent
  f32(t->dwv, M * 6, NERFDS_OVF_FP32_BACKWARD);                  // 8 screw axis cotangent
  f32(t->dxw, M * 3, NERFDS_OVF_FP32_BACKWARD);                  // 9 warped point cotangent
  f32(t->dwamb, M * 2, NERFDS_OVF_FP32_BACKWARD);                // 10 ambient coordinate cotangent
  f32(t->d_trunk_in, M * t->D.trunk_in, NERFDS_OVF_FP32_BACKWARD);   // 11 trunk input cotangent
  if (t->last_keep_tangents && t->nws && t->tws) {
    const int64_t T = std::min<int64_t>(std::max<int64_t>(t->last_tan_rows, M), 3 * M);
    scan_float(st, t->t_tin, T * t->D.trunk_in, flags, NERFDS_OVF_TANGENT, 65504.f / t->tan_x_scale);      // raw tangent inputs: f16 at the tangents' scale in their weight gradients
    scan_float(st, t->t_warp_in, T * t->D.warp_ld, flags, NERFDS_OVF_TANGENT, 65504.f / t->tan_x_scale);
    scan_float(st, t->t_hyper_in, T * t->D.hyper_ld, flags, NERFDS_OVF_TANGENT, 65504.f / t->tan_x_scale);
    f32(t->t_alpha, T * 4, NERFDS_OVF_FP32_SECOND_ORDER);       // 12 d (sigma_raw, n) / d x
    f32(t->d_t_alpha, T * 4, NERFDS_OVF_FP32_SECOND_ORDER);     // 13 its cotangent (norm loss)
    f32(t->d_t_tin, T * t->D.trunk_in, NERFDS_OVF_FP32_SECOND_ORDER);   // 14
    f32(t->d_t_wv, T * 6, NERFDS_OVF_FP32_SECOND_ORDER);        // 15
    f32(t->dwv_extra, M * 6, NERFDS_OVF_FP32_SECOND_ORDER);     // 16 exp_se3's second-derivative terms
    f32(t->du, M * 3, NERFDS_OVF_FP32_SECOND_ORDER);            // 17
    f32(t->ghat, M * 3, NERFDS_OVF_FP32_SECOND_ORDER);          // 18
    f32(t->tan_slot, 16, NERFDS_OVF_FP32_SECOND_ORDER);         // 19 device-picked scales (amax of a non-finite cotangent array)
    f32(t->tn[0], t->last_R * (int64_t)Nc * 3, NERFDS_OVF_FP32_SECOND_ORDER);   // 20 target_norm, coarse
    if (Nf > 0) f32(t->tn[1], M * 3, NERFDS_OVF_FP32_SECOND_ORDER);             // 21 target_norm, fine
  }
  unsigned f = 0;
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipMemcpy(&f, flags, sizeof f, hipMemcpyDeviceToHost) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "overflow scan failed");
  *mask_out = f;
  return NERFDS_OK;
}

int nerfds_trainer_nonfinite(nerfds_trainer* t) {
  if (!t) return NERFDS_EINVAL;
  (void)hipSetDevice(t->device);
  unsigned f = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&f, t->terms_dev + 8, sizeof f, hipMemcpyDeviceToHost) != hipSuccess)
    return t->fail(NERFDS_EDEVICE, "flag read-back failed");
  return f != 0 ? 1 : 0;
}

int nerfds_trainer_reset_optimizer(nerfds_trainer* t) {
  if (!t) return NERFDS_EINVAL;
  (void)hipSetDevice(t->device);
  (void)hipDeviceSynchronize();
  (void)hipMemset(t->m1, 0, (size_t)t->P * 4); (void)hipMemset(t->m2, 0, (size_t)t->P * 4);
  (void)hipMemset(t->adam_dev, 0, 16);
  return NERFDS_OK;
}

static void adam_update(nerfds_trainer* t, float learning_rate, hipStream_t st) {
  unsigned* flag = reinterpret_cast<unsigned*>(t->terms_dev + 8);
  (void)hipMemsetAsync(flag, 0, sizeof(unsigned), st);
  // the step count advances on the device, and only if the update is applied (a skipped update leaves parameters, moments AND the count alone)
  adam(st, t->theta, t->grad, t->m1, t->m2, t->P, learning_rate, 0.9f, 0.999f, 1e-8f, t->adam_step(), t->adam_corr(), flag);
}

int nerfds_trainer_clip_gradients(nerfds_trainer* t, float grad_max_val, float grad_max_norm, void* hip_stream) {
  if (!t) return NERFDS_EINVAL;
  if (!(grad_max_val > 0.f) && !(grad_max_norm > 0.f)) return NERFDS_OK;
  if (hipSetDevice(t->device) != hipSuccess) return t->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  clip_gradients(static_cast<hipStream_t>(hip_stream), t->grad, t->P, grad_max_val, grad_max_norm, t->loss_dev + 2);   // loss_dev[2]: scratch of the norm (the losses in [0], [1] stay readable: overflow diagnosis)
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return t->fail(NERFDS_EDEVICE, "kernel launch failed: %s", hipGetErrorString(e));
  return NERFDS_OK;
}

int nerfds_trainer_apply(nerfds_trainer* t, float learning_rate, void* hip_stream) {
  if (!t) return NERFDS_EINVAL;
  if (hipSetDevice(t->device) != hipSuccess) return t->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  adam_update(t, learning_rate, static_cast<hipStream_t>(hip_stream));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return t->fail(NERFDS_EDEVICE, "kernel launch failed: %s", hipGetErrorString(e));
  return NERFDS_OK;
}

// nerfds_trainer_step, nerfds_trainer_forward and nerfds_render_rays_bwd are one flow: forward + backward of both levels.  target_rgb == nullptr is
// allowed when the caller's cotangents (t->cot) replace the built-in squared error or when only the forward runs (t->fwd_only).
static int step_impl(nerfds_trainer* t, const nerfds_rays* rays, const float* target_rgb, const nerfds_extra* ex, const nerfds_rand* rnd,
                     const nerfds_train_objective* objective, float learning_rate, uint32_t flags, float* loss_host, void* hip_stream);

int nerfds_trainer_step(nerfds_trainer* t, const nerfds_rays* rays, const float* target_rgb, const nerfds_extra* ex, const nerfds_rand* rnd,
                        const nerfds_train_objective* objective, float learning_rate, uint32_t flags, float* loss_host, void* hip_stream) {
  if (!t) return NERFDS_EINVAL;
  if (!target_rgb) return t->fail(NERFDS_EINVAL, "null argument");
  return step_impl(t, rays, target_rgb, ex, rnd, objective, learning_rate, flags, loss_host, hip_stream);
}

namespace {
struct CallScope {      // the per-call fields of a caller-defined loss never outlive the call
  nerfds_trainer* t;
  ~CallScope() { t->cot[0] = t->cot[1] = LevelCot{0, nullptr, nullptr, nullptr}; t->lout[0] = t->lout[1] = LevelOut{nullptr, nullptr, nullptr}; t->fwd_only = false; }
};
}  // namespace

int nerfds_trainer_forward(nerfds_trainer* t, const nerfds_rays* rays, const nerfds_extra* ex, const nerfds_rand* rnd, const nerfds_level_out* fine,
                           const nerfds_level_out* coarse, void* hip_stream) {
  if (!t) return NERFDS_EINVAL;
  CallScope scope{t};
  const bool two = t->cfg.num_fine_samples > 0;
  // a single-level model has only the 'coarse' level (as the reference's out dict): its outputs go where `coarse` points
  if (coarse) t->lout[0] = LevelOut{coarse->rgb, coarse->depth, coarse->acc};
  if (fine && two) t->lout[1] = LevelOut{fine->rgb, fine->depth, fine->acc};
  if (fine && !two && (fine->rgb || fine->depth || fine->acc)) return t->fail(NERFDS_EINVAL, "this model has no fine level (num_fine_samples == 0)");
  t->fwd_only = true;
  const int rc = step_impl(t, rays, nullptr, ex, rnd, nullptr, 0.f, NERFDS_TRAIN_GRADS_ONLY, nullptr, hip_stream);
  return rc;
}

int nerfds_render_rays_bwd(nerfds_trainer* t, const nerfds_rays* rays, const nerfds_extra* ex, const nerfds_rand* rnd, const nerfds_level_cotangent* d_fine,
                           const nerfds_level_cotangent* d_coarse, void* hip_stream) {
  if (!t) return NERFDS_EINVAL;
  CallScope scope{t};
  const bool two = t->cfg.num_fine_samples > 0;
  t->cot[0] = LevelCot{1, d_coarse ? d_coarse->d_rgb : nullptr, d_coarse ? d_coarse->d_depth : nullptr, d_coarse ? d_coarse->d_acc : nullptr};
  t->cot[1] = LevelCot{1, (d_fine && two) ? d_fine->d_rgb : nullptr, (d_fine && two) ? d_fine->d_depth : nullptr, (d_fine && two) ? d_fine->d_acc : nullptr};
  if (d_fine && !two && (d_fine->d_rgb || d_fine->d_depth || d_fine->d_acc)) return t->fail(NERFDS_EINVAL, "this model has no fine level (num_fine_samples == 0)");
  return step_impl(t, rays, nullptr, ex, rnd, nullptr, 0.f, NERFDS_TRAIN_GRADS_ONLY, nullptr, hip_stream);
}

static int step_impl(nerfds_trainer* t, const nerfds_rays* rays, const float* target_rgb, const nerfds_extra* ex, const nerfds_rand* rnd,
                     const nerfds_train_objective* objective, float learning_rate, uint32_t flags, float* loss_host, void* hip_stream) {
  if (!rays || !ex || !rays->origins || !rays->directions || !rays->warp_id) return t->fail(NERFDS_EINVAL, "null argument");
  if (rays->num_rays <= 0 || rays->num_rays > t->max_rays) return t->fail(NERFDS_EINVAL, "num_rays must be in [1, max_rays = %lld]", (long long)t->max_rays);
  if (ex->mask_ratio != 1.0f && !rays->gt_mask) return t->fail(NERFDS_EINVAL, "rays_dict['mask'] is required when mask_ratio != 1");
  // train_step calls model.apply without use_sample_at_infinity / render_opts (training.py:441-455): both levels composite with the model's value
  if (ex->sample_at_infinity_override != NERFDS_TRISTATE_NONE || ex->render_opt_flags != 0)
    return t->fail(NERFDS_ENOTSUP, "the training step takes neither a use_sample_at_infinity override nor render_opts (training.py:441-455 passes neither)");
  if (hipSetDevice(t->device) != hipSuccess) return t->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  hipStream_t st = static_cast<hipStream_t>(hip_stream);
  const int R = (int)rays->num_rays, Nc = t->cfg.num_coarse_samples, Nf = t->cfg.num_fine_samples;
  Windows W;
  window(W.mask, t->D.mask_bands, ex->warp_alpha);          // models.py:967
  window(W.warp, t->D.warp_bands, ex->warp_alpha);
  window(W.hyp, t->D.hyp_bands, ex->hyper_sheet_alpha);
  window(W.sp, t->D.sp_bands, ex->nerf_alpha);
  window(W.hp, t->D.hp_bands, ex->hyper_alpha);
  window(W.nm, t->D.nm_bands, ex->norm_input_alpha);
  (void)hipMemsetAsync(t->grad, 0, (size_t)t->P * 4, st);
  if (t->grad_rep) (void)hipMemsetAsync(t->grad_rep, 0, (size_t)GRAD_REPS * t->P * 4, st);
  if (t->arena) {                      // every fragment pack recorded so far, from the current parameters, in one launch
    ++t->pack_epoch;
    if (!t->packs.empty()) {
      if (t->packs_dev_n != t->packs.size()) {
        if (t->packs_dev) (void)hipFree(t->packs_dev);
        t->packs_dev = nullptr;
        if (hipMalloc(&t->packs_dev, t->packs.size() * sizeof(PackEntry)) != hipSuccess) return t->fail(NERFDS_ENOMEM, "hipMalloc failed (pack table)");
        if (hipMemcpy(t->packs_dev, t->packs.data(), t->packs.size() * sizeof(PackEntry), hipMemcpyHostToDevice) != hipSuccess)
          return t->fail(NERFDS_EDEVICE, "pack table upload failed");
        t->packs_dev_n = t->packs.size();
      }
      pack_frags_all(st, t->theta, t->packs_dev, (int)t->packs.size(), t->max_frag_lanes, t->arena);
      std::fill(t->pack_fresh.begin(), t->pack_fresh.end(), t->pack_epoch);
    }
  }
  (void)hipMemsetAsync(t->loss_dev, 0, 2 * sizeof(float), st);
  const int strat = ex->use_stratified_sampling;
  coarse_z(st, R, Nc, ex->near, ex->far, strat, ex->use_linear_disparity, rnd ? rnd->t_rand : nullptr, rnd ? rnd->seed : 0, rnd ? rnd->first_ray : 0, t->zc);
  Objective ob{};
  const Objective* obp = nullptr;
  if (objective) {
    ob.warp_reg_weight = objective->warp_reg_loss_weight; ob.warp_reg_alpha = objective->warp_reg_loss_alpha; ob.warp_reg_scale = objective->warp_reg_loss_scale;
    ob.back_facing_weight = objective->back_facing_reg_weight; ob.mask_loss_weight = objective->predicted_mask_loss_weight;
    ob.sharp_weights_std = objective->sharp_weights_std; ob.use_sharp_weights = objective->use_mask_sharp_weights;
    ob.hyper_reg_weight = objective->hyper_reg_loss_weight;
    ob.elastic_weight = objective->elastic_loss_weight; ob.elastic_by_weight = objective->elastic_reduce_by_weight;
    ob.mask_occlusion_weight = objective->mask_occlusion_reg_loss_weight;
    if (ob.mask_loss_weight != 0.f && !rays->gt_mask) return t->fail(NERFDS_EINVAL, "the mask loss needs rays_dict['mask']");
    if (ob.use_sharp_weights && !(ob.sharp_weights_std > 0.f)) return t->fail(NERFDS_EINVAL, "sharp_weights_std must be > 0");
    // (a background-only objective keeps the plain - merged - flow of the levels: the per-ray auxiliary kernel has nothing to do)
    if (ob.warp_reg_weight != 0.f || ob.back_facing_weight != 0.f || ob.mask_loss_weight != 0.f || ob.hyper_reg_weight != 0.f || ob.elastic_weight != 0.f || ob.mask_occlusion_weight != 0.f || objective->norm_loss_weight != 0.f)
      obp = &ob;
    if (objective->background_loss_weight != 0.f) {
      if (!objective->background_points || !objective->background_ids || objective->num_background_points < 1)
        return t->fail(NERFDS_EINVAL, "the background loss needs background_points and background_ids");
      if (objective->num_background_points > (int64_t)t->max_rays * (Nc + Nf))
        return t->fail(NERFDS_EINVAL, "at most max_rays * (Nc + Nf) = %lld background points per step", (long long)t->max_rays * (Nc + Nf));
    }
  }
  (void)hipMemsetAsync(t->terms_dev, 0, 8 * sizeof(float), st);
  (void)hipMemsetAsync(t->terms_dev + 15, 0, sizeof(float), st);            // [15]: overflow-source flags (diag_scan_nerf inside the step, nerfds_trainer_overflow_sources after it)
  (void)hipMemsetAsync(t->terms_dev + 9, 0, 6 * sizeof(float), st);          // ... [12]: elastic regulariser, [13], [14]: mask occlusion regulariser (coarse / fine)
  // [9], [10]: hyper-point regulariser of the coarse / fine level, [11]: background loss ([8]: non-finite flag)
  const float norm_weight = objective ? objective->norm_loss_weight : 0.f;
  const bool elastic = objective && objective->elastic_loss_weight != 0.f;      // second order like the norm loss: needs the tangent pass and its backward
  const bool want_sg = (flags & NERFDS_TRAIN_SIGMA_GRAD) != 0 || norm_weight != 0.f || elastic;
  // the tangent passes (sigma gradient, norm loss, elastic regulariser) read the primal ReLU masks only: the primal pass stays on the fused f16 step
  // (NERFDS_TRAIN_HALF_TANGENTS=0: fp32 activations and the layer-by-layer backward for those steps, as through round 4's first half)
  static const bool half_tangents = !(getenv("NERFDS_TRAIN_HALF_TANGENTS") && std::string(getenv("NERFDS_TRAIN_HALF_TANGENTS")) == "0");
  t->half_step = t->fused_fwd && t->fused_bwd && (!want_sg || half_tangents) && !t->fp32_step;
  {
    PackBatch pb;
    if (t->fused_fwd) pack_fused_forward(*t, st, pb);
    if (t->half_step) pack_fused_backward(*t, pb);
    pack_batch(st, t->theta, t->fold, t->P, pb);
  }
  t->tan_x_scale = std::ldexp(1.f, -6 + t->tan_scale_adjust);
  t->last_R = R; t->last_tan_rows = 0; t->last_half = t->half_step;
  {   // Launch decisions of the side streams (A/Bs on one box each, interleaved: profiles/r6_ab/README.md).  The weight-gradient launches are
      // persistent - one workgroup per CU, registers for nothing beside it - so a kernel of the CALLER's stream, the step's critical path, that arrives
      // while they run waits for one of them to END (k_se3_fwd: 12 us alone, 273 us in the 4096-ray step).  Up to NERFDS_TRAIN_WGRAD_CAP_RAYS rays
      // (default 3072), and in every step with second-order terms, they take HALF of the CUs and the other half stays free for the caller's stream:
      // 128 rays 2.03 -> 1.79 ms, 256: 2.17 -> 1.98, 512: 2.55 -> 2.40, 1024: 3.70 -> 3.59, 3072: 8.61 -> 8.58, nerf_ds.gin objective -2 ... -12 %; the
      // rgb-only 4096-ray step needs the whole machine for them (11.23 -> 11.42 ms capped: not capped).  NERFDS_TRAIN_WGRAD_CUS=<n> forces a cap (0: none).
      // Streams: under the cap THREE side streams pay from 512 rays on (2.46 -> 2.40 ms; uncapped, one was 4 % faster there), ONE up to
      // NERFDS_TRAIN_SIDE_SMALL rays (default 384: 256 rays 1.98 against 2.05 ms).
    static const int small_rays = [] { const char* e = getenv("NERFDS_TRAIN_SIDE_SMALL"); return e ? atoi(e) : 384; }();
    static const int cap_rays = [] { const char* e = getenv("NERFDS_TRAIN_WGRAD_CAP_RAYS"); return e ? atoi(e) : 3072; }();
    static const int cap_env = [] { const char* e = getenv("NERFDS_TRAIN_WGRAD_CUS"); return e ? atoi(e) : -1; }();
    t->nside_eff = (R <= small_rays && t->nside > 1) ? 1 : t->nside;
    t->wgrad_cu_cap = cap_env >= 0 ? cap_env : ((R <= cap_rays || want_sg) ? t->num_cus / 2 : 0);
  }
  {
    int e = 6;
    while ((1 << (e - 6)) < R && e < 40) ++e;
    e += t->g_scale_adjust;                    // dynamic loss scaling: the host lowers it after an overflow (training.py Trainer.step)
    if (const char* s = getenv("NERFDS_TRAIN_G_SCALE_LOG2")) {      // development override of the exponent: a whole number in [-30, 40], anything else ignored
      char* end = nullptr;
      const long v = std::strtol(s, &end, 10);
      if (end != s && *end == '\0' && v >= -30 && v <= 40) e = (int)v;
    }
    e = e < -30 ? -30 : (e > 40 ? 40 : e);
    t->g_scale = (t->half_step && t->g16) ? std::ldexp(1.f, e) : 1.f;
  }
  t->keep_tangents = norm_weight != 0.f || elastic;
  if ((norm_weight != 0.f || elastic) && (!ensure_tangent_ws(*t) || !ensure_norm_ws(*t))) return t->fail(NERFDS_ENOMEM, "hipMalloc of the norm-loss workspace failed");
  if (want_sg && !ensure_tangent_ws(*t)) return t->fail(NERFDS_ENOMEM, "hipMalloc of the tangent workspace failed");
  if (want_sg && !t->half_step && !ensure_tangent_ws32(*t, t->keep_tangents)) return t->fail(NERFDS_ENOMEM, "hipMalloc of the fp32 tangent workspace failed");
  t->last_keep_tangents = t->keep_tangents; t->last_tan16 = want_sg && t->fused_tan && t->half_step;
  if (want_sg && t->fused_tan && t->half_step) {
    if (3LL * t->max_rays * (Nc + Nf) >= (1LL << 31)) return t->fail(NERFDS_EINVAL, "tangent rows are indexed in 31 bits: max_rays * samples * 3 is too large");
    // the hidden tangents: one f16 array per layer when the step differentiates the tangent pass (its weight gradients read them), one per network otherwise
    if (!ensure_tan16(*t, t->keep_tangents) || (t->keep_tangents && !ensure_tan16_g(*t))) return t->fail(NERFDS_ENOMEM, "hipMalloc of the f16 tangent workspace failed");
    pack_fused_tangents(*t, st);
  }
  // the elastic regulariser reads the tangents of the COARSE level only: without the norm loss (or the caller's flag) the fine level runs no tangent pass
  const bool want_sg_fine = (flags & NERFDS_TRAIN_SIGMA_GRAD) != 0 || norm_weight != 0.f;
  t->tn_valid = want_sg_fine;
  t->tangents_warp_only = elastic && !want_sg_fine;
  // NERFDS_TRAIN_MERGED=0: the two levels one after the other, each with its own pass over the shared networks (A/B, and every step the merged
  // flow does not cover: auxiliary losses, tangent passes, one level, the layer-by-layer kernels)
  static const bool merged_on = !(getenv("NERFDS_TRAIN_MERGED") && std::string(getenv("NERFDS_TRAIN_MERGED")) == "0");
  int rc;
  // NERFDS_TRAIN_MERGED_FULL=0: steps with auxiliary / second-order terms level by level, as through most of round 5 (A/B)
  static const bool merged_full_on = !(getenv("NERFDS_TRAIN_MERGED_FULL") && std::string(getenv("NERFDS_TRAIN_MERGED_FULL")) == "0");
  const bool mergeable = merged_on && !t->fwd_only && t->half_step && Nf > 0 && resample_has_sources(Nc, Nf) && t->g0 &&
                         64 * (int64_t)R * (Nc + Nf) <= (int64_t)t->max_rays * (Nc + Nf) * t->trunk[0].width;
  if (mergeable && !want_sg && !obp) {
    rc = run_merged(*t, st, R, t->zc, rays, target_rgb, ex, W, rnd);
    if (rc != NERFDS_OK) return rc;
  } else if (mergeable && merged_full_on && obp && !(flags & NERFDS_TRAIN_SIGMA_GRAD) && (!want_sg || (t->fused_tan && t->keep_tangents))) {
    rc = run_merged_full(*t, st, R, t->zc, rays, target_rgb, ex, W, rnd, *obp, norm_weight);
    if (rc != NERFDS_OK) return rc;
  } else {
  rc = run_level(*t, st, 0, R, Nc, t->zc, rays, target_rgb, ex, W, t->wc, want_sg, obp, norm_weight);
  if (rc != NERFDS_OK) return rc;
  if (Nf > 0 && !t->fwd_only) diag_scan_nerf(*t, st, (int64_t)R * Nc, t->keep_tangents, 3LL * R * Nc);
  if (Nf > 0) {
    resample(st, R, Nc, Nf, t->zc, t->wc, strat, rnd ? rnd->u_rand : nullptr, rnd ? rnd->seed : 0, rnd ? rnd->first_ray : 0, t->zf, t->rs_scratch);
    rc = run_level(*t, st, 1, R, Nc + Nf, t->zf, rays, target_rgb, ex, W, t->weights, want_sg_fine, obp, norm_weight);
    if (rc != NERFDS_OK) return rc;
  }
  }
  if (t->fwd_only) {
    hipError_t fe = hipGetLastError();
    if (fe != hipSuccess) return t->fail(NERFDS_EDEVICE, "kernel launch failed: %s", hipGetErrorString(fe));
    return NERFDS_OK;
  }
  if (objective && objective->background_loss_weight != 0.f) {
    rc = run_background(*t, st, *objective, W);
    if (rc != NERFDS_OK) return rc;
  }
  if (t->grad_rep) sum_partials(st, t->grad_rep, GRAD_REPS, t->P, t->grad);      // grad += the replicas of the MFMA kernels
  if (t->half_step) {
    // Fused backward: the bottleneck Dense (no activation, modules.py:255) never ran as a layer.  With S = trunk_out^T g_rgb (the
    // gradient of rgb hidden_0's trunk_out rows, just summed) and c = the bias gradient of rgb hidden_0:
    //   d K[bottleneck rows] = Wb^T S + bb (x) c,   d Wb = S K_b^T,   d bb = K_b c      (exact: bott = trunk_out Wb + bb)
    const int TW = t->trunk[0].width, VD = 6 * t->D.vd_bands;
    // (six latency-bound products on leaves of one level each - three launches per level at the serial tail of the step, then the levels side by side on two
    // streams, now ONE launch: train_kernels.hip k_bott_grads)
    BottBatch bb;
    for (int lv = 0; lv < (Nf > 0 ? 2 : 1); ++lv) {
      const LayerP& K = t->rgb_h[lv];
      bb.lv[bb.n++] = BottItem{TW, K.N, t->theta + t->bott[lv].w, t->theta + t->bott[lv].b, t->theta + K.w, t->grad + K.w + (int64_t)(TW + VD) * K.N, t->grad + K.b,
                               t->grad + K.w, t->grad + t->bott[lv].w, t->grad + t->bott[lv].b};
    }
    bott_grads(st, bb);
  }
  if (!(flags & NERFDS_TRAIN_GRADS_ONLY)) adam_update(t, learning_rate, st);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return t->fail(NERFDS_EDEVICE, "kernel launch failed: %s", hipGetErrorString(e));
  if (loss_host) {
    float l[2];
    if (hipMemcpyAsync(l, t->loss_dev, sizeof l, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      return t->fail(NERFDS_EDEVICE, "loss read-back failed");
    float tm[15];
    if (hipMemcpy(tm, t->terms_dev, sizeof tm, hipMemcpyDeviceToHost) != hipSuccess) return t->fail(NERFDS_EDEVICE, "loss read-back failed");
    unsigned nonfinite = 0;
    std::memcpy(&nonfinite, &tm[8], sizeof nonfinite);
    const int fl = Nf > 0 ? 1 : 0;
    loss_host[0] = l[fl];     // rgb loss of the fine level (the level render_image returns; coarse if there is none), of the coarse level
    loss_host[1] = l[0];
    for (int k = 0; k < 4; ++k) { loss_host[2 + k] = tm[4 * fl + k]; loss_host[6 + k] = tm[k]; }   // weighted warp_reg / back_facing / mask / norm terms: fine, coarse
    loss_host[10] = tm[9 + fl]; loss_host[11] = tm[9];                                                // weighted hyper-point regulariser: fine, coarse
    loss_host[12] = tm[11];                                                                            // weighted background regulariser
    loss_host[13] = tm[12];                                                                            // weighted elastic regulariser (coarse level)
    loss_host[14] = tm[13 + fl]; loss_host[15] = tm[13];                                              // weighted mask occlusion regulariser: fine, coarse
    if (nonfinite && !(flags & NERFDS_TRAIN_GRADS_ONLY))
      return t->fail(NERFDS_ENONFINITE, "non-finite gradient: the Adam update of this step was skipped (f16 activations / scaled f16 g overflowed? NERFDS_TRAIN_G16=0 "
                                        "keeps g in fp32, NERFDS_TRAIN_HALF_TANGENTS=0 fp32 activations for the steps with a tangent pass)");
  }
  return NERFDS_OK;
}

}  // extern "C"
