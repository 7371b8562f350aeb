// C ABI of include/nerfds.h: context, weight packing/upload, launch of the fused ray kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "nerfds.h"
#include "graphs.h"
#include "kargs.h"
#include "pack.h"
#include "camera_dev.h"

using namespace nerfds;

extern "C" {
void nerfds_launch_nerfds_bf16(const KArgs&, int, void*);
void nerfds_launch_nerfds_bf16x3(const KArgs&, int, void*);
void nerfds_launch_nerfds_f32(const KArgs&, int, void*);
void nerfds_launch_static_bf16(const KArgs&, int, void*);
void nerfds_launch_static_bf16x3(const KArgs&, int, void*);
void nerfds_launch_static_f32(const KArgs&, int, void*);
void nerfds_launch_hyper_bf16(const KArgs&, int, void*);
void nerfds_launch_hyper_bf16x3(const KArgs&, int, void*);
void nerfds_launch_hyper_f32(const KArgs&, int, void*);
void nerfds_launch_nerfds_f16(const KArgs&, int, void*);
void nerfds_launch_static_f16(const KArgs&, int, void*);
void nerfds_launch_hyper_f16(const KArgs&, int, void*);
void nerfds_launch_nerfds_mixed(const KArgs&, int, void*);
void nerfds_launch_nerfds_bf16x3f(const KArgs&, int, void*);
void nerfds_launch_hyper_bf16x3f(const KArgs&, int, void*);
void nerfds_launch_static_mixed(const KArgs&, int, void*);
void nerfds_launch_nerfds_f16x3(const KArgs&, int, void*);
void nerfds_launch_static_f16x3(const KArgs&, int, void*);
void nerfds_launch_hyper_f16x3(const KArgs&, int, void*);
void nerfds_launch_hyper_mixed(const KArgs&, int, void*);
void nerfds_launch_camera_rays(const nerfds::CameraParams&, long long, long long, const float*, float*, float*, float*, void*);
void nerfds_launch_encode_embed(const float* table, int rows, const float* meta, int channels, long long n, float* out, void* stream);
void nerfds_launch_frame_images(const float*, int, int, float, float, const double*, uint8_t*, uint8_t*, void*);
}
static_assert(sizeof(nerfds_camera) == sizeof(nerfds::CameraParams), "nerfds_camera and CameraParams must have the same layout");

namespace {

thread_local std::string g_create_error;

struct OwnedDense {
  std::vector<float> kernel, bias;
  int in_dim = 0, out_dim = 0;
  bool set(const nerfds_dense& d, int want_in, int want_out, const char* name, std::string& err) {
    if (d.kernel == nullptr || d.in_dim != want_in || d.out_dim != want_out) {
      char buf[256];
      snprintf(buf, sizeof buf, "%s: expected a [%d, %d] kernel, got %s[%d, %d]", name, want_in, want_out,
               d.kernel ? "" : "NULL ", d.in_dim, d.out_dim);
      err = buf;
      return false;
    }
    in_dim = d.in_dim;
    out_dim = d.out_dim;
    kernel.assign(d.kernel, d.kernel + (size_t)in_dim * out_dim);
    if (d.bias) bias.assign(d.bias, d.bias + out_dim); else bias.assign(out_dim, 0.f);
    return true;
  }
  DenseView view() const {
    DenseView v;
    v.kernel = kernel.data(); v.bias = bias.data(); v.in_dim = in_dim; v.out_dim = out_dim;
    return v;
  }
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t upload(const void* src, size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = n;
    hipError_t e = hipMalloc(&p, std::max<size_t>(n, 16));
    if (e != hipSuccess) return e;
    return n ? hipMemcpy(p, src, n, hipMemcpyHostToDevice) : hipSuccess;
  }
};

template <class G> bool cfg_matches(const nerfds_model_cfg& c) {
  using D = Dims<G>;
  bool ok = (c.use_warp != 0) == G::HAS_WARP && (c.use_hyper_sheet != 0) == G::HAS_HYPER &&
            (c.use_predicted_mask != 0) == G::HAS_MASK && (c.predict_norm != 0) == G::PREDICT_NORM &&
            (c.use_x_in_rgb_condition != 0) == G::X_IN_RGB && c.use_viewdirs != 0 &&
            c.nerf_trunk_depth == G::TRUNK_DEPTH && c.nerf_trunk_width == G::TRUNK_W && c.nerf_skip == G::TRUNK_SKIP &&
            c.nerf_rgb_branch_depth == 1 && c.nerf_rgb_branch_width == G::RGB_W &&
            c.spatial_point_max_deg == G::SP_BANDS && c.viewdir_max_deg == G::VD_BANDS &&
            (c.use_posenc_identity != 0) == G::IDENT;
  if (G::HAS_WARP)
    ok = ok && c.warp_max_deg == G::WARP_BANDS && c.warp_trunk_depth == G::WARP_DEPTH && c.warp_trunk_width == G::WARP_W &&
         c.warp_skip == G::WARP_SKIP && c.glo_num_dims == G::GLO && (c.use_mask_in_warp != 0) == G::HAS_MASK &&
         (c.warp_use_posenc_identity != 0) == G::WARP_IDENT;
  if (G::HAS_HYPER)
    ok = ok && c.hyper_sheet_max_deg == G::HYP_BANDS && c.hyper_sheet_depth == G::HYP_DEPTH && c.hyper_sheet_width == G::HYP_W &&
         c.hyper_sheet_skip == G::HYP_SKIP && c.hyper_num_dims == G::HYP_DIMS && c.hyper_point_max_deg == G::HP_BANDS &&
         (c.use_mask_in_hyper != 0) == G::HAS_MASK;
  if (G::HAS_MASK)
    ok = ok && c.mask_max_deg == G::MASK_BANDS && c.mask_depth == G::MASK_DEPTH && c.mask_width == G::MASK_W &&
         c.mask_skip == G::MASK_SKIP && c.mask_output_relu != 0;
  if (G::PREDICT_NORM) ok = ok && c.norm_input_max_deg == G::NM_BANDS;
  (void)sizeof(D);
  return ok;
}

int graph_of(const nerfds_model_cfg& c) {
  if (cfg_matches<GraphNerfDS>(c)) return GraphNerfDS::ID;
  if (cfg_matches<GraphStatic>(c)) return GraphStatic::ID;
  if (cfg_matches<GraphHyperNeRF>(c)) return GraphHyperNeRF::ID;
  return -1;
}

struct Weights {
  std::vector<float> warp_embed, mask_embed;
  OwnedDense mask_hidden[NERFDS_MAX_DEPTH], mask_out;
  OwnedDense warp_hidden[NERFDS_MAX_DEPTH], warp_w, warp_v;
  OwnedDense hyper_hidden[NERFDS_MAX_DEPTH], hyper_out;
  struct Nerf { OwnedDense trunk[NERFDS_MAX_DEPTH], bottleneck, alpha, rgb_hidden[NERFDS_MAX_DEPTH], rgb; } nerf[2];
  bool loaded = false;
};

template <class G> bool take_weights(Weights& W, const nerfds_model_cfg& c, const nerfds_weights& w, std::string& err) {
  using D = Dims<G>;
  char nm[64];
  auto mlp = [&](OwnedDense* dst, const nerfds_dense* src, int depth, int width, int in_dim, int skip, const char* name) {
    for (int l = 0; l < depth; ++l) {
      int k = (l == 0 ? in_dim : width) + ((l == skip && l > 0) ? in_dim : 0);
      snprintf(nm, sizeof nm, "%s/hidden_%d", name, l);
      if (!dst[l].set(src[l], k, width, nm, err)) return false;
    }
    return true;
  };
  const size_t glo = (size_t)c.num_warp_embeds * G::GLO;
  if ((G::HAS_WARP || G::HAS_MASK) && w.embed_rows != c.num_warp_embeds) {
    char buf[160];
    snprintf(buf, sizeof buf, "GLO tables have %d rows, the model configuration says num_warp_embeds = %d", w.embed_rows, c.num_warp_embeds);
    err = buf;
    return false;
  }
  if (G::HAS_WARP) {
    if (!w.warp_embed || c.num_warp_embeds <= 0) { err = "warp_embed table missing"; return false; }
    W.warp_embed.assign(w.warp_embed, w.warp_embed + glo);
    if (!mlp(W.warp_hidden, w.warp_hidden, G::WARP_DEPTH, G::WARP_W, D::WARP_IN, G::WARP_SKIP, "warp_field/trunk")) return false;
    if (!W.warp_w.set(w.warp_w, G::WARP_W, 3, "warp_field/branches_w/logit", err)) return false;
    if (!W.warp_v.set(w.warp_v, G::WARP_W, 3, "warp_field/branches_v/logit", err)) return false;
  }
  if (G::HAS_MASK) {
    if (!w.mask_embed) { err = "mask_embed table missing"; return false; }
    W.mask_embed.assign(w.mask_embed, w.mask_embed + glo);
    if (!mlp(W.mask_hidden, w.mask_hidden, G::MASK_DEPTH, G::MASK_W, D::MASK_IN, G::MASK_SKIP, "mask_mlp/MLP_0")) return false;
    if (!W.mask_out.set(w.mask_out, G::MASK_W, 1, "mask_mlp/MLP_0/logit", err)) return false;
  }
  if (G::HAS_HYPER) {
    if (!mlp(W.hyper_hidden, w.hyper_hidden, G::HYP_DEPTH, G::HYP_W, D::HYP_IN, G::HYP_SKIP, "hyper_sheet_mlp/MLP_0")) return false;
    if (!W.hyper_out.set(w.hyper_out, G::HYP_W, G::HYP_DIMS, "hyper_sheet_mlp/MLP_0/logit", err)) return false;
  }
  const int levels = c.num_fine_samples > 0 ? 2 : 1;
  for (int lv = 0; lv < levels; ++lv) {
    const nerfds_nerf_mlp& s = w.nerf[lv];
    Weights::Nerf& d = W.nerf[lv];
    const char* pre = lv ? "nerf_mlps_fine" : "nerf_mlps_coarse";
    snprintf(nm, sizeof nm, "%s/trunk_mlp", pre);
    std::string tname = nm;
    if (!mlp(d.trunk, s.trunk, G::TRUNK_DEPTH, G::TRUNK_W, D::TRUNK_IN, G::TRUNK_SKIP, tname.c_str())) return false;
    if (!d.bottleneck.set(s.bottleneck, G::TRUNK_W, G::TRUNK_W, "bottleneck", err)) return false;
    if (!d.alpha.set(s.alpha, G::TRUNK_W, D::ALPHA_OUT, "alpha_mlp/logit", err)) return false;
    if (!d.rgb_hidden[0].set(s.rgb_hidden[0], D::RGB_IN, G::RGB_W, "rgb_mlp/hidden_0", err)) return false;
    if (!d.rgb.set(s.rgb, G::RGB_W, 3, "rgb_mlp/logit", err)) return false;
  }
  W.loaded = true;
  return true;
}

bool take_weights_dispatch(int graph, Weights& W, const nerfds_model_cfg& c, const nerfds_weights& w, std::string& err) {
  if (graph == GraphNerfDS::ID) return take_weights<GraphNerfDS>(W, c, w, err);
  if (graph == GraphStatic::ID) return take_weights<GraphStatic>(W, c, w, err);
  return take_weights<GraphHyperNeRF>(W, c, w, err);
}

SharedNets shared_views(const Weights& W) {
  SharedNets n;
  for (int i = 0; i < NERFDS_MAX_DEPTH; ++i) {
    n.mask_hidden[i] = W.mask_hidden[i].view();
    n.warp_hidden[i] = W.warp_hidden[i].view();
    n.hyper_hidden[i] = W.hyper_hidden[i].view();
  }
  n.mask_out = W.mask_out.view(); n.warp_w = W.warp_w.view(); n.warp_v = W.warp_v.view(); n.hyper_out = W.hyper_out.view();
  return n;
}
NerfNet nerf_views(const Weights::Nerf& s) {
  NerfNet n;
  for (int i = 0; i < NERFDS_MAX_DEPTH; ++i) { n.trunk[i] = s.trunk[i].view(); n.rgb_hidden[i] = s.rgb_hidden[i].view(); }
  n.bottleneck = s.bottleneck.view(); n.alpha = s.alpha.view(); n.rgb = s.rgb.view();
  return n;
}

template <class G> void pack_which(StreamWriter& sw, const Weights& W, int which, int level, Plan pl) {
  // (level_plan: a plan may run the coarse level's NerfMLP in its own arithmetic - NERFDS_PREC_BF16X3_FINE)
  if (which == 0) pack_shared<G>(sw, shared_views(W), pl); else pack_nerf<G>(sw, nerf_views(W.nerf[level]), level_plan(pl, level));
}
// the kernels the Makefile builds in the two-N-tile shape (NT2FLAGS): nerf_ds / HyperNeRF graph, bf16 / f16
static int tile_pair_of(int graph, int prec) {
  return stream_tile_pair((graph == GraphNerfDS::ID || graph == GraphHyperNeRF::ID) && (prec == (int)NERFDS_PREC_BF16 || prec == (int)NERFDS_PREC_F16));
}
void pack_dispatch(int graph, StreamWriter& sw, const Weights& W, int which, int level, int prec) {
  const Plan pl = plan_of(prec);
  sw.tile_pair = tile_pair_of(graph, prec);
  sw.x3_f16 = prec == (int)NERFDS_PREC_F16X3;      // split f16: the same plan and stream geometry as split bf16, f16 hi / lo parts
  if (graph == GraphNerfDS::ID) pack_which<GraphNerfDS>(sw, W, which, level, pl);
  else if (graph == GraphStatic::ID) pack_which<GraphStatic>(sw, W, which, level, pl);
  else pack_which<GraphHyperNeRF>(sw, W, which, level, pl);
}
// level: the NerfMLP stream of that level (which == 1); -1: the larger of the two (a buffer that holds either)
template <class G> void stream_dims(int which, int level, int prec, int64_t* wbytes, int64_t* bfloats) {
  using D = Dims<G>;
  const Plan pl = plan_of(prec);
  const int nu = level < 0 ? std::max(nerf_units<G>(level_plan(pl, 0)), nerf_units<G>(level_plan(pl, 1))) : nerf_units<G>(level_plan(pl, level));
  *wbytes = (int64_t)pad_units(which == 0 ? shared_units<G>(pl) : nu) * 1024;   // zero padded to whole stages
  *bfloats = (int64_t)(which == 0 ? D::SHARED_BIAS_TILES : D::NERF_BIAS_TILES) * 32;
}
void stream_dims_dispatch(int graph, int which, int level, int prec, int64_t* wb, int64_t* bf) {
  if (graph == GraphNerfDS::ID) stream_dims<GraphNerfDS>(which, level, prec, wb, bf);
  else if (graph == GraphStatic::ID) stream_dims<GraphStatic>(which, level, prec, wb, bf);
  else stream_dims<GraphHyperNeRF>(which, level, prec, wb, bf);
}

void window(float* out, int bands, float alpha) {   // model_utils.py:420-436
  for (int b = 0; b < MAX_BANDS; ++b) {
    double x = std::min(std::max((double)alpha - b, 0.0), 1.0);
    out[b] = b < bands ? (float)(0.5 * (1.0 + std::cos(M_PI * x + M_PI))) : 0.f;
  }
}

}  // namespace

struct nerfds_ctx {
  int device = 0;
  int graph = -1;
  int num_cus = 256;
  nerfds_model_cfg cfg{};
  Weights W;
  DevBuf wstream[NUM_PLANS][3];     // [prec][shared, coarse, fine]
  DevBuf wbias[NUM_PLANS][3];   // per precision since round 6: the split-f16 packer rescales the last hidden layer of the shared networks (pack.h balance_last_hidden), biases included
  DevBuf warp_embed, mask_embed;
  DevBuf ray_scratch;       // origins | directions generated from a camera
  bool packed[NUM_PLANS] = {};
  std::string err;
  // timing: event pairs recorded around the launches since the last reset; reset returns them to the pool (no per-launch
  // hipEventCreate, nothing accumulates), and at most MAX_TIMED launches are recorded between two resets
  bool timing = false;
  static constexpr size_t MAX_TIMED = 4096;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events, event_pool;
  int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    err = buf;
    return code;
  }
};

static launch_fn launcher(int graph, uint32_t prec) {
  static_assert(NUM_PLANS == NERFDS_PREC_COUNT, "graphs.h plan_of covers every NERFDS_PREC_* value");
  static const launch_fn tab[3][NUM_PLANS] = {
      {nerfds_launch_nerfds_bf16, nerfds_launch_nerfds_bf16x3, nerfds_launch_nerfds_f32, nerfds_launch_nerfds_f16, nerfds_launch_nerfds_mixed, nerfds_launch_nerfds_bf16x3f, nerfds_launch_nerfds_f16x3},
      // (the static graph has no mixed-level kernel: nerfds_render_rays maps NERFDS_PREC_BF16X3_FINE to NERFDS_PREC_BF16X3 before it packs or launches)
      {nerfds_launch_static_bf16, nerfds_launch_static_bf16x3, nerfds_launch_static_f32, nerfds_launch_static_f16, nerfds_launch_static_mixed, nerfds_launch_static_bf16x3, nerfds_launch_static_f16x3},
      {nerfds_launch_hyper_bf16, nerfds_launch_hyper_bf16x3, nerfds_launch_hyper_f32, nerfds_launch_hyper_f16, nerfds_launch_hyper_mixed, nerfds_launch_hyper_bf16x3f, nerfds_launch_hyper_f16x3}};
  return tab[graph][prec];
}

extern "C" {

int nerfds_abi_version(void) { return NERFDS_ABI_VERSION; }
int64_t nerfds_struct_size(int which) {
  switch (which) {
    case 0: return sizeof(nerfds_model_cfg);
    case 1: return sizeof(nerfds_weights);
    case 2: return sizeof(nerfds_camera);
    case 3: return sizeof(nerfds_rays);
    case 4: return sizeof(nerfds_extra);
    case 5: return sizeof(nerfds_rand);
    case 6: return sizeof(nerfds_out);
    case 7: return sizeof(nerfds_train_objective);
    case 8: return sizeof(nerfds_train_numerics);
    default: return -1;
  }
}

int nerfds_precision_plan(uint32_t prec, int32_t plan_out[5]) {
  if (prec >= NERFDS_PREC_COUNT || !plan_out) return NERFDS_EINVAL;
  const Plan pl = plan_of((int)prec);
  plan_out[0] = pl.mask; plan_out[1] = pl.warp; plan_out[2] = pl.hyp; plan_out[3] = pl.trunk; plan_out[4] = pl.rgb;
  return NERFDS_OK;
}
int nerfds_precision_plan_level(uint32_t prec, int32_t level, int32_t plan_out[5]) {
  if (prec >= NERFDS_PREC_COUNT || !plan_out || level < 0 || level > 1) return NERFDS_EINVAL;
  const Plan pl = level_plan(plan_of((int)prec), level);
  plan_out[0] = pl.mask; plan_out[1] = pl.warp; plan_out[2] = pl.hyp; plan_out[3] = pl.trunk; plan_out[4] = pl.rgb;
  return NERFDS_OK;
}

const char* nerfds_last_error(const nerfds_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int nerfds_ctx_create(nerfds_ctx** out, int device, const nerfds_model_cfg* cfg) {
  if (!out || !cfg) { g_create_error = "null argument"; return NERFDS_EINVAL; }
  *out = nullptr;
  if (cfg->abi_version != NERFDS_ABI_VERSION) { g_create_error = "abi_version mismatch"; return NERFDS_EINVAL; }
  if (cfg->num_coarse_samples < 4 || cfg->num_fine_samples < 0 ||
      cfg->num_coarse_samples + cfg->num_fine_samples > MAX_SAMPLES) {
    g_create_error = "num_coarse_samples must be >= 4 and num_coarse_samples + num_fine_samples <= 256";
    return NERFDS_EINVAL;
  }
  const int graph = graph_of(*cfg);
  if (graph < 0) {
    g_create_error = "render graph not built as a HIP kernel: supported are the configs/nerf_ds.gin graph "
                     "(mask + SE3 warp + hyper sheet + predicted normal), the configs/base.gin HyperNeRF graph "
                     "(SE3 warp + hyper sheet, posenc identity) and the static coarse/fine NeRF graph";
    return NERFDS_ENOTSUP;
  }
  int ndev = 0;
  const hipError_t dc = hipGetDeviceCount(&ndev);
  if (dc != hipSuccess || device < 0 || device >= ndev) {
    g_create_error = std::string("no such HIP device (the HIP path has no CPU fallback): device ") + std::to_string(device) + ", hipGetDeviceCount -> " +
                     std::to_string(ndev) + " (" + hipGetErrorString(dc) + ")";
    return NERFDS_EDEVICE;
  }
  std::unique_ptr<nerfds_ctx> c(new nerfds_ctx);
  c->device = device;
  c->graph = graph;
  c->cfg = *cfg;
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "hipSetDevice failed"; return NERFDS_EDEVICE; }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
  *out = c.release();
  return NERFDS_OK;
}

int nerfds_ctx_destroy(nerfds_ctx* ctx) {
  if (!ctx) return NERFDS_OK;
  (void)hipSetDevice(ctx->device);
  for (auto* v : {&ctx->events, &ctx->event_pool})
    for (auto& e : *v) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  delete ctx;
  return NERFDS_OK;
}

int nerfds_ctx_load_weights(nerfds_ctx* ctx, const nerfds_weights* w) {
  if (!ctx || !w) return NERFDS_EINVAL;
  std::string err;
  ctx->W = Weights();
  bool ok = take_weights_dispatch(ctx->graph, ctx->W, ctx->cfg, *w, err);
  if (!ok) return ctx->fail(NERFDS_EINVAL, "load_weights: %s", err.c_str());
  for (bool& p : ctx->packed) p = false;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  if (!ctx->W.warp_embed.empty() &&
      ctx->warp_embed.upload(ctx->W.warp_embed.data(), ctx->W.warp_embed.size() * 4) != hipSuccess)
    return ctx->fail(NERFDS_ENOMEM, "upload of warp_embed failed");
  if (!ctx->W.mask_embed.empty() &&
      ctx->mask_embed.upload(ctx->W.mask_embed.data(), ctx->W.mask_embed.size() * 4) != hipSuccess)
    return ctx->fail(NERFDS_ENOMEM, "upload of mask_embed failed");
  return NERFDS_OK;
}

static int ensure_packed(nerfds_ctx* ctx, uint32_t prec) {
  if (ctx->packed[prec]) return NERFDS_OK;
  const int levels = ctx->cfg.num_fine_samples > 0 ? 2 : 1;
  for (int which = 0; which < 1 + levels; ++which) {
    int64_t wb = 0, bf = 0;
    stream_dims_dispatch(ctx->graph, which ? 1 : 0, which ? which - 1 : 0, (int)prec, &wb, &bf);
    std::vector<uint8_t> w((size_t)wb);
    std::vector<float> b((size_t)bf);
    StreamWriter sw{w.data(), b.data()};
    pack_dispatch(ctx->graph, sw, ctx->W, which ? 1 : 0, which ? which - 1 : 0, (int)prec);
    if ((int64_t)sw.wbytes > wb || wb - (int64_t)sw.wbytes >= STAGE_BYTES || (int64_t)sw.bfloats != bf)
      return ctx->fail(NERFDS_EINVAL, "internal: packed stream %d has %zu bytes / %zu bias floats, kernel expects %lld / %lld",
                       which, sw.wbytes, sw.bfloats, (long long)wb, (long long)bf);
    if (getenv("NERFDS_DEBUG_PACK")) {
      unsigned long long sum = 0; size_t nz = 0;
      for (size_t i = 0; i < w.size(); ++i) { sum += w[i]; nz += w[i] != 0; }
      fprintf(stderr, "ensure_packed: prec %u which %d bytes %zu nonzero %zu checksum %llu x3_f16 %d\n", prec, which, w.size(), nz, sum, (int)sw.x3_f16);
    }
    if (ctx->wstream[prec][which].upload(w.data(), w.size()) != hipSuccess) return ctx->fail(NERFDS_ENOMEM, "weight stream upload failed");
    if (ctx->wbias[prec][which].upload(b.data(), b.size() * 4) != hipSuccess)
      return ctx->fail(NERFDS_ENOMEM, "bias upload failed");
  }
  ctx->packed[prec] = true;
  return NERFDS_OK;
}

int nerfds_render_rays(nerfds_ctx* ctx, const nerfds_rays* rays, const nerfds_extra* extra, const nerfds_rand* rnd,
                       const nerfds_out* out, uint32_t flags, void* hip_stream) {
  if (!ctx) return NERFDS_EINVAL;
  if (!rays || !extra || !out) return ctx->fail(NERFDS_EINVAL, "null argument");
  if (!ctx->W.loaded) return ctx->fail(NERFDS_EINVAL, "nerfds_ctx_load_weights has not been called");
  uint32_t prec = flags & NERFDS_PREC_MASK;
  if (prec >= NERFDS_PREC_COUNT) return ctx->fail(NERFDS_EINVAL, "unknown precision %u", prec);
  // ONE effective precision per call, used for the packed streams, their sizes and the launcher alike.  NERFDS_PREC_BF16X3_FINE is plain split bf16
  // (a) on a single-level model - its one level IS the level render_fn returns, there is no coarse pass to run cheaply - and (b) on the static graph,
  // for which no mixed-level kernel is built (the launcher table holds the bf16x3 kernel there: it must read bf16x3-packed streams at BOTH levels)
  if (prec == NERFDS_PREC_BF16X3_FINE && (ctx->cfg.num_fine_samples == 0 || ctx->graph == GraphStatic::ID)) prec = NERFDS_PREC_BF16X3;
  if ((flags & NERFDS_FLAG_USE_WARP_OFF) && ctx->cfg.use_warp)
    return ctx->fail(NERFDS_ENOTSUP, "use_warp=False on a warp model is not runnable in the reference either (SURVEY.md 8a quirk 2)");
  if (rays->num_rays < 0 || rays->num_rays > 0x7fffffff) return ctx->fail(NERFDS_EINVAL, "num_rays out of range");
  if (rays->num_rays == 0) return NERFDS_OK;
  if (!rays->camera && (!rays->origins || !rays->directions)) return ctx->fail(NERFDS_EINVAL, "origins/directions (or a camera) are required");
  if (rays->camera && (rays->first_pixel < 0 || rays->camera->image_width <= 0 ||
                       rays->first_pixel + rays->num_rays > (int64_t)rays->camera->image_width * rays->camera->image_height))
    return ctx->fail(NERFDS_EINVAL, "camera pixel range [first_pixel, first_pixel + num_rays) is outside the image");
  if (ctx->cfg.use_warp && !rays->warp_id && !rays->encoded_warp)
    return ctx->fail(NERFDS_EINVAL, "metadata['warp'] ids (or encoded_warp vectors) are required by this graph");
  if (ctx->cfg.use_predicted_mask && !rays->warp_id && !rays->encoded_mask)
    return ctx->fail(NERFDS_EINVAL, "the mask network looks its embedding up from metadata['warp'] ids (models.py:924-926): pass warp_id or encoded_mask");
  if (extra->render_opt_flags & ~(NERFDS_OPT_DUST_THRESHOLD | NERFDS_OPT_BOUNDING_BOX)) return ctx->fail(NERFDS_EINVAL, "unknown render_opt_flags");
  if (extra->sample_at_infinity_override < NERFDS_TRISTATE_NONE || extra->sample_at_infinity_override > NERFDS_TRISTATE_FALSE)
    return ctx->fail(NERFDS_EINVAL, "sample_at_infinity_override must be a NERFDS_TRISTATE_* value");
  if (extra->mask_ratio != 1.0f && ctx->cfg.use_predicted_mask && !rays->gt_mask)
    return ctx->fail(NERFDS_EINVAL, "rays_dict['mask'] is required when mask_ratio != 1");
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  int rc = ensure_packed(ctx, prec);
  if (rc != NERFDS_OK) return rc;

  KArgs ka{};
  ka.origins = rays->origins; ka.directions = rays->directions; ka.viewdirs = rays->viewdirs;
  ka.warp_id = rays->warp_id; ka.gt_mask = rays->gt_mask;
  ka.enc_warp = ctx->cfg.use_warp ? rays->encoded_warp : nullptr;
  ka.enc_mask = ctx->cfg.use_predicted_mask ? rays->encoded_mask : nullptr;
  ka.opt_flags = (int)extra->render_opt_flags;
  ka.dust_threshold = extra->dust_threshold;
  for (int i = 0; i < 6; ++i) ka.bbox[i] = extra->bounding_box[i];
  ka.t_rand = rnd ? rnd->t_rand : nullptr;
  ka.u_rand = rnd ? rnd->u_rand : nullptr;
  ka.seed = rnd ? rnd->seed : 0;
  ka.first_ray = rnd ? rnd->first_ray : 0;
  for (int i = 0; i < 3; ++i) { ka.wstream[i] = ctx->wstream[prec][i].p; ka.bias[i] = static_cast<const float*>(ctx->wbias[prec][i].p); }
  if (ctx->cfg.num_fine_samples == 0) { ka.wstream[2] = ka.wstream[1]; ka.bias[2] = ka.bias[1]; }
#ifdef NERFDS_EXP_ONE_NERF_STREAM
  // MEASUREMENT BUILD ONLY (tools/variant.sh, results wrong by construction): the fine level walks the coarse level's weight stream, i.e. the
  // weight set every XCD's L2 has to hold shrinks from shared + 2 NerfMLPs to shared + 1 - what "L2 overflow" costs the kernel
  ka.wstream[2] = ka.wstream[1];
#endif
  ka.warp_embed = static_cast<const float*>(ctx->warp_embed.p);
  ka.mask_embed = static_cast<const float*>(ctx->mask_embed.p);
  ka.ray_fine = out->ray_fine; ka.ray_coarse = out->ray_coarse; ka.smp_fine = out->sample_fine; ka.smp_coarse = out->sample_coarse;
  ka.num_rays = (int)rays->num_rays;
  if (rays->camera) {
    // camera_to_rays ahead of sampling, on the same stream, into a library-owned device scratch (48 B/ray) - the rays
    // never cross PCIe.  (It is a separate tiny HBM-bound launch rather than a branch inside the 1000-MFMA ray kernel:
    // hipcc miscompiled the fine-level evaluation of the split-bf16 kernel when the Newton loop lived in there.)
    const size_t need = (size_t)rays->num_rays * 6 * sizeof(float);
    if (ctx->ray_scratch.bytes < need) {          // grow-only; the old block may still be in use by earlier launches
      (void)hipStreamSynchronize(static_cast<hipStream_t>(hip_stream));
      if (ctx->ray_scratch.p) { (void)hipFree(ctx->ray_scratch.p); ctx->ray_scratch.p = nullptr; ctx->ray_scratch.bytes = 0; }
      if (hipMalloc(&ctx->ray_scratch.p, need) != hipSuccess) return ctx->fail(NERFDS_ENOMEM, "hipMalloc of %zu bytes of ray scratch failed", need);
      ctx->ray_scratch.bytes = need;
    }
    float* o = static_cast<float*>(ctx->ray_scratch.p);
    float* d = o + (size_t)rays->num_rays * 3;
    nerfds::CameraParams cp;
    std::memcpy(&cp, rays->camera, sizeof cp);
    nerfds_launch_camera_rays(cp, rays->first_pixel, rays->num_rays, nullptr, o, d, nullptr, hip_stream);
    ka.origins = o; ka.directions = d; ka.viewdirs = d;
  }
  ka.num_embeds = ctx->cfg.num_warp_embeds > 0 ? ctx->cfg.num_warp_embeds : 1;
  ka.nc = ctx->cfg.num_coarse_samples; ka.nf = ctx->cfg.num_fine_samples;
  ka.stratified = extra->use_stratified_sampling;
  ka.lindisp = extra->use_linear_disparity;
  // per level (models.py:1509 vs :1544): the coarse level keeps the model's value, the per-call override reaches the fine level only
  ka.sample_at_infinity = ctx->cfg.use_sample_at_infinity;
  ka.sample_at_infinity_fine = extra->sample_at_infinity_override == NERFDS_TRISTATE_NONE ? ctx->cfg.use_sample_at_infinity
                               : (extra->sample_at_infinity_override == NERFDS_TRISTATE_TRUE ? 1 : 0);
  ka.white_bkgd = ctx->cfg.use_white_background;
  ka.near_ = extra->near; ka.far_ = extra->far;
  ka.mask_ratio = extra->mask_ratio;
  window(ka.win_mask, ctx->cfg.mask_max_deg, extra->warp_alpha);
  window(ka.win_warp, ctx->cfg.warp_max_deg, extra->warp_alpha);
  window(ka.win_hyp, ctx->cfg.hyper_sheet_max_deg, extra->hyper_sheet_alpha);
  window(ka.win_sp, ctx->cfg.spatial_point_max_deg, extra->nerf_alpha);
  window(ka.win_hp, ctx->cfg.hyper_point_max_deg, extra->hyper_alpha);
  window(ka.win_nm, ctx->cfg.norm_input_max_deg, extra->norm_input_alpha);

#ifdef NERFDS_PROF
  static unsigned long long* prof_dev = nullptr;        // MEASUREMENT BUILD ONLY (tools/prof_phases.sh)
  if (!prof_dev) (void)hipMalloc(&prof_dev, 16 * sizeof(unsigned long long));
  (void)hipMemset(prof_dev, 0, 16 * sizeof(unsigned long long));
  ka.prof = prof_dev;
#endif
  hipStream_t stream = static_cast<hipStream_t>(hip_stream);
  std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
  const bool timed = ctx->timing && ctx->events.size() < nerfds_ctx::MAX_TIMED;
  if (timed) {
    if (!ctx->event_pool.empty()) {
      ev = ctx->event_pool.back();
      ctx->event_pool.pop_back();
    } else if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) {
      return ctx->fail(NERFDS_EDEVICE, "hipEventCreate failed");
    }
    (void)hipEventRecord(ev.first, stream);
  }
  launcher(ctx->graph, prec)(ka, ctx->num_cus, stream);   // grid = min(ray groups, CUs) persistent workgroups
  hipError_t e = hipGetLastError();
  if (timed) {
    (void)hipEventRecord(ev.second, stream);
    ctx->events.push_back(ev);
  }
  if (e != hipSuccess) return ctx->fail(NERFDS_EDEVICE, "kernel launch failed: %s", hipGetErrorString(e));
#ifdef NERFDS_PROF
  {
    unsigned long long p[16] = {};
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(p, prof_dev, sizeof p, hipMemcpyDeviceToHost);
    if (p[3]) fprintf(stderr, "NERFDS_PROF rays=%lld waves=%llu cycles/wave=%.0f  layer chains %.1f%%  field evaluation outside the chains (encodings, exp_se3, heads' activations) %.1f%%  "
                              "compositing (both levels) %.1f%%  resampling: pdf / cdf / bins %.1f%%, inverse cdf %.1f%%, rank sort %.1f%%, state move %.1f%%  rest (ray setup, syncs) %.1f%%\n", (long long)rays->num_rays, p[3],
                      (double)p[0] / p[3], 100.0 * p[1] / p[0], 100.0 * ((double)p[2] - (double)p[1]) / p[0], 100.0 * p[4] / p[0], 100.0 * p[5] / p[0], 100.0 * p[6] / p[0], 100.0 * p[7] / p[0],
                      100.0 * p[8] / p[0], 100.0 * ((double)p[0] - (double)p[2] - (double)p[4] - (double)(p[5] + p[6] + p[7] + p[8])) / p[0]);
    if (p[11]) fprintf(stderr, "NERFDS_PROF_BOUND stage boundaries per wave %.0f: s_waitcnt %.1f cycles each (%.1f%% of the wave), s_barrier + one timer read %.1f cycles each (%.1f%%)\n",
                       (double)p[11] / p[3], (double)p[9] / p[11], 100.0 * p[9] / p[0], (double)p[10] / p[11], 100.0 * p[10] / p[0]);
  }
#endif
  return NERFDS_OK;
}

int nerfds_encode_embed(nerfds_ctx* ctx, int32_t table, const float* metadata, int32_t channels, int64_t num_rays, float* out, void* hip_stream) {
  if (!ctx) return NERFDS_EINVAL;
  if (!ctx->W.loaded) return ctx->fail(NERFDS_EINVAL, "nerfds_ctx_load_weights has not been called");
  if ((table != 0 && table != 1) || (channels != 1 && channels != 3) || num_rays < 0 || (num_rays > 0 && (!metadata || !out)))
    return ctx->fail(NERFDS_EINVAL, "nerfds_encode_embed: table 0 | 1, channels 1 | 3, non-null pointers");
  const float* tab = static_cast<const float*>(table == 0 ? ctx->warp_embed.p : ctx->mask_embed.p);
  if (!tab) return ctx->fail(NERFDS_EINVAL, "this graph has no %s table", table == 0 ? "warp_embed" : "mask_embed");
  if (num_rays == 0) return NERFDS_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(NERFDS_EDEVICE, "hipSetDevice failed");
  nerfds_launch_encode_embed(tab, ctx->cfg.num_warp_embeds > 0 ? ctx->cfg.num_warp_embeds : 1, metadata, channels, num_rays, out, hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return ctx->fail(NERFDS_EDEVICE, "kernel launch failed: %s", hipGetErrorString(e));
  return NERFDS_OK;
}

int nerfds_camera_to_rays(int device, const nerfds_camera* cam, int64_t first_pixel, int64_t n, const float* pixels,
                          float* origins, float* directions, float* pixels_out, void* hip_stream) {
  if (!cam || n < 0 || first_pixel < 0 || cam->image_width <= 0 || cam->image_height <= 0 || cam->focal_length == 0.f) {
    g_create_error = "nerfds_camera_to_rays: invalid argument";
    return NERFDS_EINVAL;
  }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "no such HIP device"; return NERFDS_EDEVICE; }
  nerfds::CameraParams cp;
  std::memcpy(&cp, cam, sizeof cp);
  nerfds_launch_camera_rays(cp, first_pixel, n, pixels, origins, directions, pixels_out, hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return NERFDS_EDEVICE; }
  return NERFDS_OK;
}

int nerfds_frame_images(int device, const float* ray_records, int32_t height, int32_t width, double near_, double far_,
                        const double* colormap, uint8_t* rgb_u8, uint8_t* debug_u8, void* hip_stream) {
  if (height < 0 || width < 0 || (!ray_records && (int64_t)height * width > 0) || (debug_u8 && !colormap)) {
    g_create_error = "nerfds_frame_images: invalid argument";
    return NERFDS_EINVAL;
  }
  if ((int64_t)height * width == 0 || (!rgb_u8 && !debug_u8)) return NERFDS_OK;
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "no such HIP device"; return NERFDS_EDEVICE; }
  // scale_values (visualization.py:195-196): the range is formed in double (python floats), then applied in float32
  const double range = std::max(far_ - near_, 1e-6);
  nerfds_launch_frame_images(ray_records, height, width, (float)near_, (float)range, colormap, rgb_u8, debug_u8, hip_stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return NERFDS_EDEVICE; }
  return NERFDS_OK;
}

int nerfds_kernel_time_ms(nerfds_ctx* ctx, int reset, double* total_ms) {
  if (!ctx) return NERFDS_EINVAL;
  (void)hipSetDevice(ctx->device);
  int n = 0;
  double tot = 0.0;
  for (auto& e : ctx->events) {
    if (hipEventSynchronize(e.second) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot += ms; ++n; }
    }
  }
  if (total_ms) *total_ms = tot;
  if (reset) {
    for (auto& e : ctx->events) ctx->event_pool.push_back(e);
    ctx->events.clear();
    ctx->timing = true;
  }
  return n;
}

// ---- host-only packing helpers -----------------------------------------------------------------------
int64_t nerfds_pack_stream_bytes(const nerfds_model_cfg* cfg, int which, uint32_t prec) {
  if (!cfg || prec >= NERFDS_PREC_COUNT || which < 0 || which > 1) return NERFDS_EINVAL;
  const int g = graph_of(*cfg);
  if (g < 0) return NERFDS_ENOTSUP;
  int64_t wb, bf;
  stream_dims_dispatch(g, which, -1, (int)prec, &wb, &bf);      // (the larger of the two levels' NerfMLP streams: a buffer that holds either)
  return wb;
}
int64_t nerfds_pack_stream_bytes_level(const nerfds_model_cfg* cfg, int which, int level, uint32_t prec) {
  if (!cfg || prec >= NERFDS_PREC_COUNT || which < 0 || which > 1 || level < 0 || level > 1) return NERFDS_EINVAL;
  const int g = graph_of(*cfg);
  if (g < 0) return NERFDS_ENOTSUP;
  int64_t wb, bf;
  stream_dims_dispatch(g, which, level, (int)prec, &wb, &bf);
  return wb;
}
int nerfds_pack_tile_pair(const nerfds_model_cfg* cfg, uint32_t prec) {
  if (!cfg || prec >= NERFDS_PREC_COUNT) return NERFDS_EINVAL;
  const int g = graph_of(*cfg);
  if (g < 0) return NERFDS_ENOTSUP;
  return tile_pair_of(g, (int)prec);
}
int64_t nerfds_pack_bias_floats(const nerfds_model_cfg* cfg, int which) {
  if (!cfg || which < 0 || which > 1) return NERFDS_EINVAL;
  const int g = graph_of(*cfg);
  if (g < 0) return NERFDS_ENOTSUP;
  int64_t wb, bf;
  stream_dims_dispatch(g, which, -1, 0, &wb, &bf);
  return bf;
}
int nerfds_pack_stream(const nerfds_model_cfg* cfg, const nerfds_weights* w, int which, int level, uint32_t prec,
                       void* stream_out, float* bias_out) {
  if (!cfg || !w || prec >= NERFDS_PREC_COUNT || which < 0 || which > 1 || level < 0 || level > 1) return NERFDS_EINVAL;
  const int g = graph_of(*cfg);
  if (g < 0) return NERFDS_ENOTSUP;
  Weights W;
  std::string err;
  bool ok = take_weights_dispatch(g, W, *cfg, *w, err);
  if (!ok) { g_create_error = err; return NERFDS_EINVAL; }
  StreamWriter sw{static_cast<uint8_t*>(stream_out), bias_out};
  pack_dispatch(g, sw, W, which, level, (int)prec);
  return NERFDS_OK;
}

}  // extern "C"

// ---- device self-test of the MFMA operand / accumulator maps -------------------------------------------
namespace {
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ void mfma_probe(const float* __restrict__ A /*[32][16]*/, const float* __restrict__ B /*[16][32]*/,
                           float* __restrict__ Cb, float* __restrict__ Cf) {
  const int l = threadIdx.x, m = l & 31, h = l >> 5;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) {          // k-slot (h, i) <-> k = 8h + i for BOTH operands
    a[i] = (__bf16)A[m * 16 + 8 * h + i];
    b[i] = (__bf16)B[(8 * h + i) * 32 + m];
  }
  f32x16_t acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  f32x16_t accf = {0};
  for (int i = 0; i < 8; ++i)            // fp32: one k-slot pair (h = 0, 1) per instruction
    accf = __builtin_amdgcn_mfma_f32_32x32x2f32(A[m * 16 + 8 * h + i], B[(8 * h + i) * 32 + m], accf, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // the accumulator map field.h relies on
    Cb[row * 32 + m] = acc[r];
    Cf[row * 32 + m] = accf[r];
  }
}
}  // namespace

extern "C" int nerfds_debug_mfma(int device, const float* a, const float* b, float* c_bf16, float* c_f32) {
  if (hipSetDevice(device) != hipSuccess) return NERFDS_EDEVICE;
  float *da, *db, *dc, *df;
  if (hipMalloc(&da, 32 * 16 * 4) != hipSuccess || hipMalloc(&db, 16 * 32 * 4) != hipSuccess ||
      hipMalloc(&dc, 32 * 32 * 4) != hipSuccess || hipMalloc(&df, 32 * 32 * 4) != hipSuccess)
    return NERFDS_ENOMEM;
  (void)hipMemcpy(da, a, 32 * 16 * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(db, b, 16 * 32 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, da, db, dc, df);
  hipError_t e = hipDeviceSynchronize();
  (void)hipMemcpy(c_bf16, dc, 32 * 32 * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(c_f32, df, 32 * 32 * 4, hipMemcpyDeviceToHost);
  (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(df);
  return e == hipSuccess ? NERFDS_OK : NERFDS_EDEVICE;
}
