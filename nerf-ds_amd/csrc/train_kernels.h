// Declarations shared by train_kernels.hip (element-wise / per-ray kernels) and nerfds_train.cpp (orchestration).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nerfds_train {

struct Dims {      // the nerf_ds graph's widths (from nerfds_model_cfg)
  int mask_bands, warp_bands, hyp_bands, sp_bands, hp_bands, vd_bands, nm_bands;
  int mask_in, warp_in, hyper_in, trunk_in;     // 44, 33, 45, 52
  int warp_ld, hyper_ld;                        // row strides of the warp / hyper-sheet input buffers (and their gradients / tangents): the
                                                // widths rounded up to 4 floats, pad columns zero, so the rows are 16-byte aligned (LDS-DMA)
};
struct Windows {   // posenc windows (model_utils.py:420-436), one weight per band
  float mask[8], warp[8], hyp[8], sp[8], hp[8], nm[8];
};

struct Objective {   // weights of the auxiliary losses (0 = off); mirrors nerfds_train_objective
  float warp_reg_weight, warp_reg_alpha, warp_reg_scale, back_facing_weight, mask_loss_weight, sharp_weights_std;
  int use_sharp_weights;
  float hyper_reg_weight;
  float elastic_weight;       // elastic regulariser on the warp Jacobian (coarse level), elastic_by_weight: 'weight' reduction instead of 'median'
  int elastic_by_weight;
  float mask_occlusion_weight;   // training.py:409-417
};

void coarse_z(hipStream_t, int R, int Nc, float near_, float far_, int stratified, int lindisp, const float* t_rand, uint64_t seed, long long first_ray, float* z);
void resample(hipStream_t, int R, int Nc, int Nf, const float* zc, const float* wc, int stratified, const float* u_rand, uint64_t seed, long long first_ray, float* zf, float* scratch,
              float* z_new = nullptr, int* src = nullptr);      // z_new / src: the merged step (resample_has_sources sizes only)
bool resample_has_sources(int Nc, int Nf);
void gather_rows(hipStream_t, long long M, const int* src, const float* xw, const float* wamb, const float* wv, float* xw_f, float* wamb_f, float* wv_f);
void scatter_rows(hipStream_t, long long M, const int* src, long long add_below, const float* dxw_f, const float* dwamb_f, float* dxw, float* dwamb);
void gather_cols(hipStream_t, long long M, const int* src, int rp, int C, const float* in, float* out);                              // out[i] = in[src[i]], rp x C floats per sample
void scatter_cols(hipStream_t, long long M, const int* src, long long add_below, int rp, int C, const float* in_f, float* out);    // out[src[i]] (+)= in_f[i]
void encode_inputs(hipStream_t, const Dims&, int R, int S, const float* o, const float* d, const float* z, const uint32_t* warp_id, int n_embeds,
                   const float* warp_tbl, const float* mask_tbl, const Windows&, float* x, float* mask_in, float* warp_in, float* hyper_in);
void bias_act(hipStream_t, float* y, const float* b, long long M, int N, int ld, int relu);
void mask_post(hipStream_t, const Dims&, int R, int S, const float* logit, const float* gt, float ratio, float* warp_in, float* hyper_in);
void se3_fwd(hipStream_t, long long M, const float* wv, const float* x, float* xw);
void se3_bwd(hipStream_t, long long M, const float* wv, const float* x, const float* dxw, const float* dwv_extra, float* dwv);
void trunk_in(hipStream_t, const Dims&, long long M, const float* xw, const float* wamb, const Windows&, float* tin);
void trunk_in_bwd(hipStream_t, const Dims&, long long M, const float* dtin, const float* xw, const float* wamb, const Windows&, const float* dxw_extra,
                  const float* dwamb_extra, float* dxw, float* dwamb);
void norm_loss(hipStream_t, int R, int S, float weight, const float* weights, const float* alpha, const float* t_alpha, const float* wv,
               const float* target_norm, float* term, float* d_alpha, float* d_t_alpha, float* du, float* ghat);
void trunk_in_jvp_bwd(hipStream_t, const Dims&, long long M, const float* d_t_tin, const float* xw, const float* wamb, const float* t_xw,
                      const float* t_wamb, const Windows&, float* d_t_xw, float* d_t_wamb, float* dxw_extra, float* dwamb_extra, int rp = 3);
void se3_jvp_bwd(hipStream_t, long long M, const float* wv, const float* x, const float* t_wv, const float* d_t_xw, const float* du,
                 const float* ghat, float* d_t_wv, float* dwv_extra, const float* extra_in = nullptr, const float* dir = nullptr);
void se3_rot_bwd(hipStream_t, long long M, const float* wv, const float* du, const float* ghat, float* out);
// the reverse-mode second-order path (train_kernels.hip: k_fill_head4 ...)
void fill_head4(hipStream_t, long long M, const float* scale_dev, float value, float* out);
void posenc_rev_x(hipStream_t, const Dims&, long long M, const float* x, const float* d_warp_in, const float* d_hyper_in, const Windows&, float* gx);
void sigma_grad_assemble(hipStream_t, long long M, const float* wv, const float* a, const float* gx, float* t_alpha);
void make_dir(hipStream_t, long long M, const float* d_t_alpha, const float* slot, float* dir, float* cot);
void encode_tangent_dir(hipStream_t, const Dims&, long long M, const float* x, const float* dir, const Windows&, float* t_warp_in, float* t_hyper_in);
void se3_jvp_dir(hipStream_t, long long M, const float* wv, const float* x, const float* dir, const float* t_wv, float* t_xw);
void aux_losses(hipStream_t, int R, int S, const Objective&, const float* z, const float* weights, const float* x, const float* xw, const float* alpha,
                const float* viewdirs, const float* mask_logit, const float* gt_mask, float* terms, float* dxw_reg, float* d_alpha, float* d_pm,
                const float* wamb = nullptr, float* term_hyper = nullptr, float* dwamb_reg = nullptr, float* term_occlusion = nullptr);   // hyper-point regulariser: ambient coordinates in, its term and d / d wamb out
void add_inplace(hipStream_t, float* dst, const float* src, long long n);
// slot[4] (device) <- {amax bits of x[0..n), 2^(target_log2 - floor(log2 amax)), its inverse, 1 / (scale * x_scale)}
void pick_scale(hipStream_t, const float* x, long long n, float target_log2, float x_scale, float* slot);
// overflow diagnosis after a failed gradient check: flags |= bit if the f16 / fp32 array holds an inf or a NaN (scan_half: n a multiple of 8, 16-byte aligned)
void scan_half(hipStream_t, const uint16_t* p, long long n, unsigned* flags, unsigned bit);
void scan_float(hipStream_t, const float* p, long long n, unsigned* flags, unsigned bit, float limit = 3.4028234e38f);      // |p[i]| > limit counts (an fp32 array on its way into f16)
void expand_half(hipStream_t, const uint16_t* h16, float* out, long long n);      // out[i] = float(f16 h16[i]); n a multiple of 8, both 16-byte aligned
// background regulariser (training.py:159-183): term += weight * mean_i general_loss(|xw_i - x_i|^2, alpha, scale); dxw = its gradient w.r.t. xw
// elastic regulariser (training.py:112-156 'log_svals', 274-295): t_xw = the tangents of the warped point, row 3 m + j = d x' / d x_j (the Jacobian's
// column j); term += weight * mean_rays sum_selected f * general_loss(sum_i log^2 max(s_i, 1e-6), -2, 0.03); d_t_xw += its gradient
void elastic_loss(hipStream_t, int R, int S, float weight, int by_weight, const float* weights, const float* t_xw, float* term, float* d_t_xw);
void background_loss(hipStream_t, long long B, const float* x, const float* xw, float weight, float alpha, float scale, float* term, float* dxw);
void alpha_post(hipStream_t, const Dims&, int R, int S, const float* alpha, const float* wv, const float* viewdirs, const Windows&, float* sigma, float* cond);
// cotangents of a level's per-ray outputs from a caller-defined loss (nerfds_render_rays_bwd) / where nerfds_trainer_forward wants them written
struct LevelCot { int on; const float* d_rgb; const float* d_depth; const float* d_acc; };
struct LevelOut { float* rgb; float* depth; float* acc; };
void composite_loss(hipStream_t, int R, int S, const float* z, const float* dirs, const float* sigma, const float* rgb_logit, const float* target,
                    int at_infinity, int white, float* rgb_ray, float* weights, float* loss, float* d_rgb_logit, float* d_alpha, const LevelCot& cot = LevelCot{0, nullptr, nullptr, nullptr},
                    const LevelOut& out = LevelOut{nullptr, nullptr, nullptr});
void relu_bwd(hipStream_t, float* dy, const float* y, long long n);
void relu_bwd_colsum(hipStream_t, float* dy, const float* y, long long M, int N, float* db);
void colsum_add(hipStream_t, const float* dz, long long M, int N, int ld, float* db);
void shared_in_bwd(hipStream_t, const Dims&, int R, int S, const float* d_warp_in, const float* d_hyper_in, const float* mask_logit, float ratio,
                   const float* d_pm_extra, const uint32_t* warp_id, int n_embeds, float* d_warp_tbl, float* d_mask_logit);
void mask_in_bwd(hipStream_t, const Dims&, int R, int S, const float* d_mask_in, const uint32_t* warp_id, int n_embeds, float* d_mask_tbl);
void sum_partials(hipStream_t, const float* part, int slabs, long long n, float* out);
void fill(hipStream_t, float* p, long long n, float v);
void encode_tangents(hipStream_t, const Dims&, long long M, const float* x, const Windows&, float* t_warp_in, float* t_hyper_in);
void relu_mask3(hipStream_t, float* t, const float* y, long long M, int N);
void se3_jvp(hipStream_t, long long M, const float* wv, const float* x, const float* t_wv, float* t_xw);
void trunk_in_jvp(hipStream_t, const Dims&, long long M, const float* xw, const float* wamb, const float* t_xw, const float* t_wamb, const Windows&, float* t_tin, int rp = 3);
void target_norm(hipStream_t, long long M, const float* t_alpha, const float* wv, float* out);
void clip_gradients(hipStream_t, float* g, long long n, float max_val, float max_norm, float* sumsq_scratch);
void adam(hipStream_t, float* p, const float* g, float* m1, float* m2, long long n, float lr, float b1, float b2, float eps, long long* step_dev,
          float* corr_dev, unsigned* nonfinite_flag);      // flag != 0 after the gradient check: nothing is updated
// fused forward: weight streams / biases / folded rgb layer from the current parameters (see train_kernels.hip)
// map: two units (2 KiB) per fragment; stream: split bf16 (two units), fragments [x6_lo, x6_hi) exact fp32 (wide_f32) or three units (hi | mid | lo)
// The step's packing as ONE launch: streams (n = fragments; mode = k_pack_batch's wide_f32: 0 split bf16 with the three-way range [x6_lo, x6_hi), 1 that range exact
// fp32, 2 one f16 unit per fragment) and bias vectors (mode 3, n values) collected by the host, every item gathered from the same theta / fold
struct PackItem { const int* map; void* out; int n, x6_lo, x6_hi, mode; };
struct PackBatch {
  static constexpr int MAX = 28;
  PackItem it[MAX];
  int n = 0;
  bool overflow = false;      // more than MAX items: a programming error, pack_batch aborts on it
  bool stream(const int* map, void* out, int nfrag, int x6_lo = 0, int x6_hi = 0, int wide_f32 = 0) { return add(PackItem{map, out, nfrag, x6_lo, x6_hi, wide_f32}); }
  bool bias(const int* map, float* out, int count) { return add(PackItem{map, out, count, 0, 0, 3}); }
  bool add(const PackItem& i) { if (n >= MAX) { overflow = true; return false; } it[n++] = i; return true; }
};
void pack_batch(hipStream_t, const float* theta, const float* fold, long long P, const PackBatch&);
void fold_rgb(hipStream_t, const float* B, const float* Bb, const float* K, const float* Kb, int TW, int W, int row_x, float* fold);
// gradients of the activation-free bottleneck Dense from S = trunk_out^T g_rgb [TW x W] and c = colsum(g_rgb) [W] (fused backward):
//   dKb[TW x W] = Wb^T S + bb (x) c,  dWb[TW x TW] = S Kb^T,  dbb[TW] = Kb c;  Wb [TW x TW], Kb = the first TW rows of K [.. x W]
// One launch for every level (k_bott_grads).
struct BottItem { int TW, W; const float* Wb; const float* bb; const float* K; const float* S; const float* c; float* dKb; float* dWb; float* dbb; };
struct BottBatch { BottItem lv[2]; int n = 0; };
void bott_grads(hipStream_t, const BottBatch&);

}  // namespace nerfds_train
