/*
 * nerfds.h - C ABI of the MI355X-native NeRF-DS volume-rendering core.
 *
 * The reference (JokerYan/NeRF-DS, JAX/Flax) has no FFI: its hot path is entered through three nested
 * Python callables.  This ABI sits one level below the Python layer that keeps
 * those signatures (nerf-ds_amd/nerfds_amd/model.py), and each entry point names the reference
 * interface it replaces:
 *
 *   nerfds_ctx_create        <- models.construct_nerf / NerfModel.setup      (hypernerf/models.py:2677-2741, 324-391)
 *   nerfds_ctx_load_weights  <- state.optimizer.target['model'] passed to model.apply (hypernerf/evaluation.py:119)
 *   nerfds_render_rays       <- NerfModel.__call__ via model.apply            (hypernerf/models.py:1419-1565,
 *                               called from render.py:140-154 and evaluation.py:119)
 *   nerfds_ctx_destroy       <- (garbage collection of the JAX arrays)
 *   nerfds_last_error        <- Python exceptions raised by the model        (models.py:315,328,561,744,1131)
 *
 * Conventions
 *   - Plain C types only.  "device pointer" = pointer into HIP device memory owned by the caller
 *     (e.g. torch tensors); the library never allocates outputs and never frees inputs.
 *   - Return 0 on success, a negative errno-style code otherwise; no C++ exception crosses the boundary.
 *   - One ctx per device.  Calls on a ctx are stream-ordered on the hipStream_t passed in and are not
 *     re-entrant (the reference is single-threaded on the host, SURVEY.md section 8b "Threading").
 *   - All floating-point arrays are float32, ids are uint32 - the reference's dtypes.
 */
#ifndef NERFDS_H_
#define NERFDS_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NERFDS_ABI_VERSION 7

/* error codes */
#define NERFDS_OK         0
#define NERFDS_EINVAL   (-22)
#define NERFDS_ENOMEM   (-12)
#define NERFDS_ENOTSUP  (-95)   /* graph / option not built as a HIP kernel */
#define NERFDS_EDEVICE  (-5)    /* HIP runtime error (no device, launch failure, ...) */
#define NERFDS_ENONFINITE (-34) /* trainer: the gradient vector held an inf / NaN; the Adam update of that step was skipped as a whole */

/* Arithmetic of the per-sample dense layers (flags bits 0-2 of nerfds_render_rays).  The reference's layers are
 * fp32 nn.Dense (hypernerf/modules.py:61-65,74-78); every mode accumulates in fp32. */
#define NERFDS_PREC_BF16    0u  /* bf16 x bf16 -> fp32 MFMA (v_mfma_f32_32x32x16_bf16): the throughput path   */
#define NERFDS_PREC_BF16X3  1u  /* split-bf16 (hi+lo) x3 MFMA: ~fp32 accuracy at 1/3 of the bf16 MFMA rate   */
#define NERFDS_PREC_F32     2u  /* fp32 MFMA (v_mfma_f32_32x32x2_f32): exact fp32 fma chains, parity gate    */
#define NERFDS_PREC_F16     3u  /* f16 x f16 -> fp32 MFMA (v_mfma_f32_32x32x16_f16): bf16 rate, 3 more significand bits */
#define NERFDS_PREC_MIXED   4u  /* per-network plan (csrc/graphs.h plan_of): the warp field in split bf16, the bulk of the
                                   FLOPs in one f16 MFMA per product.  Does NOT meet the 1e-4 tolerance: measured 4.6e-4 on the
                                   bench sample, bounded at 2e-3 in tests (profiles/r3_precision_budget.md: no plan with a
                                   one-MFMA network meets it; only BF16X3 and F32 do) */
#define NERFDS_PREC_BF16X3_FINE 5u  /* split bf16 everywhere EXCEPT the coarse level's NerfMLP, which runs one f16 MFMA per product: the fine
                                   * level - the one render_fn returns (evaluation.py:121-124) - sees of the coarse NerfMLP only the weights its depths
                                   * are drawn from and stays within 1e-4 (measured 3.8e-5 against 3.7e-5 for NERFDS_PREC_BF16X3 on 131 072 rays of the
                                   * bench frame); the COARSE level's own outputs are f16-grade (rgb 5e-4).  17 % fewer MFMAs than BF16X3.  A
                                   * single-level model runs plain BF16X3. */
#define NERFDS_PREC_F16X3   6u  /* split f16 (hi + lo, 11 + 11 significand bits) x3 MFMA on v_mfma_f32_32x32x16_f16: the MFMA count of NERFDS_PREC_BF16X3 at ~20 x
                                   * its accuracy on composited RGB (fp32-MFMA grade: holds 1e-4 on the badly conditioned rays where split bf16 does not);
                                   * f16's RANGE: an activation beyond 65504 is inf here, which split bf16 (fp32's exponent range) cannot produce */
#define NERFDS_PREC_COUNT   7u
#define NERFDS_PREC_MASK    7u
/* Other flags. */
#define NERFDS_FLAG_USE_WARP_OFF  (1u << 4)  /* NerfModel.__call__(use_warp=False), models.py:1468 - rejected if the graph has a warp */

/* Number of floats in one per-ray output record (see nerfds_ray_field). */
#define NERFDS_RAY_REC  26
/* Per-ray record layout, AoS [R][NERFDS_RAY_REC] so a ray is one 104-byte coalesced store and the
 * multi-GPU exchange is one all-gather of one tensor.  Keys are the reference's out-dict keys
 * (models.py:1312-1415, consumers render.py:192-193). */
enum nerfds_ray_field {
  NERFDS_RAY_RGB = 0,                 /* [3]  'rgb'                     */
  NERFDS_RAY_DEPTH = 3,               /* [1]  'depth'                   */
  NERFDS_RAY_MED_DEPTH = 4,           /* [1]  'med_depth'               */
  NERFDS_RAY_ACC = 5,                 /* [1]  'acc'                     */
  NERFDS_RAY_NORM = 6,                /* [3]  'ray_norm'                */
  NERFDS_RAY_ROTATION_FIELD = 9,      /* [3]  'ray_rotation_field'      */
  NERFDS_RAY_TRANSLATION_FIELD = 12,  /* [3]  'ray_translation_field'   */
  NERFDS_RAY_DELTA_X = 15,            /* [3]  'ray_delta_x'             */
  NERFDS_RAY_HYPER_POINTS = 18,       /* [2]  'ray_hyper_points'        */
  NERFDS_RAY_PREDICTED_MASK = 20,     /* [1]  'ray_predicted_mask'      */
  NERFDS_RAY_MED_POINTS = 21          /* [5]  'med_points'              */
};

/* Number of floats in one optional per-sample record. */
#define NERFDS_SAMPLE_REC  18
enum nerfds_sample_field {
  NERFDS_SMP_Z = 0,               /* z_vals                           */
  NERFDS_SMP_SIGMA = 1,           /* 'sigma' (after softplus)         */
  NERFDS_SMP_ALPHA = 2,           /* 'alpha'                          */
  NERFDS_SMP_ACCUM_PROD = 3,      /* 'accum_prod'                     */
  NERFDS_SMP_WEIGHTS = 4,         /* 'weights'                        */
  NERFDS_SMP_PREDICTED_MASK = 5,  /* 'predicted_mask'                 */
  NERFDS_SMP_RGB = 6,             /* [3] per-sample colour            */
  NERFDS_SMP_PREDICTED_NORM = 9,  /* [3] 'predicted_norm' (raw)       */
  NERFDS_SMP_WARPED_POINTS = 12,  /* [5] 'warped_points'              */
  NERFDS_SMP_BACK_FACING = 17     /* 'back_facing'                    */
};

/* Static description of the render graph: the gin-resolved fields of NerfModel (models.py:116-229),
 * SE3Field (warping.py:139-157), HyperSheetMLP (modules.py:354-365), MaskMLP (modules.py:396-407).
 * The library matches it against the graphs it has compiled kernels for and returns
 * NERFDS_ENOTSUP with a message otherwise. */
typedef struct nerfds_model_cfg {
  int32_t abi_version;             /* NERFDS_ABI_VERSION */
  int32_t num_coarse_samples, num_fine_samples;
  int32_t use_warp, use_hyper_sheet, use_predicted_mask, predict_norm, use_x_in_rgb_condition;
  int32_t use_mask_in_warp, use_mask_in_hyper, use_viewdirs, mask_output_relu;
  int32_t nerf_trunk_depth, nerf_trunk_width, nerf_skip, nerf_rgb_branch_depth, nerf_rgb_branch_width;
  int32_t spatial_point_max_deg, hyper_point_max_deg, viewdir_max_deg, norm_input_max_deg;
  int32_t warp_max_deg, warp_trunk_depth, warp_trunk_width, warp_skip;
  int32_t hyper_sheet_max_deg, hyper_sheet_depth, hyper_sheet_width, hyper_sheet_skip, hyper_num_dims;
  int32_t mask_max_deg, mask_depth, mask_width, mask_skip;
  int32_t glo_num_dims, num_warp_embeds;
  int32_t use_white_background, use_sample_at_infinity;
  int32_t use_posenc_identity, warp_use_posenc_identity;   /* NerfModel / SE3Field.use_posenc_identity (models.py:143, warping.py:141) */
} nerfds_model_cfg;

/* One nn.Dense: kernel is row-major [in][out] (Flax layout), y = x @ kernel + bias. HOST pointers. */
typedef struct nerfds_dense {
  const float* kernel;
  const float* bias;
  int32_t in_dim, out_dim;
} nerfds_dense;

#define NERFDS_MAX_DEPTH 16

typedef struct nerfds_nerf_mlp {           /* modules.NerfMLP, modules.py:122-152 */
  nerfds_dense trunk[NERFDS_MAX_DEPTH];    /* trunk_mlp/hidden_i */
  nerfds_dense bottleneck;                 /* bottleneck */
  nerfds_dense alpha;                      /* alpha_mlp/logit : out = 1 (+3 if predict_norm) */
  nerfds_dense rgb_hidden[NERFDS_MAX_DEPTH]; /* rgb_mlp/hidden_i */
  nerfds_dense rgb;                        /* rgb_mlp/logit */
} nerfds_nerf_mlp;

/* fp32 HOST views of the parameter tree ('params/model/...' of the Flax checkpoint). Unused nets: NULL kernels. */
typedef struct nerfds_weights {
  const float* warp_embed;                 /* warp_embed/embed/embedding  [num_warp_embeds][glo_num_dims] */
  const float* mask_embed;                 /* mask_embed/embed/embedding  [num_warp_embeds][glo_num_dims] */
  nerfds_dense mask_hidden[NERFDS_MAX_DEPTH];   /* mask_mlp/MLP_0/hidden_i */
  nerfds_dense mask_out;                        /* mask_mlp/MLP_0/logit    */
  nerfds_dense warp_hidden[NERFDS_MAX_DEPTH];   /* warp_field/trunk/hidden_i */
  nerfds_dense warp_w, warp_v;                  /* warp_field/branches_{w,v}/logit */
  nerfds_dense hyper_hidden[NERFDS_MAX_DEPTH];  /* hyper_sheet_mlp/MLP_0/hidden_i */
  nerfds_dense hyper_out;                       /* hyper_sheet_mlp/MLP_0/logit */
  nerfds_nerf_mlp nerf[2];                      /* nerf_mlps_coarse, nerf_mlps_fine */
  int32_t embed_rows;                           /* rows of the GLO tables above; must equal cfg.num_warp_embeds (the
                                                   library copies embed_rows * glo_num_dims floats from each pointer) */
} nerfds_weights;

/* Pinhole camera with radial / tangential distortion: the JSON fields of hypernerf/camera.py:140-161 (Camera.from_json;
 * the legacy key "tangential" == tangential_distortion).  orientation is the world->camera rotation, row major. */
typedef struct nerfds_camera {
  float orientation[9];
  float position[3];
  float focal_length;
  float principal_point[2];
  float skew;
  float pixel_aspect_ratio;
  float radial_distortion[3];
  float tangential_distortion[2];
  int32_t image_width, image_height;
} nerfds_camera;

/* Ray batch: DEVICE pointers, R rays.  rays_dict of models.py:1444-1478. */
typedef struct nerfds_rays {
  int64_t num_rays;
  const float* origins;       /* [R][3] */
  const float* directions;    /* [R][3] */
  const float* viewdirs;      /* [R][3]; NULL -> directions (models.py:1475-1478) */
  const uint32_t* warp_id;    /* [R] metadata['warp'] (GLO row); NULL if the graph has no warp */
  const float* gt_mask;       /* [R] rays_dict['mask']; only read when mask_ratio != 1; may be NULL */
  /* Fused camera -> rays (render.py:201 camera_to_rays + evaluation.py:78-98 flatten/chunk): when camera != NULL
   * (HOST pointer), origins/directions/viewdirs are ignored and ray r is generated on the GPU (a camera kernel on the
   * same stream, into library-owned device scratch) for the row-major pixel centre first_pixel + r
   * (origin = camera position, viewdir = direction). */
  const nerfds_camera* camera;
  int64_t first_pixel;
  /* metadata_encoded=True (models.py:898-899, 908-909): per-ray GLO vectors instead of ids - what evaluation.encode_metadata
   * (evaluation.py:29-50) produced with NerfModel._encode_embed (models.py:271-294; nerfds_encode_embed below), e.g. the interpolation of
   * two rows for 3-channel metadata (left id, right id, progression).  encoded_warp [R][glo_num_dims] replaces the warp_embed row of warp_id
   * (and is the hyper-sheet embedding too: every built graph has hyper_use_warp_embed); NULL = ids.  The mask network's embedding is
   * looked up from warp_id in the reference even then (models.py:924-926); encoded_mask [R][glo_num_dims], if given, replaces that row
   * (no reference counterpart: 3-channel metadata has no integer id to look up).  DEVICE pointers. */
  const float* encoded_warp;
  const float* encoded_mask;
} nerfds_rays;

/* Runtime scalars: state.extra_params (model_utils.py:41-52) + the kwargs of NerfModel.__call__ that the
 * render/train harness passes (render.py:150-153). */
typedef struct nerfds_extra {
  float nerf_alpha, warp_alpha, hyper_alpha, hyper_sheet_alpha, norm_input_alpha;
  float mask_ratio;           /* render.py:152: always 1 at inference */
  float near, far;
  int32_t use_stratified_sampling;   /* NerfModel.use_stratified_sampling */
  /* render_opts of NerfModel.__call__ (filter_sigma, models.py:38-66): densities below dust_threshold, and of samples whose
   * observation-space point lies outside the box [xmin, xmax, ymin, ymax, zmin, zmax], are zeroed before compositing (models.py:1288);
   * the per-sample 'sigma' output stays unfiltered (models.py:1271).  render_opt_flags = 0: render_opts is None.
   * FINE level only, as in the reference: __call__ forwards render_opts to the 'fine' render_samples call (models.py:1545) and not to
   * the 'coarse' one (models.py:1493-1517, default None at :884) - the coarse outputs and the pdf the fine depths are drawn from are
   * those of an unfiltered render; a single-level model (num_fine_samples == 0) never sees the options. */
  uint32_t render_opt_flags;         /* NERFDS_OPT_* */
  float dust_threshold;
  float bounding_box[6];
  int32_t use_linear_disparity;      /* NerfModel.use_linear_disparity: coarse depths linear in 1 / z (model_utils.py:73-76) */
  /* The use_sample_at_infinity kwarg of NerfModel.__call__ (models.py:1433, 1484-1485): NERFDS_TRISTATE_NONE = the model's
   * use_sample_at_infinity, NERFDS_TRISTATE_TRUE / _FALSE = the override.  It reaches the FINE level only (models.py:1544); the
   * coarse level always composites with the model's value (models.py:1509), as does a single-level model. */
  int32_t sample_at_infinity_override;   /* NERFDS_TRISTATE_* */
} nerfds_extra;
#define NERFDS_TRISTATE_NONE  0
#define NERFDS_TRISTATE_TRUE  1
#define NERFDS_TRISTATE_FALSE 2
#define NERFDS_OPT_DUST_THRESHOLD 1u
#define NERFDS_OPT_BOUNDING_BOX   2u

/* Sampling uniforms.  The reference draws them from JAX threefry streams (model_utils.py:84,217) which
 * cannot be reproduced outside JAX; for parity they are injected.  Either pointer NULL -> on-chip
 * Philox4x32-10 keyed by (seed, first_ray + ray index in the call, sample index) - csrc/philox.h; the same stream in
 * nerfds_render_rays and nerfds_trainer_step, so a frame rendered in chunks (first_ray = chunk offset) or a batch
 * run through the trainer in blocks draws what one call over all rays would.  DEVICE pointers. */
typedef struct nerfds_rand {
  const float* t_rand;        /* [R][num_coarse_samples] */
  const float* u_rand;        /* [R][num_fine_samples]   */
  uint64_t seed;
  int64_t first_ray;          /* Philox counter of ray 0 of this call */
} nerfds_rand;

/* Outputs: DEVICE pointers; any may be NULL (not written). */
typedef struct nerfds_out {
  float* ray_fine;            /* [R][NERFDS_RAY_REC]  (the 'fine' level, or 'coarse' if num_fine_samples == 0) */
  float* ray_coarse;          /* [R][NERFDS_RAY_REC]  ('coarse' level when a fine level exists) */
  float* sample_fine;         /* [R][Nc+Nf][NERFDS_SAMPLE_REC] */
  float* sample_coarse;       /* [R][Nc][NERFDS_SAMPLE_REC] */
} nerfds_out;

typedef struct nerfds_ctx nerfds_ctx;

int nerfds_abi_version(void);
/* sizeof the public structs as THIS library was compiled, for a binding to check its own declarations against (host only):
 * which = 0 nerfds_model_cfg, 1 nerfds_weights, 2 nerfds_camera, 3 nerfds_rays, 4 nerfds_extra, 5 nerfds_rand, 6 nerfds_out,
 * 7 nerfds_train_objective, 8 nerfds_train_numerics; anything else: -1. */
int64_t nerfds_struct_size(int which);
/* The arithmetic (NERFDS_PREC_BF16 / BF16X3 / F32 / F16) each network runs in under a NERFDS_PREC_* value of this build:
 * plan_out = {MaskMLP, SE3 warp field, hyper sheet, NerfMLP trunk (+ alpha head), rgb branch}.  Uniform for every value
 * but NERFDS_PREC_MIXED.  (No reference counterpart: the reference's layers are all fp32 nn.Dense, modules.py:61-65.) */
int nerfds_precision_plan(uint32_t prec, int32_t plan_out[5]);
/* the same for the NerfMLP of one level (0 coarse, 1 fine): differs from nerfds_precision_plan only under NERFDS_PREC_BF16X3_FINE */
int nerfds_precision_plan_level(uint32_t prec, int32_t level, int32_t plan_out[5]);
int nerfds_ctx_create(nerfds_ctx** out, int device, const nerfds_model_cfg* cfg);
int nerfds_ctx_load_weights(nerfds_ctx* ctx, const nerfds_weights* w);
int nerfds_render_rays(nerfds_ctx* ctx, const nerfds_rays* rays, const nerfds_extra* extra,
                       const nerfds_rand* rnd, const nerfds_out* out, uint32_t flags, void* hip_stream);
/* NerfModel._encode_embed (models.py:271-294) on the device, the body of evaluation.encode_metadata (evaluation.py:29-50):
 * metadata [R][channels] (float: ids are small integers) -> out [R][glo_num_dims].  channels 1: the GLO row of the id; channels 3:
 * (1 - progression) * row(left) + progression * row(right).  table: 0 = warp_embed (encode_warp_embed / encode_hyper_embed of the
 * built graphs), 1 = mask_embed.  Ids are cast like astype(uint32) and clamped like a jnp gather.  DEVICE pointers. */
int nerfds_encode_embed(nerfds_ctx* ctx, int32_t table, const float* metadata, int32_t channels, int64_t num_rays, float* out,
                        void* hip_stream);
int nerfds_ctx_destroy(nerfds_ctx* ctx);
/* Message of the last failure on this ctx (or of the last failed nerfds_ctx_create when ctx == NULL). */
const char* nerfds_last_error(const nerfds_ctx* ctx);

/* Camera -> rays on the GPU: datasets.camera_to_rays (hypernerf/datasets/core.py:51-76) = Camera.pixels_to_rays
 * (hypernerf/camera.py:245-270, Newton undistortion camera.py:75-106) on Camera.get_pixel_centers (camera.py:364-368).
 * Pixels are either given ([n][2] DEVICE floats) or, when pixels == NULL, the row-major pixel centres
 * first_pixel .. first_pixel + n - 1 of the image.  Outputs are DEVICE pointers, any may be NULL:
 * origins [n][3] (the camera position), directions [n][3] (unit, world frame), pixels_out [n][2]. */
int nerfds_camera_to_rays(int device, const nerfds_camera* cam, int64_t first_pixel, int64_t n, const float* pixels,
                          float* origins, float* directions, float* pixels_out, void* hip_stream);

/* Frame output path that follows the gather (render.py:231-268): one frame of ray records ([height * width][NERFDS_RAY_REC]
 * DEVICE floats, row major) -> rgb_u8 [height][width][3] (image_utils.image_to_uint8 of 'rgb', render.py:269) and
 * debug_u8 [2 * height][3 * width][3], the mosaic  rgb | colorize(med_depth, near, far, invert) | normal  over
 * predicted mask | |delta_x| * 10 | (med_points + 1.5) / 3  (render.py:263-268, visualization.py:199-235).
 * colormap: [256][3] DEVICE doubles (matplotlib colormaps are float64; the reference uses 'magma').  Either output may be
 * NULL.  Byte-exact with the numpy arithmetic of those lines. */
int nerfds_frame_images(int device, const float* ray_records, int32_t height, int32_t width, double near_, double far_,
                        const double* colormap, uint8_t* rgb_u8, uint8_t* debug_u8, void* hip_stream);

/* ---- Training step (BASELINE config 4; replaces training.train_step, hypernerf/training.py:198-511, for the first-order
 * objective  loss = mean((rgb_fine - gt)^2) + mean((rgb_coarse - gt)^2)  (training.py:265-274, 459-466, 481) --------------
 * A trainer owns the parameters as ONE flat fp32 device vector (leaves in a fixed order, named by their Flax paths: see
 * nerfds_trainer_leaf), their gradient, the Adam moments (flax.optim.Adam: b1 0.9, b2 0.999, eps 1e-8) and an HBM workspace
 * sized for max_rays.  nerfds_trainer_step: forward + backward of both levels into the gradient vector, then (unless
 * NERFDS_TRAIN_GRADS_ONLY) one Adam update with `learning_rate`.  rays / target_rgb ([R][3]) / rnd->t_rand,u_rand are DEVICE
 * pointers; NULL uniforms with extra->use_stratified_sampling = the on-chip Philox stream described at struct nerfds_rand - the reference always
 * draws the jitter, model_utils.py:84,217 - so pass a new seed every step.  loss_host (optional, HOST float[16]) receives
 * {[0] rgb loss fine (coarse if there is no fine level), [1] rgb loss coarse, [2..5] weighted warp_reg / back_facing / mask / norm terms of the fine
 * level, [6..9] the same four of the coarse level, [10] 0, [11] hyper-point regulariser (coarse level), [12] background regulariser, [13] elastic
 * regulariser (coarse level), [14] / [15] mask occlusion regulariser of the fine / coarse level} and synchronises the stream.
 * Only the configs/nerf_ds.gin graph is built (NERFDS_ENOTSUP otherwise).  With the widths of that gin file the forward of a level is
 * ONE launch of the fused field kernel (the render kernel's evaluation, writing every activation the backward reads; its weight streams
 * are re-packed on the device from the parameter vector at the start of every step); other widths, and the whole backward, run layer by
 * layer on the library's own MFMA kernels: a layer shape they do not cover is NERFDS_ENOTSUP (there is no library-GEMM detour).  The auxiliary losses of
 * configs/nerf_ds.gin, the hyper-point, background and elastic ('log_svals') regularisers are selected by nerfds_train_objective. */
typedef struct nerfds_trainer nerfds_trainer;
/* Weights of the auxiliary first-order losses added to the rgb loss of EACH level (0 = off): warp regulariser at the median-depth
 * sample (training.py:297-310, utils.general_loss_with_squared_residual), back-facing regulariser on the raw predicted normal
 * (training.py:334-339), 3-D predicted-mask supervision on sharpened or plain compositing weights (training.py:386-408).
 * norm_loss_weight: the norm loss mean(w |n - target_norm|) (training.py:323-332); like the reference it does NOT stop the gradient
 * at target_norm, i.e. it is second order (backward of the tangent pass); costs ~4x the step and ~3x the workspace. */
typedef struct nerfds_train_objective {
  float warp_reg_loss_weight, warp_reg_loss_alpha, warp_reg_loss_scale;
  float back_facing_reg_weight;
  float predicted_mask_loss_weight, sharp_weights_std;
  int32_t use_mask_sharp_weights;
  float norm_loss_weight;
  /* hyper-point regulariser (training.py:312-321): mean over rays of sum_s w_s * general_loss(|ambient coordinates|^2, alpha 0, scale 0.05),
   * the weights as constants - on the COARSE level only, as the reference (training.py:461-466); reported in loss_host[11] ([10], the fine
   * level's slot, stays 0) */
  float hyper_reg_loss_weight;
  /* background regulariser (training.py:159-183, 468-479): background_loss_weight * mean over the points of
   * general_loss(|warp(x) - x|^2, background_loss_alpha, background_loss_scale), warp = NerfModel.apply_warp (the SE(3) field with the GLO row
   * of background_ids and mask 0, models.py:766-773).  background_points: DEVICE [num_background_points][3] - the caller has added its noise
   * (training.py:163-165) -, background_ids: DEVICE uint32 [num_background_points] (the reference draws them at random among the warp ids);
   * at most max_rays * (Nc + Nf) points.  Reported in loss_host[12]. */
  float background_loss_weight, background_loss_alpha, background_loss_scale;
  const float* background_points;
  const uint32_t* background_ids;
  int64_t num_background_points;
  /* elastic regulariser (training.py:112-156, 274-295; 'log_svals'): elastic_loss_weight * mean over rays of general_loss(sum log^2 of the
   * singular values of the warp field's Jacobian d x' / d x, alpha -2, scale 0.03) at the median-depth sample of the COARSE level
   * (elastic_reduce_by_weight = 0, 'median') or summed over its samples with the compositing weights as constant factors (1, 'weight').
   * Second order in the warp field's weights: it runs the tangent pass like norm_loss_weight (and costs as much).  Reported in loss_host[13]. */
  float elastic_loss_weight;
  int32_t elastic_reduce_by_weight;
  /* mask occlusion regulariser (training.py:409-417, with the 3-D mask supervision): mean over rays of sum_s max(0.01 - w_s, 0) |predicted mask_s|,
   * the weights as constants, both levels; reported in loss_host[14] (fine) / [15] (coarse) */
  float mask_occlusion_reg_loss_weight;
} nerfds_train_objective;
#define NERFDS_TRAIN_GRADS_ONLY 1u
#define NERFDS_TRAIN_SIGMA_GRAD 2u   /* also evaluate the sigma gradient (models.py:1035-1077) -> nerfds_trainer_target_norm */
/* ---- A caller-defined loss on the rendered rays (training.py:441-494: jax.value_and_grad of an arbitrary _loss_fn through model.apply) -------
 * nerfds_trainer_forward: model.apply on the trainer's current parameters - rgb / depth / acc of each level (model_utils.py:138-148) into DEVICE
 * arrays, any of them NULL = not wanted; a single-level model writes its one level through `coarse`.
 * nerfds_render_rays_bwd: the vector-Jacobian product of that map - d_fine / d_coarse hold d loss / d rgb [R][3], d loss / d depth [R], d loss / d acc
 * [R] of the level (DEVICE, NULL = zero; a NULL struct = all zero) - into the trainer's gradient vector (nerfds_trainer_grads, leaves by
 * nerfds_trainer_leaf), which is overwritten; no optimizer update: follow with nerfds_trainer_clip_gradients / an all-reduce / nerfds_trainer_apply, or
 * read the vector and step any optimizer on nerfds_trainer_params.  The backward re-runs the forward (the workspace holds ONE level's activations, so
 * nothing survives between the two calls: same rays, same extra, same rnd -> same samples; the Philox stream is keyed by (seed, ray)) - a step
 * costs the forward twice; nerfds_trainer_step remains the fast path for the reference's own losses.  The fine depths are drawn from the coarse
 * weights behind a stop_gradient (model_utils.py:264), as in the reference.  med_depth and the other record fields are not differentiable outputs
 * here.  The stored f16 g of the backward is loss-scaled for gradients of the size of a mean squared error's (2 / (3 R) per unit of colour): a loss
 * reduced differently should set nerfds_trainer_set_loss_scale_adjust (nerfds_amd.autograd does it from max |cotangent|). */
typedef struct nerfds_level_out { float* rgb; float* depth; float* acc; } nerfds_level_out;
typedef struct nerfds_level_cotangent { const float* d_rgb; const float* d_depth; const float* d_acc; } nerfds_level_cotangent;
int nerfds_trainer_forward(nerfds_trainer* t, const nerfds_rays* rays, const nerfds_extra* extra, const nerfds_rand* rnd, const nerfds_level_out* fine,
                           const nerfds_level_out* coarse, void* hip_stream);
int nerfds_render_rays_bwd(nerfds_trainer* t, const nerfds_rays* rays, const nerfds_extra* extra, const nerfds_rand* rnd,
                           const nerfds_level_cotangent* d_fine, const nerfds_level_cotangent* d_coarse, void* hip_stream);
int nerfds_trainer_create(nerfds_trainer** out, int device, const nerfds_model_cfg* cfg, int64_t max_rays);
int nerfds_trainer_destroy(nerfds_trainer* t);
int64_t nerfds_trainer_param_count(const nerfds_trainer* t);
int nerfds_trainer_num_leaves(const nerfds_trainer* t);
int nerfds_trainer_leaf(const nerfds_trainer* t, int index, char* name, int name_cap, int64_t* offset, int32_t* rows, int32_t* cols);
float* nerfds_trainer_params(nerfds_trainer* t);   /* DEVICE [param_count], read/write between steps */
float* nerfds_trainer_grads(nerfds_trainer* t);    /* DEVICE [param_count], valid after a step */
/* HOST <-> DEVICE copies of a whole vector; which: 0 parameters, 1 gradients, 2 / 3 Adam first / second moments */
int nerfds_trainer_download(nerfds_trainer* t, int which, float* host);
int nerfds_trainer_upload(nerfds_trainer* t, int which, const float* host);
/* target_norm = normalize(R normalize(-d sigma_raw / d x)) per sample (models.py:1069-1077, 1273-1277, 1328; 'warped' supervision) of the
 * last step run with NERFDS_TRAIN_SIGMA_GRAD: HOST [num_rays][S][3], S = Nc (level 0) or Nc + Nf (level 1). */
int nerfds_trainer_target_norm(nerfds_trainer* t, int level, int64_t num_rays, float* host);
int nerfds_trainer_reset_optimizer(nerfds_trainer* t);   /* zero the Adam moments and the step count */
int nerfds_trainer_set_step(nerfds_trainer* t, int64_t step);   /* the optimizer's step count (resuming from a checkpoint: OptimizerState.step) */
/* The optimizer's step count = the number of updates actually APPLIED (it lives on the device and does not advance on an update skipped for a
 * non-finite gradient: flax's OptimizerState.step, what a checkpoint stores); synchronises the device. */
int nerfds_trainer_get_step(nerfds_trainer* t, int64_t* step_out);
/* Dynamic loss scaling of the plain step's stored f16 g: the scale is 2^(6 + ceil(log2 num_rays) + log2_adjust).  A step whose gradient came out
 * non-finite because g left f16's range is re-run by the caller at a lower adjust (nerfds_amd/training.py: -2 per retry, +1 back after 1000 clean
 * steps) - with a fixed scale the deterministic retry would overflow the same way.  [-40, 16]; no reference counterpart (the reference's g is fp32). */
int nerfds_trainer_set_loss_scale_adjust(nerfds_trainer* t, int32_t log2_adjust);
/* ---- Numeric policy of the trainer's f16 storage, and the diagnosis of a skipped update (ABI 7) ----------------------------------------------
 * No reference counterpart: the reference's step is fp32 throughout and cannot overflow (training.py:494-508).  This trainer stores activations,
 * the chains' g, the second-order terms' tangents and their cotangents as (power-of-two scaled) f16; an element beyond 65504 becomes inf there, the
 * gradient check of the update finds it, and the update is skipped as a whole (NERFDS_ENONFINITE).  The caller then asks WHICH array overflowed
 * (nerfds_trainer_overflow_sources) and re-runs the step - same rays, same seed - under a policy that removes that source:
 *   NERFDS_OVF_PRIMAL_G      lower loss_scale_log2_adjust            (the stored g of the primal chains left f16's range)
 *   NERFDS_OVF_TANGENT /     lower tangent_scale_log2_adjust and / or chain_arith = NERFDS_CHAINS_SPLIT_BF16
 *   NERFDS_OVF_COTANGENT     (the second-order terms' stored tangents / cotangents, or a value BETWEEN the layers of a chain run in one f16 MFMA per product)
 *   NERFDS_OVF_ACTIVATION    fp32_step = 1: fp32 activations and fp32 g, layer by layer - no f16 storage anywhere, the arithmetic range of the reference
 *   NERFDS_OVF_FP32          nothing: an fp32 value of the FORWARD pass is inf / NaN (a loss, a head output, the screw axis, a warped point) - the
 *                            reference's step would carry the same inf / NaN into its parameters; fp32_step = 1 confirms it without f16 in the way
 *   NERFDS_OVF_FP32_BACKWARD an fp32 cotangent of the BACKWARD pass (a head's, the warped points', the screw axes', the ambient coordinates') and
 *   NERFDS_OVF_FP32_SECOND_ORDER   NERFDS_OVF_FP32_SECOND_ORDER (the norm loss / elastic terms' fp32 tangents and cotangents): a CONSEQUENCE when a
 *                            TANGENT / COTANGENT array overflowed (the second-order chains' outputs feed the primal backward), genuine - as FP32 - otherwise
 * Causality runs ACTIVATION -> TANGENT -> COTANGENT -> PRIMAL_G: the earliest source set is the one to remove.
 * nerfds_amd/training.py Trainer.step implements that ladder; it ends in fp32_step, so a step the reference could take is never refused. */
#define NERFDS_CHAINS_DEFAULT 0     /* the second-order terms' chains in one f16 MFMA per product where the build does so */
#define NERFDS_CHAINS_SPLIT_BF16 1  /* every chain in split bf16: operands with fp32's exponent range */
typedef struct nerfds_train_numerics {
  int32_t loss_scale_log2_adjust;     /* [-40, 16]: as nerfds_trainer_set_loss_scale_adjust */
  int32_t tangent_scale_log2_adjust;  /* [-24, 12]: stored tangents carry 2^(-6 + adjust), the device-picked cotangent scale aims its largest element at 2^(5 + adjust) */
  int32_t chain_arith;                /* NERFDS_CHAINS_* */
  int32_t fp32_step;                  /* 0 / 1 */
  int32_t diagnose;                   /* 0 / 1: the step also scans the COARSE NerfMLP's arrays before the fine level overwrites them (a repeated attempt, to attribute its overflow) */
} nerfds_train_numerics;
int nerfds_trainer_set_numerics(nerfds_trainer* t, const nerfds_train_numerics* numerics);   /* takes effect at the next step; all zero = the default */
int nerfds_trainer_get_numerics(const nerfds_trainer* t, nerfds_train_numerics* numerics);
#define NERFDS_OVF_ACTIVATION 1u
#define NERFDS_OVF_PRIMAL_G 2u
#define NERFDS_OVF_TANGENT 4u
#define NERFDS_OVF_COTANGENT 8u
#define NERFDS_OVF_FP32 16u
#define NERFDS_OVF_FP32_SECOND_ORDER 32u
#define NERFDS_OVF_FP32_BACKWARD 64u
#define NERFDS_OVF_DETAIL_SHIFT 8     /* bits 8 .. 29: WHICH fp32 array (csrc/nerfds_train.cpp nerfds_trainer_overflow_sources lists them); diagnosis only, not part of the contract */
/* Scans the stored arrays of the LAST step for inf / NaN (synchronises the device; a pass over the workspace: call it after a skipped update only).
 * Arrays a later level of the same step overwrote (the coarse NerfMLP's, under the fine level's) are seen only if the step ran with numerics.diagnose = 1;
 * mask 0 after a skipped update = unattributed: repeat the attempt with diagnose = 1 and ask again. */
int nerfds_trainer_overflow_sources(nerfds_trainer* t, uint32_t* mask_out);
/* Development / tests: HOST copy of an internal device buffer of the last step (the f16 activations, ReLU bits and per-layer
 * gradients g_l of the fused backward, the head / input gradients): "<net>_h16_<l>", "<net>_bits_<l>", "<net>_g_<l>" with net = mask |
 * warp | hyper | trunk, "rgb_h16", "rgb_bits", "rgb_g", "d_rgb_logit", "d_alpha", "d_trunk_in", "d_hyper_in", "d_warp_in",
 * "d_mask_in", "dwamb", "dwv", "d_mask_logit".  The "<net>_g_<l>" / "rgb_g" arrays are f16 [M][width] (2 bytes per element) holding
 * g_scale * g - "g_scale" reads that power of two as one float - unless the trainer was created under NERFDS_TRAIN_G16=0 (fp32, unscaled).  Returns the bytes copied (= max_bytes) or a negative error code. */
long long nerfds_trainer_debug_read(nerfds_trainer* t, const char* name, void* host, long long max_bytes);
int nerfds_trainer_step(nerfds_trainer* t, const nerfds_rays* rays, const float* target_rgb, const nerfds_extra* extra,
                        const nerfds_rand* rnd, const nerfds_train_objective* objective /* NULL = rgb loss only */, float learning_rate,
                        uint32_t flags, float* loss_host /* HOST float[16] or NULL: all 16 used */, void* hip_stream);
/* One Adam update with the gradient vector as it stands (after a NERFDS_TRAIN_GRADS_ONLY step and, on N GPUs, after the
 * all-reduce of nerfds_trainer_grads that replaces jax.lax.pmean(grad), training.py:502). */
int nerfds_trainer_apply(nerfds_trainer* t, float learning_rate, void* hip_stream);
/* The plain step keeps activations as f16 and the weight-gradient operand g as loss-scaled f16 (a documented deviation from the reference's
 * fp32 arrays; g times 64 x the ray count rounded up to a power of two, NERFDS_TRAIN_G_SCALE_LOG2 overrides the exponent; NERFDS_TRAIN_G16=0 in the
 * environment keeps g in fp32): a value beyond 65504 turns into an inf / NaN gradient.  Every Adam update
 * first checks the whole gradient vector on the device and is SKIPPED when an element is not finite; nerfds_trainer_step reports
 * NERFDS_ENONFINITE when it reads the loss back (loss_host != NULL), and this call (it synchronises the device) returns 1 if the last update
 * was skipped, 0 if not. */
int nerfds_trainer_nonfinite(nerfds_trainer* t);
/* utils.clip_gradients (utils.py:32-47, training.py:503-504) on the gradient vector: clip by value (if > 0), then scale so that the global
 * L2 norm is at most grad_max_norm (if > 0).  Call between a NERFDS_TRAIN_GRADS_ONLY step (and the all-reduce) and nerfds_trainer_apply. */
int nerfds_trainer_clip_gradients(nerfds_trainer* t, float grad_max_val, float grad_max_norm, void* hip_stream);
const char* nerfds_trainer_last_error(const nerfds_trainer* t);

/* Timing aid for bench.py: average device time (ms) of the render kernel launches recorded with HIP events
 * on the launch stream since the last reset; returns the number of launches measured. */
int nerfds_kernel_time_ms(nerfds_ctx* ctx, int reset, double* total_ms);

/* ---- host-only helpers (no device needed): exposed so that the weight-stream packing can be tested on CPU ---- */
/* Diagnostic: the library's "dynamic LDS above 64 KiB" launch attribute is set once per (kernel, device) pair (csrc/lds_attr.h); this is
 * the table's own test-and-set on an arbitrary key - 1 the first time a pair is seen, 0 afterwards.  No device call, no reference
 * counterpart (the reference's launches are XLA's). */
int nerfds_debug_lds_attr_first_use(uint64_t kernel_key, int device);
/* Size in bytes of the packed MFMA weight stream / padded bias array for `which` (0 = shared mask+warp+hyper
 * nets, 1 = NerfMLP) at precision `prec`; negative on error. */
int64_t nerfds_pack_stream_bytes(const nerfds_model_cfg* cfg, int which, uint32_t prec);
/* the exact size of the NerfMLP stream of one level (which == 1; level 0 coarse, 1 fine): smaller than nerfds_pack_stream_bytes for the coarse
 * level under NERFDS_PREC_BF16X3_FINE (one unit per fragment); nerfds_pack_stream_bytes returns the larger of the two */
int64_t nerfds_pack_stream_bytes_level(const nerfds_model_cfg* cfg, int which, int level, uint32_t prec);
int64_t nerfds_pack_bias_floats(const nerfds_model_cfg* cfg, int which);
/* Output tiles per group in the stream of that (graph, precision) kernel: within a group the stream holds, chunk by chunk, one fragment
 * of each tile.  2, except 1 for the kernels that carry two N-tiles per wave (nerf_ds / HyperNeRF graph in bf16 / f16); negative on error. */
int nerfds_pack_tile_pair(const nerfds_model_cfg* cfg, uint32_t prec);
/* Packs into caller-provided host buffers.  level: 0 coarse, 1 fine (ignored for which == 0). */
int nerfds_pack_stream(const nerfds_model_cfg* cfg, const nerfds_weights* w, int which, int level, uint32_t prec,
                       void* stream_out, float* bias_out);

/* ---- device self-test: runs one v_mfma_f32_32x32x16_bf16 and one v_mfma_f32_32x32x2_f32 with A[m][k], B[k][n]
 * and returns C[32][32] as laid out by the accumulator map the kernels assume.  Host pointers. */
int nerfds_debug_mfma(int device, const float* a_32x16, const float* b_16x32, float* c_bf16_32x32, float* c_f32_32x32);

#ifdef __cplusplus
}
#endif
#endif  /* NERFDS_H_ */
