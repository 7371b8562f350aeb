// Kernel argument block shared by the host library and the device kernels (plain POD, passed by value).
#pragma once
#include <stdint.h>

namespace nerfds {

constexpr int MAX_SAMPLES = 256;      // Nc + Nf upper bound: 128 with 4 rays per workgroup (nerf_ds.gin 64 + 64), 256 with 2 rays (128 + 128)
constexpr int RAY_REC = 26;           // == NERFDS_RAY_REC
constexpr int SAMPLE_REC = 18;        // == NERFDS_SAMPLE_REC
constexpr int MAX_BANDS = 8;

struct KArgs {
  // rays
  const float* origins;
  const float* directions;
  const float* viewdirs;
  const uint32_t* warp_id;
  const float* gt_mask;
  // sampling uniforms (nullable -> Philox)
  const float* t_rand;
  const float* u_rand;
  uint64_t seed;
  int64_t first_ray;       // Philox counter of ray 0 (philox.h)
  // packed weights: [0] shared (mask|warp|hyper) stream, [1] coarse NerfMLP, [2] fine NerfMLP
  const void* wstream[3];
  const float* bias[3];
  const float* warp_embed;   // [N][8] fp32
  const float* mask_embed;   // [N][8] fp32
  // outputs (nullable)
  float* ray_fine;
  float* ray_coarse;
  float* smp_fine;
  float* smp_coarse;
  int num_rays;
  int num_embeds;            // rows of the GLO tables (ids are clamped like a jnp gather)
  int nc, nf;
  int stratified;
  int sample_at_infinity;
  int white_bkgd;
  float near_, far_;
  float mask_ratio;
  // posenc windows per band (model_utils.py:420-436), evaluated on the host from the extra_params alphas
  float win_mask[MAX_BANDS];   // alpha = warp_alpha        (models.py:967)
  float win_warp[MAX_BANDS];   // alpha = warp_alpha        (warping.py:213)
  float win_hyp[MAX_BANDS];    // alpha = hyper_sheet_alpha (models.py:666)
  float win_sp[MAX_BANDS];     // alpha = nerf_alpha        (models.py:507)
  float win_hp[MAX_BANDS];     // alpha = hyper_alpha       (models.py:515)
  float win_nm[MAX_BANDS];     // alpha = norm_input_alpha  (models.py:1147)
};

// Where the training forward writes what the backward pass reads (nerfds_train.cpp workspace, row-major [R * S][width] fp32).
// Passed by value to train_forward_kernel (render_kernel.hip).
struct TrainOut {
  static constexpr bool ON = true;
  float* mask_h[8];
  float* warp_h[6];
  float* hyper_h[6];
  float* trunk_h[8];
  float* rgb_h;
  float* mask_logit;   // [M]     raw logit (the ReLU is mask_post's)
  float* wv;           // [M][6]  screw axis w | v (warping.py:217-218)
  float* wamb;         // [M][2]
  float* alphav;       // [M][4]  sigma_raw | raw normal
  float* rgb_logit;    // [M][3]
  const float* z;      // [R][S]  sample depths of this level
  int level;
};

typedef void (*launch_fn)(const KArgs& ka, int num_cus, void* stream);

}  // namespace nerfds
