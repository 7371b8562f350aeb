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
  const float* enc_warp;     // [R][8] per-ray GLO vectors instead of the warp_embed row of warp_id (metadata_encoded), or null
  const float* enc_mask;     // [R][8] the same for the mask network's embedding, or null
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
  int lindisp;               // use_linear_disparity (model_utils.py:73-76)
  int sample_at_infinity;        // the 'coarse' level: the model's value (models.py:1509)
  int sample_at_infinity_fine;   // the 'fine' level: the per-call override, else the model's value (models.py:1484-1485, 1544)
  int white_bkgd;
  float near_, far_;
  float mask_ratio;
  // render_opts (filter_sigma, models.py:38-66; the FINE level only, models.py:1545): opt_flags bit 0 dust threshold, bit 1 bounding box (xmin, xmax, ymin, ymax, zmin, zmax)
  int opt_flags;
  float dust_threshold;
  float bbox[6];
  // posenc windows per band (model_utils.py:420-436), evaluated on the host from the extra_params alphas
  float win_mask[MAX_BANDS];   // alpha = warp_alpha        (models.py:967)
  float win_warp[MAX_BANDS];   // alpha = warp_alpha        (warping.py:213)
  float win_hyp[MAX_BANDS];    // alpha = hyper_sheet_alpha (models.py:666)
  float win_sp[MAX_BANDS];     // alpha = nerf_alpha        (models.py:507)
  float win_hp[MAX_BANDS];     // alpha = hyper_alpha       (models.py:515)
  float win_nm[MAX_BANDS];     // alpha = norm_input_alpha  (models.py:1147)
#ifdef NERFDS_PROF
  // MEASUREMENT BUILD ONLY (tools/prof_phases.sh): [0] kernel cycles summed over waves, [1] cycles inside dense() / head() (the MFMA chains),
  // [2] cycles inside eval_shared / eval_nerf (chains + input encodings + per-sample math), [3] waves, [4] cycles in composite() / resample()
  unsigned long long* prof;
#endif
};

// Where the training forward writes what the backward pass reads (nerfds_train.cpp workspace, row-major [R * S][width] fp32).
// Passed by value to train_forward_kernel (train_fwd_kernel.hip).
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
  // train_fwd_kernel.hip MODE: 0 shared networks + NerfMLP, 1 shared networks only, 2 NerfMLP only with the shared networks' per-sample results
  // read from in_xw [M][3] (warped point), in_wamb [M][2] (ambient coordinates), in_wv [M][6] (screw axis head) in the level's row order
  int mode;
  const float* in_xw;
  const float* in_wamb;
  const float* in_wv;
  // half_out != 0 (the plain training step): every hidden layer is written as f16 (what the weight-gradient kernels read as X: 11
  // significand bits in 2 bytes) into *_h16 plus one bit per feature "output > 0" into *_bits (what the fused backward reads as the
  // ReLU mask), instead of the fp32 arrays above (which the tangent passes of the norm loss still need).
  // Bit layout: u16 at [(row * 2 + half) * (width / 32) + tile], bit r <-> accumulator register r of the lane (half = lane >> 5).
  int half_out;
  uint16_t* mask_h16[8];  uint16_t* mask_bits[8];
  uint16_t* warp_h16[6];  uint16_t* warp_bits[6];
  uint16_t* hyper_h16[6]; uint16_t* hyper_bits[6];
  uint16_t* trunk_h16[8]; uint16_t* trunk_bits[8];
  uint16_t* rgb_h16;      uint16_t* rgb_bits;
};

// Fused backward of one network (train_bwd_kernel.hip): the data-gradient chain of a reversed MLP.  Row
// m = one sample; the kernel reads the gradient of the network's head outputs, walks the layers backwards with the transposed
// weights streamed like the forward's (dX never leaves the registers between layers), masks with the forward's ReLU bits and
// writes, for every hidden layer, g = d loss / d (pre-activation) as fp32 [M][width] - the dY the weight-gradient kernels read -
// and the gradient of the network's raw input.
struct TrainBwd {
  long long M;
  const void* wstream;          // transposed fragments in walk order (split bf16), zero padded to whole stages
  const float* d_head;          // [M][ld_head] gradient of the head outputs (nerf: d rgb logit, 3 wide)
  int ld_head;
  const float* d_head2;         // nerf only: [M][4] gradient of the alpha head outputs (sigma_raw | raw normal)
  const uint16_t* bits[9];      // ReLU bits of hidden layer l (nerf: 0..7 trunk, 8 rgb hidden)
  float* g[9];                  // out: [M][width] per hidden layer (nerf: 0..7 trunk, 8 rgb hidden)
  float* d_in;                  // out: [M][ld_in] gradient of the raw input (zero in the pad columns)
  int ld_in;
  float* sink;                  // >= 64 bytes of scratch: where lanes outside ld_in write
  // g_half != 0: g[l] is written as f16 [M][width] (the same pointers, read as uint16_t*) - the weight-gradient kernels multiply it with the
  // forward's f16 activations as stored (one MFMA per product); the chain itself keeps every g in fp32 / split bf16 registers, so only the
  // weight and bias gradients see the 11-bit rounding (unbiased, averaged over M samples)
  int g_half;
  // The whole chain runs on g_scale * g (a power of two: exact in fp32 and bf16, applied to the head gradients on their way in) so that the
  // stored g sits in f16's range (loss scaling); the raw-input gradient is multiplied by g_inv_scale on its way out and the weight-gradient
  // kernels by the same factor (WgradArgs::out_scale).  1 / 1 for fp32 g.
  float g_scale, g_inv_scale;
  // Tangent rows (the second-order terms: three tangents per sample, rows 3 m + k): mask_div = 3 reads the ReLU bits of row r / 3 - the PRIMAL
  // layer's mask of the sample (0 / 1: bits of row r).
  int mask_div;
  // scale_dev != nullptr: {g_scale, g_inv_scale} are read from DEVICE memory instead (a power of two picked on the device from the largest
  // cotangent of this launch, train_kernels.hip k_pick_scale): the cotangents of the tangent pass have no a-priori size
  const float* scale_dev;
};

typedef void (*launch_fn)(const KArgs& ka, int num_cus, void* stream);

}  // namespace nerfds
