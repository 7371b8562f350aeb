// Hand-written weight-stationary dense layer of the training path (train_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace nerfds_train {

struct DenseSeg { const float* x; int ld; int k; };
struct DenseArgs {
  DenseSeg seg[4];          // row-major fp32 inputs, concatenated along k in this order
  int nseg;
  int k_total;              // sum of seg[].k
  const void* wfrag;        // pack_frags() output for (k_total, n_out)
  const float* bias;        // [n_out] or nullptr
  float* y;                 // [M x n_out], row stride ldy
  int ldy, n_out;
  long long M;
  int relu;                 // y = max(y, 0)
  const float* mask_y;      // nullptr, or: y = 0 where mask_y[(row / mask_div) * ld_mask + n] <= 0 (ReLU backward of the consumer);
                            // with accumulate the mask (and colsum) apply to the total, i.e. this must be the last contribution
  int ld_mask, mask_div;
  int accumulate;           // y += result
  int precise;              // operands split three ways (hi / mid / lo bf16, 6 MFMAs per fragment): fp32-level products; wfrag packed with parts = 3
  float* colsum;            // nullptr, or [n_out]: += column sums of the (masked) result - the bias gradient of the layer that consumes y as dY
  long long rep_stride; int nrep;   // replicas of the colsum destination, as WgradArgs (nrep <= 1: none)
  const void* zeros;        // >= 16 zero bytes in device memory (source of the padding slots of the LDS-DMA path), or nullptr
  int vec_in, vec_out;      // set by dense_ws(): 16-byte loads / stores are legal
};

// Weight gradient dW[K x N] += X[M x K]^T dY[M x N] (contraction over the M samples).
struct WgradArgs {
  const float* x; int ldx; int k;      // K <= 256
  const float* dy; int ldy; int n;     // N <= 256
  long long M;
  float* part;                         // [grid][K x N] per-workgroup partial sums (the caller reduces them), used when dw == nullptr
  float* dw;                           // [K x N]: every workgroup adds its partial with hardware float atomics
  const void* zeros;
  int x_scalar, dy_scalar;             // set by wgrad(): the part is fetched float by float (rows not 16-byte aligned)
  // Replicas of the destination: workgroup b adds into dw + (b % nrep) * rep_stride (floats).  All workgroups finish together and
  // their atomics on one small region queue up on a few L2 channels (40-60 us per kernel, whatever its size); spread over
  // nrep copies that tail goes away, and the caller sums the copies once per step.  nrep <= 1: no replicas.
  long long rep_stride; int nrep;
  // x_half: x points to f16 [M x ldx] (what the fused forward stores for the plain training step; K, ldx multiples of 8, 16-byte
  // aligned rows) - converted exactly into bf16 hi + lo on the way to the MFMAs.
  int x_half;
  // dy_half: dy points to f16 [M x ldy] (the g arrays of the fused backward, scaled - see out_scale; N a multiple of 32, ldy of 8, 16-byte
  // aligned rows): a one-term operand.  With an f16 X both operands go to the MFMAs as they are (k_wgrad_tr: one MFMA per product); an fp32 X
  // is split into f16 hi + lo (two MFMAs per product).
  int dy_half;
  // dy_half with an fp32 X: X is multiplied by x_scale (0 = 1) before it is split into f16 hi + lo - a power of two that keeps it inside f16's range (the raw
  // inputs of the PRIMAL layers are encodings, |x| <= 1; those of the tangent pass are derivatives of encodings, up to 2^7 |t_x'|: they carry the stored
  // tangents' scale, and out_scale_dev divides it out again)
  float x_scale;
  // colsum != nullptr: += the column sums of dY (the bias gradient of the layer), a by-product of the tile conversion; replicas as dw.
  float* colsum;
  // dy_half: what reaches dw / colsum is multiplied by out_scale (0 = 1): the chains write g times a power of two so that it fits f16.
  float out_scale;
  // != nullptr: one more factor, read from DEVICE memory (the scale of a tangent cotangent array is picked on the device, train_kernels.hip
  // pick_scale; slot[3] = 1 / (scale * x_scale)); applies to every kernel, fp32 dY included
  const float* out_scale_dev;
  // nl > 1 (x_half and dy_half only): ONE launch computes nl weight gradients of the same shape and row count - the hidden layers of one MLP,
  // whose chain leaves all their g at once: workgroup b works on layer b % nl, as workgroup b / nl of gridDim / nl.  Layer i: X = mx[i],
  // dY = mdy[i] (ldx / ldy as above), destination mdw[i] / mcs[i] (replicas as above); x, dy, dw, colsum must equal entry 0.
  int nl;
  const void* mx[8]; const void* mdy[8]; float* mdw[8]; float* mcs[8];
};
bool wgrad_supported(const WgradArgs& A);
bool wgrad_multi_supported(const WgradArgs& A);     // nl layers in one launch (same shape, 16-bit operands)
// Launches on `grid` workgroups chosen by wgrad_grid(); part must hold grid * K * N floats.  false = shape not covered.
int wgrad_grid(const WgradArgs& A, int num_cus);
bool wgrad(hipStream_t st, const WgradArgs& A, int grid);

// One backward pass of a narrow hidden layer (K, N in {64, 128}): dW[K x N] += X^T dZ  AND  dX = (dZ . W^T) . 1[X > 0] with the
// column sums of dX, from ONE read of X and dZ (X is both the layer's input and, being a ReLU output, the mask of its own gradient).
struct BwdFusedArgs {
  const float* x; int ldx; int k;       // X [M x K] = ReLU output of the previous layer
  const float* dz; int lddz; int n;     // dZ [M x N], already masked with this layer's ReLU
  const void* wfrag;                    // pack_frags(W, ..., in = N, out = K, transpose = 1)
  float* dx; int lddx;                  // [M x K]
  float* colsum;                        // [K] += column sums of dX (bias gradient of the previous layer), or nullptr
  float* dw;                            // [K x N] += (float atomics)
  long long M;
  const void* zeros;
  long long rep_stride; int nrep;       // as WgradArgs
};
bool bwd_fused_supported(const BwdFusedArgs& A);
bool bwd_fused(hipStream_t st, const BwdFusedArgs& A, int num_cus);

void pack_frags(hipStream_t st, const float* W, int ldw, int row0, int in_dim, int out_dim, int transpose, void* out, int parts = 2);
size_t frag_bytes(int in_dim, int out_dim, int parts = 2);
// All fragment packs of a step in one launch: entry e packs W = theta + e.woff into arena + e.dst (bytes).
struct PackEntry { long long woff; long long dst; int ldw, row0, in_dim, out_dim, transpose, parts; };
void pack_frags_all(hipStream_t st, const float* theta, const PackEntry* entries_dev, int n, int max_frag_lanes, void* arena);
bool dense_ws_supported(const DenseArgs& A);
// false = shape not covered (the step then returns NERFDS_ENOTSUP: there is no library fallback)
bool dense_ws(hipStream_t st, const DenseArgs& A, int num_cus);

}  // namespace nerfds_train
