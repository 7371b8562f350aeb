// Frame output path: gathered ray records -> the uint8 frames render.py writes.
// One thread per pixel, HBM-bound: 104 B in (one 26-float record), 3 + 18 B out.  Every arithmetic step keeps the
// dtype and rounding of the numpy code it replaces (render.py:231-268, visualization.py:186-235,
// image_utils.py:124-131) because the outputs are bytes: float32 tiles, float64 colour-map lerp, float64 "* 255" for the
// debug mosaic, float32 "* 255" for the rgb frame.  No fused multiply-adds (numpy has none).
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace nerfds {

enum { FR_REC = 26, FR_RGB = 0, FR_MED_DEPTH = 4, FR_NORM = 6, FR_DELTA_X = 15, FR_MASK = 20, FR_MED_POINTS = 21 };

// Individually rounded operations.  (HIP's __fmul_rn-style intrinsics are inline header functions compiled under the
// default contract(fast), so hipcc still fuses a multiply with a following add/sub - seen as 1-off bytes in the depth
// tile; these are defined under contract(off) and the file is also built with -ffp-contract=off.)
__device__ __forceinline__ float fmul_(float a, float b) { return a * b; }
__device__ __forceinline__ float fadd_(float a, float b) { return a + b; }
__device__ __forceinline__ float fsub_(float a, float b) { return a - b; }
__device__ __forceinline__ double dmul_(double a, double b) { return a * b; }
__device__ __forceinline__ double dadd_(double a, double b) { return a + b; }
__device__ __forceinline__ double dsub_(double a, double b) { return a - b; }
// Correctly rounded float32 divide / sqrt, independent of compiler flags: the double result rounded once more to float
// is the correctly rounded float result (53 >= 2 * 24 + 2 bits).
__device__ __forceinline__ float div_rn(float a, float b) { return (float)((double)a / (double)b); }
__device__ __forceinline__ float sqrt_rn(float a) { return (float)sqrt((double)a); }

__device__ __forceinline__ uint8_t to_u8_f32(float v) {       // image_to_uint8 on a float32 image
  float t = fmul_(v, 255.0f);
  t = fminf(fmaxf(t, 0.0f), 255.0f);
  return (uint8_t)t;
}
__device__ __forceinline__ uint8_t to_u8_f64(double v) {      // image_to_uint8 on the float64 debug mosaic
  double t = dmul_(v, 255.0);
  t = fmin(fmax(t, 0.0), 255.0);
  return (uint8_t)t;
}

__global__ __launch_bounds__(256) void frame_images_kernel(const float* __restrict__ rec, int height, int width, float vmin, float vrange,
                                                          const double* __restrict__ lut, uint8_t* __restrict__ rgb_out,
                                                          uint8_t* __restrict__ dbg_out) {
  const long long n = (long long)height * width;
  for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    const float* r = rec + p * FR_REC;
    const int y = (int)(p / width), x = (int)(p % width);
    if (rgb_out) {
#pragma unroll
      for (int c = 0; c < 3; ++c) rgb_out[p * 3 + c] = to_u8_f32(r[FR_RGB + c]);
    }
    if (!dbg_out) continue;
    float tile[6][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) tile[0][c] = r[FR_RGB + c];
    // depth_viz = colorize(med_depth, cmin=near, cmax=far, invert=True)
    double dv[3];
    {
      const float xs = div_rn(fsub_(r[FR_MED_DEPTH], vmin), vrange);        // scale_values
      const float v = fsub_(1.0f, xs);
      const float t = fmul_(v, 255.0f);
      const float a = floorf(t);
      const float b = fminf(fadd_(a, 1.0f), 255.0f);
      const float f = fsub_(t, a);
      const int ai = (int)fminf(fmaxf(a, 0.0f), 255.0f), bi = (int)fminf(fmaxf(b, 0.0f), 255.0f);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double ca = lut[3 * ai + c], cb = lut[3 * bi + c];
        dv[c] = dadd_(ca, dmul_(dsub_(cb, ca), (double)f));
        if (xs > 1.0f) dv[c] = 0.0;
        if (xs < 0.0f) dv[c] = 1.0;
      }
    }
    {  // ray_norm = normalize_vector(ray_norm) / 2 + 0.5
      const float v0 = r[FR_NORM], v1 = r[FR_NORM + 1], v2 = r[FR_NORM + 2];
      const float n2 = fadd_(fadd_(fmul_(v0, v0), fmul_(v1, v1)), fmul_(v2, v2));
      const float d = sqrt_rn(fmaxf(n2, 1.1920929e-07f));
      tile[2][0] = fadd_(div_rn(div_rn(v0, d), 2.0f), 0.5f);
      tile[2][1] = fadd_(div_rn(div_rn(v1, d), 2.0f), 0.5f);
      tile[2][2] = fadd_(div_rn(div_rn(v2, d), 2.0f), 0.5f);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tile[3][c] = r[FR_MASK];                                                        // grayscale -> colour
      tile[4][c] = fmul_(fabsf(r[FR_DELTA_X + c]), 10.0f);
      tile[5][c] = div_rn(fadd_(r[FR_MED_POINTS + c], 1.5f), 3.0f);           // -1.5 ~ 1.5 --> 0 ~ 1
    }
    // mosaic [2H][3W][3]: row 1 = rgb | depth | normal, row 2 = mask | delta_x | med_points
    const long long W3 = 3LL * width;
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const long long o = ((long long)(y + (t / 3) * height) * W3 + (long long)(t % 3) * width + x) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) dbg_out[o + c] = (t == 1) ? to_u8_f64(dv[c]) : to_u8_f64((double)tile[t][c]);
    }
  }
}

}  // namespace nerfds

extern "C" void nerfds_launch_frame_images(const float* rec, int height, int width, float vmin, float vrange, const double* lut,
                                           uint8_t* rgb_out, uint8_t* dbg_out, void* stream) {
  const long long n = (long long)height * width;
  if (n <= 0) return;
  const int block = 256;
  const long long want = (n + block - 1) / block;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(nerfds::frame_images_kernel, dim3(grid), dim3(block), 0, static_cast<hipStream_t>(stream), rec, height, width, vmin,
                     vrange, lut, rgb_out, dbg_out);
}
