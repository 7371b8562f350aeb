// Standalone camera -> rays kernel (datasets/core.py:51-76 camera_to_rays): one thread per pixel, HBM-bound
// (0 or 8 B in, 48-56 B out per ray).  The same device function is used by the fused render path.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "camera_dev.h"

namespace nerfds {

__global__ __launch_bounds__(256) void camera_rays_kernel(const CameraParams cam, long long first_pixel, long long n,
                                                          const float* __restrict__ pixels_in, float* __restrict__ origins,
                                                          float* __restrict__ directions, float* __restrict__ pixels_out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float px, py;
    if (pixels_in != nullptr) {
      px = pixels_in[2 * i];
      py = pixels_in[2 * i + 1];
    } else {                                         // get_pixel_centers (camera.py:364-368), row major
      const long long p = first_pixel + i;
      px = (float)(p % cam.image_width) + 0.5f;
      py = (float)(p / cam.image_width) + 0.5f;
    }
    float d[3];
    camera_pixel_to_ray(cam, px, py, d);
    if (directions) { directions[3 * i] = d[0]; directions[3 * i + 1] = d[1]; directions[3 * i + 2] = d[2]; }
    if (origins) { origins[3 * i] = cam.position[0]; origins[3 * i + 1] = cam.position[1]; origins[3 * i + 2] = cam.position[2]; }
    if (pixels_out) { pixels_out[2 * i] = px; pixels_out[2 * i + 1] = py; }
  }
}

}  // namespace nerfds

extern "C" void nerfds_launch_camera_rays(const nerfds::CameraParams& cam, long long first_pixel, long long n, const float* pixels_in,
                                          float* origins, float* directions, float* pixels_out, void* stream) {
  if (n <= 0) return;
  const int block = 256;
  const long long want = (n + block - 1) / block;
  const int grid = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(nerfds::camera_rays_kernel, dim3(grid), dim3(block), 0, static_cast<hipStream_t>(stream), cam, first_pixel, n,
                     pixels_in, origins, directions, pixels_out);
}
