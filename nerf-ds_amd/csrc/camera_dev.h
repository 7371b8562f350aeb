// Camera -> ray, device side.  Restates (fp32, same operation order) the reference's
//   Camera.pixel_to_local_rays / pixels_to_rays   hypernerf/camera.py:226-270
//   _radial_and_tangential_undistort (Newton x10) hypernerf/camera.py:75-106, _compute_residual_and_jacobian :24-72
//   get_pixel_centers                             hypernerf/camera.py:364-368
#pragma once
#include <hip/hip_runtime.h>

namespace nerfds {

struct CameraParams {          // == nerfds_camera of include/nerfds.h (checked by static_assert in nerfds_host.cpp)
  float orientation[9];        // world -> camera rotation, row major
  float position[3];
  float focal_length;
  float principal_point[2];
  float skew;
  float pixel_aspect_ratio;
  float radial_distortion[3];
  float tangential_distortion[2];
  int image_width, image_height;
};

__device__ __forceinline__ void camera_pixel_to_ray(const CameraParams& c, float px, float py, float (&dir)[3]) {
  float y = (py - c.principal_point[1]) / (c.focal_length * c.pixel_aspect_ratio);
  float x = (px - c.principal_point[0] - y * c.skew) / c.focal_length;
  const float k1 = c.radial_distortion[0], k2 = c.radial_distortion[1], k3 = c.radial_distortion[2];
  const float p1 = c.tangential_distortion[0], p2 = c.tangential_distortion[1];
  if (k1 != 0.f || k2 != 0.f || k3 != 0.f || p1 != 0.f || p2 != 0.f) {
    const float xd = x, yd = y;
#pragma unroll 1
    for (int it = 0; it < 10; ++it) {
      const float r = x * x + y * y;
      const float d = 1.0f + r * (k1 + r * (k2 + k3 * r));
      const float fx = d * x + 2 * p1 * x * y + p2 * (r + 2 * x * x) - xd;
      const float fy = d * y + 2 * p2 * x * y + p1 * (r + 2 * y * y) - yd;
      const float d_r = k1 + r * (2.0f * k2 + 3.0f * k3 * r);
      const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
      const float fx_x = d + d_x * x + 2.0f * p1 * y + 6.0f * p2 * x;
      const float fx_y = d_y * x + 2.0f * p1 * x + 2.0f * p2 * y;
      const float fy_x = d_x * y + 2.0f * p2 * y + 2.0f * p1 * x;
      const float fy_y = d + d_y * y + 2.0f * p2 * x + 6.0f * p1 * y;
      const float den = fy_x * fx_y - fx_x * fy_y;
      const bool ok = fabsf(den) > 1e-9f;
      x += ok ? (fx * fy_y - fy * fx_y) / den : 0.f;
      y += ok ? (fy * fx_x - fx * fy_x) / den : 0.f;
    }
  }
  const float inv = 1.0f / sqrtf(x * x + y * y + 1.0f);        // local ray, normalised (camera.py:242-243)
  const float lx = x * inv, ly = y * inv, lz = inv;
  // orientation^T @ local (camera.py:263)
  float wx = c.orientation[0] * lx + c.orientation[3] * ly + c.orientation[6] * lz;
  float wy = c.orientation[1] * lx + c.orientation[4] * ly + c.orientation[7] * lz;
  float wz = c.orientation[2] * lx + c.orientation[5] * ly + c.orientation[8] * lz;
  const float inv2 = 1.0f / sqrtf(wx * wx + wy * wy + wz * wz);  // camera.py:267
  dir[0] = wx * inv2; dir[1] = wy * inv2; dir[2] = wz * inv2;
}

}  // namespace nerfds
