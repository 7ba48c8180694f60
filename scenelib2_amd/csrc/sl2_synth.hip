// Synthetic-input renderer entry points (host + device run the same SL2_HD source,
// sl2_synth.hpp).  Input generation only — not part of the SLAM step.
#include "sl2_common.hpp"
#include "sl2_synth.hpp"

namespace sl2 {

__global__ void __launch_bounds__(256) k_synth_render(CameraParams cam, const uint8_t* __restrict__ tex, int tex_size,
                                                      double texels_per_metre, const double* __restrict__ tex_origin,
                                                      const double* __restrict__ poses, uint8_t* __restrict__ out) {
  const int f = blockIdx.y;
  const int npix = cam.width * cam.height;
  double pose[7];
  for (int k = 0; k < 7; ++k) pose[k] = poses[(size_t)f * 7 + k];
  const double ox = tex_origin[f * 2], oy = tex_origin[f * 2 + 1];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const int v = p / cam.width, u = p % cam.width;
    out[(size_t)f * npix + p] = synth_render_pixel(cam, pose, tex, tex_size, texels_per_metre, ox, oy, u, v);
  }
}

static CameraParams to_cam(const sl2_camera* c) {
  CameraParams cam;
  cam.width = c->width; cam.height = c->height; cam.fku = c->fku; cam.fkv = c->fkv;
  cam.u0 = c->u0; cam.v0 = c->v0; cam.kd1 = c->kd1; cam.sd = c->sd;
  return cam;
}

}  // namespace sl2

using namespace sl2;

extern "C" int sl2_synth_render_host(const sl2_camera* cam, const uint8_t* tex, int tex_size, double tex_extent,
                                     const double* tex_origin, const double* poses, int count, uint8_t* out) {
  if (!cam || !tex || !tex_origin || !poses || !out || count < 0 || tex_size <= 0 || (tex_size & (tex_size - 1))) return SL2_ERR_INVALID;
  const CameraParams c = to_cam(cam);
  const double tpm = (double)tex_size / tex_extent;
  const size_t npix = (size_t)c.width * c.height;
  for (int f = 0; f < count; ++f)
    for (int v = 0; v < c.height; ++v)
      for (int u = 0; u < c.width; ++u)
        out[(size_t)f * npix + (size_t)v * c.width + u] =
            synth_render_pixel(c, poses + (size_t)f * 7, tex, tex_size, tpm, tex_origin[f * 2], tex_origin[f * 2 + 1], u, v);
  return SL2_OK;
}

extern "C" int sl2_synth_render_device(int device, void* stream, const sl2_camera* cam, const uint8_t* tex_dev, int tex_size,
                                       double tex_extent, const double* tex_origin_dev, const double* poses_dev, int count,
                                       uint8_t* out_dev) {
  if (!cam || !tex_dev || !tex_origin_dev || !poses_dev || !out_dev || count < 0 || tex_size <= 0 || (tex_size & (tex_size - 1))) return SL2_ERR_INVALID;
  if (count == 0) return SL2_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device"); return SL2_ERR_NO_DEVICE; }
  SL2_HIP(hipSetDevice(device));
  const CameraParams c = to_cam(cam);
  const int npix = c.width * c.height;
  int bx = (npix + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(k_synth_render, dim3(bx, count), dim3(256), 0, (hipStream_t)stream, c, tex_dev, tex_size,
                     (double)tex_size / tex_extent, tex_origin_dev, poses_dev, out_dev);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}
