// Synthetic-input renderer: one pixel of the view of a textured
// plane z = 0 seen by the reference's camera model.  SL2_HD: the same source is
// run by the HIP render kernel and by the host renderer, and — being built from
// IEEE + - * / sqrt floor only, with FP contraction off — both produce identical
// bytes (tests/test_synth.py checks this on the GPU box).
//
// Inverse of Camera::Project = Camera::Unproject (camera.cpp:133-154):
//   centred = (u,v) - (u0,v0); und = centred / sqrt(1 - 2 k1 |centred|^2);
//   ray_cam = (und_x / -fku, und_y / -fkv, 1); ray_world = R(q) ray_cam.
#pragma once
#include "sl2_math.hpp"

namespace sl2 {

SL2_HD uint8_t synth_render_pixel(const CameraParams& cam, const double pose[7], const uint8_t* tex, int tex_size,
                                  double texels_per_metre, double ox, double oy, int u, int v) {
  const double c0 = (double)u - cam.u0, c1 = (double)v - cam.v0;
  const double radius2 = c0 * c0 + c1 * c1;
  const double factor = sqrt(1 - 2 * cam.kd1 * radius2);
  const double rc0 = (c0 / factor) / -cam.fku;
  const double rc1 = (c1 / factor) / -cam.fkv;
  double R[9];
  quat_to_rot(&pose[3], R);
  const double rw0 = R[0] * rc0 + R[1] * rc1 + R[2];
  const double rw1 = R[3] * rc0 + R[4] * rc1 + R[5];
  const double rw2 = R[6] * rc0 + R[7] * rc1 + R[8];
  if (!(rw2 > 0.0) || !(pose[2] < 0.0)) return 0;  // plane not in front of the camera
  const double t = -pose[2] / rw2;
  const double X = pose[0] + t * rw0;
  const double Y = pose[1] + t * rw1;
  // texture coordinates (texel centres at integer + 0.5), torus-wrapped
  const double tx = (X - ox) * texels_per_metre + 0.5 * tex_size - 0.5;
  const double ty = (Y - oy) * texels_per_metre + 0.5 * tex_size - 0.5;
  const double fx0 = floor(tx), fy0 = floor(ty);
  const double ax = tx - fx0, ay = ty - fy0;
  const int mask = tex_size - 1;  // tex_size is a power of two
  const int ix0 = ((int)fx0) & mask, iy0 = ((int)fy0) & mask;
  const int ix1 = (ix0 + 1) & mask, iy1 = (iy0 + 1) & mask;
  const double t00 = tex[iy0 * tex_size + ix0], t01 = tex[iy0 * tex_size + ix1];
  const double t10 = tex[iy1 * tex_size + ix0], t11 = tex[iy1 * tex_size + ix1];
  const double top = t00 + ax * (t01 - t00);
  const double bot = t10 + ax * (t11 - t10);
  const double val = top + ay * (bot - top);
  double r = floor(val + 0.5);
  if (r < 0.0) r = 0.0;
  if (r > 255.0) r = 255.0;
  return (uint8_t)r;
}

}  // namespace sl2
