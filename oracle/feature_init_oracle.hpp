// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// CPU restatement of the image operators of the feature-initialisation path
// (SURVEY.md 8(f) rank 1), written from the behaviour of the reference (paths relative
// to /root/reference/scenelib2/):
//   * MonoSLAM::find_best_patch_inside_region / find_eigenvalues   monoslam.cpp:1070-1205
//   * SearchMultipleOverlappingEllipses (SearchDatum, add_ellipse, search)
//                                   improc/search_multiple_overlapping_ellipses.{h,cpp}
//   * the libc generator the region choice draws from (srand48(0) at monoslam.cpp:1968,
//     drand48() at :989-990): POSIX 48-bit LCG, X' = (0x5DEECE66D X + 0xB) mod 2^48,
//     srand48(s): X = (s << 16) | 0x330E, drand48() = X' / 2^48  (glibc, version unpinned).
//
// PARITY STATUS: unpinned by reference outputs (the reference ships no tests and does not
// build here, see slam_oracle.hpp).  Pinned by properties: the detector's sums are exact
// (every term is a multiple of 1/4 below 2^24, so the reference's incremental FP64 sums equal
// the direct integer box sums — tests compare against an independent integer implementation),
// and the multi-ellipse search equals independent per-ellipse scans of the same score function
// (its correlation cache cannot change a result).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

// (reached through slam_oracle.hpp -> mapping_oracle.hpp; needs correlate2_warning from slam_oracle.hpp)

namespace oracle {

// monoslam.cpp:1194-1205
inline void find_eigenvalues(double A, double B, double C, double* eval1, double* eval2) {
  const double BB = std::sqrt((A + C) * (A + C) - 4 * (A * C - B * B));
  *eval1 = (A + C + BB) / 2.0;
  *eval2 = (A + C - BB) / 2.0;
}

// monoslam.cpp:1070-1192.  Returns the three outputs through pointers exactly like the
// reference: when no position has a positive smaller eigenvalue, *ubest / *vbest are left
// untouched and *evbest = 0 (except for the empty-region early return, which sets them).
inline void find_best_patch_inside_region(const uint8_t* image, int width, int height, int* ubest, int* vbest,
                                          double* evbest, int BOXSIZE, int ustart, int vstart, int ufinish, int vfinish) {
  const int half = (BOXSIZE - 1) / 2;
  // keep one pixel of margin for the central differences (monoslam.cpp:1080-1091)
  if (ustart < half + 1) ustart = half + 1;
  if (ufinish > width - half - 1) ufinish = width - half - 1;
  if (vstart < half + 1) vstart = half + 1;
  if (vfinish > height - half - 1) vfinish = height - half - 1;
  if (vstart >= vfinish || ustart >= ufinish) {  // :1094-1099
    *ubest = ustart;
    *vbest = vstart;
    *evbest = 0;
    return;
  }
  auto px = [&](int r, int c) -> int { return image[(size_t)r * width + c]; };
  auto grad = [&](int r, int c, double* gx, double* gy) {
    *gx = (px(r, c + 1) - px(r, c - 1)) / 2.0;
    *gy = (px(r + 1, c) - px(r - 1, c)) / 2.0;
  };
  const int calc_width = ufinish - ustart + BOXSIZE - 1;
  std::vector<double> colxx(calc_width), colyy(calc_width), colxy(calc_width);
  const int cstart = ustart - half, cfinish = ufinish + half, rstart = vstart - half;
  // column sums of height BOXSIZE for the first row position (:1118-1133)
  for (int c = cstart, i = 0; c < cfinish; ++c, ++i) {
    colxx[i] = colyy[i] = colxy[i] = 0;
    for (int r = rstart; r < rstart + BOXSIZE; ++r) {
      double gx, gy;
      grad(r, c, &gx, &gy);
      colxx[i] += gx * gx;
      colyy[i] += gy * gy;
      colxy[i] += gx * gy;
    }
  }
  *evbest = 0;
  for (int v = vstart; v < vfinish; ++v) {
    double Txx = 0.0, Tyy = 0.0, Txy = 0.0;
    for (int i = 0; i < BOXSIZE; ++i) { Txx += colxx[i]; Tyy += colyy[i]; Txy += colxy[i]; }
    for (int u = ustart; u < ufinish; ++u) {
      if (u != ustart) {  // slide one column (:1150-1155)
        Txx += colxx[u - ustart + BOXSIZE - 1] - colxx[u - ustart - 1];
        Tyy += colyy[u - ustart + BOXSIZE - 1] - colyy[u - ustart - 1];
        Txy += colxy[u - ustart + BOXSIZE - 1] - colxy[u - ustart - 1];
      }
      double e1, e2;
      find_eigenvalues(Txx, Txy, Tyy, &e1, &e2);
      if (e2 > *evbest) {  // strict: the first maximum in scan order wins
        *ubest = u;
        *vbest = v;
        *evbest = e2;
      }
    }
    if (v != vfinish - 1) {  // slide the column sums one row down (:1170-1190)
      for (int c = cstart, i = 0; c < cfinish; ++c, ++i) {
        double gx, gy;
        grad(v - half, c, &gx, &gy);
        colxx[i] -= gx * gx;
        colyy[i] -= gy * gy;
        colxy[i] -= gx * gy;
        grad(v + half + 1, c, &gx, &gy);
        colxx[i] += gx * gx;
        colyy[i] += gy * gy;
        colxy[i] += gx * gy;
      }
    }
  }
}

// improc/search_multiple_overlapping_ellipses.{h,cpp}
struct MultiEllipseSearch {
  static constexpr double kCorrThresh2 = 0.40;              // h:49
  static constexpr double kCorrelationSigmaThreshold = 10.0;  // h:52
  static constexpr double kNoSigma = 3.0;                   // h:54
  static constexpr double kLowSigmaPenalty = 5.0;           // h:56

  struct Datum {  // SearchDatum, h:79-110 / cpp:43-51
    double a, b, c;       // PuInv(0,0), PuInv(0,1), PuInv(1,1)
    double cu, cv;        // search_centre
    bool result_flag = false;
    int result_u = 0, result_v = 0;
    int halfwidth, halfheight;
    double corrmax = 0.0;  // diagnostic (not stored by the reference)
    Datum(double a_, double b_, double c_, double cu_, double cv_) : a(a_), b(b_), c(c_), cu(cu_), cv(cv_) {
      halfwidth = (int)(kNoSigma / std::sqrt(a - b * b / c));
      halfheight = (int)(kNoSigma / std::sqrt(c - b * b / a));
    }
    bool inside_relative(int u, int v) const {  // cpp:88-92
      return (a * u * u + 2 * b * u * v + c * v * v < kNoSigma * kNoSigma);
    }
  };

  const uint8_t* image;
  int width, height;
  const uint8_t* patch;
  int boxsize;
  std::vector<Datum> data;
  long long correlations = 0;  // diagnostic: positions actually correlated (cache misses)

  MultiEllipseSearch(const uint8_t* image_, int w, int h, const uint8_t* patch_, int boxsize_)
      : image(image_), width(w), height(h), patch(patch_), boxsize(boxsize_) {}

  void add_ellipse(double a, double b, double c, double cu, double cv) { data.emplace_back(a, b, c, cu, cv); }

  // cpp:106-196
  void search() {
    std::vector<double> cache((size_t)width * height, -1.0);
    const int half = (boxsize - 1) / 2;
    for (Datum& d : data) {
      int urelstart = -d.halfwidth, urelfinish = d.halfwidth, vrelstart = -d.halfheight, vrelfinish = d.halfheight;
      const int ucentre = int(d.cu), vcentre = int(d.cv);  // truncation, no +0.5 here (cpp:127-128)
      if (ucentre + urelstart - half < 0) urelstart = half - ucentre;
      if (ucentre + urelfinish - half > width - boxsize) urelfinish = width - boxsize - ucentre + half;
      if (vcentre + vrelstart - half < 0) vrelstart = half - vcentre;
      if (vcentre + vrelfinish - half > height - boxsize) vrelfinish = height - boxsize - vcentre + half;
      double corrmax = 1000000.0;
      for (int urel = urelstart; urel <= urelfinish; ++urel) {
        for (int vrel = vrelstart; vrel <= vrelfinish; ++vrel) {
          if (!d.inside_relative(urel, vrel)) continue;
          double& slot = cache[(size_t)(vcentre + vrel) * width + (ucentre + urel)];
          double corr;
          if (slot != -1.0) {
            corr = slot;
          } else {
            double sdpatch, sdimage;
            corr = correlate2_warning(0, 0, boxsize, boxsize, ucentre + urel - half, vcentre + vrel - half, patch, boxsize,
                                      image, width, &sdpatch, &sdimage);
            if (sdimage < kCorrelationSigmaThreshold) corr += kLowSigmaPenalty;
            slot = corr;
            ++correlations;
          }
          if (corr <= corrmax) {
            corrmax = corr;
            d.result_u = urel + ucentre;
            d.result_v = vrel + vcentre;
          }
        }
      }
      d.corrmax = corrmax;
      d.result_flag = !(corrmax > kCorrThresh2);
    }
  }
};

// POSIX drand48 family as used by the reference (srand48(0) once in Init, drand48() in
// FindNonOverlappingRegionNoPredict).
struct Rand48 {
  uint64_t x = 0x330EULL;
  void seed(long s) { x = (((uint64_t)(uint32_t)s) << 16) | 0x330EULL; }
  double next() {
    x = (0x5DEECE66DULL * x + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)x / 281474976710656.0;  // 2^48
  }
};

}  // namespace oracle
