// Device-side building blocks of the feature-initialisation image operators, shared by the stateless
// batch operators (sl2_featureinit.hip) and the engine's mapping step (sl2_mapping.hip).  See
// sl2_featureinit.hip for the semantics and the reference lines they follow.
#pragma once
#include "sl2_common.hpp"
#include "sl2_math.hpp"

namespace sl2 {

constexpr int kDetThreads = 256;

// Shi-Tomasi detector over one region, executed by a whole workgroup of kDetThreads threads
// (MonoSLAM::find_best_patch_inside_region, monoslam.cpp:1070-1205).  uv: in/out (ubest, vbest); ev: evbest.
__device__ __forceinline__ void detect_region_wg(const uint8_t* __restrict__ img, int width, int height, int ustart, int vstart,
                                                 int ufinish, int vfinish, int* __restrict__ uv, double* __restrict__ ev) {
  const int tid = threadIdx.x;
  const int half = (kBoxSize - 1) / 2;
  if (ustart < half + 1) ustart = half + 1;                    // monoslam.cpp:1080-1091
  if (ufinish > width - half - 1) ufinish = width - half - 1;
  if (vstart < half + 1) vstart = half + 1;
  if (vfinish > height - half - 1) vfinish = height - half - 1;
  if (vstart >= vfinish || ustart >= ufinish) {                // :1094-1099
    if (tid == 0) { uv[0] = ustart; uv[1] = vstart; *ev = 0.0; }
    return;
  }
  const int nu = ufinish - ustart, nv = vfinish - vstart;
  double best = 0.0;   // *evbest = 0 (:1136): only a strictly positive eigenvalue can win
  int best_idx = -1;
  for (int idx = tid; idx < nu * nv; idx += kDetThreads) {
    const int v = vstart + idx / nu, u = ustart + idx % nu;
    int sxx = 0, syy = 0, sxy = 0;
    for (int r = v - half; r <= v + half; ++r) {
      const uint8_t* up = img + (size_t)(r - 1) * width;
      const uint8_t* mid = img + (size_t)r * width;
      const uint8_t* dn = img + (size_t)(r + 1) * width;
#pragma unroll
      for (int c = -5; c <= 5; ++c) {
        const int gx2 = (int)mid[u + c + 1] - (int)mid[u + c - 1];   // 2 gx
        const int gy2 = (int)dn[u + c] - (int)up[u + c];             // 2 gy
        sxx += gx2 * gx2; syy += gy2 * gy2; sxy += gx2 * gy2;
      }
    }
    const double A = sxx / 4.0, Bq = sxy / 4.0, C = syy / 4.0;       // exact
    const double BB = sqrt((A + C) * (A + C) - 4 * (A * C - Bq * Bq));  // find_eigenvalues, :1194-1205
    const double e2 = (A + C - BB) / 2.0;
    if (e2 > best) { best = e2; best_idx = idx; }
  }
  __shared__ double s_best[kDetThreads];
  __shared__ int s_idx[kDetThreads];
  s_best[tid] = best;
  s_idx[tid] = best_idx;
  __syncthreads();
  for (int off = kDetThreads / 2; off > 0; off >>= 1) {
    if (tid < off) {
      const double ob = s_best[tid + off];
      const int oi = s_idx[tid + off];
      const double mb = s_best[tid];
      const int mi = s_idx[tid];
      // larger eigenvalue wins; among equals the earlier scan position (a lane without a candidate has idx -1)
      if (oi >= 0 && (mi < 0 || ob > mb || (ob == mb && oi < mi))) { s_best[tid] = ob; s_idx[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    *ev = s_idx[0] >= 0 ? s_best[0] : 0.0;
    if (s_idx[0] >= 0) {                 // otherwise *ubest / *vbest keep the caller's values
      uv[0] = ustart + s_idx[0] % nu;
      uv[1] = vstart + s_idx[0] / nu;
    }
  }
}

// SearchDatum + the clipping of SearchMultipleOverlappingEllipses::search (cpp:43-51, 117-149):
// d = uc, vc, urelstart, nu, vrelstart, nv, halfwidth, halfheight
__device__ __forceinline__ void me_describe(double a, double b, double c, double cu, double cv, int width, int height, int* d) {
  const int hw = (int)(kNoSigma / sqrt(a - b * b / c));   // cpp:49-50
  const int hh = (int)(kNoSigma / sqrt(c - b * b / a));
  const int uc = int(cu), vc = int(cv);                   // truncation, no +0.5 (cpp:127-128)
  const int half = (kBoxSize - 1) / 2;
  int us = -hw, uf = hw, vs = -hh, vf = hh;
  if (uc + us - half < 0) us = half - uc;                               // cpp:131-149
  if (uc + uf - half > width - kBoxSize) uf = width - kBoxSize - uc + half;
  if (vc + vs - half < 0) vs = half - vc;
  if (vc + vf - half > height - kBoxSize) vf = height - kBoxSize - vc + half;
  d[0] = uc; d[1] = vc; d[2] = us; d[3] = uf - us + 1; d[4] = vs; d[5] = vf - vs + 1; d[6] = hw; d[7] = hh;
}

__device__ __forceinline__ bool me_visits(const int* __restrict__ d, const double* __restrict__ pu, int x, int y) {
  const int urel = x - d[0], vrel = y - d[1];
  if (urel < d[2] || urel >= d[2] + d[3] || vrel < d[4] || vrel >= d[4] + d[5]) return false;
  return in_ellipse(pu[0], pu[1], pu[2], urel, vrel);
}

// correlate2_warning at window centre (x, y) against the template in s_patch (121 ints), plus the
// low-image-sigma penalty (cpp:164-175)
__device__ __forceinline__ double me_score_position(const uint8_t* __restrict__ img, int width, const int* s_patch, int Sg0, int Sg0sq,
                                                    int x, int y) {
  const uint8_t* p1 = img + (size_t)(y - 5) * width + (x - 5);
  int Sg1 = 0, Sg0g1 = 0, Sg1sq = 0;
  for (int r = 0; r < 11; ++r)
#pragma unroll
    for (int cc = 0; cc < 11; ++cc) {
      const int g0 = s_patch[r * 11 + cc];
      const int g1 = p1[r * width + cc];
      Sg1 += g1; Sg0g1 += g0 * g1; Sg1sq += g1 * g1;
    }
  double sd0, sd1;
  double corr = ncc_score(Sg0, Sg1, Sg0g1, Sg0sq, Sg1sq, &sd0, &sd1);
  if (sd1 < kCorrelationSigmaThreshold) corr += 5.0;     // LOW_SIGMA_PENALTY, h:56 / cpp:173-175
  return corr;
}

// The union of a job's ellipses is scored once (the reference's per-call score cache, cpp:114,160-181).  Pass 1: every
// ellipse stamps the positions it visits in an int map (plain stores - the stamp only says "somebody visits this",
// so which store lands last is irrelevant; an atomicMin here cost 0.75 ms per search at batch 1024: the ellipses overlap
// heavily and the atomics serialise).  Pass 2: ONE scan over the union's bounding box per job finds the stamped
// positions, clears the stamps and scores each position once.  Pass 3: per-ellipse arg-min over the score map.
// (History: testing every earlier ellipse of the job for every position - 90 % of the search with 100 particle ellipses;
// then one workgroup per ellipse scoring the positions it owned - each re-scanned its own box, 0.31 ms of 0.55.)
constexpr int kOwnerFree = 0x7f7f7f7f;    // = memset(0x7f): above every ellipse index

// idx = q * n + r for 0 <= idx < 2^22, 0 < n: a float reciprocal estimate, corrected (the integer division sequence is ~40
// instructions, and the box scans below do little else)
__device__ __forceinline__ void box_divmod(int idx, int n, float rcp, int* q, int* r) {
  int qq = (int)((float)idx * rcp);
  int rr = idx - qq * n;
  if (rr < 0) { --qq; rr += n; }
  if (rr >= n) { ++qq; rr -= n; }
  *q = qq; *r = rr;
}

// One wavefront stamps one ellipse.
__device__ __forceinline__ void me_mark_ellipse_wave(const int* __restrict__ d, const double* __restrict__ pu, int width,
                                                     int* __restrict__ owner, int index) {
  const int nu = d[3], nv = d[5];
  if (nu <= 0 || nv <= 0) return;
  const float rcp = 1.0f / (float)nv;
  const double a = pu[0], b = pu[1], c = pu[2];
  for (int idx = threadIdx.x & 63; idx < nu * nv; idx += 64) {
    int q, r;
    box_divmod(idx, nv, rcp, &q, &r);
    const int urel = d[2] + q, vrel = d[4] + r;
    if (!in_ellipse(a, b, c, urel, vrel)) continue;
    owner[(size_t)(d[1] + vrel) * width + (d[0] + urel)] = index;     // racy on purpose: any visitor's stamp will do
  }
}

// Workgroup-collective (256 threads): score every stamped position of rows slice / nslices of the bounding box of the
// job's n_ell ellipses (descriptors desc[8 e]), clearing the stamps on the way.
__device__ __forceinline__ void me_score_union_wg(const uint8_t* __restrict__ img, int width, const uint8_t* __restrict__ patch121,
                                                  const int* __restrict__ desc, int n_ell, int* __restrict__ owner,
                                                  double* __restrict__ map, int slice, int nslices) {
  const int tid = threadIdx.x;
  __shared__ int s_patch[121];
  __shared__ int s_sums[2];
  __shared__ int s_box[4];
  if (tid < 121) s_patch[tid] = patch121[tid];
  if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -1; s_box[3] = -1; }
  __syncthreads();
  for (int e = tid; e < n_ell; e += 256) {
    const int* d = desc + 8 * (size_t)e;
    if (d[3] <= 0 || d[5] <= 0) continue;
    atomicMin(&s_box[0], d[0] + d[2]);
    atomicMin(&s_box[1], d[1] + d[4]);
    atomicMax(&s_box[2], d[0] + d[2] + d[3]);
    atomicMax(&s_box[3], d[1] + d[4] + d[5]);
  }
  if (tid == 0) {
    int s0 = 0, s0q = 0;
    for (int p = 0; p < 121; ++p) { s0 += s_patch[p]; s0q += s_patch[p] * s_patch[p]; }
    s_sums[0] = s0; s_sums[1] = s0q;
  }
  __syncthreads();
  if (s_box[2] < 0) return;
  const int Sg0 = s_sums[0], Sg0sq = s_sums[1];
  const int x0 = s_box[0], bw = s_box[2] - s_box[0];
  const int rows = s_box[3] - s_box[1], per = (rows + nslices - 1) / nslices;
  const int ys = s_box[1] + slice * per;
  const int ye = (ys + per < s_box[3]) ? ys + per : s_box[3];
  const int total = bw * (ye - ys);
  // Stamped positions are first compacted into an LDS list so that the 121-tap correlations run on full wavefronts.
  constexpr int kListCap = 1024;
  __shared__ int s_list[kListCap];
  __shared__ int s_n;
  for (int base = 0; base < total; base += kListCap) {
    if (tid == 0) s_n = 0;
    __syncthreads();
    const int end = (base + kListCap < total) ? base + kListCap : total;
    for (int idx = base + tid; idx < end; idx += 256) {
      const int y = ys + idx / bw, x = x0 + idx % bw;
      const size_t pos = (size_t)y * width + x;
      if (owner[pos] >= kOwnerFree) continue;
      owner[pos] = kOwnerFree;          // the stamp map is clean again for the next search
      s_list[atomicAdd(&s_n, 1)] = (y << 16) | x;
    }
    __syncthreads();
    const int n = s_n;
    for (int k = tid; k < n; k += 256) {
      const int x = s_list[k] & 0xffff, y = s_list[k] >> 16;
      map[(size_t)y * width + x] = me_score_position(img, width, s_patch, Sg0, Sg0sq, x, y);
    }
    __syncthreads();
  }
}

// Arg-min of ellipse e over its positions in scan order (u outer, v inner), "corr <= corrmax" => last minimum
// wins.  One wavefront.  out = (flag, u, v); returns the best score in *best_out (lane 0).
__device__ __forceinline__ void me_argmin_wave(int width, const int* __restrict__ d, const double* __restrict__ pu,
                                               const double* __restrict__ map, int* __restrict__ out, double* best_out) {
  const int lane = threadIdx.x & 63;
  const int nu = d[3], nv = d[5];
  double best = 1000000.0;   // cpp:156
  int order = -1;
  if (nu > 0 && nv > 0) {
    const float rcp = 1.0f / (float)nv;
    const double a = pu[0], b = pu[1], c = pu[2];
    for (int idx = lane; idx < nu * nv; idx += 64) {
      int q, r;
      box_divmod(idx, nv, rcp, &q, &r);
      const int urel = d[2] + q, vrel = d[4] + r;
      if (!in_ellipse(a, b, c, urel, vrel)) continue;
      const size_t pos = (size_t)(d[1] + vrel) * width + (d[0] + urel);
      const double corr = map[pos];
      if (corr <= best) { best = corr; order = idx; }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int oo = __shfl_xor(order, off, 64);
    if (oo >= 0 && (order < 0 || ob < best || (ob == best && oo > order))) { best = ob; order = oo; }
  }
  if (lane == 0) {
    out[0] = (order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;           // cpp:187-191
    out[1] = order >= 0 ? d[0] + d[2] + order / nv : 0;                  // result_u_ / result_v_ start at 0 (cpp:45-46)
    out[2] = order >= 0 ? d[1] + d[4] + order % nv : 0;
    if (best_out) *best_out = best;
  }
}

}  // namespace sl2
