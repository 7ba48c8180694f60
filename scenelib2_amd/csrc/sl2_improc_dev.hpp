// Device-side building blocks of the feature-initialisation image operators, shared by the stateless
// batch operators (sl2_featureinit.hip) and the engine's mapping step (sl2_mapping.hip).  See
// sl2_featureinit.hip for the semantics and the reference lines they follow.
#pragma once
#include "sl2_common.hpp"
#include "sl2_math.hpp"

namespace sl2 {

#ifdef SL2_ME_TRACE   // development build (scripts/me_trace.py): cycles per phase of me_search_fused_wg, summed over workgroups
__device__ unsigned long long g_me_trace[16];
#define METR(slot) do { if (threadIdx.x == 0) { const long long now_ = (long long)__builtin_readcyclecounter(); atomicAdd(&g_me_trace[slot], (unsigned long long)(now_ - t_me_)); t_me_ = now_; } } while (0)
#define METR_BEGIN long long t_me_ = (long long)__builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_me_trace[15], 1ull)
#else
#define METR(slot) do { } while (0)
#define METR_BEGIN do { } while (0)
#endif
constexpr int kDetThreads = 1024;     // (round 4: 256 until the mapping step showed that a region's job has a CU to itself - what counts is its latency)

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
  METR_BEGIN;
  const int nu = ufinish - ustart;
  double best = 0.0;   // *evbest = 0 (:1136): only a strictly positive eigenvalue can win
  int best_idx = -1;
  // The three 11 x 11 sums of gradient products are separable, like the reference's own sliding column sums
  // (monoslam.cpp:1120-1192): per tile of kDetTW x kDetTH positions the image bytes go to LDS once, every (row, column)
  // gets its three HORIZONTAL 11-sums (33 multiply-adds on bytes from LDS), every position the VERTICAL sum of eleven of
  // those - ~45 taps per position instead of 121 x 3 (rounds 1-3; 0.15 ms per mapping step at batch 1024).  All integer,
  // so the order of the sums is immaterial; the eigenvalue is the reference's FP64 expression on exact inputs.
  constexpr int kDetTW = 80, kDetTH = 60;       // (the engine's regions are 80 x 60: one tile, one round trip for its image; 20 rows until round 4)
  __shared__ uint8_t s_img[(kDetTH + 12) * (kDetTW + 12)];
  __shared__ int s_h[3][(kDetTH + 10) * kDetTW];
  for (int v0 = vstart; v0 < vfinish; v0 += kDetTH)
    for (int u0 = ustart; u0 < ufinish; u0 += kDetTW) {
      const int tw = min(kDetTW, ufinish - u0), th = min(kDetTH, vfinish - v0);
      const int iw = tw + 12, ih = th + 12;                     // image bytes: rows v0 - 6 .. v0 + th + 5, columns u0 - 6 .. u0 + tw + 5
      __syncthreads();                                          // the previous tile's sums have been consumed
      {
        // every byte of the tile is requested before the first one is parked (a load - store - load chain was one memory
        // round trip per byte and thread: 18 us of a 35 us job)
        constexpr int kPer = ((kDetTH + 12) * (kDetTW + 12) + kDetThreads - 1) / kDetThreads;
        uint8_t pix[kPer];
        int at[kPer];
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          const int i = tid + q * kDetThreads;
          const int r = i / iw, c = i - r * iw;
          at[q] = i < iw * ih ? r * (kDetTW + 12) + c : -1;
          pix[q] = img[(size_t)(v0 - 6 + min(r, ih - 1)) * width + (u0 - 6 + c)];              // inside the frame: the region is clamped 6 px in
        }
#pragma unroll
        for (int q = 0; q < kPer; ++q)
          if (at[q] >= 0) s_img[at[q]] = pix[q];
      }
      __syncthreads();
      METR(5);
      // horizontal sums for rows v0 - 5 .. v0 + th + 4 (local row index 1 .. th + 10 of s_img), columns u0 .. u0 + tw - 1:
      // a thread takes eight consecutive columns of a row - the first window in full (eleven taps), the next seven by
      // sliding (one tap in, one out): 25 taps for eight sums instead of 88 (integers: the order of the sums is immaterial)
      {
        const int nseg = (tw + 7) >> 3;
        for (int i = tid; i < (th + 10) * nseg; i += kDetThreads) {
          const int rr = i / nseg, c0 = (i - rr * nseg) * 8;
          const uint8_t* up = s_img + rr * (kDetTW + 12) + c0 + 1;        // row above, at column u - 5 of the segment's first window
          const uint8_t* mid = up + (kDetTW + 12);
          const uint8_t* dn = mid + (kDetTW + 12);
          auto tap = [&](int c, int& xx, int& yy, int& xy) {
            const int gx2 = (int)mid[c + 1] - (int)mid[c - 1];            // 2 gx
            const int gy2 = (int)dn[c] - (int)up[c];                      // 2 gy
            xx = gx2 * gx2; yy = gy2 * gy2; xy = gx2 * gy2;
          };
          int sxx = 0, syy = 0, sxy = 0;
#pragma unroll
          for (int c = 0; c < 11; ++c) { int a, b, d; tap(c, a, b, d); sxx += a; syy += b; sxy += d; }
          int* h0 = &s_h[0][rr * kDetTW + c0];
          int* h1 = &s_h[1][rr * kDetTW + c0];
          int* h2 = &s_h[2][rr * kDetTW + c0];
          h0[0] = sxx; h1[0] = syy; h2[0] = sxy;
          const int nc = min(8, tw - c0);
          for (int k = 1; k < nc; ++k) {
            int a, b, d, a0, b0, d0;
            tap(k + 10, a, b, d);
            tap(k - 1, a0, b0, d0);
            sxx += a - a0; syy += b - b0; sxy += d - d0;
            h0[k] = sxx; h1[k] = syy; h2[k] = sxy;
          }
        }
      }
      __syncthreads();
      METR(6);
      for (int i = tid; i < th * tw; i += kDetThreads) {
        const int rv = i / tw, cu = i - rv * tw;
        int sxx = 0, syy = 0, sxy = 0;
#pragma unroll
        for (int r = 0; r < 11; ++r) {
          sxx += s_h[0][(rv + r) * kDetTW + cu]; syy += s_h[1][(rv + r) * kDetTW + cu]; sxy += s_h[2][(rv + r) * kDetTW + cu];
        }
        const double A = sxx / 4.0, Bq = sxy / 4.0, C = syy / 4.0;       // exact
        const double BB = sqrt((A + C) * (A + C) - 4 * (A * C - Bq * Bq));  // find_eigenvalues, :1194-1205
        const double e2 = (A + C - BB) / 2.0;
        // the reference scans v outer / u inner and keeps the FIRST maximum (strict '>'): here positions arrive tile by
        // tile, so the scan index decides among equals
        const int idx = (v0 - vstart + rv) * nu + (u0 - ustart + cu);
        if (e2 > best || (e2 == best && best_idx >= 0 && idx < best_idx)) { best = e2; best_idx = idx; }
      }
    }
  METR(7);
  // larger eigenvalue wins; among equals the earlier scan position (a lane without a candidate has idx -1)
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(best_idx, off, 64);
    if (oi >= 0 && (best_idx < 0 || ob > best || (ob == best && oi < best_idx))) { best = ob; best_idx = oi; }
  }
  __shared__ double s_best[kDetThreads / 64];
  __shared__ int s_idx[kDetThreads / 64];
  if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_idx[tid >> 6] = best_idx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < kDetThreads / 64; ++w) {
      const double ob = s_best[w];
      const int oi = s_idx[w];
      if (oi >= 0 && (s_idx[0] < 0 || ob > s_best[0] || (ob == s_best[0] && oi < s_idx[0]))) { s_best[0] = ob; s_idx[0] = oi; }
    }
  }
  METR(8);
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

// The same score with the window in LDS (a byte tile of pitch iw whose allocation runs at least 16 bytes past the last
// byte a window can touch) and the template as 33 packed dwords (11 rows x 12 bytes, byte 11 = 0): a window row is twelve
// bytes out of four aligned dwords (v_alignbyte), the three sums are v_dot4_u32_u8 - 20 instructions a row instead of 66
// for eleven byte taps.  Integer sums: identical.
__device__ __forceinline__ double me_score_position_lds(const uint8_t* s_img, int iw, const unsigned* s_tpl, int Sg0, int Sg0sq,
                                                        int x, int y) {
  unsigned Sg1 = 0, Sg0g1 = 0, Sg1sq = 0;
  int addr = (y - 5) * iw + (x - 5);
#pragma unroll
  for (int r = 0; r < 11; ++r) {
    const unsigned* base = (const unsigned*)(s_img + (addr & ~3));
    const unsigned sh = (unsigned)addr & 3u;
    const unsigned d0 = base[0], d1 = base[1], d2 = base[2], d3 = base[3];
    const unsigned w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    const unsigned w2 = __builtin_amdgcn_alignbyte(d3, d2, sh) & 0x00ffffffu;
    const unsigned t0 = s_tpl[3 * r], t1 = s_tpl[3 * r + 1], t2 = s_tpl[3 * r + 2];
    Sg1 = __builtin_amdgcn_udot4(w0, 0x01010101u, Sg1, false);
    Sg1 = __builtin_amdgcn_udot4(w1, 0x01010101u, Sg1, false);
    Sg1 = __builtin_amdgcn_udot4(w2, 0x01010101u, Sg1, false);
    Sg0g1 = __builtin_amdgcn_udot4(w0, t0, Sg0g1, false);
    Sg0g1 = __builtin_amdgcn_udot4(w1, t1, Sg0g1, false);
    Sg0g1 = __builtin_amdgcn_udot4(w2, t2, Sg0g1, false);
    Sg1sq = __builtin_amdgcn_udot4(w0, w0, Sg1sq, false);
    Sg1sq = __builtin_amdgcn_udot4(w1, w1, Sg1sq, false);
    Sg1sq = __builtin_amdgcn_udot4(w2, w2, Sg1sq, false);
    addr += iw;
  }
  double sd0, sd1;
  double corr = ncc_score(Sg0, (int)Sg1, (int)Sg0g1, Sg0sq, (int)Sg1sq, &sd0, &sd1);
  if (sd1 < kCorrelationSigmaThreshold) corr += 5.0;     // LOW_SIGMA_PENALTY, h:56 / cpp:173-175
  return corr;
}

// The union of a job's ellipses is scored once (the reference's per-call score cache, cpp:114,160-181): every position of
// the union's bounding box gets its score, then every ellipse takes the arg-min over the positions it visits.
// (History: testing every earlier ellipse of the job for every position - 90 % of the search with 100 particle ellipses;
// one workgroup per ellipse scoring the positions it owned - each re-scanned its own box, 0.31 ms of 0.55; ellipses
// stamping the positions they visit in an int map, one scan per job compacting the stamped positions into lists - round 4
// dropped the stamps: scoring a position nobody visits is cheaper than finding out that nobody does.)

// idx = q * n + r for 0 <= idx < 2^22, 0 < n: a float reciprocal estimate, corrected (the integer division sequence is ~40
// instructions, and the box scans below do little else)
__device__ __forceinline__ void box_divmod(int idx, int n, float rcp, int* q, int* r) {
  int qq = (int)((float)idx * rcp);
  int rr = idx - qq * n;
  if (rr < 0) { --qq; rr += n; }
  if (rr >= n) { ++qq; rr -= n; }
  *q = qq; *r = rr;
}

// Workgroup-collective (256 threads): the scores of rows slice / nslices of the bounding box of the job's n_ell ellipses
// (descriptors desc[8 e]) go to the job's image-sized score map - EVERY position of the box, whether an ellipse visits it
// or not (round 4 stamped the visited ones first, a launch of its own over every ellipse's box, and compacted them into
// lists: a frame-sized box is 77 k positions of ~250 instructions, nothing for the chip once it is spread over workgroups;
// nobody reads a score outside its ellipse).  The image under a band of rows goes through LDS (me_score_position_lds).
constexpr int kMeBandBytes = 16384;
__device__ __forceinline__ void me_score_box_wg(const uint8_t* __restrict__ img, int width, const uint8_t* __restrict__ patch121,
                                                const int* __restrict__ desc, int n_ell, double* __restrict__ map, int slice,
                                                int nslices) {
  const int tid = threadIdx.x, lane = tid & 63;
  __shared__ int s_patch[121];
  __shared__ unsigned s_tpl[33];
  __shared__ int s_sums[2];
  __shared__ int s_box[4];
  __shared__ __attribute__((aligned(16))) uint8_t s_band[kMeBandBytes + 16];
  const int pv = patch121[min(tid, 120)];
  int lo_x = 0x7fffffff, lo_y = 0x7fffffff, hi_x = -1, hi_y = -1;
  for (int e = tid; e < n_ell; e += 256) {
    const int* d = desc + 8 * (size_t)e;
    const int nu = d[3], nv = d[5];
    if (nu <= 0 || nv <= 0) continue;
    lo_x = min(lo_x, d[0] + d[2]); lo_y = min(lo_y, d[1] + d[4]);
    hi_x = max(hi_x, d[0] + d[2] + nu); hi_y = max(hi_y, d[1] + d[4] + nv);
  }
  if (tid < 121) s_patch[tid] = pv;
  if (tid == 0) { s_box[0] = 0x7fffffff; s_box[1] = 0x7fffffff; s_box[2] = -1; s_box[3] = -1; }
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) {
    lo_x = min(lo_x, __shfl_xor(lo_x, off, 64)); lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
    hi_x = max(hi_x, __shfl_xor(hi_x, off, 64)); hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
  }
  if (lane == 0 && hi_x >= 0) { atomicMin(&s_box[0], lo_x); atomicMin(&s_box[1], lo_y); atomicMax(&s_box[2], hi_x); atomicMax(&s_box[3], hi_y); }
  if (tid >= 64 && tid < 64 + 33) {                      // the template packed for me_score_position_lds
    const int row = (tid - 64) / 3, k = (tid - 64) % 3;
    unsigned w = 0;
    for (int e = 0; e < 4; ++e) { const int col = 4 * k + e; if (col < 11) w |= (unsigned)s_patch[row * 11 + col] << (8 * e); }
    s_tpl[tid - 64] = w;
  }
  if (tid >= 128 && tid < 192) {                         // the template's sums
    int s0 = 0, s0q = 0;
    for (int p = lane; p < 121; p += 64) { s0 += s_patch[p]; s0q += s_patch[p] * s_patch[p]; }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s0q += __shfl_xor(s0q, off, 64); }
    if (lane == 0) { s_sums[0] = s0; s_sums[1] = s0q; }
  }
  __syncthreads();
  if (s_box[2] < 0) return;
  const int Sg0 = s_sums[0], Sg0sq = s_sums[1];
  const int x0 = s_box[0], bw = s_box[2] - s_box[0];
  const int rows = s_box[3] - s_box[1], per = (rows + nslices - 1) / nslices;
  const int ys = s_box[1] + slice * per;
  const int ye = (ys + per < s_box[3]) ? ys + per : s_box[3];
  const int iw = bw + 10;
  const int band = kMeBandBytes / iw - 10;               // rows of positions whose windows fit the LDS band (iw <= width + 10)
  if (band <= 0) return;                                 // (a frame wider than ~1600 pixels: not reachable, me_big is bounded by sl2_create's checks)
  const float rcpi = 1.0f / (float)iw, rcpw = 1.0f / (float)bw;
  for (int yb = ys; yb < ye; yb += band) {
    const int nr = min(band, ye - yb);
    __syncthreads();                                     // the previous band has been scored
    for (int i = tid; i < iw * (nr + 10); i += 256) {
      int r, q;
      box_divmod(i, iw, rcpi, &r, &q);
      s_band[i] = img[(size_t)(yb - 5 + r) * width + (x0 - 5 + q)];
    }
    __syncthreads();
    for (int idx = tid; idx < bw * nr; idx += 256) {
      int r, q;
      box_divmod(idx, bw, rcpw, &r, &q);
      map[(size_t)(yb + r) * width + (x0 + q)] = me_score_position_lds(s_band, iw, s_tpl, Sg0, Sg0sq, q + 5, r + 5);
    }
  }
}

// Every position (column q, row r) of an ellipse's nu x nv bounding box that lies inside the ellipse, one wavefront:
// use(fetch(q, r), q, r).  A lane keeps ONE column (or one per block of 64 columns) and walks rows, 64 / W rows of W columns
// per step, W the power of two that holds nu: the column's terms of the reference's expression ((a u) u, (2 b) u) are formed
// once, a position costs the row's terms and the compare - twelve instructions where the flat index walk (position = step *
// 64 + lane, divided by nu) took 35.  Four rows are worked on together: a job has a workgroup to itself and the chip is
// mostly idle (~300 jobs a step), so what counts is the length of a wavefront's dependent chain - four predicates and four
// fetches (unconditional, from a row clamped into the box) in flight instead of one.  Same operations in the same order as
// in_ellipse; use() is called in increasing row order.
template <typename Fetch, typename Use>
__device__ __forceinline__ void me_for_each_inside(double a, double b, double c, int us, int nu, int vs, int nv, int lane, Fetch fetch,
                                                   Use use, int r_first = 0, int r_end = 0x7fffffff, int lanes = 64) {
  // (lanes = 32, lane = 0 .. 31: HALF a wavefront walks the box - two small ellipses side by side; needs nu <= 32)
  const int lg = nu <= 16 ? 4 : (nu <= 32 ? 5 : 6);
  const int ql = lane & ((1 << lg) - 1), rs = lane >> lg, rp = lanes >> lg;
  const double b2 = 2 * b;
  for (int cb = 0; cb < nu; cb += 64) {
    const int q = cb + ql;
    if (q >= nu) continue;
    const double du = (double)(us + q);
    const double t1 = a * du * du, bu = b2 * du;
    const int rend = min(nv, r_end);                      // (rows r_first .. r_end - 1 only: a tall box shared by several wavefronts)
    for (int r0 = r_first + rs; r0 < rend; r0 += 4 * rp) {
      bool in[4];
      decltype(fetch(0, 0)) val[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int r = r0 + k * rp;
        const double dv = (double)(vs + r);
        in[k] = r < rend && (t1 + bu * dv + c * dv * dv < kNoSigma * kNoSigma);
        val[k] = fetch(q, min(r, rend - 1));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (in[k]) use(val[k], q, r0 + k * rp);
    }
  }
}

// Arg-min of ellipse e over its positions in scan order (u outer, v inner), "corr <= corrmax" => last minimum
// wins.  One wavefront.  out = (flag, u, v); returns the best score in *best_out (lane 0).
// me_argmin_part: the wave-reduced (best, order) of rows part * ceil(nv / parts) ... of the box, valid on every lane
__device__ __forceinline__ void me_argmin_part(int width, const int* __restrict__ d, const double* __restrict__ pu,
                                               const double* __restrict__ map, int part, int parts, double* best_o, int* order_o) {
  const int lane = threadIdx.x & 63;
  const int nu = d[3], nv = d[5];
  double best = 1000000.0;   // cpp:156
  int order = -1;
  if (nu > 0 && nv > 0) {
    // A lane keeps a column and walks rows (me_for_each_inside: consecutive lanes along an image row = coalesced reads of the
    // score map, four rows' loads in flight); it meets its candidates in increasing scan order (u outer, v inner), so "corr <=
    // best: take it" is the whole rule inside a lane and the order index only decides between lanes.
    const int rows_per = (nv + parts - 1) / parts;
    const double* m0 = map + (size_t)(d[1] + d[4]) * width + (d[0] + d[2]);
    me_for_each_inside(pu[0], pu[1], pu[2], d[2], nu, d[4], nv, lane, [&](int q, int r) { return m0[(size_t)r * width + q]; },
                       [&](double corr, int q, int r) { if (corr <= best) { best = corr; order = q * nv + r; } },
                       part * rows_per, (part + 1) * rows_per);
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int oo = __shfl_xor(order, off, 64);
    if (oo >= 0 && (order < 0 || ob < best || (ob == best && oo > order))) { best = ob; order = oo; }
  }
  *best_o = best;
  *order_o = order;
}
__device__ __forceinline__ void me_argmin_wave(int width, const int* __restrict__ d, const double* __restrict__ pu,
                                               const double* __restrict__ map, int* __restrict__ out, double* best_out) {
  double best;
  int order;
  me_argmin_part(width, d, pu, map, 0, 1, &best, &order);
  if ((threadIdx.x & 63) == 0) {
    const int nv = d[5];
    out[0] = (order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;           // cpp:187-191
    out[1] = order >= 0 ? d[0] + d[2] + order / nv : 0;                  // result_u_ / result_v_ start at 0 (cpp:45-46)
    out[2] = order >= 0 ? d[1] + d[4] + order % nv : 0;
    if (best_out) *best_out = best;
  }
}

// ---------------------------------------------------------------------------
// The whole multi-ellipse search of ONE job by ONE workgroup (any multiple of 64 threads; the callers use 1024), round 4.
//
// A job's ellipses - the depth particles of one partially initialised feature - lie along one epipolar line and overlap
// almost completely: in the mapping workload of bench.py --mapping about 95 ellipses of ~310 positions each (29.6 k box
// positions per job) cover a union whose BOUNDING BOX has ~480 positions (p90 810, largest seen 2 200).  The three-kernel
// form walked every ellipse's box twice through image-sized maps in HBM, in the reference's scan order (u outer, v inner:
// consecutive lanes = consecutive image ROWS, one cache line per lane): 0.46 + 0.40 + 0.45 ms per step at batch 1024.
// Here the stamps and the scores of the union's bounding box live in LDS (up to kMeCap positions), every box is walked
// with consecutive lanes along a row, and nothing but the image and the results touches memory.  Order matters to the
// result in one place only - "corr <= corrmax" lets the LAST minimum in scan order win - and that is carried as an
// explicit order index (u-major) in the arg-min.  A bounding box beyond kMeCap positions (a freshly created feature under a
// weak pose estimate can have ellipses as large as the frame: 95 x 71 k box positions) is NOT done here - one workgroup
// would take milliseconds - : the function returns false and the caller runs the three-pass form, spread over many
// workgroups, for those jobs (a handful per step).
//   pu_of(e)  -> pointer to (PuInv(0,0), PuInv(0,1), PuInv(1,1)) of ellipse e
//   emit(e, flag, u, v, best)   called by one lane per ellipse
// ---------------------------------------------------------------------------
constexpr int kMeCap = 2048;
constexpr int kMeEllCap = 256;       // ellipses of a job whose records fit the one-workgroup form
constexpr int kMeImgCap = 6144;      // bytes of image under a union's bounding box (+ 5 pixels all round) kept in LDS
template <typename PuFn, typename EmitFn>
__device__ __forceinline__ bool me_search_fused_wg(const uint8_t* __restrict__ img, int width, const uint8_t* __restrict__ patch121,
                                                   const int* __restrict__ desc, int n_ell, PuFn pu_of, EmitFn emit) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = (int)blockDim.x, nwave = nthr >> 6;
  METR_BEGIN;
  __shared__ int f_patch[121];
  __shared__ int f_sums[2];
  __shared__ int f_box[4];
  __shared__ double f_score[kMeCap];
  __shared__ __attribute__((aligned(16))) uint8_t f_img[kMeImgCap + 16];      // (+ 16: me_score_position_lds reads whole dwords)
  __shared__ unsigned f_tpl[33];
  // A job has a workgroup to itself on a chip that is mostly idle (~300 jobs a step): what counts is the length of its
  // chain of memory round trips.  Everything the search reads besides the image - template, the ellipses' boxes and S^-1 -
  // is requested HERE, by all threads at once, and parked in LDS: the passes below walk ~7 ellipses per wavefront twice,
  // and fetching each one's record when its turn came was two dependent round trips per ellipse and pass; loading the
  // template, the packed template, the boxes and S^-1 in four separate blocks was four round trips in a row (10 us of a
  // 40 us job, scripts/me_trace.py).  (More ellipses than kMeEllCap, or a workgroup of another size: the job takes the
  // path of the oversized unions, which has neither limit.)
  if (n_ell > kMeEllCap || nthr != 1024) return false;
  __shared__ int f_desc[kMeEllCap * 8];
  __shared__ double f_pu[kMeEllCap * 3];
  {
    const int nd = n_ell * 8, np = n_ell * 3;
    const int ip = max(min(tid, np - 1), 0);            // (clamped: every thread loads, only the owners store)
    const int pv = patch121[min(tid, 120)];
    const int dA = desc[max(min(tid, nd - 1), 0)], dB = desc[max(min(tid + 1024, nd - 1), 0)];
    const double puv = pu_of(ip / 3)[ip % 3];
    if (tid < 121) f_patch[tid] = pv;
    if (tid < nd) f_desc[tid] = dA;
    if (tid + 1024 < nd) f_desc[tid + 1024] = dB;
    if (tid < np) f_pu[tid] = puv;
    if (tid == 0) { f_box[0] = 0x7fffffff; f_box[1] = 0x7fffffff; f_box[2] = -1; f_box[3] = -1; }
  }
  __syncthreads();
  if (tid < kMeEllCap) {                                 // waves 0-3: the union's bounding box, one ellipse per thread
    int lo_x = 0x7fffffff, lo_y = 0x7fffffff, hi_x = -1, hi_y = -1;
    if (tid < n_ell) {
      const int* d = f_desc + 8 * tid;
      if (d[3] > 0 && d[5] > 0) { lo_x = d[0] + d[2]; lo_y = d[1] + d[4]; hi_x = lo_x + d[3]; hi_y = lo_y + d[5]; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      lo_x = min(lo_x, __shfl_xor(lo_x, off, 64)); lo_y = min(lo_y, __shfl_xor(lo_y, off, 64));
      hi_x = max(hi_x, __shfl_xor(hi_x, off, 64)); hi_y = max(hi_y, __shfl_xor(hi_y, off, 64));
    }
    if (lane == 0 && hi_x >= 0) { atomicMin(&f_box[0], lo_x); atomicMin(&f_box[1], lo_y); atomicMax(&f_box[2], hi_x); atomicMax(&f_box[3], hi_y); }
  } else if (wave == 4) {                                // the template's sums
    int s0 = 0, s0q = 0;
    for (int p = lane; p < 121; p += 64) { s0 += f_patch[p]; s0q += f_patch[p] * f_patch[p]; }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s0q += __shfl_xor(s0q, off, 64); }
    if (lane == 0) { f_sums[0] = s0; f_sums[1] = s0q; }
  } else if (wave == 5 && lane < 33) {                   // the template packed for me_score_position_lds
    const int row = lane / 3, k = lane % 3;
    unsigned w = 0;
    for (int e = 0; e < 4; ++e) { const int col = 4 * k + e; if (col < 11) w |= (unsigned)f_patch[row * 11 + col] << (8 * e); }
    f_tpl[lane] = w;
  }
  __syncthreads();
  METR(0);
  const bool any = f_box[2] >= 0;
  const int x0 = f_box[0], y0 = f_box[1], bw = any ? f_box[2] - f_box[0] : 0, bh = any ? f_box[3] - f_box[1] : 0;
  const int area = bw * bh;
  if (area > kMeCap) return false;      // the caller spreads such a job over many workgroups (me_score_box_wg / me_argmin_part)
  // ---- every position of the union's bounding box is scored, whether an ellipse visits it or not: finding out which ones
  // are visited (a pass of all ellipses over their boxes that stamped them) cost 9 us of a 32 us job, scoring the unvisited
  // third of at most kMeCap positions costs ~1 (scripts/me_trace.py); nobody reads a score outside its ellipse.
  METR(1);
  {
    // the image under the union (+ 5 pixels all round) goes to LDS once: the 121 taps of a position then cost no memory
    // round trips (a thread scores one or two positions; from memory its eleven rows were eleven dependent round trips each)
    const int Sg0 = f_sums[0], Sg0sq = f_sums[1];
    const float rcpw = 1.0f / (float)bw;
    const int iw = bw + 10, ih = bh + 10;
    const bool tile = area > 0 && iw * ih <= kMeImgCap;      // (area == 0: a job none of whose ellipses has a valid box - x0 / y0 are unset)
    if (tile) {
      const float rcpi = 1.0f / (float)iw;
      for (int i = tid; i < iw * ih; i += nthr) {
        int r, q;
        box_divmod(i, iw, rcpi, &r, &q);
        f_img[i] = img[(size_t)(y0 - 5 + r) * width + (x0 - 5 + q)];
      }
      __syncthreads();
    }
    METR(2);
    for (int idx = tid; idx < area; idx += nthr) {
      int r, q;
      box_divmod(idx, bw, rcpw, &r, &q);
      f_score[idx] = tile ? me_score_position_lds(f_img, iw, f_tpl, Sg0, Sg0sq, q + 5, r + 5)
                          : me_score_position(img, width, f_patch, Sg0, Sg0sq, x0 + q, y0 + r);
    }
  }
  __syncthreads();
  METR(3);
  // ---- per-ellipse arg-min: smallest score, among equals the LARGEST scan-order index (u outer, v inner; cpp:151-185).
  // A lane meets its candidates in increasing scan order (its column, or columns 64 apart, rows upwards), so "corr <=
  // best: take it" is the whole rule inside a lane (a first candidate above 1e6 is not taken: "corr <= corrmax", cpp:156);
  // the order index only decides between lanes.
  // Two ellipses per wavefront, a half each, when both boxes are at most 32 columns wide (they nearly always are: ~20 x 15):
  // fetching an ellipse's record, forming its column terms and the cross-lane reduction are as long as the walk itself, and
  // the two halves share them.
  for (int e0 = 2 * wave; e0 < n_ell; e0 += 2 * nwave) {
    const int half = lane >> 5, l32 = lane & 31;
    const bool have = e0 + half < n_ell;
    const int e = have ? e0 + half : e0;
    const int* d = f_desc + 8 * e;
    const int nu = d[3], nv = d[5];
    if (!__any(nu > 32)) {
      double best = 1000000.0;   // cpp:156
      int order = -1;
      if (have && nu > 0 && nv > 0) {
        const double* pu = f_pu + 3 * e;
        const int bx = d[0] + d[2] - x0, by = d[1] + d[4] - y0;
        me_for_each_inside(pu[0], pu[1], pu[2], d[2], nu, d[4], nv, l32, [&](int q, int r) { return f_score[(by + r) * bw + bx + q]; },
                           [&](double corr, int q, int r) { if (corr <= best) { best = corr; order = q * nv + r; } }, 0, 0x7fffffff, 32);
      }
      for (int off = 16; off > 0; off >>= 1) {              // (stays inside the halves)
        const double ob = __shfl_xor(best, off, 64);
        const int oo = __shfl_xor(order, off, 64);
        if (oo >= 0 && (order < 0 || ob < best || (ob == best && oo > order))) { best = ob; order = oo; }
      }
      if (l32 == 0 && have)
        emit(e, (order >= 0 && !(best > kCorrThresh2)) ? 1 : 0, order >= 0 ? d[0] + d[2] + order / nv : 0,
             order >= 0 ? d[1] + d[4] + order % nv : 0, best);
      continue;
    }
    for (int h = 0; h < 2 && e0 + h < n_ell; ++h) {          // a wide box: the whole wavefront, one ellipse after the other
      const int ee = e0 + h;
      const int* dd = f_desc + 8 * ee;
      const int nuu = dd[3], nvv = dd[5];
      double best = 1000000.0;
      int order = -1;
      if (nuu > 0 && nvv > 0) {
        const double* pu = f_pu + 3 * ee;
        const int bx = dd[0] + dd[2] - x0, by = dd[1] + dd[4] - y0;
        me_for_each_inside(pu[0], pu[1], pu[2], dd[2], nuu, dd[4], nvv, lane, [&](int q, int r) { return f_score[(by + r) * bw + bx + q]; },
                           [&](double corr, int q, int r) { if (corr <= best) { best = corr; order = q * nvv + r; } });
      }
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, 64);
        const int oo = __shfl_xor(order, off, 64);
        if (oo >= 0 && (order < 0 || ob < best || (ob == best && oo > order))) { best = ob; order = oo; }
      }
      if (lane == 0)
        emit(ee, (order >= 0 && !(best > kCorrThresh2)) ? 1 : 0, order >= 0 ? dd[0] + dd[2] + order / nvv : 0,
             order >= 0 ? dd[1] + dd[4] + order % nvv : 0, best);
    }
  }
  METR(4);
  return true;
}

// ---------------------------------------------------------------------------
// The jobs me_search_fused_wg declined (union beyond kMeCap positions), spread over many workgroups: the three passes through
// the image-sized maps.  `Jobs` describes where a job's data lives (the engine's particle records / the arrays of the
// stateless operator); list / count = the declined jobs of this step.  Fixed small grids that stride over the list - a
// step has a handful of such jobs, usually none.
//   Jobs: img(job), patch(job), desc(job), n_ell(job), pu(job, e), map(job), emit(job, e, flag, u, v, best)
// ---------------------------------------------------------------------------
template <typename Jobs>
__global__ void __launch_bounds__(256) k_me_big_scores(Jobs J, const int* __restrict__ list, const int* __restrict__ count, int width) {
  const int n = *count;
  for (int li = blockIdx.y; li < n; li += gridDim.y) {
    const int job = list[li];
    me_score_box_wg(J.img(job), width, J.patch(job), J.desc(job), J.n_ell(job), J.map(job), blockIdx.x, gridDim.x);
    __syncthreads();
  }
}
constexpr int kMeBigArgWaves = 16;      // wavefronts per ellipse (a step has a handful of such jobs on an idle chip: a frame-sized box is 240 rows)
template <typename Jobs>
__global__ void __launch_bounds__(64 * kMeBigArgWaves) k_me_big_argmin(Jobs J, const int* __restrict__ list, const int* __restrict__ count, int width) {
  const int n = *count, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __shared__ int s_ord[kMeBigArgWaves];
  __shared__ double s_bst[kMeBigArgWaves];
  for (int li = blockIdx.y; li < n; li += gridDim.y) {
    const int job = list[li], ne = J.n_ell(job);
    for (int e = blockIdx.x; e < ne; e += gridDim.x) {            // one ellipse per workgroup, its rows split over the waves
      const int* d = J.desc(job) + 8 * (size_t)e;
      double best;
      int order;
      me_argmin_part(width, d, J.pu(job, e), J.map(job), wave, kMeBigArgWaves, &best, &order);
      if (lane == 0) { s_bst[wave] = best; s_ord[wave] = order; }
      __syncthreads();
      if (threadIdx.x == 0) {
        for (int w = 1; w < kMeBigArgWaves; ++w)
          if (s_ord[w] >= 0 && (order < 0 || s_bst[w] < best || (s_bst[w] == best && s_ord[w] > order))) { best = s_bst[w]; order = s_ord[w]; }
        const int nv = d[5];
        J.emit(job, e, (order >= 0 && !(best > kCorrThresh2)) ? 1 : 0, order >= 0 ? d[0] + d[2] + order / nv : 0,
               order >= 0 ? d[1] + d[4] + order % nv : 0, best);
      }
      __syncthreads();
    }
  }
}
constexpr int kMeBigGridX = 128, kMeBigGridY = 8, kMeBigSlices = 64;
template <typename Jobs>
inline void me_big_launch(Jobs J, const int* list, const int* count, int width, hipStream_t st) {
  hipLaunchKernelGGL(k_me_big_scores<Jobs>, dim3(kMeBigSlices, kMeBigGridY), dim3(256), 0, st, J, list, count, width);
  hipLaunchKernelGGL(k_me_big_argmin<Jobs>, dim3(kMeBigGridX, kMeBigGridY), dim3(64 * kMeBigArgWaves), 0, st, J, list, count, width);
}

}  // namespace sl2
