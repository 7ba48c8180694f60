// JPEG (sequential and progressive DCT) -> 8-bit grey, host only: the third container of sl2_read_image (sl2_ingest.hip).
//
// The reference decodes frames with cv::imread(path, 0) (framegrabber/filegrabber.cpp:106-109); for a JPEG that is libjpeg
// with out_color_space = JCS_GRAYSCALE: the luminance component of a YCbCr file (or the only component of a grey one) is
// delivered as it comes out of the inverse DCT - no colour conversion, no chroma involved - and libjpeg's default inverse DCT
// is the "slow but accurate" integer one (jpeg_idct_islow).  That is restated here from the published algorithm (Loeffler,
// Ligtenberg and Moschytz, 13-bit constants, two passes with 2 extra bits between them), so the bytes are libjpeg's, not an
// approximation of them: tests/test_ingest.py compares against an independent libjpeg build (Pillow, draft('L')).
// Decoded: sequential DCT (SOF0 / SOF1) and progressive DCT (SOF2: spectral selection and successive approximation, the
// luminance coefficients collected over all scans and reconstructed once at the end - what libjpeg delivers for a complete
// file), 8 bits, Huffman, one or three components in any scan arrangement, any sampling factors as long as the luminance
// component has the largest, restart intervals, 8- or 16-bit quantisation tables.
// Refused with an error, not guessed: lossless processes, arithmetic coding, 12-bit samples, four components (CMYK / YCCK), a
// subsampled first component.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace sl2 {
namespace jpeg {

static const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {
  bool defined = false;
  int mincode[17], maxcode[18], valptr[17];
  uint8_t vals[256];
  bool build(const uint8_t counts[16], const uint8_t* symbols, int nsym) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
      valptr[l] = k;
      mincode[l] = code;
      code += counts[l - 1];
      k += counts[l - 1];
      maxcode[l] = counts[l - 1] ? code - 1 : -1;
      if (code > (1 << l)) return false;                 // more codes of this length than the prefix property allows
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    if (k != nsym || nsym > 256) return false;
    memcpy(vals, symbols, nsym);
    defined = true;
    return true;
  }
};

struct BitReader {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc = 0;
  int nbits = 0;
  bool hit_marker = false;
  void fill() {
    while (nbits <= 24) {
      int b = 0;
      if (!hit_marker && p < end) {
        b = *p++;
        if (b == 0xFF) {
          if (p < end && *p == 0x00) ++p;                 // a stuffed zero: the data byte is 0xFF
          else { --p; hit_marker = true; b = 0; }         // a marker: the entropy-coded segment ends here, feed zeros
        }
      }
      acc |= (uint32_t)b << (24 - nbits);
      nbits += 8;
    }
  }
  int bit() {
    if (nbits == 0) fill();
    const int v = (int)(acc >> 31);
    acc <<= 1; --nbits;
    return v;
  }
  int bits(int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) v = (v << 1) | bit();
    return v;
  }
  void restart() { acc = 0; nbits = 0; hit_marker = false; }
};

static inline int decode_symbol(BitReader& br, const Huff& h) {
  int code = 0;
  for (int l = 1; l <= 16; ++l) {
    code = (code << 1) | br.bit();
    if (h.maxcode[l] >= 0 && code <= h.maxcode[l] && code >= h.mincode[l]) return h.vals[h.valptr[l] + code - h.mincode[l]];
  }
  return -1;
}
static inline int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }

// jpeg_idct_islow: dequantised coefficients in natural order -> 8 x 8 samples.  The zero-AC shortcuts of the original are
// left out: they produce the same values as the full expressions.
static inline void idct_islow(const int* in, uint8_t* out, int stride) {
  constexpr int CB = 13, P1 = 2;
  constexpr int F_0_298631336 = 2446, F_0_390180644 = 3196, F_0_541196100 = 4433, F_0_765366865 = 6270, F_0_899976223 = 7373,
                F_1_175875602 = 9633, F_1_501321110 = 12299, F_1_847759065 = 15137, F_1_961570560 = 16069,
                F_2_053119869 = 16819, F_2_562915447 = 20995, F_3_072711026 = 25172;
  auto descale = [](long long x, int n) { return (int)((x + ((long long)1 << (n - 1))) >> n); };
  int ws[64];
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < 8; ++i) {
      long long c[8];
      for (int k = 0; k < 8; ++k) c[k] = pass == 0 ? in[k * 8 + i] : ws[i * 8 + k];      // pass 1: column i; pass 2: row i
      long long z2 = c[2], z3 = c[6];
      long long z1 = (z2 + z3) * F_0_541196100;
      long long tmp2 = z1 + z3 * (-F_1_847759065);
      long long tmp3 = z1 + z2 * F_0_765366865;
      long long tmp0 = (c[0] + c[4]) << CB;
      long long tmp1 = (c[0] - c[4]) << CB;
      const long long tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
      tmp0 = c[7]; tmp1 = c[5]; tmp2 = c[3]; tmp3 = c[1];
      z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
      long long z4 = tmp1 + tmp3;
      const long long z5 = (z3 + z4) * F_1_175875602;
      tmp0 *= F_0_298631336; tmp1 *= F_2_053119869; tmp2 *= F_3_072711026; tmp3 *= F_1_501321110;
      z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
      z3 += z5; z4 += z5;
      tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
      const long long o[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
      if (pass == 0) {
        for (int k = 0; k < 8; ++k) ws[k * 8 + i] = descale(o[k], CB - P1);
      } else {
        for (int k = 0; k < 8; ++k) {
          int v = descale(o[k], CB + P1 + 3) + 128;
          out[i * stride + k] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
      }
    }
  }
}

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0; };

// Returns false with *err set.  px: width * height bytes, row-major.
inline bool decode_grey(const std::vector<uint8_t>& file, std::vector<uint8_t>& px, int* w, int* h, std::string* err, size_t max_pixels) {
  auto fail = [&](const char* m) { *err = m; return false; };
  const uint8_t* d = file.data();
  const size_t n = file.size();
  if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail("not a JPEG file");
  uint16_t qt[4][64];
  bool qt_def[4] = {false, false, false, false};
  Huff hdc[4], hac[4];
  Component comp[3];
  int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart_interval = 0;
  bool have_frame = false, y_done = false, progressive = false;
  int scans = 0;
  std::vector<uint8_t> plane;                      // the first component, padded to whole MCUs
  std::vector<int> ycoef;                          // progressive: the first component's coefficients (natural order, before dequantisation)
  int pw = 0, ph = 0;
  size_t pos = 2;
  while (pos + 4 <= n) {
    if (d[pos] != 0xFF) { ++pos; continue; }       // (fill bytes / garbage between segments)
    const int m = d[pos + 1];
    if (m == 0xFF) { ++pos; continue; }
    pos += 2;
    if (m == 0xD9) break;                          // EOI
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    if (pos + 2 > n) return fail("truncated JPEG segment");
    const size_t len = ((size_t)d[pos] << 8) | d[pos + 1];
    if (len < 2 || pos + len > n) return fail("truncated JPEG segment");
    const uint8_t* s = d + pos + 2;
    const size_t sl = len - 2;
    if (m == 0xDB) {                               // DQT
      size_t k = 0;
      while (k < sl) {
        const int pq = s[k] >> 4, tq = s[k] & 15;
        ++k;
        if (tq > 3 || pq > 1 || k + (pq ? 128 : 64) > sl) return fail("bad JPEG quantisation table");
        for (int i = 0; i < 64; ++i) {
          qt[tq][kZigzag[i]] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k];
          k += pq ? 2 : 1;
        }
        qt_def[tq] = true;
      }
    } else if (m == 0xC4) {                        // DHT
      size_t k = 0;
      while (k + 17 <= sl) {
        const int tc = s[k] >> 4, th = s[k] & 15;
        int nsym = 0;
        for (int i = 0; i < 16; ++i) nsym += s[k + 1 + i];
        if (tc > 1 || th > 3 || k + 17 + nsym > sl) return fail("bad JPEG Huffman table");
        if (!(tc ? hac[th] : hdc[th]).build(s + k + 1, s + k + 17, nsym)) return fail("bad JPEG Huffman table");
        k += 17 + nsym;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {   // SOF0 / SOF1: sequential DCT, SOF2: progressive DCT; Huffman
      if (sl < 6) return fail("bad JPEG frame header");
      if (s[0] != 8) return fail("JPEG: only 8-bit samples are decoded");
      H = (s[1] << 8) | s[2]; W = (s[3] << 8) | s[4]; ncomp = s[5];
      if (ncomp != 1 && ncomp != 3) return fail("JPEG: only one- or three-component files are decoded (no CMYK / YCCK)");
      if (W <= 0 || H <= 0 || (size_t)W * H > max_pixels || sl < (size_t)6 + 3 * ncomp) return fail("bad JPEG frame header");
      for (int i = 0; i < ncomp; ++i) {
        comp[i].id = s[6 + 3 * i]; comp[i].h = s[7 + 3 * i] >> 4; comp[i].v = s[7 + 3 * i] & 15; comp[i].tq = s[8 + 3 * i];
        if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) return fail("bad JPEG frame header");
        if (comp[i].h > hmax) hmax = comp[i].h;
        if (comp[i].v > vmax) vmax = comp[i].v;
      }
      if (comp[0].h != hmax || comp[0].v != vmax) return fail("JPEG: a subsampled first component is not decoded");
      pw = (W + 8 * hmax - 1) / (8 * hmax) * 8 * hmax;
      ph = (H + 8 * vmax - 1) / (8 * vmax) * 8 * vmax;
      plane.assign((size_t)pw * ph, 0);
      progressive = (m == 0xC2);
      if (progressive) ycoef.assign((size_t)pw * ph, 0);            // 64 coefficients per 8 x 8 block = one per sample
      have_frame = true;
    } else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return fail("JPEG: only the Huffman DCT processes (SOF0 / SOF1 / SOF2) are decoded, not lossless / arithmetic");
    } else if (m == 0xDD) {                        // DRI
      if (sl < 2) return fail("bad JPEG restart interval");
      restart_interval = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {                        // SOS + entropy-coded data
      if (!have_frame) return fail("JPEG scan before the frame header");
      if (sl < 1) return fail("bad JPEG scan header");
      // every scan walks the MCU grid: a crafted file of thousands of empty scans would cost scans x pixels (libjpeg's own
      // progressive scripts use about ten; 3 components x (1 DC + 63 AC bands) x 14 bit positions bounds any legal file)
      if (++scans > 1024) return fail("JPEG: too many scans");
      const int ns = s[0];
      if (ns < 1 || ns > ncomp || sl < (size_t)1 + 2 * ns + 3) return fail("bad JPEG scan header");
      int sc[3];
      for (int i = 0; i < ns; ++i) {
        int ci = -1;
        for (int c = 0; c < ncomp; ++c) if (comp[c].id == s[1 + 2 * i]) ci = c;
        if (ci < 0) return fail("bad JPEG scan header");
        comp[ci].td = s[2 + 2 * i] >> 4; comp[ci].ta = s[2 + 2 * i] & 15;
        if (comp[ci].td > 3 || comp[ci].ta > 3 || !qt_def[comp[ci].tq]) return fail("JPEG scan refers to an undefined table");
        comp[ci].pred = 0;
        sc[i] = ci;
      }
      const int Ss = s[1 + 2 * ns], Se = s[2 + 2 * ns], Ah = s[3 + 2 * ns] >> 4, Al = s[3 + 2 * ns] & 15;
      BitReader br;
      br.p = d + pos + len;
      br.end = d + n;
      auto next_marker = [&]() {                     // behind the entropy-coded segment: the next marker that is not RSTn / a stuffed byte
        const uint8_t* q = br.p;
        while (q + 1 < br.end && !(q[0] == 0xFF && q[1] != 0x00 && q[1] != 0xFF && !(q[1] >= 0xD0 && q[1] <= 0xD7))) ++q;
        return (size_t)(q - d);
      };
      if (progressive) {
        // ---- one scan of a progressive file (ITU T.81 annex G): a DC scan (Ss = Se = 0, possibly interleaved) or an AC band
        // of ONE component; Ah = 0: first pass at bit position Al, Ah = Al + 1: one more bit of every coefficient
        if (Ss > Se || Se > 63 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1) || (Ah != 0 && Ah != Al + 1) || Al > 13)
          return fail("bad progressive JPEG scan header");
        if (Ss > 0 && sc[0] != 0) { pos = next_marker(); continue; }       // an AC band of a chroma component: not needed for grey
        for (int i = 0; i < ns; ++i) {
          if (Ss == 0 && Ah == 0 && !hdc[comp[sc[i]].td].defined) return fail("JPEG scan refers to an undefined table");
          if (Ss > 0 && !hac[comp[sc[i]].ta].defined) return fail("JPEG scan refers to an undefined table");
        }
        int mcux, mcuy;
        if (ns > 1) { mcux = pw / (8 * hmax); mcuy = ph / (8 * vmax); }
        else {
          const Component& c = comp[sc[0]];
          const int cw = (W * c.h + hmax - 1) / hmax, chh = (H * c.v + vmax - 1) / vmax;
          mcux = (cw + 7) / 8; mcuy = (chh + 7) / 8;
        }
        const int bpr = pw / 8;                        // blocks per row of the stored (first) component
        int since_restart = 0, next_rst = 0, eobrun = 0;
        const int p1 = 1 << Al, m1 = -(1 << Al);
        for (int my = 0; my < mcuy; ++my)
          for (int mx = 0; mx < mcux; ++mx) {
            if (restart_interval && since_restart == restart_interval) {
              const uint8_t* q = br.p;
              while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
              if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) return fail("JPEG restart marker missing");
              br.p = q + 2;
              br.restart();
              next_rst = (next_rst + 1) & 7;
              since_restart = 0;
              eobrun = 0;
              for (int i = 0; i < ns; ++i) comp[sc[i]].pred = 0;
            }
            for (int i = 0; i < ns; ++i) {
              Component& c = comp[sc[i]];
              const int bh = ns > 1 ? c.h : 1, bv = ns > 1 ? c.v : 1;
              for (int by = 0; by < bv; ++by)
                for (int bx = 0; bx < bh; ++bx) {
                  const int BX = ns > 1 ? mx * c.h + bx : mx, BY = ns > 1 ? my * c.v + by : my;
                  int* blk = (sc[i] == 0 && BX < bpr && BY < ph / 8) ? ycoef.data() + ((size_t)BY * bpr + BX) * 64 : nullptr;
                  if (Ss == 0) {
                    if (Ah == 0) {                     // DC, first pass: the difference as in a sequential file, scaled by 2^Al
                      const int t = decode_symbol(br, hdc[c.td]);
                      if (t < 0 || t > 11) return fail("corrupt JPEG data (DC)");
                      c.pred += extend(br.bits(t), t);
                      if (c.pred > 32767 || c.pred < -32768) return fail("corrupt JPEG data (DC range)");
                      if (blk) blk[0] = c.pred * (1 << Al);
                    } else {                           // DC, refinement: one bit
                      if (br.bit() && blk) blk[0] |= p1;
                    }
                    continue;
                  }
                  int scratch[64];
                  if (!blk) { memset(scratch, 0, sizeof(scratch)); blk = scratch; }     // (a block outside the stored plane: decoded and dropped)
                  int k = Ss;
                  if (Ah == 0) {                       // AC, first pass (figure G.3 ff.): runs of zeros, values, end-of-band runs over blocks
                    if (eobrun > 0) { --eobrun; continue; }
                    while (k <= Se) {
                      const int rs = decode_symbol(br, hac[c.ta]);
                      if (rs < 0) return fail("corrupt JPEG data (AC)");
                      const int r = rs >> 4, sz = rs & 15;
                      if (sz == 0) {
                        if (r == 15) { k += 16; continue; }
                        eobrun = (1 << r) - 1;
                        if (r) eobrun += br.bits(r);
                        break;
                      }
                      k += r;
                      if (k > Se) return fail("corrupt JPEG data (run)");
                      blk[kZigzag[k]] = extend(br.bits(sz), sz) * (1 << Al);
                      ++k;
                    }
                  } else {                             // AC, refinement (figure G.7): a correction bit for every coefficient that
                    // is already non-zero, new coefficients of magnitude 2^Al in between
                    auto refine = [&](int& coefv) {
                      if (br.bit() && (coefv & p1) == 0) coefv += coefv >= 0 ? p1 : m1;
                    };
                    if (eobrun == 0) {
                      while (k <= Se) {
                        const int rs = decode_symbol(br, hac[c.ta]);
                        if (rs < 0) return fail("corrupt JPEG data (AC)");
                        int r = rs >> 4;
                        const int sz = rs & 15;
                        int newv = 0;
                        if (sz == 0) {
                          if (r != 15) {
                            eobrun = 1 << r;
                            if (r) eobrun += br.bits(r);
                            break;                     // (the rest of this block is refined below, as the first block of the run)
                          }
                        } else {
                          if (sz != 1) return fail("corrupt JPEG data (refinement)");
                          newv = br.bit() ? p1 : m1;
                        }
                        while (k <= Se) {              // pass over r zero coefficients, refining the non-zero ones on the way
                          int& cv = blk[kZigzag[k]];
                          if (cv != 0) refine(cv);
                          else { if (r == 0) break; --r; }
                          ++k;
                        }
                        if (newv) { if (k > Se) return fail("corrupt JPEG data (run)"); blk[kZigzag[k]] = newv; }
                        ++k;
                      }
                    }
                    if (eobrun > 0) {
                      for (; k <= Se; ++k) { int& cv = blk[kZigzag[k]]; if (cv != 0) refine(cv); }
                      --eobrun;
                    }
                  }
                }
            }
            ++since_restart;
          }
        if (Ss == 0) for (int i = 0; i < ns; ++i) if (sc[i] == 0) y_done = true;
        pos = next_marker();
        continue;
      }
      for (int i = 0; i < ns; ++i)
        if (!hdc[comp[sc[i]].td].defined || !hac[comp[sc[i]].ta].defined) return fail("JPEG scan refers to an undefined table");
      if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) return fail("bad sequential JPEG scan header");
      // MCU geometry: interleaved (every component's h x v blocks per MCU) or a single component block by block
      int mcux, mcuy;
      if (ns > 1) { mcux = pw / (8 * hmax); mcuy = ph / (8 * vmax); }
      else {
        const Component& c = comp[sc[0]];
        const int cw = (W * c.h + hmax - 1) / hmax, chh = (H * c.v + vmax - 1) / vmax;
        mcux = (cw + 7) / 8; mcuy = (chh + 7) / 8;
      }
      int coef[64], since_restart = 0, next_rst = 0;
      for (int my = 0; my < mcuy; ++my)
        for (int mx = 0; mx < mcux; ++mx) {
          if (restart_interval && since_restart == restart_interval) {
            // byte-align, expect RSTn
            const uint8_t* q = br.p;
            while (q + 1 < br.end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
            if (q + 1 >= br.end || q[1] != 0xD0 + next_rst) return fail("JPEG restart marker missing");
            br.p = q + 2;
            br.restart();
            next_rst = (next_rst + 1) & 7;
            since_restart = 0;
            for (int i = 0; i < ns; ++i) comp[sc[i]].pred = 0;
          }
          for (int i = 0; i < ns; ++i) {
            Component& c = comp[sc[i]];
            const int bh = ns > 1 ? c.h : 1, bv = ns > 1 ? c.v : 1;
            for (int by = 0; by < bv; ++by)
              for (int bx = 0; bx < bh; ++bx) {
                memset(coef, 0, sizeof(coef));
                const int t = decode_symbol(br, hdc[c.td]);
                if (t < 0 || t > 11) return fail("corrupt JPEG data (DC)");
                c.pred += extend(br.bits(t), t);
                if (c.pred > 32767 || c.pred < -32768) return fail("corrupt JPEG data (DC range)");
                coef[0] = c.pred * qt[c.tq][0];
                for (int k = 1; k < 64;) {
                  const int rs = decode_symbol(br, hac[c.ta]);
                  if (rs < 0) return fail("corrupt JPEG data (AC)");
                  const int r = rs >> 4, sz = rs & 15;
                  if (sz == 0) {
                    if (r == 15) { k += 16; continue; }
                    break;
                  }
                  k += r;
                  if (k > 63) return fail("corrupt JPEG data (run)");
                  coef[kZigzag[k]] = extend(br.bits(sz), sz) * qt[c.tq][kZigzag[k]];
                  ++k;
                }
                if (sc[i] == 0) {                    // only the first component is reconstructed
                  const int X = (ns > 1 ? mx * c.h + bx : mx) * 8, Y = (ns > 1 ? my * c.v + by : my) * 8;
                  if (X + 8 <= pw && Y + 8 <= ph) idct_islow(coef, plane.data() + (size_t)Y * pw + X, pw);
                }
              }
          }
          ++since_restart;
        }
      for (int i = 0; i < ns; ++i) if (sc[i] == 0) y_done = true;
      pos = next_marker();
      continue;
    }
    pos += len;
  }
  if (!have_frame || !y_done) return fail("JPEG without a decodable luminance scan");
  if (progressive) {                                 // every scan is in: dequantise and reconstruct the first component
    int coef[64];
    const int bpr = pw / 8;
    for (int by = 0; by < ph / 8; ++by)
      for (int bx = 0; bx < bpr; ++bx) {
        const int* blk = ycoef.data() + ((size_t)by * bpr + bx) * 64;
        for (int i = 0; i < 64; ++i) coef[i] = blk[i] * qt[comp[0].tq][i];
        idct_islow(coef, plane.data() + (size_t)by * 8 * pw + bx * 8, pw);
      }
  }
  px.resize((size_t)W * H);
  for (int y = 0; y < H; ++y) memcpy(px.data() + (size_t)y * W, plane.data() + (size_t)y * pw, W);
  *w = W; *h = H;
  return true;
}

}  // namespace jpeg
}  // namespace sl2
