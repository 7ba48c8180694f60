// Elliptical normalised-SSD patch search - MonoSLAM::elliptical_search
// (monoslam.cpp:401-477) + correlate2_warning (improc/improc.cpp:55-134).
//
// One 64-lane wavefront per search.  Integer sums are exact int32; the deciding score is the reference's FP64
// expression (ncc_score in sl2_math.hpp, -ffp-contract=off) so the arg-min, the "<=" last-candidate-wins tie rule (Q2),
// the sigma >= 10 tests (Q3) and the 0.40 threshold are decided on bit-identical numbers.
//
// Two search cores:
//   search_core_v0    ("exact"): lane = candidate, 121-pixel loop straight from global / L2, full FP64 epilogue per
//                     candidate.  The simple, obviously faithful kernel; also the fallback of the matrix-core walk.
//   search_core_mfma  ("matrix-core walk", production): the three 11x11 window sums of 256 candidates at a time come out
//                     of the int8 matrix cores; candidates are ranked in FP32 from the exact integers and the unique
//                     near-best one is scored in FP64; whatever that cannot decide exactly goes to search_core_v0.
// (The column-walk kernels of rounds 1-2 - one feature per wavefront, and several features packed into one - and the
// first matrix-core kernel were removed in round 3; profiles/r03_search_v3_v4_* holds the last side-by-side run.)
#include "sl2_common.hpp"
#include "sl2_score_dev.hpp"

namespace sl2 {

struct SearchResult {
  int code;   // 0 = resolved here (exact), 1 = unique winner, exact score deferred, -1 = caller must fall back
  int ok, found, u, v, ncand;
  int S1, S2, X;  // integer sums of the deferred winner
  double score;
};

// Wave-wide arg-min with the reference's sequential semantics: scanning in
// candidate order with "corr <= corrmax" means the winner is the candidate with
// the smallest score and, among equals, the LARGEST order index.
__device__ __forceinline__ void wave_argmin(double& best, int& best_order) {
  for (int off = 32; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64);
    const int oo = __shfl_xor(best_order, off, 64);
    if (ob < best || (ob == best && oo > best_order)) { best = ob; best_order = oo; }
  }
}

__device__ __forceinline__ SearchBounds bounds_from_desc(const int* si) {
  SearchBounds sb;
  sb.ucentre = si[0]; sb.vcentre = si[1]; sb.urelstart = si[2]; sb.urelfinish = si[2] + si[3] - 1;
  sb.vrelstart = si[4]; sb.vrelfinish = si[4] + si[5] - 1; sb.halfwidth = si[6]; sb.halfheight = si[7];
  return sb;
}

__device__ __forceinline__ SearchResult search_core_v0(const uint8_t* __restrict__ image, int width,
                                                       const uint8_t* __restrict__ patch, const SearchBounds sb, double a,
                                                       double b, double c) {
  const int lane = threadIdx.x & 63;
  const int nu = sb.urelfinish - sb.urelstart + 1;
  const int nv = sb.vrelfinish - sb.vrelstart + 1;
  // template sums (wave-uniform; every lane computes them redundantly)
  int Sg0 = 0, Sg0sq = 0;
  for (int p = 0; p < 121; ++p) { const int g = patch[p]; Sg0 += g; Sg0sq += g * g; }
  double best = 1000000.0;  // corrmax initial value, monoslam.cpp:444
  int best_order = -1;
  int ncand = 0;
  if (nu > 0 && nv > 0) {
    const int total = nu * nv;
    for (int idx = lane; idx < total; idx += 64) {
      const int urel = sb.urelstart + idx / nv;
      const int vrel = sb.vrelstart + idx % nv;
      if (!in_ellipse(a, b, c, urel, vrel)) continue;
      ++ncand;
      const int x1 = sb.ucentre + urel - 5, y1 = sb.vcentre + vrel - 5;
      const uint8_t* p1 = image + (size_t)y1 * width + x1;
      int Sg1 = 0, Sg0g1 = 0, Sg1sq = 0;
      for (int r = 0; r < 11; ++r)
        for (int cc = 0; cc < 11; ++cc) {
          const int g0 = patch[r * 11 + cc];
          const int g1 = p1[r * width + cc];
          Sg1 += g1; Sg0g1 += g0 * g1; Sg1sq += g1 * g1;
        }
      double sd0, sd1;
      const double corr = ncc_score(Sg0, Sg1, Sg0g1, Sg0sq, Sg1sq, &sd0, &sd1);
      // accept rule of monoslam.cpp:457-466; within a lane the scan is in order
      if (corr <= best && !(sd0 < kCorrelationSigmaThreshold) && !(sd1 < kCorrelationSigmaThreshold)) {
        best = corr;
        best_order = idx;
      }
    }
  }
  wave_argmin(best, best_order);
  for (int off = 32; off > 0; off >>= 1) ncand += __shfl_xor(ncand, off, 64);
  SearchResult r;
  r.code = 0;
  r.ncand = ncand;
  r.score = best;
  r.u = 0; r.v = 0; r.S1 = r.S2 = r.X = 0;
  r.found = best_order >= 0;
  if (best_order >= 0) {
    r.u = sb.ucentre + sb.urelstart + best_order / nv;
    r.v = sb.vcentre + sb.vrelstart + best_order % nv;
  }
  r.ok = (best_order >= 0 && !(best > kCorrThresh2)) ? 1 : 0;
  return r;
}

__device__ __forceinline__ unsigned udot4(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// full-rate 24-bit multiply: the patch sums are < 2^24 (S1 <= 121*255, S2, X <= 121*255^2), products < 2^31
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }

// ---------------------------------------------------------------------------
// The matrix-core walk.
//
// For a 16 (rows v) x 16 (columns u) tile of candidates the cross term is a sum of eleven Toeplitz products,
//     X[v][u] = sum_i  I[v+i][0..31] . Tt_i[0..31][u],     Tt_i[k][u] = T[i][k-u]  (0 <= k-u < 11, else 0),
// i.e. a 16 x (11*32) by (11*32) x 16 integer GEMM: six v_mfma_i32_16x16x64_i8 (two template rows per instruction).
// The same A operand against a Toeplitz matrix of ones gives sum(g1); sum(g1^2) uses two more byte planes, the high
// and the low byte of the squared pixel (g^2 = 256 H + L), against the ones matrix.  Pixels and template are offset by
// 128 to fit the signed int8 operands: 24 MFMAs per 256 candidates, and a lane ends up holding the three exact int32
// sums of four candidates (C/D layout: column = lane & 15, row = 4 (lane >> 4) + register).
//
// Which byte of an operand register pairs with which k of the instruction does not matter here: lane group g = lane >> 4
// and byte e of A always meet lane group g, byte e of B, and both operands are built from (g, e) -> (template row
// 2p + (g >> 1), window byte 16 (g & 1) + e).
//
// Candidates are ranked on rho_f = cov / sqrt(var0 var1) in FP32 from the exact integers (error < 1e-6).  Only a
// candidate within 4e-6 of the best can be the reference's winner; if there is exactly one such candidate its integer
// sums are handed to the scoring pass (k_search_score, or scored in place by the stateless kernel), which evaluates the
// reference's FP64 score and thresholds.  Several near-best candidates, or a candidate on the sigma == 10 boundary,
// return code -1 and the caller runs search_core_v0 - same results, just slower.
// ---------------------------------------------------------------------------
#ifdef SL2_SEARCH_TRACE
__device__ long long* g_search_trace = nullptr;     // development only: 16 values per workgroup (scripts/search_trace.py)
#define STR(slot) do { if (g_search_trace && threadIdx.x == 0) g_search_trace[(size_t)blockIdx.x * 16 + (slot)] = (long long)__builtin_readcyclecounter(); } while (0)
// phase accumulators of the position loop: PH(k) adds the cycles since the previous stamp to phase k
#define PH_DECL long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long ph_prev = (long long)__builtin_readcyclecounter()
#define PH(k) do { const long long ph_t = (long long)__builtin_readcyclecounter(); ph_acc[k] += ph_t - ph_prev; ph_prev = ph_t; } while (0)
#define PH_WAIT(what) asm volatile("s_waitcnt " what ::: "memory")
#define PH_STORE do { if (g_search_trace && threadIdx.x == 0) for (int q = 0; q < 8; ++q) g_search_trace[(size_t)blockIdx.x * 16 + 8 + q] = ph_acc[q]; } while (0)
#else
#define STR(slot) do { } while (0)
#define PH_DECL do { } while (0)
#define PH(k) do { } while (0)
#define PH_WAIT(what) do { } while (0)
#define PH_STORE do { } while (0)
#endif

constexpr int kMfPitchDw = 12;                 // 48-byte template rows: 16 zeros + 11 bytes + zeros
// Four copies of the padded template, copy c = the rows moved left by c bytes (round 4).  A lane group of a template read
// (32 lanes: 16 candidate columns x 2 halves of the window bytes) touches dwords 1..8 of copy 0 and 0..7 of copies 1-3 of
// one row, so the copies start at dwords 3, 172, 340, 508 = banks 3, 12, 20, 28 (mod 32): the group's 32 addresses fall on
// 32 different banks.  (With equal strides one bank was used twice: SQ_LDS_BANK_CONFLICT +4.8e6 per launch.)
constexpr int kMfCopyDw = 168;                 // 13 rows x 12 dwords + 12 of padding
__host__ __device__ constexpr int mf_copy_base(int c) { return 3 + c * kMfCopyDw + (c ? 1 : 0); }
constexpr int kMfTplDw = mf_copy_base(3) + kMfCopyDw;          // 676 dwords (a multiple of four: zeroed in 16-byte stores)
static_assert(kMfTplDw % 4 == 0, "s_T is cleared with ds_write_b128");
typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef unsigned short mf_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned mf_sq_pairs(unsigned packed_u16x2) {   // (x, y) -> (x^2, y^2), v_pk_mul_lo_u16
  mf_us2 v;
  __builtin_memcpy(&v, &packed_u16x2, 4);
  v = v * v;
  unsigned r;
  __builtin_memcpy(&r, &v, 4);
  return r;
}

// 16 bytes of padded template row `row` (copy 0) starting at byte `off` (1..32) of its 48-byte LDS row: five aligned dwords
// and four v_alignbyte.  Used for the operands of the ones matrix only (once per wavefront).  (gfx950 runs in
// unaligned-access mode and a single ds_read_b128 at a byte address works but is slow: 0.150 ms against 0.119 ms.)
__device__ __forceinline__ mf_v4i mf_load_b(const unsigned* s_T, int row, int off) {
  mf_v4i r;
  const unsigned* p = s_T + mf_copy_base(0) + row * kMfPitchDw + (off >> 2);
  const int sh = off & 3;
  const unsigned q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
  r.x = (int)__builtin_amdgcn_alignbyte(q1, q0, sh);
  r.y = (int)__builtin_amdgcn_alignbyte(q2, q1, sh);
  r.z = (int)__builtin_amdgcn_alignbyte(q3, q2, sh);
  r.w = (int)__builtin_amdgcn_alignbyte(q4, q3, sh);
  return r;
}
// The template operand of a lane is the 16 bytes at byte `off` = 16 + 16 (g & 1) - j of a padded row: a different
// alignment for every candidate column j.  Rounds 2-3 read five dwords and shifted them with four v_alignbyte per operand
// (24 vector instructions per candidate tile, in a kernel bound by vector issue).  Now the padded rows are kept four
// times, copy c moved left by c bytes, so the operand is four ALIGNED dwords of copy off & 3 (two ds_read2_b32, no vector
// instruction); the copies cost three v_alignbyte and three more ds_write_b32 per feature.
__device__ __forceinline__ const unsigned* mf_b_base(const unsigned* s_T, int g, int off) {
  const int c = off & 3;
  return s_T + (3 + c * kMfCopyDw + (c ? 1 : 0)) + (g >> 1) * kMfPitchDw + (off >> 2);
}
__device__ __forceinline__ mf_v4i mf_load_b4(const unsigned* bp, int p) {     // template rows 2 p + (g >> 1)
  mf_v4i r;
  r.x = (int)bp[24 * p]; r.y = (int)bp[24 * p + 1]; r.z = (int)bp[24 * p + 2]; r.w = (int)bp[24 * p + 3];
  return r;
}

// padded template rows in LDS, copy 0: row r = [16 x 0][T[r][0..10] - 128][21 x 0]; row 11 = zeros; row 12 = ones (copy 0
// only).  mf_tpl_init clears the array, mf_tpl_ones writes the ones row (both once per wavefront), mf_tpl_store the
// 4 x 44 data dwords of a template.  Lane layout of a template in registers (mf_tpl_index): lane 4 r + q holds dword q - 1 of row r
// (q = 0: nothing, the 16 zero bytes in front of the row), lanes 48 / 49 / 50 hold sum g0, sum g0^2 and the patch-sigma
// flag.  Row r of a copy is then one quad of lanes, and the dword that follows a lane's own is its right neighbour's
// (v_mov_dpp row_shl:1; the last lane of a row's quad meets the zero of the next row's q = 0 lane, lane 15 of a DPP row
// reads zero by bound_ctrl).
__device__ __forceinline__ void mf_tpl_init(unsigned* s_T, int lane) {      // (a barrier must follow before mf_tpl_store)
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 z = {0u, 0u, 0u, 0u};
  for (int i = lane; i < kMfTplDw / 4; i += 64) ((u4*)s_T)[i] = z;
}
__device__ __forceinline__ void mf_tpl_ones(unsigned* s_T, int lane) {      // after the barrier that follows mf_tpl_init
  if (lane < 3) s_T[mf_copy_base(0) + 12 * kMfPitchDw + 4 + lane] = (lane == 2) ? 0x00010101u : 0x01010101u;
}
__device__ __forceinline__ int mf_tpl_index(int lane) {       // which dword of the packed record (sl2_common.hpp) a lane loads
  const int r = lane >> 2, q = lane & 3;
  return lane < 44 ? (q ? 3 * r + q - 1 : 0) : (lane >= 48 && lane <= 50 ? 33 + (lane - 48) : 0);
}
__device__ __forceinline__ unsigned mf_tpl_mask(int lane) {   // bytes of the lane's dword that are template pixels
  const int q = lane & 3;
  return lane < 44 ? (q == 0 ? 0u : (q == 3 ? 0x00ffffffu : 0xffffffffu)) : 0u;
}
__device__ __forceinline__ void mf_tpl_store(unsigned tv, unsigned tmask, unsigned* s_T, int lane) {
  const unsigned x = (tv ^ 0x80808080u) & tmask;
  const unsigned nx = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x101 /* row_shl:1 */, 0xf, 0xf, true);
  if (lane < 44) {
    unsigned* o = s_T + (lane >> 2) * kMfPitchDw + 3 + (lane & 3);
    o[mf_copy_base(0)] = x;
    o[mf_copy_base(1)] = __builtin_amdgcn_alignbyte(nx, x, 1);
    o[mf_copy_base(2)] = __builtin_amdgcn_alignbyte(nx, x, 2);
    o[mf_copy_base(3)] = __builtin_amdgcn_alignbyte(nx, x, 3);
  }
}

// Round 4 (profiles/r04_search_*): the band's LDS layout and the four pre-shifted template copies are bank-conflict free
// (SQ_LDS_BANK_CONFLICT 9.07e6 -> 3.0e5 per launch), the template operand costs no vector instruction, the byte-plane
// offsets come back through the accumulators' start values, key and pixel sum share a register, the sigma == 10 boundary
// is checked for the winner only, and the ellipse is classified in FP32 with an exact FP64 fallback: 193 -> 137 vector
// instructions per candidate tile, 332 -> 318 per search (per-search fixed work - staging, byte planes, decision - is now
// the larger part), 0.092 -> 0.0887 ms per launch on the same box.
// ---------------------------------------------------------------------------
// Round 3 re-costed this kernel for vector-instruction issue, which is what bounds it (round 2: 551 vector instructions
// per searched feature with the matrix pipes 23 % busy; now ~380, PMC SQ_INSTS_VALU in profiles/r03_search_*):
//   * the window band is staged in 16-byte pieces (one global_load_dwordx4 and three ds_write_b128 per lane: a 26 x 26
//     byte band is ONE pass of 52 lanes instead of three passes of dwords), the byte planes come out of v_perm_b32;
//   * the Toeplitz operand of the ones matrix is loaded once per wavefront;
//   * the offsets of the signed operands are never taken out per candidate: variance and covariance are shift
//     invariant, so D1 = 121 w - (S1^2 - K) and Nc = 121 x + (15488 - sum g0) s1 come straight from the
//     accumulators (s1 = sum(g - 128), w = 256 H' + L', x = the raw cross term); only the winner's sums are converted;
//   * candidate counts and the sigma == 10 flag live in scalar registers (popcount / or of compare masks: m4_mask below),
//     a tile none of whose 256 slots lies inside the ellipse is skipped, the wave-wide maximum is six DPP instructions,
//     and the winning lane stores its own record.
// ---------------------------------------------------------------------------
// LDS layout of a band (round 4): plane -> 16-byte column chunk (0..2: 32 candidate columns + 10) -> row (32 slots of 16
// bytes, 27 used: 16 candidate rows + 10 + the partner row of template row 10).  An A operand is the 16-byte piece
// (row j + (g >> 1) + 2 p, chunk tile + (g & 1)); ds_read_b128 is serviced in four 16-lane groups ({0-3, 12-15, 20-27},
// ...) over 16 slots of 16 bytes, and with the chunk stride a multiple of 16 slots the lanes of a group - eight rows of
// one chunk, the other eight rows of the next - fall on 16 different slots.  (Rounds 2-3 kept a row's three pieces side
// by side, pitch 48: every group had five slots used twice - SQ_LDS_BANK_CONFLICT 9.1e6 per launch, 2.3 per LDS instruction,
// profiles/r03_final_pmc_summary.txt.)  The band is staged chunk-major too (consecutive lanes = consecutive rows of one
// chunk), so the eight lanes of a ds_write_b128 group write eight different slots.
constexpr int kM4RowSlots = 32;                  // 16-byte slots per chunk column
constexpr int kM4Chunk = kM4RowSlots * 16;       // bytes per chunk column
constexpr int kM4Plane = 3 * kM4Chunk;           // bytes per plane
typedef unsigned m4_u4 __attribute__((ext_vector_type(4)));

struct M4Band { int base, rows_needed, nchunk, ntask; };
__device__ __forceinline__ M4Band m4_band(int ucentre, int vcentre, int urelstart, int vrelstart, int nu_all, int nv_all, int up,
                                          int vt, int width) {
  M4Band bd;
  bd.base = (vcentre + vrelstart + 16 * vt - 5) * width + (ucentre + urelstart + 16 * up - 5);   // < 2^31
  bd.rows_needed = min(nv_all - 16 * vt, 16) + 10;             // <= 26
  const int bytes_needed = min(nu_all - 16 * up, 32) + 10;     // 11 .. 42
  bd.nchunk = (bytes_needed + 15) >> 4;                        // 1 .. 3
  bd.ntask = bd.rows_needed * bd.nchunk;                       // <= 78
  return bd;
}
// the band's 16-byte pieces in flight: two passes of 64 lanes; lds = byte offset of the piece inside a plane, -1 = this
// lane has no piece in that pass, bit 30 = the piece runs past the end of the frame (see m4_band_fix)
struct M4Pf { m4_u4 v[2]; int lds[2]; };
__device__ __forceinline__ bool m4_band_loads(const uint8_t* __restrict__ img, int width, int frame_bytes, const M4Band bd,
                                              int lane, M4Pf& pf) {
  bool over = false;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if (ps * 64 < bd.ntask) {                                  // wave-uniform
      if (ps) asm volatile("" ::: "memory");                   // (a real branch: the second pass is the exception, not to be speculated)
      const int idx = ps * 64 + lane;
      const int chunk = (idx >= bd.rows_needed ? 1 : 0) + (idx >= 2 * bd.rows_needed ? 1 : 0);     // chunk-major: idx = chunk * rows + row
      const int row = idx - mul24(chunk, bd.rows_needed);
      const bool valid = idx < bd.ntask;
      const int off = bd.base + mul24(row, width) + 16 * chunk;
      // A piece may reach up to 37 bytes beyond its window row; inside the frame that is harmless (the bytes land in
      // columns no candidate uses), but the last image row must not be over-read: the address is clamped and the piece is
      // re-read byte by byte by m4_band_fix (windows that touch the last image row only).
      const bool ov = valid && off > frame_bytes - 16;
      over |= ov;
      __builtin_memcpy(&pf.v[ps], img + min(off, frame_bytes - 16), 16);
      pf.lds[ps] = valid ? ((16 * row + kM4Chunk * chunk) | (ov ? 0x40000000 : 0)) : -1;
    }
  }
  return __any(over);
}
// a piece that would run past the end of the frame, read byte by byte (bytes beyond the frame are zero: they belong to
// columns no candidate uses)
__device__ __noinline__ m4_u4 m4_piece_bytes(const uint8_t* __restrict__ img, int off, int frame_bytes) {
  unsigned w[4] = {0u, 0u, 0u, 0u};
  for (int e = 0; e < 16; ++e)
    if (off + e < frame_bytes) w[e >> 2] |= (unsigned)img[off + e] << (8 * (e & 3));
  m4_u4 r;
  r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3];
  return r;
}
__device__ __forceinline__ void m4_band_fix(const uint8_t* __restrict__ img, int width, int frame_bytes, const M4Band bd, int lane,
                                            M4Pf& pf) {
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if (ps * 64 < bd.ntask && pf.lds[ps] >= 0 && (pf.lds[ps] & 0x40000000)) {
      const int idx = ps * 64 + lane;
      const int chunk = (idx >= bd.rows_needed ? 1 : 0) + (idx >= 2 * bd.rows_needed ? 1 : 0);
      const int row = idx - chunk * bd.rows_needed;
      pf.v[ps] = m4_piece_bytes(img, bd.base + row * width + 16 * chunk, frame_bytes);
      pf.lds[ps] &= ~0x40000000;
    }
  }
}
// four pixels -> the three signed byte planes (g - 128, high and low byte of g^2, each - 128)
__device__ __forceinline__ void m4_planes(unsigned d, unsigned& I, unsigned& H, unsigned& L) {
  const unsigned lo = d & 0x00ff00ffu;                                  // pixels 0, 2 as 16-bit lanes
  const unsigned hi = __builtin_amdgcn_perm(0u, d, 0x0c030c01u);        // pixels 1, 3
  const unsigned sqlo = mf_sq_pairs(lo), sqhi = mf_sq_pairs(hi);
  H = __builtin_amdgcn_perm(sqhi, sqlo, 0x07030501u) ^ 0x80808080u;
  L = __builtin_amdgcn_perm(sqhi, sqlo, 0x06020400u) ^ 0x80808080u;
  I = d ^ 0x80808080u;
}
__device__ __forceinline__ void m4_band_store(const M4Pf& pf, const M4Band bd, char* s_pl) {
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    if (ps * 64 < bd.ntask && pf.lds[ps] >= 0) {
      m4_u4 I, H, L;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unsigned iq, hq, lq;
        m4_planes(pf.v[ps][q], iq, hq, lq);
        I[q] = iq; H[q] = hq; L[q] = lq;
      }
      char* p = s_pl + pf.lds[ps];
      *(m4_u4*)p = I;
      *(m4_u4*)(p + kM4Plane) = H;
      *(m4_u4*)(p + 2 * kM4Plane) = L;
    }
  }
}

// wave-wide maximum in seven instructions: v_max_f32 with the DPP modifier (the compiler spells each of these steps as
// v_mov + v_mov_dpp + two canonicalising v_max: 27 instructions).  Two wait states between a VALU write and the DPP read.
__device__ __forceinline__ float m4_wave_max(float x) {
  asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(x));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// per-lane state of one search (raw accumulator values of the best candidate) ...
// best_ks = ((tile column << 8 | tile row << 2 | accumulator register) << 15) | sum g of the best candidate: which of the
// lane's candidates it is (the lane itself gives column j and row group g) and its 15-bit pixel sum in one register - one
// v_add with a wave-uniform constant per candidate, one v_cndmask less per candidate than separate key and sum.
struct M4State {
  float best_q, second_q;
  int best_ks, best_w, best_x;
  __device__ __forceinline__ void reset() { best_q = -3.0e38f; second_q = -3.0e38f; best_ks = 0; best_w = best_x = 0; }
};
constexpr int kM4MaxTU = 128, kM4MaxTV = 64;     // tile columns / rows the packed key has room for (windows up to 2048 x 1024)
__device__ __forceinline__ int m4_ks_u(int ks, int j) { return 16 * (ks >> 23) + j; }                       // candidate column in the window
__device__ __forceinline__ int m4_ks_v(int ks, int g) { return 16 * ((ks >> 17) & 63) + 4 * g + ((ks >> 15) & 3); }
__device__ __forceinline__ int m4_ks_S1(int ks) { return ks & 0x7fff; }
// 121 sum g^2 - (sum g)^2 of a stored candidate (exact)
__device__ __forceinline__ int m4_D1(int S1, int S2) { return __mul24(121, S2) - __mul24(S1, S1); }
// wave-uniform constants of a search's scoring loop.  c_s1 / c_s2 are the start values of the sum-g and sum-g^2 (low byte
// plane) accumulators; they live in registers the compiler cannot see through (m4_opaque), or it would rebuild the two
// quads with eight v_mov in front of every tile.
struct M4Quads { mf_v4i c_s1, c_s2; };                 // once per wavefront
struct M4Const { int kS, c121, c_nc; float af, b2f, cf, lo, hi; };   // once per search
__device__ __forceinline__ void m4_opaque(mf_v4i& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ M4Quads m4_quads() {
  M4Quads q;
  q.c_s1 = mf_v4i{15488, 15488, 15488, 15488};
  q.c_s2 = mf_v4i{15488 * 257, 15488 * 257, 15488 * 257, 15488 * 257};
  m4_opaque(q.c_s1); m4_opaque(q.c_s2);
  return q;
}
// Ellipse membership is first decided in FP32: val_f = fma(fma(c, dv, 2 b du), dv, (a du) du) differs from the reference's
// FP64 value by less than 1e-6 R, R = 9 a c / (a c - b^2) >= every |term| (|du| <= halfwidth = floor(3 / sqrt(a - b^2 / c))
// gives a du^2 <= R, likewise c dv^2, and the cross term is at most their sum): a candidate with val_f outside
// [9 - tol, 9 + tol], tol = 4e-6 R, is classified for certain; any other candidate sends its tile down the reference's FP64
// expression (a degenerate S^-1 makes tol infinite or NaN: everything takes the FP64 path).
__device__ __forceinline__ M4Const m4_const(int Sg0, double a, double b2, double c) {
  M4Const k;
  k.kS = 15488 - Sg0;
  k.c121 = 121;
  k.c_nc = -15488 * k.kS;                  // |.| <= 15488 * 15367 < 2^31
  k.af = (float)a; k.b2f = (float)b2; k.cf = (float)c;
  const float ac = k.af * k.cf, det = ac - 0.25f * k.b2f * k.b2f;
  const float tol = 3.6e-5f * ac * __builtin_amdgcn_rcpf(det);         // 4e-6 * 9 a c / (a c - b^2)
  k.lo = 9.0f - tol; k.hi = 9.0f + tol;
  return k;
}
// a * b + c with a, b 24-bit: ONE v_mad_i32_i24 (b in a scalar register, c in a vector register: GFX9's VOP3 takes one
// scalar source and no literal, and left to itself the compiler splits the constant term off into a separate add)
__device__ __forceinline__ int m4_mad24(int v_a, int s_b, int v_c) {
  int r;
  asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(v_a), "s"(s_b), "v"(v_c));
  return r;
}
// scalar a + immediate on the scalar unit (the compiler re-associates S1 + (ks0 + imm) into two vector adds)
__device__ __forceinline__ int m4_sadd(int s_a, int imm) {
  int r;
  asm("s_add_i32 %0, %1, %2" : "=s"(r) : "s"(s_a), "n"(imm) : "scc");
  return r;
}
// ... and its wave-uniform part
struct M4Uni { int ncand; };

// Lane predicates as 64-bit masks in scalar registers.  The compiler keeps a predicate that is both selected on and
// counted (popcount of a ballot) as a 0 / 1 vector register and re-compares it - two more vector instructions per use, in
// a kernel bound by vector issue - so the compares and selects of the scoring loop are spelled out: a compare writes an
// SGPR pair, masks are combined and counted on the scalar unit, v_cndmask takes the pair.
typedef unsigned long long m4_mask;
__device__ __forceinline__ m4_mask m4_gt_i32(int s_a, int v_b) { m4_mask m; asm("v_cmp_gt_i32_e64 %0, %1, %2" : "=s"(m) : "s"(s_a), "v"(v_b)); return m; }   // a > b
__device__ __forceinline__ m4_mask m4_lt_i32(int s_a, int v_b) { m4_mask m; asm("v_cmp_lt_i32_e64 %0, %1, %2" : "=s"(m) : "s"(s_a), "v"(v_b)); return m; }   // a < b
__device__ __forceinline__ m4_mask m4_eq_i32(int s_a, int v_b) { m4_mask m; asm("v_cmp_eq_u32_e64 %0, %1, %2" : "=s"(m) : "s"(s_a), "v"(v_b)); return m; }
__device__ __forceinline__ m4_mask m4_gt_f64(double s_a, double v_b) { m4_mask m; asm("v_cmp_gt_f64_e64 %0, %1, %2" : "=s"(m) : "s"(s_a), "v"(v_b)); return m; }   // a > b
__device__ __forceinline__ m4_mask m4_lt_f32s(float v_a, float s_b) { m4_mask m; asm("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v_a), "s"(s_b)); return m; }   // a < b
__device__ __forceinline__ m4_mask m4_gt_f32s(float v_a, float s_b) { m4_mask m; asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v_a), "s"(s_b)); return m; }   // a > b
__device__ __forceinline__ m4_mask m4_gt_f32(float v_a, float v_b) { m4_mask m; asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v_a), "v"(v_b)); return m; }   // a > b
__device__ __forceinline__ int m4_sel(m4_mask m, int t, int f) { int r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m)); return r; }
__device__ __forceinline__ float m4_self(m4_mask m, float t, float f) { float r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(f), "v"(t), "s"(m)); return r; }

// the (up to two) 16 x 16 candidate tiles of the band in LDS
__device__ __forceinline__ void m4_band_tiles(const char* s_pl, const unsigned* s_T, const mf_v4i b_ones, const mf_v4i b_ones_last,
                                              int up, int vt, int TU, int nu_all, int nv_all, int urelstart, int vrelstart,
                                              const double* __restrict__ abc, const M4Quads& kq, const M4Const& k, bool patch_ok,
                                              int j, int g, M4State& st, M4Uni& un) {
  const int vi0 = 16 * vt + 4 * g;
  const unsigned* bp = mf_b_base(s_T, g, 16 + 16 * (g & 1) - j);      // this lane's template operands (offset 1..32)
  for (int ut = up; ut < min(up + 2, TU); ++ut) {
    // ellipse membership: FP32 with a guard band (m4_const), the reference's FP64 expression where that cannot decide.
    // (Round 4 also tried the matrix-core work FIRST and this classification behind it, to fill the MFMAs' ~400 cycles with
    // vector work of the same wavefront: 0.0947 against 0.0865 ms on the same box - the operand reads' LDS latency is then
    // exposed in front of the first MFMA, where this code used to cover it.  profiles/r04_search_ab_pmc.txt.)
    const int ui = 16 * ut + j;
    const m4_mask m_col = m4_gt_i32(nu_all, ui);
    m4_mask m_cand[4];
    {
      const float duf = (float)(urelstart + ui);
      const float euu = k.af * duf * duf, eu = k.b2f * duf;
      // rows vi0 .. vi0 + 3 of this lane as floats (exact: small integers); the row bound is tested on them too
      float dv0 = (float)(vrelstart + vi0);
      asm volatile("" : "+v"(dv0));          // (keeps the per-row terms out of registers across the matrix-core section)
      const float dvend = (float)(vrelstart + nv_all);
      m4_mask m_amb = 0ull;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float dvf = reg == 0 ? dv0 : dv0 + (float)reg;
        const float val = __builtin_fmaf(__builtin_fmaf(k.cf, dvf, eu), dvf, euu);
        const m4_mask in = m4_lt_f32s(val, k.lo);
        m_amb |= ~(in | m4_gt_f32s(val, k.hi));
        m_cand[reg] = m_col & m4_gt_f32(dvend, dvf) & in;
      }
      if (m_amb != 0ull) {
        // the reference's expression ((a u) u) + (((2 b) u) v) + ((c v) v) < 9, same values, same order
        const double a = abc[0], b2 = 2 * abc[1], c = abc[2];
        const double du = (double)(urelstart + ui);
        const double e_uu = a * du * du, e_u = b2 * du;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          int vi = vi0 + reg;
          asm volatile("" : "+v"(vi));
          const double dv = (double)(vrelstart + vi);
          m_cand[reg] = m_col & m4_gt_i32(nv_all, vi) & m4_gt_f64(kNoSigma * kNoSigma, e_uu + e_u * dv + c * dv * dv);
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) un.ncand += __popcll(m_cand[reg]);
    }
    // (a flat template: every candidate is skipped, they are only counted; a corner tile of a tilted ellipse may be empty)
    if (!patch_ok || (m_cand[0] | m_cand[1] | m_cand[2] | m_cand[3]) == 0ull) continue;
    mf_v4i accX, acc1, accH, accL;
    const mf_v4i zero = {0, 0, 0, 0};
    const char* ap = s_pl + 16 * (j + (g >> 1)) + kM4Chunk * ((ut - up) + (g & 1));
#pragma unroll
    for (int p = 0; p < 6; ++p) {
      const mf_v4i aI = *(const mf_v4i*)(ap + 32 * p);
      const mf_v4i aH = *(const mf_v4i*)(ap + 32 * p + kM4Plane);
      const mf_v4i aL = *(const mf_v4i*)(ap + 32 * p + 2 * kM4Plane);
      const mf_v4i bX = mf_load_b4(bp, p);
      const mf_v4i bo = (p == 5) ? b_ones_last : b_ones;
      // the offsets of the signed byte planes come back through the accumulators' start values: sum g = sum (g - 128) +
      // 15488 and sum g^2 = 256 sum (H - 128) + sum (L - 128) + 15488 * 257 leave the matrix cores as they are needed
      accX = __builtin_amdgcn_mfma_i32_16x16x64_i8(aI, bX, p ? accX : zero, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(aI, bo, p ? acc1 : kq.c_s1, 0, 0, 0);
      accH = __builtin_amdgcn_mfma_i32_16x16x64_i8(aH, bo, p ? accH : zero, 0, 0, 0);
      accL = __builtin_amdgcn_mfma_i32_16x16x64_i8(aL, bo, p ? accL : kq.c_s2, 0, 0, 0);
    }
    // The first readers of the accumulators below are inline-asm vector instructions (m4_mad24).  The compiler's hazard
    // recogniser inserts the wait states a matrix-core result needs before a VALU read only for instructions it can see
    // into - not for inline asm - so whether the scoring read finished accumulators used to depend on what the scheduler
    // happened to place in between (round 5: passing PuInv by pointer moved the first v_mad_i32_i24 up against the last MFMA
    // and the engine's searches came back one pixel off).  The wait states are spelled out: 16 cover an 8-pass result.
    // The count belongs to THIS instruction on THIS target - v_mfma_i32_16x16x64_i8 is 8 passes on gfx950 - and to nothing else:
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "the MFMA -> inline-asm wait states below (2 x s_nop 7) are counted for v_mfma_i32_16x16x64_i8 on gfx950 only"
#endif
    static_assert(sizeof(mf_v4i) == 16, "accumulators of a 16x16 i32 MFMA: four dwords per lane (the 8-pass shape the wait states are for)");
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(accX), "+v"(acc1), "+v"(accH), "+v"(accL));
    // Ranking value q = Nc / sqrt(D1) = rho * sqrt(D0) (D0 = 121 sum g0^2 - (sum g0)^2 is the same for every candidate of
    // a search, so it is left out: one multiply less per candidate; the callers scale the guard band by sqrt(D0) instead).
    // A candidate whose image sigma is EXACTLY 10 (D1 == 1464100) is ranked like a valid one; whether the reference skips
    // it is an FP64 matter that only counts if it is the winner or ties with it: then the caller takes the exact walk.
    const int ks0 = ((ut << 8) | (vt << 2)) << 15;               // wave-uniform: + (reg << 15) + sum g = the packed key | sum
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int S1 = acc1[reg];                                  // sum g
      const int S2 = (accH[reg] << 8) + accL[reg];               // sum g^2
      const int D1 = mul24(121, S2) - mul24(S1, S1);             // 121 sum g^2 - (sum g)^2, exact
      // 121 sum g0 g - sum g0 sum g = 121 x + kS (S1 - 15488), x = the raw cross term of the offset operands (exact)
      const int Nc = m4_mad24(S1, k.kS, m4_mad24(accX[reg], k.c121, k.c_nc));
      const float q = (float)Nc * __builtin_amdgcn_rsqf((float)D1);
      const float qq = m4_self(m_cand[reg] & m4_lt_i32(1464099, D1), q, -3.0e38f);
      const m4_mask better = m4_gt_f32(qq, st.best_q);
      st.second_q = __builtin_amdgcn_fmed3f(st.best_q, st.second_q, qq);    // second of {best, second, new}
      st.best_q = m4_self(better, qq, st.best_q);
      st.best_ks = m4_sel(better, S1 + m4_sadd(ks0, reg << 15), st.best_ks);
      st.best_w = m4_sel(better, S2, st.best_w);
      st.best_x = m4_sel(better, accX[reg], st.best_x);
    }
  }
}

// One search, one wavefront, nothing in flight across calls: the stateless batch API (templates as raw 121 bytes).
__device__ __forceinline__ SearchResult search_core_mfma(const uint8_t* __restrict__ image, int width, int frame_bytes,
                                                       const uint8_t* __restrict__ patch, const SearchBounds sb, double a,
                                                       double b, double c, char* s_pl, unsigned* s_T) {
  const double abc_v[3] = {a, b, c}; const double* abc_ = abc_v;
  const int lane = threadIdx.x & 63;
  const int nu_all = sb.urelfinish - sb.urelstart + 1;
  const int nv_all = sb.vrelfinish - sb.vrelstart + 1;
  SearchResult res;
  res.code = 0; res.ok = 0; res.found = 0; res.u = 0; res.v = 0; res.ncand = 0; res.score = 1000000.0;
  res.S1 = res.S2 = res.X = 0;
  if (nu_all <= 0 || nv_all <= 0) return res;
  const int TU = (nu_all + 15) >> 4, TV = (nv_all + 15) >> 4;
  if (TU > kM4MaxTU || TV > kM4MaxTV) { res.code = -1; return res; }       // beyond the packed key: the exact walk
  unsigned tv = 0;                                  // the template in the lane layout of mf_tpl_store
  const int tr = lane >> 2, tq = lane & 3;
  if (lane < 44 && tq) {
    for (int kk = 0; kk < 4; ++kk) {
      const int col = 4 * (tq - 1) + kk;
      const unsigned byte = (col < 11) ? patch[tr * 11 + col] : 0u;
      tv |= byte << (8 * kk);
    }
  }
  unsigned s1 = udot4(tv, 0x01010101u, 0u), s2 = udot4(tv, tv, 0u);
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
  const int Sg0 = (int)s1, Sg0sq = (int)s2;
  // patch sigma test, exactly as correlate2_warning + elliptical_search evaluate it
  const double g0bar = (double)Sg0 / 121.0;
  const double varg0 = (double)Sg0sq / 121.0 - (g0bar * g0bar);
  const double sigmag0 = sqrt(varg0);
  const bool patch_ok = !(sigmag0 < kCorrelationSigmaThreshold);
  const float d0f = (float)(121 * Sg0sq - Sg0 * Sg0);
  mf_tpl_init(s_T, lane);
  __syncthreads();
  mf_tpl_ones(s_T, lane);
  mf_tpl_store(tv, mf_tpl_mask(lane), s_T, lane);
  __syncthreads();
  const int j = lane & 15, g = lane >> 4;
  const int boff = 16 + 16 * (g & 1) - j;
  const mf_v4i b_ones = mf_load_b(s_T, 12, boff);
  const mf_v4i b_ones_last = mf_load_b(s_T, (g >> 1) ? 11 : 12, boff);   // template row 11 does not exist: the zero row
  M4State st;
  st.reset();
  M4Uni un;
  un.ncand = 0;
  const M4Quads kq = m4_quads();
  const M4Const kc = m4_const(Sg0, a, 2 * b, c);
  for (int vt = 0; vt < TV; ++vt)
    for (int up = 0; up < TU; up += 2) {
      if (vt + up > 0) __syncthreads();                                // the previous band has been consumed
      const M4Band bd = m4_band(sb.ucentre, sb.vcentre, sb.urelstart, sb.vrelstart, nu_all, nv_all, up, vt, width);
      M4Pf pf;
      if (m4_band_loads(image, width, frame_bytes, bd, lane, pf)) m4_band_fix(image, width, frame_bytes, bd, lane, pf);
      m4_band_store(pf, bd, s_pl);
      __syncthreads();
      m4_band_tiles(s_pl, s_T, b_ones, b_ones_last, up, vt, TU, nu_all, nv_all, sb.urelstart, sb.vrelstart, abc_, kq, kc,
                    patch_ok, j, g, st, un);
    }
  res.ncand = un.ncand;
  if (!patch_ok) return res;
  const float gmax = m4_wave_max(st.best_q);
  if (!(gmax > -1.0e38f)) return res;                                     // nothing passed the sigma test
  const float thr = gmax - 4.1e-6f * __builtin_amdgcn_sqrtf(d0f);         // q = rho sqrt(D0): the 4e-6 guard band on rho
  const unsigned long long near_mask = __ballot(st.best_q >= thr);
  if (__any(st.second_q >= thr) || __popcll(near_mask) > 1) { res.code = -1; return res; }
  const int wl = __ffsll((long long)near_mask) - 1;
  const int ks = __shfl(st.best_ks, wl, 64), ww = __shfl(st.best_w, wl, 64), wx = __shfl(st.best_x, wl, 64);
  const int wS1 = m4_ks_S1(ks);
  if (m4_D1(wS1, ww) == 1464100) { res.code = -1; return res; }           // image sigma exactly 10: decided in FP64 by the exact walk
  res.S1 = wS1; res.S2 = ww; res.X = wx + 128 * (wS1 - 15488) + 128 * Sg0;
  res.found = 1;
  res.u = sb.ucentre + sb.urelstart + m4_ks_u(ks, wl & 15);
  res.v = sb.vcentre + sb.vrelstart + m4_ks_v(ks, wl >> 4);
  double sd0, sd1;
  const double corr = ncc_score(Sg0, res.S1, res.X, Sg0sq, res.S2, &sd0, &sd1);
  if (sd0 < kCorrelationSigmaThreshold || sd1 < kCorrelationSigmaThreshold) {   // cannot happen (D1 > bound), kept exact
    res.found = 0; res.u = res.v = 0;
    return res;
  }
  res.score = corr;
  res.ok = !(corr > kCorrThresh2) ? 1 : 0;
  return res;
}

// ---------------------------------------------------------------------------
// Engine kernels.  k_search_exact: one wave per (sequence, selected position) on search_core_v0, XCD-mapped so that all
// windows of one frame go through one XCD's L2 (search variant 0: the cross-check of the matrix-core walk).  Both search
// kernels write a compact result record; k_search_score (one THREAD per selected position, all lanes busy) evaluates the
// deferred FP64 scores and does the reference's bookkeeping (successful_/failed_measurement_of_feature,
// monoslam.cpp:479-496).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_search_exact(const uint8_t* __restrict__ frames, size_t seq_stride, int width,
                                                     const uint8_t* __restrict__ patch, const int* __restrict__ srch_i,
                                                     const double* __restrict__ srch_d, const int* __restrict__ sel_idx,
                                                     const int* __restrict__ n_sel, int* __restrict__ srch_res,
                                                     double* __restrict__ meas_score, int N, int nsel_max, int B) {
  int b, k;
  if (!xcd_map(nsel_max, B, &b, &k)) return;
  if (k >= n_sel[b]) return;
  const int f = sel_idx[(size_t)b * N + k];
  const size_t fi = (size_t)b * N + f;
  const SearchBounds sb = bounds_from_desc(srch_i + fi * 8);
  const double a = srch_d[fi * 4], bq = srch_d[fi * 4 + 1], c = srch_d[fi * 4 + 2];
  const SearchResult r = search_core_v0(frames + (size_t)b * seq_stride, width, patch + fi * kPatchStride, sb, a, bq, c);
  if ((threadIdx.x & 63) == 0) {
    int* o = srch_res + ((size_t)b * N + k) * 8;
    o[0] = r.code; o[1] = r.u; o[2] = r.v; o[3] = r.S1; o[4] = r.S2; o[5] = r.X; o[6] = r.ncand;
    o[7] = (r.found ? 1 : 0) | (r.ok ? 2 : 0);
    meas_score[(size_t)b * N + k] = r.score;
  }
}

#ifndef SL2_MF_WAVES
#define SL2_MF_WAVES 4
#endif

// ---------------------------------------------------------------------------
// Large windows (round 4).  A feature whose position is poorly constrained - a fresh one under a weak pose estimate - has
// a window of up to the whole frame: 150 bands at 320 x 240, each a synchronous round trip for the one wavefront that owns
// it.  In the mapping workload that tail WAS the kernel: 0.12 ms per launch on average and 0.55 at worst for 0.03 ms of
// typical work (profiles/r04_mapping_kernel_stats.csv).  k_select therefore cuts every window of at least
// sl2_engine::search_split bands into UNITS of a few bands (srch_unit_bands: four, more for windows beyond 256 bands, so
// that a window has at most kSrchBigSlots units), puts them on the step's list (srch_big) and marks the window's record;
// the position's own wavefront skips it, and the last workgroups of the launch, which own no positions, take units off
// the list - one atomic counter for the whole list, a unit per grab:
//   * a unit leaves a partial result: the best ranking value among its candidates, the second best, the best
//     candidate's position and sums, the number of candidates inside the ellipse;
//   * the wavefront that finishes a window's LAST unit (a counter per window) combines the partial results.  The decision
//     is the one a single wavefront takes - a unique candidate within the guard band of the maximum is the reference's
//     winner, anything else goes to the exact walk: "unique" is "the second largest ranking value of the whole window lies
//     below the band", and the second largest of a union follows from each part's two largest.
// Nobody waits for anybody; the counters are returned to zero by k_search_score (the next launch).  Partial results cross
// XCDs: they are written and read with agent-scope atomics around __threadfence().
// ---------------------------------------------------------------------------
__device__ __forceinline__ int m4_wave_atomic_add(int* p, int v, int lane) {
  int r = 0;
  if (lane == 0) r = atomicAdd(p, v);
  return __builtin_amdgcn_readfirstlane(r);
}
__device__ __forceinline__ int m4_coherent_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void m4_coherent_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void m4_big_windows(const uint8_t* __restrict__ frames, size_t seq_stride, int width, int frame_bytes,
                                               const uint8_t* __restrict__ patch, const int* __restrict__ srch_sel,
                                               int* __restrict__ srch_res, double* __restrict__ meas_score, int N,
                                               int* __restrict__ big, int nunits, char* s_pl, unsigned* s_T) {
  const int lane = threadIdx.x;
  const int j = lane & 15, g = lane >> 4;
  int u = m4_wave_atomic_add(big + 2, 1, lane);
  if (u >= nunits) return;                            // (the usual end of a wavefront when a list exists: everything is handed out)
  __syncthreads();                                    // whatever the caller was doing with the LDS is over
  mf_tpl_init(s_T, lane);
  __syncthreads();
  mf_tpl_ones(s_T, lane);
  __syncthreads();
  const int boff = 16 + 16 * (g & 1) - j;
  const mf_v4i b_ones = mf_load_b(s_T, 12, boff);
  const mf_v4i b_ones_last = mf_load_b(s_T, (g >> 1) ? 11 : 12, boff);
  const M4Quads kq = m4_quads();
  const int tidx = mf_tpl_index(lane);
  const bool tload = lane < 44 ? (lane & 3) != 0 : (lane >= 48 && lane <= 50);
  const unsigned tmask = mf_tpl_mask(lane);
  for (; u < nunits; u = m4_wave_atomic_add(big + 2, 1, lane)) {
    const int4 en = *(const int4*)(big + kSrchBigEntries + 4 * u);       // sequence, selected position, the window's first unit, its units
    const int b = en.x, k = en.y, u0 = en.z, units = en.w;
    if (b < 0) continue;                                                   // (allocated for a window that found no room)
    const int* rec = srch_sel + ((size_t)b * N + k) * 16;
    const int f = rec[0], uc = rec[1], vc = rec[2], us = rec[3], nu_all = rec[14], vs = rec[5], nv_all = rec[6];   // (rec[4] = kSrchSharedNu: see k_select)
    const int TU = (nu_all + 15) >> 4, TV = (nv_all + 15) >> 4, nbu = (TU + 1) >> 1, nbands = nbu * TV;
    const int per = srch_unit_bands(nbands);
    // (an entry that no longer describes its record - the selection ran twice before a search - is nobody's work: the
    // record's own entries, or its own wavefront, cover it)
    if (rec[4] != kSrchSharedNu || nu_all <= 0 || nv_all <= 0 || (nbands + per - 1) / per != units) continue;
    const double* recd = (const double*)(rec + 8);
    const double pa = recd[0], b2 = 2 * recd[1], pc = recd[2];
    const uint8_t* img = frames + (size_t)b * seq_stride;
    const unsigned* tpl = (const unsigned*)(patch + ((size_t)b * N + f) * kPatchStride + kPatchPackedOffset);
    const unsigned tv = tload ? tpl[tidx] : 0u;
    const int Sg0 = (int)__builtin_amdgcn_readlane(tv, 48), Sg0sq = (int)__builtin_amdgcn_readlane(tv, 49);
    const bool patch_ok = __builtin_amdgcn_readlane(tv, 50) != 0;
    __syncthreads();                                  // the previous unit's last tile has been read
    mf_tpl_store(tv, tmask, s_T, lane);
    const M4Const kc = m4_const(Sg0, pa, b2, pc);
    M4State st;
    st.reset();
    M4Uni un;
    un.ncand = 0;
    const int band0 = (u - u0) * per, band1 = min(band0 + per, nbands);
    for (int bi = band0; bi < band1; ++bi) {
      const int vt = bi / nbu, up = 2 * (bi - vt * nbu);
      __syncthreads();
      const M4Band bd = m4_band(uc, vc, us, vs, nu_all, nv_all, up, vt, width);
      M4Pf pf;
      if (m4_band_loads(img, width, frame_bytes, bd, lane, pf)) m4_band_fix(img, width, frame_bytes, bd, lane, pf);
      m4_band_store(pf, bd, s_pl);
      __syncthreads();
      m4_band_tiles(s_pl, s_T, b_ones, b_ones_last, up, vt, TU, nu_all, nv_all, us, vs, recd, kq, kc, patch_ok, j, g, st, un);
    }
    // ---- the unit's partial result: maximum, runner-up, the best candidate's position and sums, candidate count
    const float gmax = m4_wave_max(st.best_q);
    const int wl = __ffsll((long long)__ballot(st.best_q == gmax)) - 1;  // (every lane holds -3e38 if nothing qualified: lane 0)
    const float top2 = m4_wave_max(lane == wl ? st.second_q : st.best_q);
    const int w_ks = __builtin_amdgcn_readlane(st.best_ks, wl), w_S2 = __builtin_amdgcn_readlane(st.best_w, wl);
    const int w_x = __builtin_amdgcn_readlane(st.best_x, wl);
    const int w_S1 = m4_ks_S1(w_ks);
    if (lane == 0) {
      int* pr = big + kSrchBigParts + 8 * u;
      m4_coherent_store(pr + 0, __float_as_int(gmax));
      m4_coherent_store(pr + 1, __float_as_int(top2));
      m4_coherent_store(pr + 2, uc + us + m4_ks_u(w_ks, wl & 15));
      m4_coherent_store(pr + 3, vc + vs + m4_ks_v(w_ks, wl >> 4));
      m4_coherent_store(pr + 4, w_S1);
      m4_coherent_store(pr + 5, w_S2);
      m4_coherent_store(pr + 6, w_x + 128 * w_S1 + 128 * (Sg0 - 15488));
      m4_coherent_store(pr + 7, un.ncand);
    }
    __threadfence();                                                     // the partial result is visible before the count says so
    if (m4_wave_atomic_add(big + kSrchBigDone + u0, 1, lane) != units - 1) continue;
    // ---- this was the window's last unit: combine (a lane per unit)
    __threadfence();
    float pg = -3.0e38f, pt = -3.0e38f;
    int pu = 0, pv = 0, pS1 = 0, pS2 = 0, pX = 0, pn = 0;
    if (lane < units) {
      const int* pr = big + kSrchBigParts + 8 * (u0 + lane);
      pg = __int_as_float(m4_coherent_load(pr + 0)); pt = __int_as_float(m4_coherent_load(pr + 1));
      pu = m4_coherent_load(pr + 2); pv = m4_coherent_load(pr + 3); pS1 = m4_coherent_load(pr + 4);
      pS2 = m4_coherent_load(pr + 5); pX = m4_coherent_load(pr + 6); pn = m4_coherent_load(pr + 7);
    }
    int ncand = pn;
    for (int off = 32; off > 0; off >>= 1) ncand += __shfl_xor(ncand, off, 64);
    int* o = srch_res + ((size_t)b * N + k) * 8;
    int code = 0;
    unsigned long long near_units = 0ull;                                  // units that hold a candidate within the guard band
    if (patch_ok) {
      const float wmax = m4_wave_max(pg);
      if (wmax > -1.0e38f) {
        const float d0f = (float)(121 * Sg0sq - Sg0 * Sg0);
        const float thr = wmax - 4.1e-6f * __builtin_amdgcn_sqrtf(d0f);  // q = rho sqrt(D0): the 4e-6 guard band on rho
        const int hl = __ffsll((long long)__ballot(pg == wmax)) - 1;
        const float second = m4_wave_max(lane == hl ? pt : pg);          // second largest ranking value of the whole window
        const bool boundary = __builtin_amdgcn_readlane(m4_D1(pS1, pS2), hl) == 1464100;
        if (boundary) code = -1;                                           // sigma == 10 for the best: the exact walk of the whole window
        else if (second >= thr) { code = -2; near_units = __ballot(lane < units && pg >= thr); }   // several near-best candidates
        else {
          code = 1;
          if (lane == hl) {
            int4 o0, o1;
            o0.x = 1; o0.y = pu; o0.z = pv; o0.w = pS1;
            o1.x = pS2; o1.y = pX; o1.z = ncand; o1.w = 1;
            *(int4*)o = o0; *(int4*)(o + 4) = o1;                          // (k_search_score writes this position's score)
          }
        }
      }
    }
    if (code == -2) {
      // The reference's winner is the smallest FP64 score, among equals the last in its scan order (u outer, v inner).  Every
      // candidate that can be it lies within the guard band of the best ranking value, i.e. in one of `near_units`: only
      // their bands take the exact walk (a window of the whole frame is 150 bands; the ones that hold the tie are one or
      // two), and the bands' winners are compared by the same rule.
      double best = 1000000.0;
      int bu = 0, bv = 0, found = 0;
      while (near_units) {
        const int h = __ffsll((long long)near_units) - 1;
        near_units &= near_units - 1;
        for (int bi = h * per; bi < min(h * per + per, nbands); ++bi) {
          const int vt = bi / nbu, up = 2 * (bi - vt * nbu);
          SearchBounds sb;
          sb.ucentre = uc; sb.vcentre = vc; sb.halfwidth = sb.halfheight = 0;
          sb.urelstart = us + 16 * up; sb.urelfinish = min(us + 16 * up + 31, us + nu_all - 1);
          sb.vrelstart = vs + 16 * vt; sb.vrelfinish = min(vs + 16 * vt + 15, vs + nv_all - 1);
          const SearchResult r = search_core_v0(img, width, patch + ((size_t)b * N + f) * kPatchStride, sb, recd[0], recd[1], recd[2]);
          if (r.found && (!found || r.score < best || (r.score == best && (r.u > bu || (r.u == bu && r.v > bv))))) {
            found = 1; best = r.score; bu = r.u; bv = r.v;
          }
        }
      }
      if (lane == 0) {
        o[0] = 0; o[1] = found ? bu : 0; o[2] = found ? bv : 0; o[3] = 0; o[4] = 0; o[5] = 0; o[6] = ncand;
        o[7] = (found ? 1 : 0) | ((found && !(best > kCorrThresh2)) ? 2 : 0) | 4;
        meas_score[(size_t)b * N + k] = best;
      }
    } else if (code < 0) {
      SearchBounds sb;
      sb.ucentre = uc; sb.vcentre = vc; sb.urelstart = us; sb.urelfinish = us + nu_all - 1;
      sb.vrelstart = vs; sb.vrelfinish = vs + nv_all - 1; sb.halfwidth = sb.halfheight = 0;
      const SearchResult r = search_core_v0(img, width, patch + ((size_t)b * N + f) * kPatchStride, sb, recd[0], recd[1], recd[2]);
      if (lane == 0) {
        o[0] = r.code; o[1] = r.u; o[2] = r.v; o[3] = r.S1; o[4] = r.S2; o[5] = r.X; o[6] = r.ncand;
        o[7] = (r.found ? 1 : 0) | (r.ok ? 2 : 0) | 4;
        meas_score[(size_t)b * N + k] = r.score;
      }
    } else if (code == 0 && lane == 0) {                                   // nothing found
      int4 o0, o1;
      o0.x = 0; o0.y = 0; o0.z = 0; o0.w = 0; o1.x = 0; o1.y = 0; o1.z = ncand; o1.w = 0;
      *(int4*)o = o0; *(int4*)(o + 4) = o1;
      meas_score[(size_t)b * N + k] = 1000000.0;
    }
  }
}

constexpr int kMfChunk = 4;
constexpr int kSearchBigWaves = 2048;   // workgroups of k_search_mfma that work off the large windows' units, at most: two per SIMD of the chip
// Engine kernel.  One wavefront works through `chunk` consecutive selected positions of one sequence (XCD-mapped: a
// sequence's frame stays in one XCD's L2).  A feature's search is a chain of dependent memory round trips - record,
// template + window, LDS - so the loop is software-pipelined: while position i is in the matrix cores and being scored,
// the template and the first window band of position i + 1 are already in flight.  Held to the register budget of
// SL2_MF_WAVES wavefronts per SIMD.
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SL2_MF_WAVES, 8)))
k_search_mfma(const uint8_t* __restrict__ frames, size_t seq_stride, int width, int frame_bytes, const uint8_t* __restrict__ patch,
            const int* __restrict__ srch_sel, const int* __restrict__ n_sel, int* __restrict__ srch_res,
            double* __restrict__ meas_score, int N, int nchunks, int B, int chunk, int* __restrict__ srch_big) {
  __shared__ __attribute__((aligned(16))) char s_pl[3 * kM4Plane];
  __shared__ __attribute__((aligned(16))) unsigned s_T[kMfTplDw];
  // The last workgroups of the grid (launch_search: up to kSearchBigWaves) own no positions: they work off the units of the step's large windows
  // (m4_big_windows above; with no window on the list - the usual step of a well-constrained map - they read the list's
  // length and end).  Being the last to be dispatched they run while the launch drains.  Measured alternatives
  // (profiles/r04_search_shared_ab.txt): the same work at the END of every wavefront costs the common path 5-9 % (the
  // arguments' scalar registers stay alive across the wavefront's own positions, the list length is read at the start or
  // at the end) and ran the units slower (0.134 against 0.082 ms in the mapping workload); a launch of its own costs ~3 us
  // and moved the placement of the launches behind it (k_build_AS + 13 us at the headline shape).
  if (srch_big && (int)blockIdx.x >= nchunks * ((B + kXcds - 1) / kXcds * kXcds)) {      // (= xcd_grid(nchunks, B), the workgroups that own positions)
    const int nunits = min(srch_big[0], kSrchBigUnits);
    if (nunits > 0)
      m4_big_windows(frames, seq_stride, width, frame_bytes, patch, srch_sel, srch_res, meas_score, N, srch_big, nunits, s_pl, s_T);
    return;
  }
  int b, ch;
  if (!xcd_map(nchunks, B, &b, &ch)) return;
  STR(0);
  const int lane = threadIdx.x;
  const int k0 = ch * chunk;
  const int nsel = n_sel[b];
  if (k0 >= nsel) return;
  const int nf = min(chunk, nsel - k0);
  const uint8_t* img = frames + (size_t)b * seq_stride;
  const int j = lane & 15, g = lane >> 4;

  // the record k_select wrote for selected position k (one 64-byte line; uniform address: scalar loads).  The window
  // integers are fetched one position ahead (the prefetch needs them), PuInv when its position is worked on.
  // (A window that k_select put on the step's list of large windows has nu = kSrchSharedNu in its record - its true width
  // sits in rec[14] - and reads as an empty window here: nothing is loaded for it, and the position's result is left to
  // the workgroups that work off the list.)
  struct Rec { int f, uc, vc, us, nu, vs, nv; };
  auto record = [&](int i) {
    const int* rec = srch_sel + ((size_t)b * N + k0 + i) * 16;
    Rec r;
    r.f = rec[0]; r.uc = rec[1]; r.vc = rec[2]; r.us = rec[3]; r.nu = rec[4]; r.vs = rec[5]; r.nv = rec[6];
    return r;
  };

  mf_tpl_init(s_T, lane);
  const int tidx = mf_tpl_index(lane);               // this lane's dword of a packed template record ...
  const bool tload = lane < 44 ? (lane & 3) != 0 : (lane >= 48 && lane <= 50);
  const unsigned tmask = mf_tpl_mask(lane);          // ... and which of its bytes are pixels
  M4Pf pf;
  pf.lds[0] = pf.lds[1] = -1;
  bool pf_over = false;
  unsigned pf_tv = 0;
  Rec rc = record(0);
  {
    if (rc.nu > 0 && rc.nv > 0)
      pf_over = m4_band_loads(img, width, frame_bytes, m4_band(rc.uc, rc.vc, rc.us, rc.vs, rc.nu, rc.nv, 0, 0, width), lane, pf);
    const unsigned* tpl = (const unsigned*)(patch + ((size_t)b * N + rc.f) * kPatchStride + kPatchPackedOffset);
    pf_tv = tload ? tpl[tidx] : 0u;
  }
  // operands of the ones matrix: they depend on the lane alone
  __syncthreads();
  mf_tpl_ones(s_T, lane);
  __syncthreads();
  const int boff = 16 + 16 * (g & 1) - j;
  const mf_v4i b_ones = mf_load_b(s_T, 12, boff);
  const mf_v4i b_ones_last = mf_load_b(s_T, (g >> 1) ? 11 : 12, boff);   // template row 11 does not exist: the zero row
  const M4Quads kq = m4_quads();
  STR(1);
  PH_DECL;
  for (int i = 0; i < nf; ++i) {
    // the next position's record: scalar loads issued here, consumed after the barrier below (clamped index: the last
    // iteration re-reads its own record instead of branching)
    const Rec rn = record(min(i + 1, nf - 1));
    const double* recd = (const double*)(srch_sel + ((size_t)b * N + k0 + i) * 16 + 8);
    const double pa = recd[0], pb = recd[1], pc = recd[2];
    const int nu_all = rc.nu, nv_all = rc.nv;
    const bool geom_ok = nu_all > 0 && nv_all > 0;
    const unsigned tv = pf_tv;
    const int Sg0 = (int)__builtin_amdgcn_readlane(tv, 48), Sg0sq = (int)__builtin_amdgcn_readlane(tv, 49);
    const bool patch_ok = __builtin_amdgcn_readlane(tv, 50) != 0;
    const float d0f = (float)(121 * Sg0sq - Sg0 * Sg0);
    PH_WAIT("lgkmcnt(0)"); PH(0);                     // (trace build: the record's scalar loads)
    PH_WAIT("vmcnt(0)"); PH(1);                       // (trace build: the prefetched band / template - and the previous position's stores)
    if (i > 0) __syncthreads();                       // feature i - 1 is done with the LDS
    if (geom_ok) {
      mf_tpl_store(tv, tmask, s_T, lane);
      const M4Band bd0 = m4_band(rc.uc, rc.vc, rc.us, rc.vs, nu_all, nv_all, 0, 0, width);
      if (pf_over) m4_band_fix(img, width, frame_bytes, bd0, lane, pf);
      m4_band_store(pf, bd0, s_pl);
    }
    __syncthreads();
    PH_WAIT("lgkmcnt(0)"); PH(2);                     // (trace build: planes computed, band and template in LDS)
    pf_over = false;
    if (i + 1 < nf) {                                 // next feature's template and first band: in flight from here on
      if (rn.nu > 0 && rn.nv > 0)
        pf_over = m4_band_loads(img, width, frame_bytes, m4_band(rn.uc, rn.vc, rn.us, rn.vs, rn.nu, rn.nv, 0, 0, width), lane, pf);
      const unsigned* tpl = (const unsigned*)(patch + ((size_t)b * N + rn.f) * kPatchStride + kPatchPackedOffset);
      pf_tv = tload ? tpl[tidx] : 0u;
    }
    PH(3);
    int* o = srch_res + ((size_t)b * N + k0 + i) * 8;
    M4State st;
    st.reset();
    M4Uni un;
    un.ncand = 0;
    int code = 0;
    const int TU = (nu_all + 15) >> 4, TV = (nv_all + 15) >> 4;
    if (geom_ok && (TU > kM4MaxTU || TV > kM4MaxTV)) code = -1;      // a window beyond the packed key: the exact walk
    else if (geom_ok) {
      const double b2 = 2 * pb;
      const M4Const kc = m4_const(Sg0, pa, b2, pc);
      m4_band_tiles(s_pl, s_T, b_ones, b_ones_last, 0, 0, TU, nu_all, nv_all, rc.us, rc.vs, recd, kq, kc, patch_ok, j, g,
                    st, un);
      PH(4);                                          // (trace build: ellipse + matrix cores + scoring of the first band)
      if (TU > 2 || TV > 1)                           // (most windows are one band: keep the loop set-up off their path)
      for (int vt = 0; vt < TV; ++vt)                 // the other bands of a large window: staged synchronously
        for (int up = (vt == 0 ? 2 : 0); up < TU; up += 2) {
          __syncthreads();
          const M4Band bd = m4_band(rc.uc, rc.vc, rc.us, rc.vs, nu_all, nv_all, up, vt, width);
          M4Pf p2;
          if (m4_band_loads(img, width, frame_bytes, bd, lane, p2)) m4_band_fix(img, width, frame_bytes, bd, lane, p2);
          m4_band_store(p2, bd, s_pl);
          __syncthreads();
          m4_band_tiles(s_pl, s_T, b_ones, b_ones_last, up, vt, TU, nu_all, nv_all, rc.us, rc.vs, recd, kq, kc, patch_ok,
                        j, g, st, un);
        }
      PH(5);                                          // (trace build: further bands)
      // ---- decision: a unique near-best candidate goes on to k_search_score with its exact sums
      if (patch_ok) {
        const float gmax = m4_wave_max(st.best_q);
        if (gmax > -1.0e38f) {
          const float thr = gmax - 4.1e-6f * __builtin_amdgcn_sqrtf(d0f);   // q = rho sqrt(D0): the 4e-6 guard band on rho
          const bool near = st.best_q >= thr;
          const unsigned long long near_mask = __ballot(near);
          const int wS1 = m4_ks_S1(st.best_ks);
          // several near-best candidates, or the one there is sits on the sigma == 10 boundary: the exact walk decides
          if (__any(st.second_q >= thr) || __popcll(near_mask) > 1 || __any(near && m4_D1(wS1, st.best_w) == 1464100)) code = -1;
          else {
            code = 1;
            if (near) {                               // the one lane that holds the only possible winner stores its record
              int4 o0, o1;
              o0.x = 1; o0.y = rc.uc + rc.us + m4_ks_u(st.best_ks, j); o0.z = rc.vc + rc.vs + m4_ks_v(st.best_ks, g); o0.w = wS1;
              o1.x = st.best_w; o1.y = st.best_x + 128 * wS1 + 128 * (Sg0 - 15488); o1.z = un.ncand; o1.w = 1;
              *(int4*)o = o0; *(int4*)(o + 4) = o1;     // (k_search_score writes this position's score)
            }
          }
        }
      }
    }
    if (code < 0) {
      SearchBounds sb;
      sb.ucentre = rc.uc; sb.vcentre = rc.vc; sb.urelstart = rc.us; sb.urelfinish = rc.us + nu_all - 1;
      sb.vrelstart = rc.vs; sb.vrelfinish = rc.vs + nv_all - 1; sb.halfwidth = sb.halfheight = 0;
      const double* recd2 = (const double*)(srch_sel + ((size_t)b * N + k0 + i) * 16 + 8);     // (rare path: fetched again)
      const SearchResult r = search_core_v0(img, width, patch + ((size_t)b * N + rc.f) * kPatchStride, sb, recd2[0], recd2[1], recd2[2]);
      if (lane == 0) {
        o[0] = r.code; o[1] = r.u; o[2] = r.v; o[3] = r.S1; o[4] = r.S2; o[5] = r.X; o[6] = r.ncand;
        o[7] = (r.found ? 1 : 0) | (r.ok ? 2 : 0) | 4;
        meas_score[(size_t)b * N + k0 + i] = r.score;
      }
    } else if (code == 0 && lane == 0 && nu_all != kSrchSharedNu) {      // nothing found (or nothing to search)
      int4 o0, o1;
      o0.x = 0; o0.y = 0; o0.z = 0; o0.w = 0; o1.x = 0; o1.y = 0; o1.z = un.ncand; o1.w = 0;
      *(int4*)o = o0; *(int4*)(o + 4) = o1;
      meas_score[(size_t)b * N + k0 + i] = 1000000.0;
    }
    PH(6);                                            // (trace build: decision and result record)
    rc = rn;
  }
  PH_STORE;
  STR(6);
}

// k_search_score: one workgroup per sequence, one thread per selected position (body: sl2_score_dev.hpp).
__global__ void __launch_bounds__(1024) k_search_score(const int* __restrict__ srch_res, const int* __restrict__ srch_i,
                                                       const uint8_t* __restrict__ patch, const double* __restrict__ f_h,
                                                       const int* __restrict__ sel_idx, const int* __restrict__ n_sel,
                                                       int* __restrict__ f_flags, double* __restrict__ f_z,
                                                       double* __restrict__ f_nu, int* __restrict__ attempted,
                                                       int* __restrict__ successful, int* __restrict__ meas_ok,
                                                       double* __restrict__ meas_score, double* __restrict__ work,
                                                       int* __restrict__ succ_idx, int* __restrict__ f_arow,
                                                       int* __restrict__ m_count, const int* __restrict__ n_slots,
                                                       const int* __restrict__ pos_err, const int* __restrict__ pos_err_any,
                                                       int* __restrict__ f_hcol, const int* __restrict__ ps_i, int kpart, int ppos0,
                                                       int N, int* __restrict__ srch_big, int* __restrict__ status) {
  extern __shared__ int s_flag[];
  search_score_body(blockIdx.x, srch_res, srch_i, patch, f_h, sel_idx, n_sel, f_flags, f_z, f_nu, attempted, successful, meas_ok,
                    meas_score, work, succ_idx, f_arow, m_count, n_slots, pos_err, pos_err_any, f_hcol, ps_i, kpart, ppos0, N,
                    srch_big, status, s_flag);
}

// Stateless batch kernel (C-ABI seam S1): grid (count), one wave per search.  VARIANT 0 = exact, 1 = matrix-core walk.
template <int VARIANT>
__global__ void __launch_bounds__(64) k_search_batch(const uint8_t* __restrict__ images, int width, int height,
                                                     const int* __restrict__ image_index, const uint8_t* __restrict__ patches,
                                                     const double* __restrict__ centre, const double* __restrict__ puinv,
                                                     int* __restrict__ ok, int* __restrict__ uv, double* __restrict__ score) {
  const int i = blockIdx.x;
  __shared__ __attribute__((aligned(16))) char s_pl[VARIANT == 1 ? 3 * kM4Plane : 16];
  __shared__ __attribute__((aligned(16))) unsigned s_tpl[VARIANT == 1 ? kMfTplDw : 4];
  const double ce[2] = {centre[i * 2], centre[i * 2 + 1]};
  const double a = puinv[i * 3], b = puinv[i * 3 + 1], c = puinv[i * 3 + 2];
  const SearchBounds sb = search_bounds(ce, a, b, c, width, height);
  const uint8_t* img = images + (size_t)image_index[i] * width * height;
  SearchResult r;
  r.code = -1;
  if (VARIANT == 1) r = search_core_mfma(img, width, width * height, patches + (size_t)i * 121, sb, a, b, c, s_pl, s_tpl);
  if (r.code < 0) r = search_core_v0(img, width, patches + (size_t)i * 121, sb, a, b, c);
  if ((threadIdx.x & 63) == 0) {
    ok[i] = r.ok;
    score[i] = r.score;
    if (r.found) { uv[i * 2] = r.u; uv[i * 2 + 1] = r.v; }  // untouched if nothing qualified (Q4)
  }
}

#ifdef SL2_SEARCH_TRACE
}  // namespace sl2
extern "C" int sl2_debug_search_trace(long long* dev_buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(sl2::g_search_trace), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : 2;
}
namespace sl2 {
#endif

int launch_search(sl2_engine* e) {
  int rc = launch_search_kernel(e);
  if (rc != SL2_OK) return rc;
  return launch_search_score(e);
}

int launch_search_kernel(sl2_engine* e) {
  {
    if (e->root->search_variant == 0) {
      LaunchScope ls(e, "k_search_exact", true);
      hipLaunchKernelGGL(k_search_exact, dim3(xcd_grid(e->nsel_max, e->B)), dim3(64), 0, e->stream, e->cur_frames, e->cur_stride,
                         e->cam.width, e->patch, e->srch_i, e->srch_d, e->sel_idx, e->n_sel, e->srch_res, e->meas_score, e->N,
                         e->nsel_max, e->B);
    } else {
      LaunchScope ls(e, "k_search_mfma", true);
      // positions per wavefront: kMfChunk when the batch fills the chip's wave slots (4096 at four per SIMD), fewer at
      // small batches - the positions of a wavefront run one after the other (18 us for four at batch 1)
      int chunk = (int)(((long long)e->B * e->nsel_max + 4095) / 4096);
      chunk = chunk < 1 ? 1 : (chunk > kMfChunk ? kMfChunk : chunk);
      if (e->root->search_chunk > 0) chunk = e->root->search_chunk;        // experiments (TEST build: SL2_SEARCH_CHUNK)
      const int nchunks = (e->nsel_max + chunk - 1) / chunk;
      // Whether k_select listed windows was decided at SELECT time (its split threshold); the list itself says so here:
      // the trailing workgroups read its length and end when it is empty.  (Deciding again from search_split at this point
      // let sl2_set_search_split(0) between the split-phase calls leave marked windows unsearched.)
      // (build_groups always allocates the list, so the helpers are always dispatched: 128 .. kSearchBigWaves single-wave
      // workgroups that read one word and exit when k_select listed nothing - sl2_set_search_split(0) included; ~1 us at batch 1)
      const bool shared = e->srch_big != nullptr;
      // trailing workgroups for the large windows' units: a quarter of the launch, 128 to kSearchBigWaves (a single sequence
      // should not pay for dispatching two thousand empty wavefronts; its frame-sized window is 38 units)
      int helpers = xcd_grid(nchunks, e->B) / 4;
      helpers = helpers < 128 ? 128 : (helpers > kSearchBigWaves ? kSearchBigWaves : helpers);
      hipLaunchKernelGGL(k_search_mfma, dim3(xcd_grid(nchunks, e->B) + (shared ? helpers : 0)), dim3(64), (size_t)e->root->search_lds_pad, e->stream, e->cur_frames, e->cur_stride,
                         e->cam.width, e->cam.width * e->cam.height, e->patch, e->srch_sel, e->n_sel, e->srch_res, e->meas_score,
                         e->N, nchunks, e->B, chunk, shared ? e->srch_big : nullptr);
    }
    SL2_HIP(hipGetLastError());
  }
  return SL2_OK;
}

int launch_search_score(sl2_engine* e) {
  {
    LaunchScope ls(e, "k_search_score");
    // One wavefront per sequence at large batches, one thread per position at small ones (latency).  The block shape matters
    // to the NEXT launch: k_build_AS at batch 1024 is exactly one round of four 5-wave workgroups per CU, and it runs 0.33-0.35 ms
    // behind a kernel of 1024 single-wave workgroups but 0.41-0.44 ms behind one of 1024 two-wave workgroups (the hardware
    // dispatcher's placement; measured on four boxes with this kernel at 64 / 128 threads and with an empty kernel in
    // between, profiles/r02_probes.txt).
    int threads = (e->nsel_max + 63) / 64 * 64;
    if (threads > 1024) threads = 1024;
    if (e->B >= 256) threads = 64;
    if (e->root->score_threads > 0) threads = e->root->score_threads;     // experiments (SL2_SCORE_THREADS)
    hipLaunchKernelGGL(k_search_score, dim3(e->B), dim3(threads), sizeof(int) * (e->N + 8), e->stream, e->srch_res, e->srch_i, e->patch, e->f_h, e->sel_idx,
                       e->n_sel, e->f_flags, e->f_z, e->f_nu, e->attempted, e->successful, e->meas_ok, e->meas_score, e->work,
                       e->succ_idx, e->f_arow, e->m_count, e->n_slots, e->pos_err, e->pos_err_any, e->f_hcol, e->ps_i, e->kpart, e->ppos, e->N, e->srch_big, e->status);
    SL2_HIP(hipGetLastError());
  }
  return SL2_OK;
}

}  // namespace sl2

extern "C" int sl2_elliptical_search_batch(int device, const uint8_t* images, int nimages, int width, int height,
                                           const int32_t* image_index, const uint8_t* patches, const double* centre,
                                           const double* puinv, int count, int32_t* ok, int32_t* uv, double* score,
                                           int variant) {
  using namespace sl2;
  if (!images || !image_index || !patches || !centre || !puinv || !ok || !uv || !score || count < 0 || nimages <= 0) {
    set_error("sl2_elliptical_search_batch: null pointer or bad count");
    return SL2_ERR_INVALID;
  }
  if (variant < 0 || variant > 1) { set_error("sl2_elliptical_search_batch: variant must be 0 (exact kernel) or 1 (matrix-core walk)"); return SL2_ERR_INVALID; }
  if (width < kBoxSize || height < kBoxSize) { set_error("sl2_elliptical_search_batch: image smaller than the 11x11 patch"); return SL2_ERR_INVALID; }
  for (int i = 0; i < count; ++i)
    if (image_index[i] < 0 || image_index[i] >= nimages) {
      set_error("sl2_elliptical_search_batch: image_index out of range");
      return SL2_ERR_INVALID;
    }
  if (count == 0) return SL2_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device (the engine has no CPU fallback)"); return SL2_ERR_NO_DEVICE; }
  SL2_HIP(hipSetDevice(device));
  // device temporaries, released on every exit path
  struct Temps {
    std::vector<void*> p;
    ~Temps() { for (void* q : p) if (q) (void)hipFree(q); }
    int alloc(void** out, size_t bytes) {
      *out = nullptr;
      const hipError_t e = hipMalloc(out, bytes);
      if (e == hipSuccess) p.push_back(*out);
      return e == hipSuccess ? SL2_OK : SL2_ERR_HIP;
    }
  } tmp;
  uint8_t *d_img = nullptr, *d_pat = nullptr;
  int *d_idx = nullptr, *d_ok = nullptr, *d_uv = nullptr;
  double *d_ce = nullptr, *d_pu = nullptr, *d_sc = nullptr;
  const size_t img_bytes = (size_t)nimages * width * height;
  if (tmp.alloc((void**)&d_img, img_bytes) || tmp.alloc((void**)&d_pat, (size_t)count * 121 + 16) ||
      tmp.alloc((void**)&d_idx, sizeof(int) * count) || tmp.alloc((void**)&d_ok, sizeof(int) * count) ||
      tmp.alloc((void**)&d_uv, sizeof(int) * 2 * count) || tmp.alloc((void**)&d_ce, sizeof(double) * 2 * count) ||
      tmp.alloc((void**)&d_pu, sizeof(double) * 3 * count) || tmp.alloc((void**)&d_sc, sizeof(double) * count)) {
    set_error("sl2_elliptical_search_batch: device allocation failed");
    return SL2_ERR_HIP;
  }
  SL2_HIP(hipMemcpy(d_img, images, img_bytes, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pat, patches, (size_t)count * 121, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_idx, image_index, sizeof(int) * count, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_uv, uv, sizeof(int) * 2 * count, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_ce, centre, sizeof(double) * 2 * count, hipMemcpyHostToDevice));
  SL2_HIP(hipMemcpy(d_pu, puinv, sizeof(double) * 3 * count, hipMemcpyHostToDevice));
  if (variant == 0)
    hipLaunchKernelGGL(k_search_batch<0>, dim3(count), dim3(64), 0, 0, d_img, width, height, d_idx, d_pat, d_ce, d_pu, d_ok, d_uv, d_sc);
  else
    hipLaunchKernelGGL(k_search_batch<1>, dim3(count), dim3(64), 0, 0, d_img, width, height, d_idx, d_pat, d_ce, d_pu, d_ok, d_uv, d_sc);
  SL2_HIP(hipGetLastError());
  SL2_HIP(hipDeviceSynchronize());
  SL2_HIP(hipMemcpy(ok, d_ok, sizeof(int) * count, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(uv, d_uv, sizeof(int) * 2 * count, hipMemcpyDeviceToHost));
  SL2_HIP(hipMemcpy(score, d_sc, sizeof(double) * count, hipMemcpyDeviceToHost));
  return SL2_OK;
}
