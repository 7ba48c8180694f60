// Feature initialisation inside the per-frame step: the tail of
// MonoSLAM::GoOneStep (monoslam.cpp:152-170) for the whole batch, up to kMaxPartial partially initialised
// features per sequence (params.max_features_to_init_at_once; the shipped value is 1):
//
//   k_map_find       speed gate + AutoInitialiseFeature's region choice     monoslam.cpp:159-165, 823-1032
//                    + set_image_selection_automatically (Shi-Tomasi)       :1043-1205
//   k_map_create     InitialiseFeature + partially-initialised Feature ctor :1211-1276, feature.cpp:45-104
//   k_map_particles  predict_partially_initialised_feature_measurements     :1349-1401
//   k_map_me_search  measure_feature_with_multiple_priors                   :1411-1439 (+ improc/search_multiple...)
//   k_map_update     update_partially_initialised_feature_probabilities, conversion test,
//                    convert_from_partially_to_fully_initialised, sell-by deletion, trajectory push
//                                                                           :1299-1342, 1449-1538, feature.cpp:204-269
//   k_map_finish     k_map_create + k_map_update in one launch (one sequence, no partial feature at the start of the frame)
// Which of them a step consists of: launch_mapping, from what the host knows about the partial features (parts_state).
//
// State layout: the six states (r_W, hhat_W) of the partial feature in partial slot k live in columns ppos + 6 k ..
// ppos + 6 k + 5 of x / P, ppos = 13 + 3N (behind the map: the update's algebra does not care where a state sits, and the
// accessors put them back at their feature's place in feature_list_ order); its label slot is reserved at creation
// (next_free_label_++) and receives the 3-D point at conversion.  A deleted or converted partial feature leaves its six
// rows / columns zero.  feature_init_info_vector_'s order (creation order, entries erased on conversion / deletion) is the
// list part_i[kPartOrder ..]: the reference walks that vector with its erase-inside-the-loop quirks, and so do we.
#include "sl2_improc_dev.hpp"
#include "sl2_mapmath.hpp"

namespace sl2 {

struct MapParams {
  int enable_mapping, save_trajectory;
  int force;      // InitialiseAutoFeature (monoslam.cpp:1535-1541): no speed gate, no visible-feature count
  int keep_visible, n_particles, min_particles, erase_after;
  double min_lambda, max_lambda, sd_ratio, prune_threshold, dt;
  int pcap;       // particle slots per partial feature in `particles` / `me_desc` (sl2_engine::pcap)
  int kpart;      // partial slots per sequence (sl2_engine::kpart)
  // One-sequence engines (launch_mapping): k_map_update reports how many partially initialised features the step leaves, and a
  // step issued while the report says "none" runs without k_map_particles / k_map_me_search / k_me_big.
  int parts_skipped = 0;    // this step runs without them: k_map_create does the one thing k_map_particles does to a feature of this frame
  int publish_parts = 0;    // k_map_update writes (steps completed << 32) | partial features left to parts_mail
};

// ---------------------------------------------------------------------------
// k_map_find = FindNonOverlappingRegion + the detector over the region it picked, one workgroup of kDetThreads per sequence
// (rounds 1-5: k_map_region and k_map_detect, two launches - at one sequence each launch is a link of the frame's chain).
// region_body is the first wavefront's work; the other waves only keep its barriers company (every thread of the workgroup
// calls it).  On return the region, or "no region", is in part_i for every thread to read.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void region_body(const double* __restrict__ x, const int* __restrict__ f_flags,
                                            const int* __restrict__ n_slots, const int* __restrict__ n_vis,
                                            const double* __restrict__ prev_r, int* __restrict__ part_i,
                                            unsigned long long* __restrict__ rand48, double* __restrict__ last_r,
                                            int* __restrict__ status, const CameraParams& cam, const MapParams& mp, int N, int ld,
                                            double* s_uv /* [2 * N] projections of the known features in front of the camera */) {
  __shared__ int s_cnt, s_go, s_safe[4];
  const int b = blockIdx.x, lane = threadIdx.x;
  const bool first_wave = lane < 64;
  const double* xb = x + (size_t)b * ld;
  int* pi = part_i + (size_t)b * kPartInts;
  const int ns = n_slots[b];
  if (lane == 0) {
    pi[kPartRegionValid] = 0;
    pi[kPartCreated] = 0;
    s_cnt = 0;
    s_go = 0;
    // camera speed estimate (monoslam.cpp:152-157)
    const double vx = (xb[0] - prev_r[b * 3 + 0]) / mp.dt, vy = (xb[1] - prev_r[b * 3 + 1]) / mp.dt,
                 vz = (xb[2] - prev_r[b * 3 + 2]) / mp.dt;
    const double speed = sqrt(vx * vx + vy * vy + vz * vz);
    // feature_init_info_vector_.size() < kMaxFeaturesToInitAtOnce_ (monoslam.cpp:163-165; the buttons have no such gate in the
    // reference, but a sequence has only kpart partial slots here)
    if ((mp.force || (speed > 0.2 && mp.enable_mapping && n_vis[b] < mp.keep_visible)) && pi[kPartCount] < mp.kpart) {
      if (ns >= N) {
        status[b] |= 2;            // the map is full: cannot reserve a label (capacity chosen at sl2_create)
      } else {
        status[b] &= ~2;           // (the bit tells about the LAST attempt: room again after deletions / the slot squeeze)
        // FindNonOverlappingRegion (:867-943): where will the image centre be in ten steps?
        double x0[13], xv[13];
        for (int i = 0; i < 13; ++i) x0[i] = xb[i];
        motion_f_repeated(x0, mp.dt, 10, xv);
        double R[9], yW[3], xp[7], zeroed[3], h[2], Hx[14], Hy[6], Rn;
        quat_to_rot(&xv[3], R);
        for (int i = 0; i < 3; ++i) {
          double acc = 0.0;
          acc += R[i * 3 + 0] * 0.0;
          acc += R[i * 3 + 1] * 0.0;
          acc += R[i * 3 + 2] * 2.5;   // FEATURE_INIT_DEPTH_HYPOTHESIS
          yW[i] = xv[i] + acc;
        }
        for (int i = 0; i < 7; ++i) xp[i] = xb[i];
        measurement_model(cam, xp, yW, zeroed, h, Hx, Hy, &Rn);
        const double pmu = cam.width / 2.0 - h[0], pmv = cam.height / 2.0 - h[1];
        int sus = (int)(-pmu), svs = (int)(-pmv), suf = (int)(cam.width - pmu), svf = (int)(cam.height - pmv);
        const int m = (kBoxSize - 1) / 2 + 1;
        if (sus < m) sus = m;
        if (suf > cam.width - m) suf = cam.width - m;
        if (svs < m) svs = m;
        if (svf > cam.height - m) svf = cam.height - m;
        s_safe[0] = sus; s_safe[1] = svs; s_safe[2] = suf; s_safe[3] = svf;
        // rRES_ now holds the CURRENT position (func_hi -> func_zeroedyi -> func_r), Q12
        for (int k = 0; k < 3; ++k) last_r[b * 3 + k] = xb[k];
        s_go = (suf - sus > 80 && svf - svs > 60) ? 1 : 0;   // :957-958
      }
    }
  }
  __syncthreads();
  if (!s_go) return;
  // image positions of the fully initialised features in front of the camera (:968-985)
  if (first_wave) {
    double xp[7];
    for (int i = 0; i < 7; ++i) xp[i] = xb[i];
    for (int i = lane; i < ns; i += 64) {
      if (!(f_flags[(size_t)b * N + i] & FF_ACTIVE)) continue;
      double y[3], zeroed[3], h[2], Hx[14], Hy[6], Rn;
      for (int k = 0; k < 3; ++k) y[k] = xb[13 + 3 * i + k];
      measurement_model(cam, xp, y, zeroed, h, Hx, Hy, &Rn);
      if (zeroed[2] > 0) {
        const int k = atomicAdd(&s_cnt, 1);
        s_uv[2 * k] = h[0];
        s_uv[2 * k + 1] = h[1];
      }
    }
  }
  __syncthreads();
  if (lane == 0) {
    unsigned long long st = rand48[b];
    const int sus = s_safe[0], svs = s_safe[1], suf = s_safe[2], svf = s_safe[3];
    int us = 0, vs = 0, uf = 0, vf = 0, i = 0;
    while (i < 5) {   // NUMBER_OF_RANDOM_INIT_FEATURE_SEARCH_REGION_TRIES
      const int u_offset = int((suf - sus - 80) * rand48_next(&st));
      const int v_offset = int((svf - svs - 60) * rand48_next(&st));
      us = sus + u_offset; uf = us + 80;
      vs = svs + v_offset; vf = vs + 60;
      bool found = false;
      for (int k = 0; k < s_cnt; ++k) {
        const double fu = s_uv[2 * k], fv = s_uv[2 * k + 1];
        if (fu >= us - 10 && fu < uf + 10 && fv >= vs - 10 && fv < vf + 10) { found = true; break; }
      }
      if (!found) break;
      ++i;
    }
    rand48[b] = st;
    pi[kPartRegion + 0] = us; pi[kPartRegion + 1] = vs; pi[kPartRegion + 2] = uf; pi[kPartRegion + 3] = vf;
    pi[kPartRegionValid] = (i != 5) ? 1 : 0;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(kDetThreads) k_map_find(const double* __restrict__ x, const int* __restrict__ f_flags,
                                                          const int* __restrict__ n_slots, const int* __restrict__ n_vis,
                                                          const double* __restrict__ prev_r, int* __restrict__ part_i,
                                                          double* __restrict__ part_d, unsigned long long* __restrict__ rand48,
                                                          double* __restrict__ last_r, int* __restrict__ status,
                                                          const uint8_t* __restrict__ frames, size_t seq_stride, CameraParams cam,
                                                          MapParams mp, int N, int ld) {
  extern __shared__ double s_uv[];
  region_body(x, f_flags, n_slots, n_vis, prev_r, part_i, rand48, last_r, status, cam, mp, N, ld, s_uv);
  const int b = blockIdx.x;
  int* pi = part_i + (size_t)b * kPartInts;
  if (!pi[kPartRegionValid]) return;
  detect_region_wg(frames + (size_t)b * seq_stride, cam.width, cam.height, pi[kPartRegion + 0], pi[kPartRegion + 1], pi[kPartRegion + 2],
                   pi[kPartRegion + 3], pi + kPartUU, part_d + (size_t)b * kPartDoubles + 2);
}

// ---------------------------------------------------------------------------
// k_map_create: one wavefront per sequence.  The arguments of the feature-initialisation tail's per-sequence kernels travel
// in one struct: create_body and update_body run as kernels of their own, or one after the other in k_map_finish.
// ---------------------------------------------------------------------------
struct MapArrays {
  double *x, *P;
  const uint8_t* frames; size_t seq_stride;
  uint8_t* patch; int* patch_sums; double* xp_org;
  int *f_flags, *n_slots, *attempted, *successful, *f_label, *next_label, *part_i;
  double* part_d; int* ps_i; double* ps_d; int *pos_err, *pos_err_any;
  double *particles, *last_r, *traj; int* traj_count; const int* pos_count; unsigned long long* parts_mail;
  int N, ld, ppos0;
};

__device__ __forceinline__ void create_body(const MapArrays& a, const CameraParams& cam, const MapParams& mp) {
  double* __restrict__ x = a.x; double* __restrict__ P = a.P; const uint8_t* __restrict__ frames = a.frames;
  const size_t seq_stride = a.seq_stride;
  uint8_t* __restrict__ patch = a.patch; int* __restrict__ patch_sums = a.patch_sums; double* __restrict__ xp_org = a.xp_org;
  int* __restrict__ f_flags = a.f_flags; int* __restrict__ n_slots = a.n_slots; int* __restrict__ attempted = a.attempted;
  int* __restrict__ successful = a.successful; int* __restrict__ f_label = a.f_label; int* __restrict__ next_label = a.next_label;
  int* __restrict__ part_i = a.part_i; double* __restrict__ part_d = a.part_d; int* __restrict__ ps_i = a.ps_i;
  double* __restrict__ ps_d = a.ps_d; int* __restrict__ pos_err = a.pos_err; double* __restrict__ particles = a.particles;
  double* __restrict__ last_r = a.last_r;
  const int N = a.N, ld = a.ld, ppos0 = a.ppos0;
  const int b = blockIdx.x, lane = threadIdx.x;
  int* pi = part_i + (size_t)b * kPartInts;
  double* pd = part_d + (size_t)b * kPartDoubles;
  if (!pi[kPartRegionValid]) return;
  if (!(pd[2] > 20000)) return;        // SUITABLE_PATCH_SCORE_THRESHOLD (:837, 850-858)
  // the partial slot this feature takes: the first free one (k_map_find / k_map_manual checked that there is one)
  int* psb = ps_i + (size_t)b * mp.kpart * kPsInts;
  int ks = 0;
  while (ks < mp.kpart - 1 && psb[ks * kPsInts + kPsActive]) ++ks;
  if (psb[ks * kPsInts + kPsActive]) return;
  const int ppos = ppos0 + 6 * ks;
  // (`label` below is the SLOT the feature takes - the end of the list; its Feature::label_ comes from next_label)
  double* xb = x + (size_t)b * ld;
  double* Pb = P + (size_t)b * ld * ld;
  const int label = n_slots[b];
  const int uu = pi[kPartUU], vv = pi[kPartVV];
  __shared__ double s_T[6 * 13], s_D[12], s_col[6 * 13], s_y[6], s_Ri;
  __shared__ int s_extra[6 * kMaxPartial], s_nextra;
  if (lane == 0) {
    double xp[7], ypi[6], Tq[12], Dh[6], Ri;
    for (int i = 0; i < 7; ++i) xp[i] = xb[i];
    const double hi[2] = {(double)uu, (double)vv};
    part_create_model(cam, xp, hi, ypi, Tq, Dh, &Ri);
    for (int i = 0; i < 6 * 13; ++i) s_T[i] = 0.0;
    for (int k = 0; k < 3; ++k) s_T[k * 13 + k] = 1.0;
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 4; ++j) s_T[(3 + k) * 13 + 3 + j] = Tq[k * 4 + j];
    for (int i = 0; i < 12; ++i) s_D[i] = 0.0;
    for (int k = 0; k < 3; ++k) { s_D[(3 + k) * 2 + 0] = Dh[k * 2 + 0]; s_D[(3 + k) * 2 + 1] = Dh[k * 2 + 1]; }
    for (int i = 0; i < 6; ++i) s_y[i] = ypi[i];
    s_Ri = Ri;
    // the other partially initialised features are earlier entries of feature_list_ too: their six columns get cross terms
    int ne = 0;
    for (int j = 0; j < mp.kpart; ++j)
      if (j != ks && psb[j * kPsInts + kPsActive])
        for (int k = 0; k < 6; ++k) s_extra[ne++] = ppos0 + 6 * j + k;
    s_nextra = ne;
  }
  __syncthreads();
  // new columns: P[c][ppos + k] = sum_i T[k][i] P[i][c]  (Pxy = Pxx T^T and (T Pxy_j)^T, feature.cpp:82-103)
  const int n_rows = 13 + 3 * label;
  for (int cc = lane; cc < n_rows + s_nextra; cc += 64) {
    const int c = cc < n_rows ? cc : s_extra[cc - n_rows];
    double pc[13];
    for (int i = 0; i < 13; ++i) pc[i] = (c < 13) ? Pb[(size_t)c * ld + i] : Pb[(size_t)i * ld + c];
    for (int k = 0; k < 6; ++k) {
      double acc = 0.0;
      for (int i = 0; i < 13; ++i) acc += pc[i] * s_T[k * 13 + i];
      Pb[(size_t)c * ld + ppos + k] = acc;
      Pb[(size_t)(ppos + k) * ld + c] = acc;
      if (c < 13) s_col[k * 13 + c] = acc;
    }
  }
  __syncthreads();
  // (From here on lane 0 used to work alone - 121 template pixels one dependent load after the other, 1200 particle
  // numbers, the 36 entries of Pyy: 20 of the kernel's 26 us.  Every piece is spread over the lanes now, each entry still
  // formed by one lane in the reference's order.)
  const size_t fi = (size_t)b * N + label;
  if (lane < 36) {
    // Pyy = (T Pxx) T^T + (D Ri) D^T
    const int k = lane / 6, l = lane % 6;
    double a1 = 0.0;
    for (int j = 0; j < 13; ++j) a1 += s_col[k * 13 + j] * s_T[l * 13 + j];
    double a2 = 0.0;
    for (int c = 0; c < 2; ++c) a2 += (s_D[k * 2 + c] * s_Ri) * s_D[l * 2 + c];
    Pb[(size_t)(ppos + k) * ld + ppos + l] = a1 + a2;
  }
  // template: copy_into_patch (:1240-1251) + the packed form the search kernels read
  __shared__ int s_pix[121];
  __shared__ double s_lambda[kMaxParticles];
  {
    const uint8_t* img = frames + (size_t)b * seq_stride;
    uint8_t* pt = patch + fi * kPatchStride;
    const int p0 = lane, p1 = lane + 64;
    const int g0 = img[(size_t)(p0 / 11 + vv - 5) * cam.width + p0 % 11 + uu - 5];
    const int g1 = p1 < 121 ? img[(size_t)(p1 / 11 + vv - 5) * cam.width + p1 % 11 + uu - 5] : 0;
    pt[p0] = (uint8_t)g0; s_pix[p0] = g0;
    if (p1 < 121) { pt[p1] = (uint8_t)g1; s_pix[p1] = g1; }
    if (lane < kPatchPackedOffset - 121) pt[121 + lane] = 0;
    int s0 = g0 + g1, s0sq = g0 * g0 + g1 * g1;             // (integer sums: the order is immaterial)
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off, 64); s0sq += __shfl_xor(s0sq, off, 64); }
    if (lane == 0) {
      // particle depths: lambda_i by repeated addition, as the reference forms them (:1222-1236)
      const double lambda_step = (1.0 / double(mp.n_particles)) * (mp.max_lambda - mp.min_lambda);
      double lambda = mp.min_lambda;
      for (int i = 0; i < mp.n_particles; ++i) { s_lambda[i] = lambda; lambda += lambda_step; }
    }
    __syncthreads();
    unsigned* packed = (unsigned*)(pt + kPatchPackedOffset);
    if (lane < 33) {
      const int r = lane / 3, d = lane % 3;
      unsigned v = 0;
      for (int k = 0; k < 4; ++k) {
        const int col = 4 * d + k;
        if (col < 11) v |= (unsigned)s_pix[r * 11 + col] << (8 * k);
      }
      packed[lane] = v;
    } else if (lane >= 36 && lane < (kPatchStride - kPatchPackedOffset) / 4) {
      packed[lane] = 0u;
    }
    if (lane == 0) {
      const double g0bar = (double)s0 / 121.0;
      const double varg0 = (double)s0sq / 121.0 - (g0bar * g0bar);
      const double sigmag0 = sqrt(varg0);
      packed[33] = (unsigned)s0; packed[34] = (unsigned)s0sq;
      packed[35] = (sigmag0 < kCorrelationSigmaThreshold) ? 0u : 1u;
      patch_sums[fi * 2] = s0; patch_sums[fi * 2 + 1] = s0sq;
    }
  }
  {
    // particle set: uniform prior over [min_lambda, max_lambda) (:1222-1236)
    const double uniform_probability = 1.0 / double(mp.n_particles);
    double* pp = particles + ((size_t)b * mp.kpart + ks) * mp.pcap * kParticleDoubles;
    for (int e = lane; e < mp.n_particles * kParticleDoubles; e += 64) {
      const int i = e / kParticleDoubles, k = e - i * kParticleDoubles;
      pp[e] = k == 0 ? s_lambda[i] : (k == 1 ? uniform_probability : 0.0);
    }
  }
  if (lane == 0) {
    for (int i = 0; i < 6; ++i) xb[ppos + i] = s_y[i];
    // label slot: reserved, not yet a 3-D point
    for (int k = 0; k < 7; ++k) xp_org[fi * 8 + k] = xb[k];
    xp_org[fi * 8 + 7] = 0.0;
    f_flags[fi] = FF_USED | FF_PARTIAL;
    attempted[fi] = 0; successful[fi] = 0;
    pos_err[fi] = 0;
    f_label[fi] = next_label[b];           // label_ = next_free_label_++ (monoslam.cpp:1306-1307)
    next_label[b] += 1;
    n_slots[b] = label + 1;
    int* ps = psb + ks * kPsInts;
    // (a feature made in this frame is not matched in it, Q29: all k_map_particles does to it is number_of_match_attempts_++)
    ps[kPsActive] = 1; ps[kPsLabel] = label; ps[kPsAttempts] = mp.parts_skipped ? 1 : 0; ps[kPsNp] = mp.n_particles; ps[kPsMaking] = 0;
    pi[kPartOrder + pi[kPartCount]] = ks;          // feature_init_info_vector_.push_back
    pi[kPartCount] += 1;
    pi[kPartCreated] = 1;
    pi[kPartInitialised] += 1;
    ps_d[((size_t)b * mp.kpart + ks) * kPsDoubles + 0] = 0.0; ps_d[((size_t)b * mp.kpart + ks) * kPsDoubles + 1] = 0.0;
    for (int k = 0; k < 3; ++k) last_r[b * 3 + k] = xb[k];
  }
}

__global__ void __launch_bounds__(64) k_map_create(MapArrays a, CameraParams cam, MapParams mp) { create_body(a, cam, mp); }

// ---------------------------------------------------------------------------
// k_map_particles: one workgroup per sequence, one thread per particle.
// ---------------------------------------------------------------------------
// THREADS = the launch's block size (particle slots of the engine, a multiple of 64): the shipped 100 particles run at
// __launch_bounds__(128) - the thread holds about 130 doubles of covariance blocks and Jacobians, which a 1024-thread bound
// (128 registers) spills - and only engines created for more particles take the wider, spilling instantiations.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_map_particles(const double* __restrict__ x, const double* __restrict__ P,
                                                                 int* __restrict__ ps_i, double* __restrict__ particles,
                                                                 int* __restrict__ me_desc, double* __restrict__ last_r,
                                                                 int* __restrict__ me_big_count, int* __restrict__ part_i, int clear_region,
                                                                 CameraParams cam, int ld, int ppos0, int pcap, int kpart) {
  const int b = blockIdx.x, ks = blockIdx.y, tid = threadIdx.x;       // one workgroup per (sequence, partial slot)
  int* ps = ps_i + ((size_t)b * kpart + ks) * kPsInts;
  if (b == 0 && ks == 0 && tid == 0) *me_big_count = 0;     // the step's list of oversized multi-ellipse searches starts empty
  if (clear_region && ks == 0 && tid == 0) {                // this step runs without k_map_find (launch_mapping: parts_state 2): its two per-step flags
    part_i[(size_t)b * kPartInts + kPartRegionValid] = 0;
    part_i[(size_t)b * kPartInts + kPartCreated] = 0;
  }
  if (!ps[kPsActive]) return;
  __shared__ int s_making;
  if (tid == 0) {
    const int att = ps[kPsAttempts];
    ps[kPsAttempts] = att + 1;                  // number_of_match_attempts_++ != 0  (Q29)
    s_making = (att != 0) ? 1 : 0;
    ps[kPsMaking] = s_making;
  }
  __syncthreads();
  if (!s_making) return;
  const int ppos = ppos0 + 6 * ks;
  const double* xb = x + (size_t)b * ld;
  const double* Pb = P + (size_t)b * ld * ld;
  if (tid == 0)
    for (int k = 0; k < 3; ++k) last_r[b * 3 + k] = xb[k];   // func_zeroedyi -> func_r(xp), Q12
  const int np = ps[kPsNp];
  if (tid >= np) return;
  double xp[7], ypi[6];
  for (int i = 0; i < 7; ++i) xp[i] = xb[i];
  for (int i = 0; i < 6; ++i) ypi[i] = xb[ppos + i];
  double* o = particles + (((size_t)b * kpart + ks) * pcap + tid) * kParticleDoubles;
  double h[2], Hx[14], Hy[12], Rn;
  part_measurement_model(cam, xp, ypi, o[0], h, Hx, Hy, &Rn);
  double Pxx7[49], Pxy7[42], Pyy[36], S[4];
  for (int r = 0; r < 7; ++r) {
    for (int c = 0; c < 7; ++c) Pxx7[r * 7 + c] = Pb[(size_t)r * ld + c];
    for (int c = 0; c < 6; ++c) Pxy7[r * 6 + c] = Pb[(size_t)r * ld + ppos + c];
  }
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) Pyy[r * 6 + c] = Pb[(size_t)(ppos + r) * ld + ppos + c];
  innovation_cov6(Hx, Hy, Rn, Pxx7, Pxy7, Pyy, S);
  double a, bq, c;
  sinv_from_S(S, &a, &bq, &c);
  o[3] = h[0]; o[4] = h[1];
  o[7] = a; o[8] = bq; o[9] = c;
  o[10] = det2_partial_pivot_lu(S);
  me_describe(a, bq, c, h[0], h[1], cam.width, cam.height, me_desc + (((size_t)b * kpart + ks) * pcap + tid) * 8);
}

// measure_feature_with_multiple_priors (monoslam.cpp:1411-1439) + SearchMultipleOverlappingEllipses::search: the whole
// multi-ellipse search of a sequence's partially initialised feature by one workgroup, stamps and scores of the union's
// bounding box in LDS (me_search_fused_wg; rounds 1-3 ran three launches - k_map_me_mark, k_map_me_scores, k_map_me_argmin -
// through image-sized maps in HBM, walking every ellipse's box column by column: 1.3 ms of a 1.35 ms mapping step at batch
// 1024, profiles/r04_mapping_*).  A sequence whose union is too large for that goes on the list of k_me_big_*.
struct MeJobsEngine {      // job = sequence * kpart + partial slot
  const uint8_t* frames; size_t seq_stride; const uint8_t* patch_base; const int* ps_i; const int* me_desc; double* particles;
  double* map_base; int N, pcap, width, height, kpart;
  __device__ const int* ps(int j) const { return ps_i + (size_t)j * kPsInts; }
  __device__ const uint8_t* img(int j) const { return frames + (size_t)(j / kpart) * seq_stride; }
  __device__ const uint8_t* patch(int j) const { return patch_base + ((size_t)(j / kpart) * N + ps(j)[kPsLabel]) * kPatchStride; }
  __device__ const int* desc(int j) const { return me_desc + (size_t)j * pcap * 8; }
  __device__ int n_ell(int j) const { return ps(j)[kPsNp]; }
  __device__ const double* pu(int j, int e) const { return particles + ((size_t)j * pcap + e) * kParticleDoubles + 7; }
  __device__ double* map(int j) const { return map_base + (size_t)j * width * height; }
  __device__ void emit(int j, int e, int flag, int u, int v, double) const {
    double* o = particles + ((size_t)j * pcap + e) * kParticleDoubles;
    if (flag) {           // the measurement is stored only on success (:1429-1437)
      o[5] = (double)u;
      o[6] = (double)v;
      o[11] = 1.0;
    } else {
      o[11] = 0.0;
    }
  }
};
__global__ void __launch_bounds__(1024) k_map_me_search(MeJobsEngine J, int* __restrict__ big_list, int* __restrict__ big_count) {
  const int j = blockIdx.x;
  const int* ps = J.ps(j);
  if (!ps[kPsActive] || !ps[kPsMaking]) return;
  const bool done = me_search_fused_wg(J.img(j), J.width, J.patch(j), J.desc(j), J.n_ell(j), [&](int e) { return J.pu(j, e); },
                                       [&](int e, int flag, int u, int v, double best) { J.emit(j, e, flag, u, v, best); });
  if (!done && threadIdx.x == 0) big_list[atomicAdd(big_count, 1)] = j;
}

// ---------------------------------------------------------------------------
// k_map_update: one wavefront per sequence; the rest of MatchPartiallyInitialisedFeatures (monoslam.cpp:1299-1342) over the
// sequence's partially initialised features IN THE ORDER OF feature_init_info_vector_ (part_i[kPartOrder ..]):
//   A  update_partially_initialised_feature_probabilities (:1449-1497): Bayes update, normalise, prune, normalise, mean /
//      covariance per feature; a feature whose matches all failed is deleted - inside a `for (; feat < end; ++feat)` loop, so
//      the entry that slides into its place is SKIPPED in this pass (it keeps last frame's weights, mean and covariance);
//   B  the conversion test (:1316-1333) on every entry that was measured, erase(feat--) + ++feat: nothing is skipped;
//   C  delete_partially_initialised_features_past_sell_by_date (:1506-1521).
// Per feature the particle list lives in LDS: the per-particle arithmetic (likelihood, division by the total, pruning test,
// compaction) runs on all lanes; every SUM is formed by lane 0 in list order, which is the reference's order (the sums
// decide the bits of the weights, and through them pruning and conversion).  The covariance surgery of a conversion /
// deletion is done by all lanes.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void update_body(const MapArrays& a, const MapParams& mp, double* s_p /* [number_of_particles][kParticleDoubles] */) {
  double* __restrict__ x = a.x; double* __restrict__ P = a.P; int* __restrict__ f_flags = a.f_flags;
  const int* __restrict__ n_slots = a.n_slots; int* __restrict__ part_i = a.part_i; int* __restrict__ ps_i = a.ps_i;
  double* __restrict__ ps_d = a.ps_d; int* __restrict__ pos_err = a.pos_err; int* __restrict__ pos_err_any = a.pos_err_any;
  double* __restrict__ particles = a.particles; double* __restrict__ traj = a.traj; int* __restrict__ traj_count = a.traj_count;
  const double* __restrict__ last_r = a.last_r; const int* __restrict__ pos_count = a.pos_count;
  unsigned long long* __restrict__ parts_mail = a.parts_mail;
  const int N = a.N, ld = a.ld, ppos0 = a.ppos0;
  const int b = blockIdx.x, lane = threadIdx.x;
  int* pi = part_i + (size_t)b * kPartInts;
  double* xb = x + (size_t)b * ld;
  double* Pb = P + (size_t)b * ld * ld;
  const int ns = n_slots[b];
  __shared__ int s_count, s_order[kMaxPartial];
  __shared__ int s_flag;           // scratch decision of lane 0
  __shared__ double s_total;
  // Everything the passes below read - the records, the particle lists (k_map_me_search left them in another XCD's cache),
  // the trajectory cursor - is touched HERE, by all lanes at once: the passes are a chain of small dependent reads, each a
  // round trip to memory when it is the first to meet its line (13-15 us for one sequence's kernel), a hit behind this one.
  {
    int wi = 0;
    double wd = 0.0;
    if (lane < kPartInts) wi += pi[lane];
    for (int ks = 0; ks < mp.kpart; ++ks) {
      if (lane < kPsInts) wi += ps_i[((size_t)b * mp.kpart + ks) * kPsInts + lane];
      if (lane < kPsDoubles) wd += ps_d[((size_t)b * mp.kpart + ks) * kPsDoubles + lane];
      const double* pp = particles + ((size_t)b * mp.kpart + ks) * mp.pcap * kParticleDoubles;
      for (int i = lane * 16; i < mp.n_particles * kParticleDoubles; i += 64 * 16) wd += pp[i];      // one read per 128-byte line
    }
    if (lane == 0) { wi += traj_count[b] + pos_count[b] + n_slots[b]; wd += last_r[b * 3]; }
    asm volatile("" ::"v"(wi), "v"(wd));
  }
  if (lane == 0) {
    s_count = pi[kPartCount];
    for (int k = 0; k < kMaxPartial; ++k) s_order[k] = pi[kPartOrder + k];
  }
  __syncthreads();

  // vector::erase(begin + idx)
  auto erase_entry = [&](int idx) {
    if (lane == 0) {
      for (int k = idx; k + 1 < s_count; ++k) s_order[k] = s_order[k + 1];
      s_count -= 1;
    }
    __syncthreads();
  };
  // the six rows / columns of partial slot ks are released (zero = absent), its record cleared
  auto release_slot = [&](int ks) {
    const int ppos = ppos0 + 6 * ks;
    for (int c = lane; c < ld; c += 64)
      for (int k = 0; k < 6; ++k) {
        Pb[(size_t)c * ld + ppos + k] = 0.0;
        Pb[(size_t)(ppos + k) * ld + c] = 0.0;
      }
    if (lane == 0) {
      for (int k = 0; k < 6; ++k) xb[ppos + k] = 0.0;
      int* ps = ps_i + ((size_t)b * mp.kpart + ks) * kPsInts;
      ps[kPsActive] = 0; ps[kPsMaking] = 0; ps[kPsNp] = 0;
    }
    __syncthreads();
  };
  // delete_partially_initialised_feature (:1523-1538): delete_feature() consumes the label, the entry leaves the vector
  auto delete_partial = [&](int idx) {
    const int ks = s_order[idx];
    if (lane == 0) {
      const int label = ps_i[((size_t)b * mp.kpart + ks) * kPsInts + kPsLabel];
      f_flags[(size_t)b * N + label] = FF_USED;
      pi[kPartDeleted] += 1;
    }
    release_slot(ks);
    erase_entry(idx);
  };

  // ---- A: update_partially_initialised_feature_probabilities
  for (int idx = 0; idx < s_count; ++idx) {
    const int ks = s_order[idx];
    int* ps = ps_i + ((size_t)b * mp.kpart + ks) * kPsInts;
    double* pd = ps_d + ((size_t)b * mp.kpart + ks) * kPsDoubles;
    if (!ps[kPsMaking]) continue;
    double* pp = particles + ((size_t)b * mp.kpart + ks) * mp.pcap * kParticleDoubles;
    int np = ps[kPsNp];
    __syncthreads();                               // (the previous feature is done with s_p)
    for (int i = lane; i < np * kParticleDoubles; i += 64) s_p[i] = pp[i];
    __syncthreads();
    for (int i = lane; i < np; i += 64) {
      double* o = s_p + i * kParticleDoubles;
      double likelihood = 0.0;
      if (o[11] != 0.0) likelihood = particle_likelihood(o + 5, o + 3, o + 7, o[10]);
      o[1] = o[1] * likelihood;
    }
    __syncthreads();
    if (lane == 0) {
      double total = 0.0;
      for (int i = 0; i < np; ++i) total += s_p[i * kParticleDoubles + 1];
      s_total = total;
      s_flag = (total == 0.0) ? 1 : 0;             // every match failed: the feature goes (:1490-1494)
    }
    __syncthreads();
    if (s_flag) {
      delete_partial(idx);                         // the loop's ++idx now steps over the entry that moved into this place
      continue;
    }
    double total = s_total;
    for (int i = lane; i < np; i += 64) s_p[i * kParticleDoubles + 1] = s_p[i * kParticleDoubles + 1] / total;
    __syncthreads();
    // (the cumulative column is rewritten after pruning; only its final values are ever read)
    // prune_particle_vector (feature_init_info.cpp:131-147): order-preserving compaction, 64 particles per round; a
    // round's destinations lie below its own sources and below every later round's sources
    const double prune_threshold = mp.prune_threshold / double(np);
    int kept = 0;
    for (int i0 = 0; i0 < np; i0 += 64) {
      const int i = i0 + lane;
      double v[kParticleDoubles];
      bool keep = false;
      if (i < np) {
#pragma unroll
        for (int k = 0; k < kParticleDoubles; ++k) v[k] = s_p[i * kParticleDoubles + k];
        keep = !(v[1] < prune_threshold);
      }
      const unsigned long long mask = __ballot(keep);
      __syncthreads();
      if (keep) {
        const int dst = kept + __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
        for (int k = 0; k < kParticleDoubles; ++k) s_p[dst * kParticleDoubles + k] = v[k];
      }
      kept += __popcll(mask);
      __syncthreads();
    }
    np = kept;
    if (lane == 0) {
      double tot2 = 0.0;
      for (int i = 0; i < np; ++i) tot2 += s_p[i * kParticleDoubles + 1];
      s_total = tot2;
    }
    __syncthreads();
    total = s_total;
    if (total != 0.0)
      for (int i = lane; i < np; i += 64) s_p[i * kParticleDoubles + 1] = s_p[i * kParticleDoubles + 1] / total;
    __syncthreads();
    if (lane == 0) {
      double cum = 0.0;
      for (int i = 0; i < np; ++i) {
        cum += s_p[i * kParticleDoubles + 1];
        s_p[i * kParticleDoubles + 2] = cum;
      }
      // (a zero total after pruning would need a zero pruning threshold, which prunes nothing: not reachable)
      // calculate_mean_and_covariance (feature_init_info.cpp:157-174)
      double mean = 0.0, e2 = 0.0;
      for (int i = 0; i < np; ++i) {
        const double* o = s_p + i * kParticleDoubles;
        mean += o[1] * o[0];
        e2 += o[1] * (o[0] * o[0]);
      }
      pd[0] = mean;
      pd[1] = e2 - (mean * mean);
      ps[kPsNp] = np;
    }
    __syncthreads();
    for (int i = lane; i < np * kParticleDoubles; i += 64) pp[i] = s_p[i];
  }
  __syncthreads();

  // ---- B: conversion (:1316-1333)
  for (int idx = 0; idx < s_count;) {
    const int ks = s_order[idx];
    int* ps = ps_i + ((size_t)b * mp.kpart + ks) * kPsInts;
    double* pd = ps_d + ((size_t)b * mp.kpart + ks) * kPsDoubles;
    if (lane == 0) {
      s_flag = 0;
      if (ps[kPsMaking]) {
        const double mean_sd_ratio = sqrt(pd[1]) / pd[0];
        if (mean_sd_ratio < mp.sd_ratio && ps[kPsNp] > mp.min_particles) s_flag = 1;
      }
    }
    __syncthreads();
    if (!s_flag) { ++idx; continue; }
    // convert_from_partially_to_fully_initialised (feature.cpp:204-269): J = [I3 | lambda I3], d = hhat
    const int ppos = ppos0 + 6 * ks;
    const int label = ps[kPsLabel];
    const int fpos = 13 + 3 * label;
    const double lam = pd[0], plam = pd[1];
    // every column that holds state: the pose, the feature slots, the other partial slots (their rows / columns are zero
    // when unused, so they need no test)
    const int n_rows = 13 + 3 * ns;
    for (int cc = lane; cc < n_rows + 6 * mp.kpart; cc += 64) {
      const int c = cc < n_rows ? cc : ppos0 + (cc - n_rows);
      if ((c >= fpos && c < fpos + 3) || (c >= ppos && c < ppos + 6)) continue;
      for (int k = 0; k < 3; ++k) {
        // sum over the six partial states; only m = k and m = 3 + k are non-zero in J
        double acc = 0.0;
        for (int m = 0; m < 6; ++m) {
          const double j = (m == k) ? 1.0 : ((m == 3 + k) ? lam : 0.0);
          acc += Pb[(size_t)c * ld + ppos + m] * j;
        }
        Pb[(size_t)c * ld + fpos + k] = acc;
        Pb[(size_t)(fpos + k) * ld + c] = acc;
      }
    }
    if (lane == 0) {
      double Pyy6[36], JP[18], d[3], y3[3];
      for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) Pyy6[r * 6 + c] = Pb[(size_t)(ppos + r) * ld + ppos + c];
      for (int k = 0; k < 3; ++k)
        for (int c = 0; c < 6; ++c) {
          double acc = 0.0;
          for (int m = 0; m < 6; ++m) {
            const double j = (m == k) ? 1.0 : ((m == 3 + k) ? lam : 0.0);
            acc += j * Pyy6[m * 6 + c];
          }
          JP[k * 6 + c] = acc;
        }
      for (int k = 0; k < 3; ++k) { d[k] = xb[ppos + 3 + k]; y3[k] = xb[ppos + k] + lam * xb[ppos + 3 + k]; }
      for (int k = 0; k < 3; ++k)
        for (int l = 0; l < 3; ++l) {
          double a1 = 0.0;
          for (int m = 0; m < 6; ++m) {
            const double j = (m == l) ? 1.0 : ((m == 3 + l) ? lam : 0.0);
            a1 += JP[k * 6 + m] * j;
          }
          const double a2 = (d[k] * plam) * d[l];
          Pb[(size_t)(fpos + k) * ld + fpos + l] = a1 + a2;
        }
      for (int k = 0; k < 3; ++k) xb[fpos + k] = y3[k];
      f_flags[(size_t)b * N + label] = FF_USED | FF_ACTIVE;
      pi[kPartConverted] += 1;
    }
    // Q28 (feature.cpp:254): every LATER feature of feature_list_ has its position_in_total_state_vector_ moved by 6, the
    // partial model's size, where the state shrank by 3: from now on its dh_by_dy lands three columns early in H
    {
      bool any = false;
      for (int f = label + 1 + lane; f < ns; f += 64)
        if (f_flags[(size_t)b * N + f] & (FF_ACTIVE | FF_PARTIAL)) { pos_err[(size_t)b * N + f] += 3; any = true; }
      if (__any(any) && lane == 0) pos_err_any[b] = 1;
    }
    __syncthreads();
    release_slot(ks);
    erase_entry(idx);                    // erase(feat--) then ++feat: the entry that moved up is examined next
  }

  // ---- C: delete_partially_initialised_features_past_sell_by_date (:1506-1521)
  for (int idx = 0; idx < s_count;) {
    const int ks = s_order[idx];
    const int* ps = ps_i + ((size_t)b * mp.kpart + ks) * kPsInts;
    if (ps[kPsAttempts] > mp.erase_after || ps[kPsNp] <= mp.min_particles) delete_partial(idx);
    else ++idx;
  }
  if (lane == 0) {
    pi[kPartCount] = s_count;
    for (int k = 0; k < kMaxPartial; ++k) pi[kPartOrder + k] = s_order[k];
    if (mp.save_trajectory) {   // monoslam.cpp:172-177, after the mapping tail (stale rRES_, Q12)
      const int c = traj_count[b];
      double* t = traj + ((size_t)b * kTrajCapacity + (c % kTrajCapacity)) * 3;
      for (int k = 0; k < 3; ++k) t[k] = last_r[b * 3 + k];
      traj_count[b] = c + 1;
    }
    // the host's next step (index pos_count: k_finalize has counted this one) may leave the partial-feature launches out if none is left
    if (mp.publish_parts)
      __hip_atomic_store(parts_mail, ((unsigned long long)(unsigned)pos_count[b] << 32) | (unsigned)s_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the word is the whole message)
  }
}

__global__ void __launch_bounds__(64) k_map_update(MapArrays a, MapParams mp) {
  extern __shared__ double s_particles[];
  update_body(a, mp, s_particles);
}

// A step that starts without a partially initialised feature (launch_mapping: parts_none) has nothing between the creation of
// a feature and the end-of-frame bookkeeping: both in one launch.
__global__ void __launch_bounds__(64) k_map_finish(MapArrays a, CameraParams cam, MapParams mp) {
  extern __shared__ double s_particles[];
  create_body(a, cam, mp);
  __syncthreads();           // what lane 0 wrote about the new feature is read by every lane below
  update_body(a, mp, s_particles);
}

// ---------------------------------------------------------------------------
// k_map_compact_slots: one workgroup per sequence; does nothing unless the sequence has used all N slots and some of them
// are retired (features deleted by delete_bad_features, the sell-by date of a partial feature or sl2_delete_features keep
// their slot with zeroed rows / columns of P).  The reference erases such features from feature_list_ and its
// next_free_label_ is unbounded: here the live features are squeezed down to the front of the slot range IN LIST ORDER
// (so every "in feature_list_ order" rule - selection ties, the deletion walk, the total state layout - is untouched) and
// the freed slots at the end take the next initialisations.  Labels live in f_label and are never reused.  Everything
// indexed by slot moves: the feature's entries of x, its rows and columns of P, template, counters, flags, the per-frame
// scratch the accessors read, and the slot numbers held by the selection lists and the partial feature's record.
// ---------------------------------------------------------------------------
struct SlotArrays {
  double *x, *P, *xp_org, *f_h, *f_Hx, *f_Hy, *f_R, *f_S, *f_score, *f_z, *f_nu, *srch_d;
  uint8_t* patch;
  int *patch_sums, *f_flags, *attempted, *successful, *f_label, *srch_i, *sel_idx, *succ_idx, *f_arow, *n_sel, *m_count, *n_slots, *ps_i, *pos_err;
  int kpart;
};
__global__ void __launch_bounds__(256) k_map_compact_slots(SlotArrays a, int N, int ld, int ppos, int need) {
  extern __shared__ int s_map[];        // [N] new slot -> old slot, then [N] old slot -> new slot (-1: retired)
  int* s_src = s_map;
  int* s_new = s_map + N;
  __shared__ int s_live;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const int ns = a.n_slots[b];
  if (ns + need <= N) return;                                    // room for what is about to be added: nothing to do
  int* flags = a.f_flags + (size_t)b * N;
  if (tid == 0) {
    int nl = 0;
    for (int f = 0; f < ns; ++f) {
      const bool keep = (flags[f] & (FF_ACTIVE | FF_PARTIAL)) != 0;
      s_new[f] = keep ? nl : -1;
      if (keep) s_src[nl++] = f;
    }
    s_live = nl;
  }
  __syncthreads();
  const int nl = s_live;
  if (nl == ns) return;                                          // no retired slot to give back
  const size_t o = (size_t)b * N;
  // ---- per-slot records: new slot d <- old slot s_src[d] >= d.  A thread takes one slot and holds its whole record (63
  // numbers) between ONE barrier's reads and writes; slots go in rounds of nt, increasing, so a round's sources are never
  // slots an earlier round rewrote.  The templates (288 bytes a slot) move as dwords, sixteen a thread and round.  (Until
  // round 4 every slot was moved by the whole workgroup with a barrier behind it: 20 small dependent copies per slot.)
  for (int d0 = 0; d0 < nl; d0 += nt) {
    const int d = d0 + tid;
    const bool mine = d < nl;
    const int f = mine ? s_src[d] : 0;
    int r_sums[2], r_flags, r_att, r_succ, r_label, r_si[8], r_arow, r_perr;
    double r_xo[8], r_h[2], r_Hx[14], r_Hy[6], r_R, r_S[4], r_score, r_z[2], r_nu[2], r_sd[4], r_x[3];
    if (mine) {
      for (int k = 0; k < 2; ++k) r_sums[k] = a.patch_sums[(o + f) * 2 + k];
      for (int k = 0; k < 8; ++k) r_xo[k] = a.xp_org[(o + f) * 8 + k];
      r_flags = a.f_flags[o + f]; r_att = a.attempted[o + f]; r_succ = a.successful[o + f]; r_label = a.f_label[o + f];
      for (int k = 0; k < 2; ++k) r_h[k] = a.f_h[(o + f) * 2 + k];
      for (int k = 0; k < 14; ++k) r_Hx[k] = a.f_Hx[(o + f) * 14 + k];
      for (int k = 0; k < 6; ++k) r_Hy[k] = a.f_Hy[(o + f) * 6 + k];
      r_R = a.f_R[o + f];
      for (int k = 0; k < 4; ++k) r_S[k] = a.f_S[(o + f) * 4 + k];
      r_score = a.f_score[o + f];
      for (int k = 0; k < 2; ++k) { r_z[k] = a.f_z[(o + f) * 2 + k]; r_nu[k] = a.f_nu[(o + f) * 2 + k]; }
      for (int k = 0; k < 8; ++k) r_si[k] = a.srch_i[(o + f) * 8 + k];
      for (int k = 0; k < 4; ++k) r_sd[k] = a.srch_d[(o + f) * 4 + k];
      r_arow = a.f_arow[o + f];      // (slot-indexed like the rest: a squeeze between make_measurements and the update)
      r_perr = a.pos_err[o + f];
      for (int k = 0; k < 3; ++k) r_x[k] = a.x[(size_t)b * ld + 13 + 3 * f + k];
    }
    __syncthreads();
    if (mine && f != d) {
      for (int k = 0; k < 2; ++k) a.patch_sums[(o + d) * 2 + k] = r_sums[k];
      for (int k = 0; k < 8; ++k) a.xp_org[(o + d) * 8 + k] = r_xo[k];
      a.f_flags[o + d] = r_flags; a.attempted[o + d] = r_att; a.successful[o + d] = r_succ; a.f_label[o + d] = r_label;
      for (int k = 0; k < 2; ++k) a.f_h[(o + d) * 2 + k] = r_h[k];
      for (int k = 0; k < 14; ++k) a.f_Hx[(o + d) * 14 + k] = r_Hx[k];
      for (int k = 0; k < 6; ++k) a.f_Hy[(o + d) * 6 + k] = r_Hy[k];
      a.f_R[o + d] = r_R;
      for (int k = 0; k < 4; ++k) a.f_S[(o + d) * 4 + k] = r_S[k];
      a.f_score[o + d] = r_score;
      for (int k = 0; k < 2; ++k) { a.f_z[(o + d) * 2 + k] = r_z[k]; a.f_nu[(o + d) * 2 + k] = r_nu[k]; }
      for (int k = 0; k < 8; ++k) a.srch_i[(o + d) * 8 + k] = r_si[k];
      for (int k = 0; k < 4; ++k) a.srch_d[(o + d) * 4 + k] = r_sd[k];
      a.f_arow[o + d] = r_arow;
      a.pos_err[o + d] = r_perr;
      for (int k = 0; k < 3; ++k) a.x[(size_t)b * ld + 13 + 3 * d + k] = r_x[k];
    }
    // (no second barrier: the next round reads slots >= its own first slot, none of which this round wrote)
  }
  {
    static_assert(kPatchStride % 4 == 0, "templates move as dwords");
    constexpr int kDw = kPatchStride / 4, kPer = 16;
    unsigned* pw = (unsigned*)(a.patch + o * kPatchStride);          // (hipMalloc'ed base, stride 288: dword aligned)
    const int total = nl * kDw;
    for (int c0 = 0; c0 < total; c0 += kPer * nt) {
      unsigned v[kPer];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int e = c0 + tid + q * nt;
        if (e < total) { const int d = e / kDw, k = e - d * kDw; v[q] = pw[(size_t)s_src[d] * kDw + k]; }
      }
      __syncthreads();
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        const int e = c0 + tid + q * nt;
        if (e < total) pw[e] = v[q];
      }
      // (a round's last slot may be half written: the next round reads the other half of it, or slots above it)
    }
  }
  __syncthreads();
  for (int f = nl + tid; f < ns; f += nt) {                      // the freed slots: unused again
    flags[f] = 0;
    a.f_arow[o + f] = -1;
    a.pos_err[o + f] = 0;
    a.attempted[o + f] = 0; a.successful[o + f] = 0;
    for (int k = 0; k < 3; ++k) a.x[(size_t)b * ld + 13 + 3 * f + k] = 0.0;
  }
  // slot numbers held elsewhere
  for (int k = tid; k < a.n_sel[b]; k += nt) { const int f = a.sel_idx[o + k]; a.sel_idx[o + k] = (f >= 0 && f < ns) ? s_new[f] : -1; }
  for (int k = tid; k < a.m_count[b]; k += nt) { const int f = a.succ_idx[o + k]; a.succ_idx[o + k] = (f >= 0 && f < ns) ? s_new[f] : -1; }
  if (tid < a.kpart) {
    int* ps = a.ps_i + ((size_t)b * a.kpart + tid) * kPsInts;
    if (ps[kPsActive] && ps[kPsLabel] >= 0 && ps[kPsLabel] < ns) ps[kPsLabel] = s_new[ps[kPsLabel]];
  }
  // ---- P: new index i <- old index src(i); pose rows and the partial feature's six rows / the innovation row stay where
  // they are, the vacated feature rows / columns become zero.  In blocks of rows, increasing: src(i) >= i, so a block's
  // source rows are never rows already rewritten; within a block every read completes (barrier) before the writes.  A
  // block is as many rows as sixteen elements per thread hold (32 rows of 128 columns: a squeeze used to walk the rows one
  // by one, two barriers each - 145 us for the worst launch of the mapping workload, profiles/r04_mapping_kernel_stats.csv).
  double* Pb = a.P + (size_t)b * ld * ld;
  auto src_index = [&](int i) -> int {
    if (i < 13 || i >= 13 + 3 * N) return i;
    const int d = (i - 13) / 3, c = (i - 13) % 3;
    return d < nl ? 13 + 3 * s_src[d] + c : -1;
  };
  constexpr int kPerThread = 16;
  const int rows_per = max(1, (kPerThread * nt) / ld);            // (ld <= 2048 = 8 * 256: at least two rows)
  const float rcp_ld = 1.0f / (float)ld;
  for (int i0 = 0; i0 < ld; i0 += rows_per) {
    const int nelem = min(rows_per, ld - i0) * ld;
    double v[kPerThread];
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const int e = tid + q * nt;
      v[q] = 0.0;
      if (e < nelem) {
        int r = (int)((float)e * rcp_ld), j = e - r * ld;           // e = r * ld + j, e < 2^12: the float estimate is off by one at most
        if (j < 0) { --r; j += ld; }
        if (j >= ld) { ++r; j -= ld; }
        const int si = src_index(i0 + r), sj = src_index(j);
        if (si >= 0 && sj >= 0) v[q] = Pb[(size_t)si * ld + sj];
      }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const int e = tid + q * nt;
      if (e < nelem) Pb[(size_t)i0 * ld + e] = v[q];
    }
    // (no second barrier: the next block reads rows >= its own first row, none of which this block wrote)
  }
  if (tid == 0) a.n_slots[b] = nl;
}

int launch_compact_slots(sl2_engine* e, int need) {
  if ((size_t)e->ld > 8 * 256) return SL2_OK;                     // (maps beyond 2048 states: slots are not squeezed)
  LaunchScope ls(e, "k_map_compact_slots");
  SlotArrays a;
  a.x = e->x; a.P = e->P; a.xp_org = e->xp_org; a.f_h = e->f_h; a.f_Hx = e->f_Hx; a.f_Hy = e->f_Hy; a.f_R = e->f_R; a.f_S = e->f_S;
  a.f_score = e->f_score; a.f_z = e->f_z; a.f_nu = e->f_nu; a.srch_d = e->srch_d; a.patch = e->patch; a.patch_sums = e->patch_sums;
  a.f_flags = e->f_flags; a.attempted = e->attempted; a.successful = e->successful; a.f_label = e->f_label; a.srch_i = e->srch_i;
  a.sel_idx = e->sel_idx; a.succ_idx = e->succ_idx; a.f_arow = e->f_arow; a.n_sel = e->n_sel; a.m_count = e->m_count; a.n_slots = e->n_slots;
  a.ps_i = e->ps_i; a.pos_err = e->pos_err; a.kpart = e->kpart;
  hipLaunchKernelGGL(k_map_compact_slots, dim3(e->B), dim3(256), sizeof(int) * 2 * e->N, e->stream, a, e->N, e->ld, e->ppos, need);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

// MonoSLAM::InitialiseFeature at a caller-chosen pixel (monoslam.cpp:1211-1235; the GUI sets uu_ / vv_ by mouse click):
// per sequence, publish the selection for k_map_create exactly as k_map_find would have.
__global__ void __launch_bounds__(64) k_map_manual(const int* __restrict__ uv, const int* __restrict__ n_slots, int* __restrict__ part_i,
                                                   double* __restrict__ part_d, int* __restrict__ status, int N, int width,
                                                   int height, int B, int kpart) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  int* pi = part_i + (size_t)b * kPartInts;
  pi[kPartRegionValid] = 0;
  pi[kPartCreated] = 0;
  const int u = uv[2 * b], v = uv[2 * b + 1];
  if (u < 0) return;                                                   // this sequence is left alone
  if (u < 5 || v < 5 || u > width - 6 || v > height - 6) return;       // the 11 x 11 patch must lie inside the frame
  if (pi[kPartCount] >= kpart) return;                                 // every partial slot of the sequence is taken
  if (n_slots[b] >= N) { status[b] |= 2; return; }
  status[b] &= ~2;
  pi[kPartUU] = u; pi[kPartVV] = v;
  pi[kPartRegionValid] = 1;
  part_d[(size_t)b * kPartDoubles + 2] = 1.0e300;                      // no score threshold on a manual selection
}

static MapParams map_params(const sl2_engine* e, int enable_mapping, int save_trajectory, int force) {
  MapParams mp;
  mp.enable_mapping = enable_mapping; mp.save_trajectory = save_trajectory; mp.force = force;
  mp.keep_visible = e->prm.number_of_features_to_keep_visible;
  mp.n_particles = e->prm.number_of_particles;
  mp.min_particles = e->prm.min_number_of_particles;
  mp.erase_after = e->prm.erase_partially_init_feature_after_this_many_attempts;
  mp.min_lambda = e->prm.min_lambda; mp.max_lambda = e->prm.max_lambda;
  mp.sd_ratio = e->prm.standard_deviation_depth_ratio; mp.prune_threshold = e->prm.prune_probability_threshold;
  mp.dt = e->prm.delta_t;
  mp.pcap = e->root->pcap;
  mp.kpart = e->root->kpart;
  return mp;
}

static MapArrays map_arrays(const sl2_engine* e) {
  MapArrays a;
  a.x = e->x; a.P = e->P; a.frames = e->cur_frames; a.seq_stride = e->cur_stride; a.patch = e->patch; a.patch_sums = e->patch_sums;
  a.xp_org = e->xp_org; a.f_flags = e->f_flags; a.n_slots = e->n_slots; a.attempted = e->attempted; a.successful = e->successful;
  a.f_label = e->f_label; a.next_label = e->next_label; a.part_i = e->part_i; a.part_d = e->part_d; a.ps_i = e->ps_i; a.ps_d = e->ps_d;
  a.pos_err = e->pos_err; a.pos_err_any = e->pos_err_any; a.particles = e->particles; a.last_r = e->last_r; a.traj = e->traj;
  a.traj_count = e->traj_count; a.pos_count = e->pos_count; a.parts_mail = e->root->parts_mail_dev;
  a.N = e->N; a.ld = e->ld; a.ppos0 = e->ppos;
  return a;
}

static int launch_create(sl2_engine* e, const MapParams& mp) {
  LaunchScope ls(e, "k_map_create");
  hipLaunchKernelGGL(k_map_create, dim3(e->B), dim3(64), 0, e->stream, map_arrays(e), e->cam, mp);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

static int launch_find(sl2_engine* e, const MapParams& mp) {
  LaunchScope ls(e, "k_map_find");
  hipLaunchKernelGGL(k_map_find, dim3(e->B), dim3(kDetThreads), sizeof(double) * 2 * e->N, e->stream, e->x, e->f_flags, e->n_slots, e->n_vis,
                     e->prev_r, e->part_i, e->part_d, e->rand48, e->last_r, e->status, e->cur_frames, e->cur_stride, e->cam, mp, e->N, e->ld);
  SL2_HIP(hipGetLastError());
  return SL2_OK;
}

// InitialiseFeature(frame) with (uu_, vv_) = uv[b] (device array [B][2]; u < 0 = skip the sequence)
int launch_manual_init(sl2_engine* e, const int* d_uv) {
  const MapParams mp = map_params(e, 1, 0, 1);
  { int rc = launch_compact_slots(e, 1); if (rc != SL2_OK) return rc; }
  hipLaunchKernelGGL(k_map_manual, dim3((e->B + 63) / 64), dim3(64), 0, e->stream, d_uv, e->n_slots, e->part_i, e->part_d, e->status,
                     e->N, e->cam.width, e->cam.height, e->B, e->kpart);
  SL2_HIP(hipGetLastError());
  return launch_create(e, mp);
}

// InitialiseAutoFeature(frame) = AutoInitialiseFeature(frame, 0) (monoslam.cpp:1535-1541, 823-865): region, detector, creation
int launch_auto_init(sl2_engine* e) {
  const MapParams mp = map_params(e, 1, 0, 1);
  { int rc = launch_compact_slots(e, 1); if (rc != SL2_OK) return rc; }
  { int rc = launch_find(e, mp); if (rc != SL2_OK) return rc; }
  return launch_create(e, mp);
}

int launch_mapping(sl2_engine* e, int enable_mapping, int save_trajectory, int slots_bound, int parts_state) {
  const int parts_none = parts_state == 1, parts_full = parts_state == 2;
  const int B = e->B;
  MapParams mp;
  mp.force = 0;
  mp.enable_mapping = enable_mapping; mp.save_trajectory = save_trajectory;
  mp.keep_visible = e->prm.number_of_features_to_keep_visible;
  mp.n_particles = e->prm.number_of_particles;
  mp.min_particles = e->prm.min_number_of_particles;
  mp.erase_after = e->prm.erase_partially_init_feature_after_this_many_attempts;
  mp.min_lambda = e->prm.min_lambda; mp.max_lambda = e->prm.max_lambda;
  mp.sd_ratio = e->prm.standard_deviation_depth_ratio; mp.prune_threshold = e->prm.prune_probability_threshold;
  mp.dt = e->prm.delta_t;
  mp.pcap = e->root->pcap;
  mp.kpart = e->root->kpart;
  // The three launches that serve partially initialised features are dead weight while there is none, and at one sequence each
  // is a link of the frame's dependent chain: k_map_update's report (sl2_engine.hip: parts_state_for_step) lets the host leave
  // them out (state 1).  me_big_count keeps its last value meanwhile; k_map_particles zeroes it before anything reads it again.
  // The other way round (state 2): every partial slot is taken, so FindNonOverlappingRegion's gate (k_map_find: kPartCount <
  // kpart, monoslam.cpp:163-165) is shut whatever the camera does - no region, no detector, no creation: k_map_find and
  // k_map_create are left out, and k_map_particles clears the two per-step flags k_map_find would have.
  mp.parts_skipped = parts_none ? 1 : 0;
  mp.publish_parts = (e->root->B == 1 && e->root->parts_mail_dev) ? 1 : 0;
  const int W = e->cam.width, H = e->cam.height;
  if (!e->score_map) { set_error("launch_mapping: score map not allocated"); return SL2_ERR_INVALID; }
  // Retired slots are squeezed out only when a sequence is about to run out of slots; the host's upper bound on the slots in use
  // (sl2_engine.hip: slots_upper_bound) says when none can be: the launch - one of the step's dependent chain, 7 us at one
  // sequence, 0.03 ms at 1024 - is then left out altogether.
  if (enable_mapping && slots_bound + 1 > e->N) { int rc = launch_compact_slots(e, 1); if (rc != SL2_OK) return rc; }
  if (!parts_full) { int rc = launch_find(e, mp); if (rc != SL2_OK) return rc; }
  const size_t shm_particles = sizeof(double) * kParticleDoubles * (size_t)mp.n_particles;
  if (parts_none) {
    LaunchScope ls(e, "k_map_finish");
    hipLaunchKernelGGL(k_map_finish, dim3(B), dim3(64), shm_particles, e->stream, map_arrays(e), e->cam, mp);
    SL2_HIP(hipGetLastError());
    return SL2_OK;
  }
  if (!parts_full) { int rc = launch_create(e, mp); if (rc != SL2_OK) return rc; }
  {
    LaunchScope ls(e, "k_map_particles");
    const int pc = e->root->pcap;
#define SL2_PARTICLES(T) hipLaunchKernelGGL(k_map_particles<T>, dim3(B, mp.kpart), dim3(pc), 0, e->stream, e->x, e->P, e->ps_i, e->particles, \
                                            e->me_desc, e->last_r, e->me_big_count, e->part_i, parts_full, e->cam, e->ld, e->ppos, pc, mp.kpart)
    if (pc <= 128) SL2_PARTICLES(128);
    else if (pc <= 256) SL2_PARTICLES(256);
    else if (pc <= 512) SL2_PARTICLES(512);
    else SL2_PARTICLES(1024);
#undef SL2_PARTICLES
    SL2_HIP(hipGetLastError());
  }
  {
    MeJobsEngine J;
    J.frames = e->cur_frames; J.seq_stride = e->cur_stride; J.patch_base = e->patch; J.ps_i = e->ps_i; J.me_desc = e->me_desc;
    J.particles = e->particles; J.map_base = e->score_map; J.N = e->N; J.pcap = e->root->pcap;
    J.width = W; J.height = H; J.kpart = mp.kpart;
    {
      LaunchScope ls(e, "k_map_me_search");
      hipLaunchKernelGGL(k_map_me_search, dim3(B * mp.kpart), dim3(1024), 0, e->stream, J, e->me_big_list, e->me_big_count);
      SL2_HIP(hipGetLastError());
    }
    {
      LaunchScope ls(e, "k_me_big");
      me_big_launch(J, e->me_big_list, e->me_big_count, W, e->stream);
      SL2_HIP(hipGetLastError());
    }
  }
  {
    LaunchScope ls(e, "k_map_update");
    hipLaunchKernelGGL(k_map_update, dim3(B), dim3(64), shm_particles, e->stream, map_arrays(e), mp);
    SL2_HIP(hipGetLastError());
  }
  return SL2_OK;
}

}  // namespace sl2

#ifdef SL2_ME_TRACE
extern "C" int sl2_debug_me_trace(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(sl2::g_me_trace), sizeof(unsigned long long) * 16) != hipSuccess) return 2;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(sl2::g_me_trace), z, sizeof(z)) != hipSuccess) return 2; }
  return 0;
}
#endif
