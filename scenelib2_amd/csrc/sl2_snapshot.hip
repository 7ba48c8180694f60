// sl2_snapshot: the one-call read-back of a sequence's public state (include/scenelib2_amd.h).
//
// The reference's callers read the MonoSLAM object's public members after every GoOneStep at zero cost
// (examples/MonoSlamSceneLib1.cpp:132-151; graphic/graphictool.cpp:130-167, 290-347 walk xv_, Pxx_ and every Feature's y_,
// Pxy_, Pyy_, h_, z_, S_, flags).  Here those members live in HBM, spread over the engine's per-slot arrays and the dense
// covariance.  k_snapshot gathers them for ONE sequence into a packed blob in a device staging buffer (per-feature records
// in feature_list_ order, deleted features removed, the Pxy / Pyy blocks cut out of P) and then streams the blob - exactly
// its size, in coalesced 16-byte pieces - into a pinned, mapped host buffer owned by the engine: one launch, one stream
// synchronisation, no hipMemcpy, no allocation per call.
#include "sl2_common.hpp"

#include <chrono>

namespace sl2 {

struct SnapArrays {
  const double *x, *P, *xp_org, *f_h, *f_z, *f_nu, *f_R, *f_S, *f_Hx, *f_Hy, *traj, *ps_d, *particles;
  const int *f_flags, *f_label, *n_slots, *next_label, *attempted, *successful, *sel_idx, *n_sel, *n_vis, *m_count, *traj_count,
      *status, *part_i, *ps_i, *pos_err;
  const uint8_t* patch;
};

constexpr int kSnapMagic = 0x53324c53;
constexpr int kSnapThreads = 256;
constexpr int kFeatureInfoBytes = (int)sizeof(sl2_feature_info);
static_assert(sizeof(sl2_snapshot_header) == 256, "sl2_snapshot_header is 64 ints");
static_assert(sizeof(sl2_feature_info) % 8 == 0, "feature records keep the sections 8-byte aligned");
static_assert(sizeof(sl2_partial_info) == 32, "sl2_partial_info");

__device__ __forceinline__ int up8(int v) { return (v + 7) & ~7; }

// One workgroup.  LDS: per slot flags, label, list index (-1: not in feature_list_), state size, first double of its
// covariance record, patch index (-1: not included); then the selection's labels.
__global__ void __launch_bounds__(kSnapThreads) k_snapshot(SnapArrays a, int seq, int N, int ld, int ppos, int pcap, int kpart, int traj_cursor,
                                                           int patch_from_label, long long steps_done, unsigned char* __restrict__ stage,
                                                           uint4* __restrict__ host_out, unsigned long long* __restrict__ done_word,
                                                           unsigned long long ticket) {
  extern __shared__ int sm[];
  int* s_flags = sm;
  int* s_label = sm + N;
  int* s_li = sm + 2 * N;
  int* s_cov = sm + 3 * N;
  int* s_pi = sm + 4 * N;
  int* s_pos = sm + 5 * N;
  int* s_sel = sm + 6 * N;
  int* s_col = sm + 7 * N;                     // first state column of a live slot's feature
  __shared__ sl2_snapshot_header hd;
  const int tid = threadIdx.x;
  const size_t o = (size_t)seq * N;
  const int ns = a.n_slots[seq];
  const int nsel_raw = a.n_sel[seq];
  const int* pi = a.part_i + (size_t)seq * kPartInts;
  const int* psb = a.ps_i + (size_t)seq * kpart * kPsInts;
  __shared__ int s_ps[kMaxPartial][kPsInts];
  if (tid < kpart * kPsInts) s_ps[tid / kPsInts][tid % kPsInts] = psb[tid];
  // the partial slot whose label sits in feature slot f (-1: none)
  auto pslot_of = [&](int f) { int r = -1; for (int k = 0; k < kpart; ++k) if (s_ps[k][kPsActive] && s_ps[k][kPsLabel] == f) r = k; return r; };
  for (int f = tid; f < ns; f += kSnapThreads) { s_flags[f] = a.f_flags[o + f]; s_label[f] = a.f_label[o + f]; }
  for (int k = tid; k < nsel_raw; k += kSnapThreads) s_sel[k] = a.sel_idx[o + k];
  __syncthreads();
  // list positions, state positions, covariance offsets and the compacted selection: one wavefront, ballots and popcounts
  // (a single thread walking the slots through LDS was 15 us of a 40 us kernel at 100 features)
  __shared__ int s_tot[5];                     // nf, total state size, covariance doubles, patches, kept selection entries
  if (tid < 64) {
    int nfull = 0, npart = 0, npatch = 0;      // running counts in front of the current 64 slots (wave-uniform)
    for (int base = 0; base < ns; base += 64) {
      const int f = base + tid;
      const int fl = f < ns ? s_flags[f] : 0;
      const bool part = (fl & FF_PARTIAL) != 0, full = !part && (fl & FF_ACTIVE) != 0;
      const bool wants_patch = (part || full) && s_label[f < ns ? f : 0] >= patch_from_label;
      const unsigned long long m_full = __ballot(full), m_part = __ballot(part), m_pat = __ballot(wants_patch);
      const unsigned long long below = (1ull << tid) - 1ull;
      const int cf = nfull + __popcll(m_full & below), cp = npart + __popcll(m_part & below);
      if (f < ns) {
        s_li[f] = (part || full) ? cf + cp : -1;
        s_pos[f] = 13 + 3 * cf + 6 * cp;
        s_cov[f] = (13 * 3 + 9) * cf + (13 * 6 + 36) * cp;
        s_pi[f] = wants_patch ? npatch + __popcll(m_pat & below) : -1;
      }
      nfull += __popcll(m_full); npart += __popcll(m_part); npatch += __popcll(m_pat);
    }
    // delete_feature() deselects the feature it removes (monoslam.cpp:800-801)
    int kept = 0;
    for (int base = 0; base < nsel_raw; base += 64) {
      const int k = base + tid;
      const int f = k < nsel_raw ? s_sel[k] : -1;
      const bool ok = f >= 0 && f < ns && (s_flags[f] & FF_ACTIVE);
      const int lab = ok ? s_label[f] : 0;
      const unsigned long long m = __ballot(ok);
      if (ok) s_sel[kept + __popcll(m & ((1ull << tid) - 1ull))] = lab;      // (destinations lie at or below this round's sources)
      kept += __popcll(m);
    }
    if (tid == 0) { s_tot[0] = nfull + npart; s_tot[1] = 13 + 3 * nfull + 6 * npart; s_tot[2] = 48 * nfull + 114 * npart; s_tot[3] = npatch; s_tot[4] = kept; }
  }
  __syncthreads();
  if (tid == 0) {
    const int nf = s_tot[0], pos = s_tot[1], cov = s_tot[2], np = s_tot[3], kept = s_tot[4];
    const int total = a.traj_count[seq];
    int first = total - kTrajCapacity;
    if (first < 0) first = 0;
    if (first < traj_cursor) first = traj_cursor;
    if (first > total) first = total;
    const int n_partial = pi[kPartCount];
    int n_particles = 0;                       // of all entries of feature_init_info_vector_ together
    for (int q = 0; q < n_partial; ++q) n_particles += s_ps[pi[kPartOrder + q]][kPsNp];
    int* h = reinterpret_cast<int*>(&hd);
    for (int k = 0; k < 64; ++k) h[k] = 0;
    hd.magic = kSnapMagic; hd.api_version = SL2_API_VERSION; hd.seq = seq;
    hd.n_features = nf; hd.total_state_size = pos;
    hd.number_of_visible_features = a.n_vis[seq];
    hd.n_selected = kept;
    hd.successful_measurement_vector_size = 2 * a.m_count[seq];
    hd.next_free_label = a.next_label[seq];
    hd.status_flags = a.status[seq];
    hd.traj_total = total; hd.traj_first = first; hd.traj_count = total - first;
    hd.n_partial = n_partial; hd.n_patches = np;
    hd.uu = pi[kPartUU]; hd.vv = pi[kPartVV];
    hd.location_selected_flag = pi[kPartCreated];
    hd.init_feature_search_region_defined_flag = pi[kPartRegionValid];
    for (int k = 0; k < 4; ++k) hd.init_feature_search_region[k] = pi[kPartRegion + k];
    hd.steps_done = (int)(steps_done & 0x7fffffff);
    int off = (int)sizeof(sl2_snapshot_header);
    hd.off_xv = off; off += 13 * 8;
    hd.off_Pxx = off; off += 169 * 8;
    hd.off_features = off; off += nf * kFeatureInfoBytes;
    hd.off_cov = off; off += cov * 8;
    hd.off_selection = off; off += up8(kept * 4);
    hd.off_traj = off; off += (total - first) * 24;
    hd.off_partial = off; off += n_partial * (int)sizeof(sl2_partial_info) + n_particles * kParticleDoubles * 8;
    hd.off_patches = off; off += np * 128;
    hd.bytes = off;
  }
  __syncthreads();
  const double* xb = a.x + (size_t)seq * ld;
  const double* Pb = a.P + (size_t)seq * ld * ld;
  // header, xv_, Pxx_
  for (int k = tid; k < 64; k += kSnapThreads) reinterpret_cast<int*>(stage)[k] = reinterpret_cast<const int*>(&hd)[k];
  double* o_xv = reinterpret_cast<double*>(stage + hd.off_xv);
  double* o_Pxx = reinterpret_cast<double*>(stage + hd.off_Pxx);
  for (int k = tid; k < 13; k += kSnapThreads) o_xv[k] = xb[k];
  for (int k = tid; k < 169; k += kSnapThreads) o_Pxx[k] = Pb[(size_t)(k / 13) * ld + (k % 13)];
  // feature records + covariance blocks
  for (int f = tid; f < ns; f += kSnapThreads) {
    const int li = s_li[f];
    if (li < 0) continue;
    const int fl = s_flags[f];
    const bool partial = (fl & FF_PARTIAL) != 0;
    const int d = partial ? 6 : 3;
    const int col = partial ? ppos + 6 * (pslot_of(f) < 0 ? 0 : pslot_of(f)) : 13 + 3 * f;
    sl2_feature_info fi;
    fi.label = s_label[f];
    fi.active = 1;
    fi.selected_flag = (fl & FF_SELECTED) ? 1 : 0;
    fi.successful_measurement_flag = (fl & FF_SUCCESS) ? 1 : 0;
    fi.attempted_measurements_of_feature = a.attempted[o + f];
    fi.successful_measurements_of_feature = a.successful[o + f];
    fi.position_in_total_state_vector = s_pos[f] - a.pos_err[o + f];      // (Q28: what the reference has on record)
    fi.visible = (fl & FF_VISIBLE) ? 1 : 0;
    for (int k = 0; k < 3; ++k) fi.y[k] = xb[col + k];
    for (int k = 0; k < 2; ++k) { fi.h[k] = a.f_h[(o + f) * 2 + k]; fi.z[k] = a.f_z[(o + f) * 2 + k]; fi.nu[k] = a.f_nu[(o + f) * 2 + k]; }
    fi.R = a.f_R[o + f];
    for (int k = 0; k < 4; ++k) fi.S[k] = a.f_S[(o + f) * 4 + k];
    for (int k = 0; k < 14; ++k) fi.dh_by_dxp[k] = a.f_Hx[(o + f) * 14 + k];
    for (int k = 0; k < 6; ++k) fi.dh_by_dy[k] = a.f_Hy[(o + f) * 6 + k];
    for (int k = 0; k < 7; ++k) fi.xp_org[k] = a.xp_org[(o + f) * 8 + k];
    fi.fully_initialised_flag = partial ? 0 : 1;
    fi.state_size = d;
    for (int k = 0; k < 3; ++k) fi.y_direction[k] = partial ? xb[col + 3 + k] : 0.0;
    *reinterpret_cast<sl2_feature_info*>(stage + hd.off_features + (size_t)li * kFeatureInfoBytes) = fi;
    s_col[f] = col;
  }
  __syncthreads();
  // Pxy_ / Pyy_ blocks and the new templates: one (feature, element) pair per thread and round, so that the ~50 loads of a
  // feature's blocks are spread over lanes instead of queued on one
  for (int idx = tid; idx < ns * 128; idx += kSnapThreads) {
    const int f = idx >> 7, el = idx & 127;
    if (s_li[f] < 0) continue;
    const int d = (s_flags[f] & FF_PARTIAL) ? 6 : 3, col = s_col[f];
    if (el >= 13 * d + d * d) continue;
    double* oc = reinterpret_cast<double*>(stage + hd.off_cov) + s_cov[f];
    if (el < 13 * d) oc[el] = Pb[(size_t)(el / d) * ld + col + el % d];
    else { const int q = el - 13 * d; oc[el] = Pb[(size_t)(col + q / d) * ld + col + q % d]; }
  }
  if (hd.n_patches)
    for (int idx = tid; idx < ns * 128; idx += kSnapThreads) {
      const int f = idx >> 7, el = idx & 127;
      const int pidx = s_li[f] >= 0 ? s_pi[f] : -1;
      if (pidx < 0) continue;
      unsigned char* op = stage + hd.off_patches + (size_t)pidx * 128;
      if (el < 4) op[el] = (unsigned char)((unsigned)s_label[f] >> (8 * el));            // int32 label, little endian
      else if (el < 4 + SL2_PATCH_BYTES) op[el] = a.patch[(o + f) * kPatchStride + el - 4];
      else op[el] = 0;
    }
  // selected_feature_list_ as labels
  int* o_sel = reinterpret_cast<int*>(stage + hd.off_selection);
  for (int k = tid; k < hd.n_selected; k += kSnapThreads) o_sel[k] = s_sel[k];
  if (tid == 0 && (hd.n_selected & 1)) o_sel[hd.n_selected] = 0;
  // trajectory_store_: the entries the caller has not seen yet
  double* o_tr = reinterpret_cast<double*>(stage + hd.off_traj);
  for (int k = tid; k < hd.traj_count * 3; k += kSnapThreads) {
    const int logical = hd.traj_first + k / 3;
    o_tr[k] = a.traj[((size_t)seq * kTrajCapacity + (logical % kTrajCapacity)) * 3 + (k % 3)];
  }
  // feature_init_info_vector_, in the vector's order
  {
    unsigned char* out = stage + hd.off_partial;
    for (int q = 0; q < hd.n_partial; ++q) {
      const int ks = pi[kPartOrder + q];
      const int npart = s_ps[ks][kPsNp];
      sl2_partial_info* pinfo = reinterpret_cast<sl2_partial_info*>(out);
      if (tid == 0) {
        const int slot = s_ps[ks][kPsLabel];
        pinfo->label = (slot >= 0 && slot < ns) ? s_label[slot] : -1;
        pinfo->number_of_match_attempts = s_ps[ks][kPsAttempts];
        pinfo->n_particles = npart;
        pinfo->making_measurement_on_this_step_flag = s_ps[ks][kPsMaking];
        pinfo->mean = a.ps_d[((size_t)seq * kpart + ks) * kPsDoubles + 0];
        pinfo->covariance = a.ps_d[((size_t)seq * kpart + ks) * kPsDoubles + 1];
      }
      double* opp = reinterpret_cast<double*>(pinfo + 1);
      const double* src = a.particles + ((size_t)seq * kpart + ks) * pcap * kParticleDoubles;
      for (int k = tid; k < npart * kParticleDoubles; k += kSnapThreads) opp[k] = src[k];
      out += sizeof(sl2_partial_info) + (size_t)npart * kParticleDoubles * 8;
    }
  }
  // the blob leaves for the host: the workgroup's own stores above are visible to it after the barrier
  __threadfence();
  __syncthreads();
  const int n16 = (hd.bytes + 15) >> 4;
  const uint4* src16 = reinterpret_cast<const uint4*>(stage);
  for (int k = tid; k < n16; k += kSnapThreads) host_out[k] = src16[k];
  __threadfence_system();
  // The blob is in host memory: say so in a word of its own, so that the caller can pick it up the moment it is complete instead
  // of waiting for the launch to be retired (sl2_snapshot spins on this word first: ~8 us less per call than a stream wait).
  __syncthreads();
  if (tid == 0) __hip_atomic_store(done_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace sl2

using namespace sl2;

extern "C" size_t sl2_snapshot_capacity(const sl2_engine* e) {
  if (!e) return 0;
  const size_t N = e->N;
  size_t b = sizeof(sl2_snapshot_header) + 13 * 8 + 169 * 8 + N * sizeof(sl2_feature_info) + (N * (13 * 3 + 9) + (size_t)e->kpart * (13 * 6 + 36)) * 8 +
             ((N * 4 + 7) & ~(size_t)7) + (size_t)kTrajCapacity * 24 + (size_t)e->kpart * (sizeof(sl2_partial_info) + (size_t)e->pcap * kParticleDoubles * 8) +
             N * 128;
  return (b + 15) & ~(size_t)15;
}

extern "C" int sl2_snapshot(sl2_engine* e, int seq, int traj_cursor, int patch_from_label, const void** blob, size_t* bytes) {
  if (!e || seq < 0 || seq >= e->B || !blob || !bytes || traj_cursor < 0) return SL2_ERR_INVALID;
  SL2_HIP(hipSetDevice(e->device));
  if (!e->snap_host) {   // first use: a device staging buffer and a pinned, mapped host buffer, both owned by the engine
    const size_t cap = sl2_snapshot_capacity(e);
    SL2_HIP(hipMalloc(&e->snap_stage, cap));
    // (+ 64 bytes: the completion word behind the blob, on a cache line of its own)
    SL2_HIP(hipHostMalloc(&e->snap_host, cap + 64, hipHostMallocMapped | hipHostMallocCoherent));
    SL2_HIP(hipHostGetDevicePointer(&e->snap_host_dev, e->snap_host, 0));
    e->snap_cap = cap;
    *(volatile unsigned long long*)((char*)e->snap_host + cap) = 0ull;
  }
  // the groups' streams (sl2_set_groups > 1) join the root stream at the end of every stepping call; the kernel below is
  // queued behind them
  SnapArrays a;
  a.x = e->x; a.P = e->P; a.xp_org = e->xp_org; a.f_h = e->f_h; a.f_z = e->f_z; a.f_nu = e->f_nu; a.f_R = e->f_R; a.f_S = e->f_S;
  a.f_Hx = e->f_Hx; a.f_Hy = e->f_Hy; a.traj = e->traj; a.ps_d = e->ps_d; a.particles = e->particles;
  a.f_flags = e->f_flags; a.f_label = e->f_label; a.n_slots = e->n_slots; a.next_label = e->next_label; a.attempted = e->attempted;
  a.successful = e->successful; a.sel_idx = e->sel_idx; a.n_sel = e->n_sel; a.n_vis = e->n_vis; a.m_count = e->m_count;
  a.traj_count = e->traj_count; a.status = e->status; a.part_i = e->part_i; a.ps_i = e->ps_i; a.pos_err = e->pos_err; a.patch = e->patch;
  hipLaunchKernelGGL(k_snapshot, dim3(1), dim3(kSnapThreads), sizeof(int) * 8 * e->N, e->stream, a, seq, e->N, e->ld, e->ppos, e->pcap, e->kpart,
                     traj_cursor, patch_from_label, e->steps_done, (unsigned char*)e->snap_stage, (uint4*)e->snap_host_dev,
                     (unsigned long long*)((char*)e->snap_host_dev + e->snap_cap), ++e->snap_ticket);
  SL2_HIP(hipGetLastError());
  {
    // the kernel announces the finished blob in the word behind it; a short spin on that word, then the ordinary wait (a long
    // step in front of the snapshot, a large batch: no point in burning a core)
    const volatile unsigned long long* done = (const volatile unsigned long long*)((const char*)e->snap_host + e->snap_cap);
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (;;) {
      for (int i = 0; i < 64 && !seen; ++i) { seen = __atomic_load_n(done, __ATOMIC_ACQUIRE) == e->snap_ticket; if (!seen) __builtin_ia32_pause(); }
      if (seen || std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(300)) break;
    }
    if (!seen) SL2_HIP(hipStreamSynchronize(e->stream));
  }
  const sl2_snapshot_header* h = (const sl2_snapshot_header*)e->snap_host;
  if (h->magic != kSnapMagic || h->bytes <= 0 || (size_t)h->bytes > sl2_snapshot_capacity(e)) {
    set_error("sl2_snapshot: malformed blob");
    return SL2_ERR_HIP;
  }
  *blob = e->snap_host;
  *bytes = (size_t)h->bytes;
  return SL2_OK;
}
