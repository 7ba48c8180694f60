// Engine-internal declarations shared by the HIP translation units.
// gfx950 (MI355X / CDNA4) only; wavefront = 64.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/scenelib2_amd.h"
#include "sl2_math.hpp"

namespace sl2 {

void set_error(const std::string& s);
const char* hip_err_text(hipError_t e);

#define SL2_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t _e = (call);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      char _buf[512];                                                                          \
      snprintf(_buf, sizeof(_buf), "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(_e)); \
      sl2::set_error(_buf);                                                                    \
      return SL2_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// feature flag bits (f_flags)
enum : int {
  FF_ACTIVE = 1,      // slot holds a live feature
  FF_SCHEDULED = 2,   // scheduled_for_termination_flag_
  FF_SELECTED = 4,    // selected_flag_
  FF_SUCCESS = 8,     // successful_measurement_flag_
  FF_VISIBLE = 16,    // passed visibility_test in the last selection
  FF_USED = 32,       // slot was ever used (deleted features keep FF_USED)
  FF_PARTIAL = 64,    // label reserved by a partially initialised feature (its 6 states live at ppos + 6 k, k = its partial slot)
};

constexpr int kTrajCapacity = 1000;  // monoslam.cpp:174
// feature initialisation (one partially initialised feature per sequence)
constexpr int kMaxParticles = 1024;      // upper bound of params.number_of_particles (k_map_particles: one thread per particle)
constexpr int kParticleDoubles = 12;     // lambda, probability, cumulative, h[2], z[2], SInv(00,01,11), detS, success
// Large search windows are cut into UNITS of a few bands (32 x 16 candidate positions each) that any wavefront of the search
// launch may take (sl2_search.hip: m4_big_windows).  sl2_engine::srch_big, ints: [0] units allocated this step (k_select),
// [1] windows shared out in the last completed step (sl2_get_step_work), [2] next unit to hand out, [3] windows shared out this
// step; then per unit: its entry (sequence, selected position, first unit of its window, units of its window), the count
// of finished units of the window (kept at the window's first unit) and its partial result (8 ints).
// Default threshold: 8 bands at 320 x 240 (150 bands a frame; mapping workload, step: 0.734 ms never, 0.687 at 24, 0.665 at 8,
// 0.661 at 4), the same twentieth of the frame at other sizes - what is shared out must be the rare window that is far larger
// than the rest, not the typical one: at 1280 x 720 x 500 features a threshold of 8 bands listed 8192 windows a step and
// the 2048 trailing workgroups became the search (2.19 against 1.24 ms, profiles/r04_search_shared_ab.txt).
constexpr int kSrchSplitDefault = 8;
__host__ __device__ inline int srch_split_default(int width, int height) {
  const int frame_bands = ((width + 31) / 32) * ((height + 15) / 16);
  const int t = frame_bands * kSrchSplitDefault / 150;
  return t > kSrchSplitDefault ? t : kSrchSplitDefault;
}
constexpr int kSrchBigUnits = 16384;                  // units per step and sequence group (a 320 x 240 window: 38); windows beyond stay with their own wavefront
constexpr int kSrchBigSlots = 64;                     // units per window at most: its partial results are combined by one wavefront, a lane each
constexpr int kSrchBigMinBands = 4;                   // bands per unit at least
constexpr int kSrchSharedNu = -(1 << 30);             // the width a shared window's record carries (srch_sel[..][4]; the true one: [14])
constexpr int kSrchBigEntries = 4;                    // int offsets into srch_big
constexpr int kSrchBigDone = kSrchBigEntries + 4 * kSrchBigUnits;
constexpr int kSrchBigParts = kSrchBigDone + kSrchBigUnits;
constexpr int kSrchBigInts = kSrchBigParts + 8 * kSrchBigUnits;
__host__ __device__ inline int srch_unit_bands(int bands) { const int g = (bands + kSrchBigSlots - 1) / kSrchBigSlots; return g > kSrchBigMinBands ? g : kSrchBigMinBands; }
// Partially initialised features: up to kMaxPartial per sequence (params.max_features_to_init_at_once, monoslam.cpp:163-167).
// part_i / part_d = the per-SEQUENCE record: feature_init_info_vector_.size(), the partial slots in the vector's order (a
// conversion or deletion erases an entry, the later ones move up), the image selection and the counters; ps_i / ps_d = one
// record per PARTIAL SLOT k (FeatureInitInfo): its six states live in columns ppos + 6 k of x / P.
constexpr int kMaxPartial = 4;
constexpr int kPartInts = 16, kPartDoubles = 4;
enum : int { kPartCount = 0, kPartOrder /* kMaxPartial ints */, kPartUU = 5, kPartVV, kPartRegionValid,
             kPartRegion /* 4 ints */, kPartInitialised = 12, kPartConverted, kPartDeleted, kPartCreated };
constexpr int kPsInts = 8, kPsDoubles = 4;          // ps_d: mean, covariance of lambda
enum : int { kPsActive = 0, kPsLabel /* the feature SLOT that holds its label */, kPsAttempts, kPsNp, kPsMaking };
constexpr int kWorkDoubles = 5;     // per-sequence work counters of a step (work[]): window bytes, searches, candidates, exact
                                    // fallbacks, 16 x 16 candidate tiles of the matrix-core search
constexpr int kCholBlock = 32;       // block size of the blocked Cholesky / forward substitution
constexpr int kPatchStride = 288;    // bytes per stored template: 121 raw bytes (+7 pad), then at byte
                                     // 128 the packed form: 33 dwords (11 rows x 12 bytes, byte 11 = 0),
                                     // sum g0, sum g0^2, flag (patch sigma >= 10), pad
constexpr int kPatchPackedOffset = 128;

// XCD-aware block -> (sequence, tile) mapping.  MI355X dispatches workgroup L to XCD L % 8 and
// every XCD has its own 4 MB L2; all tiles of one sequence share that sequence's operands
// (V^T, L, the frame), so they are placed on ONE XCD: the grid is 1-D with
// tiles * roundup(B, 8) blocks and block L works on sequence (L/8/tiles)*8 + L%8, tile (L/8)%tiles.
// (Placement is a speed matter only; any mapping is correct.)
constexpr int kXcds = 8;
inline int xcd_grid(int tiles, int B) { return tiles * round_up(B, kXcds); }
#if defined(__HIPCC__)
__device__ __forceinline__ bool xcd_map(int tiles, int B, int* seq, int* tile) {
  const int L = blockIdx.x;
  const int xcd = L % kXcds, slot = L / kXcds;
  *seq = (slot / tiles) * kXcds + xcd;
  *tile = slot % tiles;
  return *seq < B;
}
#endif

struct KernelTimer {
  std::string name;
  double total_ms = 0.0;
  int64_t launches = 0;
};
struct PendingEvent {
  int timer;
  hipEvent_t start, stop;
};

}  // namespace sl2

// The opaque engine object of the C ABI.
struct sl2_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  sl2::CameraParams cam;
  sl2_params prm;
  int B = 0;         // sequences
  int N = 0;         // feature capacity per sequence
  int ld = 0;        // leading dimension of x / P / At / Vt rows (>= 13 + 3N + 1, multiple of 64)
  int nsel_max = 0;  // max features measured per frame
  int mld = 0;       // leading dimension of the innovation system (>= 2 nsel_max, multiple of 32)
  int nblk_max = 0;  // mld / 32

  // ---- persistent SLAM state (device) ----
  double* x = nullptr;        // [B][ld]        total state: xv(13), y_0(3), y_1(3) ... ; x[ld-1] unused
  double* P = nullptr;        // [B][ld][ld]    total covariance, dense; row/col ld-1 always zero
  uint8_t* patch = nullptr;   // [B][N][128]    11x11 templates
  int* patch_sums = nullptr;  // [B][N][2]      (sum g0, sum g0^2) of each template
  double* xp_org = nullptr;   // [B][N][8]      xp_org_ (7 used)
  int* f_flags = nullptr;     // [B][N]
  int* n_slots = nullptr;     // [B]            slots in use (live, reserved or retired features), list order = slot order
  int* f_label = nullptr;     // [B][N]         Feature::label_ of the slot (slots are compacted when they run out, labels never reused)
  int* next_label = nullptr;  // [B]            next_free_label_
  int* attempted = nullptr;   // [B][N]
  int* successful = nullptr;  // [B][N]
  double* traj = nullptr;     // [B][kTrajCapacity][3]
  int* traj_count = nullptr;  // [B]  total pushes
  double* last_r = nullptr;   // [B][3] scratch motion_model_->rRES_ (Q12)
  int* status = nullptr;      // [B]
  double* pos_log = nullptr;  // [B][kTrajCapacity][3] xv[0:3] after every step (the true trajectory, cf. Q12)
  int* pos_count = nullptr;   // [B] steps logged so far (device-side, so that a captured step needs no per-step argument)
  long long steps_done = 0;
  int chol_variant = 1;       // 1 = one-launch left-looking Cholesky (k_chol_left; the product's only path); TEST build: 0 = launch-per-block kernels
  void* chol_trace = nullptr; // development only (SL2_CHOL_TRACE builds): per-wave cycle stamps of k_chol_fused4
  int build_variant = 1;      // 1 = k_build_AS (A and S in one pass over the measured features' rows of P; the product's only path); TEST build: 0 = k_build_A then k_build_S
  int fwd_variant = 1;        // forward substitution: 1 = L through LDS + solved rows in registers (<= 8 blocks, default), 0 = operands re-read from memory (any size)
  // ---- feature initialisation ----
  int ppos = 0;                          // first column of the partial features' states (13 + 3N); slot k at ppos + 6 k
  int kpart = 1;                         // partial slots per sequence: params.max_features_to_init_at_once, 1 .. kMaxPartial
  int* part_i = nullptr;                 // [B][kPartInts]
  double* part_d = nullptr;              // [B][kPartDoubles]  [2] = evbest of the last detection
  int* ps_i = nullptr;                   // [B][kpart][kPsInts]
  double* ps_d = nullptr;                // [B][kpart][kPsDoubles]
  int pcap = 128;                        // particle slots per partial feature: roundup(params.number_of_particles, 64)
  double* particles = nullptr;           // [B][kpart][pcap][kParticleDoubles]
  // Q28 (feature.cpp:254): a conversion moves the LATER features' position_in_total_state_vector_ by 6 instead of 3, and the
  // reference then places their dh_by_dy blocks three columns early in H (monoslam.cpp:564).  pos_err = how far a slot's
  // recorded position lies below its true one; f_hcol = the engine column its H block therefore lands on (k_search_score
  // recomputes it for sequences that carry such an error; every other sequence uses 13 + 3 slot).
  int* pos_err = nullptr;                // [B][N]
  int* pos_err_any = nullptr;            // [B]
  int* f_hcol = nullptr;                 // [B][N]
  unsigned long long* rand48 = nullptr;  // [B]  drand48 state (srand48(0) at Init, monoslam.cpp:1968)
  double* prev_r = nullptr;              // [B][3] camera position before the prediction (speed estimate, :121-124)
  int* me_desc = nullptr;                // [B][kpart][pcap][8] search ellipses of the particles
  double* score_map = nullptr;           // [B][kpart][H][W] score cache of an OVERSIZED multi-ellipse search (allocated on first use)
  int* me_big_list = nullptr;            // [B * kpart] (sequence, partial slot) jobs too large for the one-workgroup form, this step
  int* me_big_count = nullptr;           // [1]
  bool mapping_used = false;
  int* init_uv = nullptr;                // [B][2] pixel selections of sl2_initialise_feature (allocated on first use)
  // ---- whole-step HIP graphs (small batches are launch-bound: ~12 kernels per step) ----
  struct StepGraph { const void* frames; size_t stride; int save_trajectory, enable_mapping, tail, small; hipGraphExec_t exec; };
  bool graph_mode = false;
  std::vector<StepGraph> step_graphs;
  int build_split = 0;        // development switches (TEST build only: SL2_BUILD_SPLIT, SL2_SCORE_THREADS, SL2_NO_KSPLIT read
  int score_threads = 0;      // once at sl2_create): 0 = the engine's own choice
  int no_ksplit = 0;
  int panel_from = 16;        // systems of more 32-blocks than this are factored panel-wise (launch_chol_panels) ...
  int group_from = 13;        // ... and substituted in groups of eight block rows (launch_fwdsub_grouped)
  int fwd_group = 8;          // block rows per group of the grouped substitution (TEST build: SL2_FWD_GROUP = 4 | 8)
  int chol_panel = 8;         // 32-blocks per panel of the large-map Cholesky: 8 (256 columns; 4.3 ms against 5.05 with 4 at
                              // 512 x n = 1513, profiles/r03_c5_chol_panel_ab.txt); TEST build: SL2_CHOL_PANEL = 4 | 8
  int build_lds_min = 0;      // minimum dynamic LDS bytes per k_build_AS workgroup (caps the workgroups a CU takes; TEST build: SL2_BUILD_LDS_MIN)
  int search_lds_pad = 0;     // extra dynamic LDS bytes per k_search_mfma workgroup (TEST build: SL2_SEARCH_LDS_PAD): an occupancy probe
  int search_chunk = 0;       // selected positions per wavefront of k_search_mfma (TEST build: SL2_SEARCH_CHUNK); 0 = the engine's own choice
  int search_variant = 1;     // 0 = exact kernel (one candidate per lane), 1 = int8 matrix-core walk (default)
  // How large the live maps are, known on the host WITHOUT a synchronisation (the choice of the step's kernels): finalize
  // publishes the batch's maximum of n_slots per step to pinned memory (sl2_frontend_dev.hpp: finalize_body); exact values
  // are taken wherever a call synchronises anyway (sl2_add_known_features, the "initialise feature" buttons).
  int* slots_max_dev = nullptr;                  // [2] device: maxima of the steps in flight (step parity)
  unsigned long long* slots_mail = nullptr;      // pinned + mapped host word: (step << 32) | max n_slots of the step before it
  unsigned long long* slots_mail_dev = nullptr;  // its device address
  unsigned long long* parts_mail = nullptr;      // the word after it, one-sequence engines: (steps completed << 32) | partially initialised features left by k_map_update
  unsigned long long* parts_mail_dev = nullptr;
  int place_candidates = 0;                      // sl2_create's placement of the large matrices (sl2_engine.hip: place_large_matrices): sets probed,
  float place_syrk_ms[2] = {0, 0};               // k_syrk on the kept (P, V^T) and on the slowest pair probed
  float place_kept_ms[4] = {0, 0, 0, 0};         // probe time of the kept P, V^T, A^T, S and of the slowest candidate of each size
  float place_worst_ms[3] = {0, 0, 0};
  long long parts_block_step = -1;               // a step of this index must not trust parts_mail (a feature was initialised by hand since)
  int slots_exact = 0;                           // max n_slots over the batch when last read back ...
  long long slots_exact_step = 0;                // ... and steps_done at that point
  int step_fusion = 1;        // small maps (sl2_small.hip: ld <= 128, one 32-row innovation block) step in three launches instead of ten; 0 = never (sl2_set_step_fusion)
  int search_split = sl2::kSrchSplitDefault;   // windows of at least this many 32 x 16 bands are shared out over wavefronts (0 = never); sl2_create: srch_split_default, then sl2_set_search_split
  // ---- large search windows (round 4): the step's units of work for every wavefront of k_search_mfma (layout: kSrchBig* above) ----
  int* srch_big = nullptr;    // per sequence GROUP (allocated by build_groups)

  // ---- per-frame feature scratch (device), indexed [B][N] ----
  double* f_h = nullptr;      // [..][2]
  double* f_Hx = nullptr;     // [..][14]
  double* f_Hy = nullptr;     // [..][6]
  double* f_R = nullptr;      // [..]
  double* f_S = nullptr;      // [..][4]
  double* f_score = nullptr;  // [..]
  double* f_z = nullptr;      // [..][2]  (persistent: untouched on failure, Q4)
  double* f_nu = nullptr;     // [..][2]
  int* sel_idx = nullptr;     // [B][N]   selected feature slots in selection order
  int* n_sel = nullptr;       // [B]
  int* n_vis = nullptr;       // [B]
  int* meas_ok = nullptr;     // [B][N]   per selected position k
  double* meas_score = nullptr;  // [B][N]
  int* succ_idx = nullptr;    // [B][N]   successful feature slots, ascending (slot order)
  int* f_arow = nullptr;      // [B][N]   per slot: first row of A^T / S of its measurement (2 x rank among the successes), -1 = none this frame
  int* m_count = nullptr;     // [B]      number of successful features (m = 2 * m_count)
  double* work = nullptr;     // [B][kWorkDoubles]   window bytes, searched, candidates, exact-fallback searches, candidate tiles
  int* srch_i = nullptr;      // [B][N][8]  per-feature search window: ucentre, vcentre, urelstart, nu, vrelstart, nv, hw, hh
  double* srch_d = nullptr;   // [B][N][4]  PuInv (a, b, c), pad
  int* srch_res = nullptr;    // [B][N][8]  per selected position: code, u, v, S1, S2, X, ncand, pad
  int* srch_sel = nullptr;    // [B][N][16] per selected position k (written by k_select): slot f, the 7 window ints of srch_i, then PuInv (a, b, c) as 3 doubles, pad - one 64-byte line, so that the search kernel needs ONE round trip for it

  // ---- EKF update workspaces (device) ----
  double* At = nullptr;    // [B][mld][ld]   (P H^T)^T, k-major; column ld-1 carries nu
  double* Vt = nullptr;    // [B][mld][ld]   L^-1 (P H^T)^T
  double* St = nullptr;    // [B][mld][mld]  St[c][r] = S[r][c]; overwritten by L (same layout)
  double* LinvT = nullptr; // [B][nblk_max][32][32]  LinvT[p][k] = (L_JJ^-1)[k][p]

  // ---- engine-owned staging of the state accessors (no allocation per call; released by sl2_destroy) ----
  void* snap_stage = nullptr;     // device: the packed blob of sl2_snapshot
  void* snap_host = nullptr;      // pinned + mapped host memory the blob is streamed into (what sl2_snapshot returns)
  void* snap_host_dev = nullptr;  // its device-side address
  size_t snap_cap = 0;            // bytes of the blob area; the completion word the kernel writes last sits right behind it
  unsigned long long snap_ticket = 0;
  void* acc_dev = nullptr;        // device scratch of the get / set accessors (grown on demand)
  size_t acc_dev_bytes = 0;
  void* acc_host = nullptr;       // pinned host scratch of the same calls
  size_t acc_host_bytes = 0;

  uint8_t* frames_buf = nullptr;  // [B][W*H] staging for host frames
  const uint8_t* cur_frames = nullptr;
  size_t cur_stride = 0;

  // profiling
  bool profiling = false;
  int profile_level = 2;
  std::string profile_focus;  // level 1: bracket only these kernels (",name,name,"); empty = the four major ones
  std::vector<sl2::KernelTimer> timers;
  std::vector<sl2::PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;

  // ---- sequence groups: the batch is split into G contiguous groups, each stepped on its own
  // HIP stream, so that the latency-bound kernels of one group (blocked Cholesky, selection,
  // bookkeeping) overlap with the throughput-bound kernels of another (SYRK, substitution, search).
  // A group is a shallow copy of the root engine whose per-sequence pointers are offset.
  sl2_engine* root = nullptr;           // self for the root object
  std::vector<sl2_engine*> groups;      // root only
  int group_first = 0;                  // first sequence of this group
  hipEvent_t fork_event = nullptr;

  int timer_id(const char* name);
  void prof_begin(int id, hipStream_t st);
  void prof_end(hipStream_t st);
  int fold_events();
  int sync_all();
};

namespace sl2 {

// RAII-less helper: bracket a launch with events when profiling is on.
struct LaunchScope {
  sl2_engine* e;
  bool on;
  size_t slot = 0;
  // level 1 = only the roofline kernels (cheap: 4 event pairs per step), level 2 = every launch
  static bool wanted(const sl2_engine* r, const char* name, bool major) {
    if (!r->profiling) return false;
    if (r->profile_level >= 2) return true;
    if (!major) return false;
    if (r->profile_focus.empty()) return true;
    return r->profile_focus.find(std::string(",") + name + ",") != std::string::npos;
  }
  LaunchScope(sl2_engine* eng, const char* name, bool major = false) : e(eng), on(wanted(eng->root, name, major)) {
    if (on) { e->root->prof_begin(e->root->timer_id(name), e->stream); slot = e->root->pending.size() - 1; }
  }
  ~LaunchScope() {
    if (on) hipEventRecord(e->root->pending[slot].stop, e->stream);
  }
};

// launchers implemented in the kernel translation units (all asynchronous on e->stream)
int launch_predict(sl2_engine* e);
int launch_feature_prediction(sl2_engine* e);
int launch_select(sl2_engine* e, int n);
int launch_search(sl2_engine* e);            // the search kernel, then k_search_score
int launch_search_kernel(sl2_engine* e);     // the search kernel alone (the fused small-map step scores in k_small_back)
int launch_search_score(sl2_engine* e);
int small_step_mode(const sl2_engine* e, int slots_bound);       // sl2_small.hip: 0 = ten launches, 1 = three (both sides of the search fused), 2 = the back side only
int launch_small_front(sl2_engine* e, int n);                 // predict + feature prediction + selection in one launch
int launch_small_back(sl2_engine* e, int save_trajectory, int slots_bound);   // scoring + EKF update + normalise / delete / symmetrise in one launch
int launch_update(sl2_engine* e);
int launch_syrk_on(sl2_engine* e, const double* Vt, double* P);
int launch_finalize(sl2_engine* e, int save_trajectory);
int launch_mapping(sl2_engine* e, int enable_mapping, int save_trajectory, int slots_bound, int parts_state);
int launch_manual_init(sl2_engine* e, const int* d_uv);
int launch_auto_init(sl2_engine* e);
int launch_compact_slots(sl2_engine* e, int need);   // sl2_mapping.hip: retired slots squeezed out when a sequence lacks room for `need` more features
int write_grey_image(const char* path, const uint8_t* px, int w, int h);   // sl2_ingest.hip (PGM / PNG)

}  // namespace sl2
