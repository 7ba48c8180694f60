// Independent sequences sharded over the GPUs of one node from plain host C++: one host thread, one
// sl2_engine and one sl2_comm per GPU; no communication inside a step, RCCL only at the edges (include/scenelib2_amd_comm.h):
//   * the frames of a step originate on rank 0 (one grabber) and are scattered to the ranks that own the sequences;
//   * after the last step every rank gathers every sequence's vehicle state (and its covariance block).
// The reference has none of this (a single instance, monoslam.h:158-218); the per-frame work is the loop of
// examples/MonoSlamSceneLib1.cpp:132-142 for every sequence.
//
//   sharded_monoslam --cfg scene.cfg --frames dir [--gpus N] [--per-gpu B] [--steps K] [--dump states.txt]
//
// Every sequence is the scene of the cfg on the frames of `dir` (the demo has one camera): all rows of the gathered result must
// therefore be equal - which is what the dump lets a test check against a single-sequence run.
#include <scenelib2_amd_comm.h>

#include <atomic>
#include <thread>

#include "scene_cfg.hpp"

#include <hip/hip_runtime_api.h>

struct Rank {
  int rank = 0, rc = 0;
  std::string err;
};

#define RCHECK(call)                                                                                             \
  do {                                                                                                           \
    const int rc_ = (call);                                                                                      \
    if (rc_ != SL2_OK) { r.rc = rc_; r.err = std::string(#call) + ": " + sl2_last_error() + " / " + sl2_comm_last_error(); return; } \
  } while (0)
#define HCHECK(call)                                                                                             \
  do {                                                                                                           \
    const hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) { r.rc = SL2_ERR_HIP; r.err = std::string(#call) + ": " + hipGetErrorString(e_); return; } \
  } while (0)

int main(int argc, char** argv) {
  std::string cfg, frames_dir, dump;
  int gpus = 1, per_gpu = 4, steps = -1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--cfg" && i + 1 < argc) cfg = argv[++i];
    else if (a == "--frames" && i + 1 < argc) frames_dir = argv[++i];
    else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
    else if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
    else if (a == "--gpus" && i + 1 < argc) gpus = atoi(argv[++i]);
    else if (a == "--per-gpu" && i + 1 < argc) per_gpu = atoi(argv[++i]);
    else { fprintf(stderr, "usage: %s --cfg scene.cfg --frames dir [--gpus N] [--per-gpu B] [--steps K] [--dump file]\n", argv[0]); return 2; }
  }
  if (cfg.empty() || frames_dir.empty() || gpus < 1 || per_gpu < 1) { fprintf(stderr, "need --cfg, --frames, --gpus >= 1, --per-gpu >= 1\n"); return 2; }
  if (sl2_device_count() < gpus) { fprintf(stderr, "--gpus %d but %d HIP device(s) visible\n", gpus, sl2_device_count()); return 3; }
  Scene sc;
  if (int rc = load_scene(cfg, sc)) return rc;
  const int total = gpus * per_gpu;
  const size_t fb = (size_t)sc.cam.width * sc.cam.height;

  std::vector<sl2_comm*> comms(gpus, nullptr);
  if (sl2_comm_create_all(gpus, nullptr, comms.data()) != SL2_OK) { fprintf(stderr, "sl2_comm_create_all: %s\n", sl2_comm_last_error()); return 1; }

  // rank 0 owns the grabber; its frame count is what every rank steps
  const char* dirs[1] = {frames_dir.c_str()};
  sl2_ingest* grab = nullptr;
  if (sl2_ingest_open(dirs, 1, sc.cam.width, sc.cam.height, 0, 8, &grab) != SL2_OK) { fprintf(stderr, "%s\n", sl2_last_error()); return 1; }
  int n = sl2_ingest_frame_count(grab);
  if (steps >= 0 && steps < n) n = steps;
  printf("%d GPU(s) x %d sequences, %d known features, %d frames\n", gpus, per_gpu, sc.n_known, n);

  const int row = sl2_gather_row_doubles(SL2_GATHER_VEHICLE_PXX, 32);
  std::vector<double> gathered((size_t)total * row);
  std::vector<Rank> ranks(gpus);
  // a step's hand-over between the ranks' threads: rank 0 publishes the step whose frames it has staged, the others follow
  std::atomic<int> staged{-1}, arrived{0};
  auto body = [&](int rk) {
    Rank& r = ranks[rk];
    r.rank = rk;
    HCHECK(hipSetDevice(rk));
    hipStream_t st;
    HCHECK(hipStreamCreate(&st));
    sl2_engine* eng = nullptr;
    RCHECK(sl2_create(&sc.cam, &sc.prm, per_gpu, 32, rk, st, &eng));
    std::vector<double> xv((size_t)13 * per_gpu), Pxx((size_t)169 * per_gpu);
    for (int b = 0; b < per_gpu; ++b) { memcpy(&xv[13 * b], sc.xv, sizeof(sc.xv)); memcpy(&Pxx[169 * b], sc.Pxx, sizeof(sc.Pxx)); }
    RCHECK(sl2_set_vehicle_state(eng, 0, per_gpu, xv.data(), Pxx.data()));
    for (int b = 0; b < per_gpu; ++b)
      for (int k = 0; k < sc.n_known; ++k) RCHECK(sl2_add_known_features(eng, b, 1, 1, &sc.y[3 * k], &sc.xp[7 * k], &sc.patches[121 * k]));
    uint8_t *mine = nullptr, *all = nullptr;
    HCHECK(hipMalloc((void**)&mine, fb * per_gpu));
    if (rk == 0) HCHECK(hipMalloc((void**)&all, fb * total));
    for (int k = 0; k < n; ++k) {
      if (rk == 0) {
        const uint8_t* d_frame = nullptr;
        size_t stride = 0;
        RCHECK(sl2_ingest_next(grab, st, &d_frame, &stride));
        for (int s = 0; s < total; ++s) HCHECK(hipMemcpyAsync(all + (size_t)s * fb, d_frame, fb, hipMemcpyDeviceToDevice, st));   // one camera, `total` sequences
        staged.store(k);
      } else {
        while (staged.load() < k && !ranks[0].rc) std::this_thread::yield();
        if (ranks[0].rc) return;
      }
      RCHECK(sl2_scatter_frames(comms[rk], 0, all, fb, total, mine, st));
      RCHECK(sl2_go_one_step(eng, mine, fb, /*frames_on_device=*/1, /*save_trajectory=*/0, /*enable_mapping=*/0));
    }
    double* d_out = nullptr;
    HCHECK(hipMalloc((void**)&d_out, sizeof(double) * gathered.size()));
    RCHECK(sl2_gather_states(comms[rk], eng, SL2_GATHER_VEHICLE_PXX, d_out, st));
    HCHECK(hipStreamSynchronize(st));
    if (rk == 0) HCHECK(hipMemcpy(gathered.data(), d_out, sizeof(double) * gathered.size(), hipMemcpyDeviceToHost));
    arrived.fetch_add(1);
    (void)hipFree(d_out); (void)hipFree(mine); if (all) (void)hipFree(all);
    sl2_destroy(eng);
    (void)hipStreamDestroy(st);
  };
  std::vector<std::thread> th;
  for (int rk = 0; rk < gpus; ++rk) th.emplace_back(body, rk);
  for (auto& t : th) t.join();
  int bad = 0;
  for (const Rank& r : ranks) if (r.rc) { fprintf(stderr, "rank %d: %s\n", r.rank, r.err.c_str()); bad = 1; }
  sl2_ingest_close(grab);
  for (sl2_comm* c : comms) sl2_comm_destroy(c);
  if (bad) return 1;
  for (int s = 0; s < total; s += (total > 8 ? total / 8 : 1))
    printf("sequence %4d (rank %d)  r = (% .6f % .6f % .6f)  tr Pxx = %.3e\n", s, s / per_gpu, gathered[(size_t)s * row], gathered[(size_t)s * row + 1],
           gathered[(size_t)s * row + 2], gathered[(size_t)s * row + 13] + gathered[(size_t)s * row + 13 + 14] + gathered[(size_t)s * row + 13 + 28]);
  if (!dump.empty()) {
    FILE* f = fopen(dump.c_str(), "w");
    if (!f) { fprintf(stderr, "cannot write %s\n", dump.c_str()); return 5; }
    for (int s = 0; s < total; ++s) {
      for (int i = 0; i < row; ++i) fprintf(f, "%.17g ", gathered[(size_t)s * row + i]);
      fprintf(f, "\n");
    }
    fclose(f);
  }
  return 0;
}
