// Headless driver over the C ABI: what examples/MonoSlamSceneLib1.cpp:132-142 of the
// reference does per frame — GetFrame, GoOneStep(frame, save_trajectory, enable_mapping) — without Pangolin,
// OpenCV or Eigen.  Plain C++, links only libscenelib2_amd.so.
//
//   headless_monoslam --cfg scene.cfg --frames frame_dir [--mapping] [--steps N] [--dump state.txt]
//
// scene.cfg uses the keys of the reference's data/SceneLib2.cfg ("name = value;", '#' comments): cam.*, params.*,
// state.* (rw_*, qwr_*, vw_*, ww_*, pxxR_C), fK.yi_*, fK.xp_org_J, fK.identifier (an 11x11 binary PGM, looked up
// beside the cfg).  Keys that are absent read as 0, like pangolin::Var<T>(key, 0).
#include "scene_cfg.hpp"

#define CHECK(call)                                                                        \
  do {                                                                                     \
    const int rc_ = (call);                                                                \
    if (rc_ != SL2_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, sl2_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  std::string cfg, frames_dir, dump;
  int mapping = 0, steps = -1;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--cfg" && i + 1 < argc) cfg = argv[++i];
    else if (a == "--frames" && i + 1 < argc) frames_dir = argv[++i];
    else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
    else if (a == "--steps" && i + 1 < argc) steps = atoi(argv[++i]);
    else if (a == "--mapping") mapping = 1;
    else { fprintf(stderr, "usage: %s --cfg scene.cfg --frames dir [--mapping] [--steps N] [--dump file]\n", argv[0]); return 2; }
  }
  if (cfg.empty() || frames_dir.empty()) { fprintf(stderr, "need --cfg and --frames\n"); return 2; }
  // MonoSLAM::Init (monoslam.cpp:1574-1969): camera, constants, xv_, Pxx_, known features
  Scene sc;
  if (int rc = load_scene(cfg, sc)) return rc;
  const sl2_camera& cam = sc.cam;
  if (sl2_device_count() < 1) { fprintf(stderr, "no HIP device: this engine has no CPU path\n"); return 3; }
  sl2_engine* eng = nullptr;
  const int max_features = 128;
  CHECK(sl2_create(&sc.cam, &sc.prm, 1, max_features, 0, nullptr, &eng));
  CHECK(sl2_set_vehicle_state(eng, 0, 1, sc.xv, sc.Pxx));
  const int n_known = sc.n_known;
  for (int k = 0; k < n_known; ++k)
    CHECK(sl2_add_known_features(eng, 0, 1, 1, &sc.y[3 * k], &sc.xp[7 * k], &sc.patches[121 * k]));

  // FrameGrabber / FileGrabber
  const char* dirs[1] = {frames_dir.c_str()};
  sl2_ingest* grab = nullptr;
  CHECK(sl2_ingest_open(dirs, 1, cam.width, cam.height, 0, 8, &grab));
  int n = sl2_ingest_frame_count(grab);
  if (steps >= 0 && steps < n) n = steps;
  printf("%d known features, %d frames, mapping %s\n", n_known, n, mapping ? "on" : "off");

  for (int k = 0; k < n; ++k) {                     // the loop of examples/MonoSlamSceneLib1.cpp:132-142
    const uint8_t* d_frame = nullptr;
    size_t stride = 0;
    CHECK(sl2_ingest_next(grab, sl2_get_stream(eng), &d_frame, &stride));     // (the copy of the frame after this one starts here, under the step below)
    CHECK(sl2_go_one_step(eng, d_frame, stride, /*frames_on_device=*/1, /*save_trajectory=*/1, mapping));
    if (k % 10 == 9 || k + 1 == n) {
      double x13[13], P[169];
      int32_t counters[3], labels[128];
      CHECK(sl2_get_vehicle_state(eng, 0, 1, x13, P));
      CHECK(sl2_get_selection(eng, 0, labels, 128, counters));
      int32_t size = 0;
      CHECK(sl2_get_total_state_sizes(eng, 0, 1, &size));
      printf("frame %4d  r = (% .4f % .4f % .4f)  visible %d  measured %d  state size %d\n", k, x13[0], x13[1], x13[2], counters[0],
             counters[2] / 2, size);
    }
  }
  if (!dump.empty()) {
    int32_t size = 0;
    CHECK(sl2_get_total_state_sizes(eng, 0, 1, &size));
    std::vector<double> x(size);
    CHECK(sl2_get_total_state(eng, 0, x.data(), size));
    FILE* f = fopen(dump.c_str(), "w");
    if (!f) { fprintf(stderr, "cannot write %s\n", dump.c_str()); return 5; }
    for (double v : x) fprintf(f, "%.17g\n", v);
    fclose(f);
  }
  sl2_ingest_close(grab);
  sl2_destroy(eng);
  return 0;
}
