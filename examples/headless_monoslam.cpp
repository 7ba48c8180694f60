// Headless driver over the C ABI: what examples/MonoSlamSceneLib1.cpp:132-142 of the
// reference does per frame — GetFrame, GoOneStep(frame, save_trajectory, enable_mapping) — without Pangolin,
// OpenCV or Eigen.  Plain C++, links only libscenelib2_amd.so.
//
//   headless_monoslam --cfg scene.cfg --frames frame_dir [--mapping] [--steps N] [--dump state.txt]
//
// scene.cfg uses the keys of the reference's data/SceneLib2.cfg ("name = value;", '#' comments): cam.*, params.*,
// state.* (rw_*, qwr_*, vw_*, ww_*, pxxR_C), fK.yi_*, fK.xp_org_J, fK.identifier (an 11x11 binary PGM, looked up
// beside the cfg).  Keys that are absent read as 0, like pangolin::Var<T>(key, 0).
#include <scenelib2_amd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

static std::map<std::string, std::string> parse_vars(const std::string& path) {
  std::map<std::string, std::string> kv;
  std::ifstream in(path);
  std::string line;
  while (std::getline(in, line)) {
    const size_t hash = line.find('#');
    if (hash != std::string::npos) line.erase(hash);
    const size_t eq = line.find('=');
    if (eq == std::string::npos) continue;
    auto trim = [](std::string s) {
      const char* ws = " \t\r\n;";
      const size_t a = s.find_first_not_of(ws);
      if (a == std::string::npos) return std::string();
      const size_t b = s.find_last_not_of(ws);
      return s.substr(a, b - a + 1);
    };
    kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
  }
  return kv;
}

static double num(const std::map<std::string, std::string>& kv, const std::string& k, double dflt = 0.0) {
  auto it = kv.find(k);
  return it == kv.end() ? dflt : atof(it->second.c_str());
}

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
  const auto kv = parse_vars(cfg);
  const std::string base = cfg.find('/') == std::string::npos ? "." : cfg.substr(0, cfg.rfind('/'));

  // MonoSLAM::Init (monoslam.cpp:1574-1969): camera, constants, xv_, Pxx_, known features
  sl2_camera cam;
  cam.width = (int)num(kv, "cam.width"); cam.height = (int)num(kv, "cam.height");
  cam.fku = (int)num(kv, "cam.fku"); cam.fkv = (int)num(kv, "cam.fkv");          // read as Var<int> (monoslam.cpp:1597-1602)
  cam.u0 = (int)num(kv, "cam.u0"); cam.v0 = (int)num(kv, "cam.v0");
  cam.kd1 = num(kv, "cam.kd1"); cam.sd = (int)num(kv, "cam.sd");
  sl2_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.delta_t = num(kv, "params.delta_t");
  prm.number_of_features_to_select = (int)num(kv, "params.number_of_features_to_select");
  prm.number_of_features_to_keep_visible = (int)num(kv, "params.number_of_features_to_keep_visible");
  prm.max_features_to_init_at_once = (int)num(kv, "params.max_features_to_init_at_once");
  prm.min_lambda = num(kv, "params.min_lambda"); prm.max_lambda = num(kv, "params.max_lambda");
  prm.number_of_particles = (int)num(kv, "params.number_of_particles");
  prm.standard_deviation_depth_ratio = num(kv, "params.standard_deviation_depth_ratio");
  prm.min_number_of_particles = (int)num(kv, "params.min_number_of_particles");
  prm.prune_probability_threshold = num(kv, "params.prune_probability_threshold");
  prm.erase_partially_init_feature_after_this_many_attempts = (int)num(kv, "params.erase_partially_init_feature_after_this_many_attempts");
  prm.minimum_attempted_measurements_of_feature = 10;   // monoslam.cpp:1875-1876
  prm.successful_match_fraction = 0.5;

  if (sl2_device_count() < 1) { fprintf(stderr, "no HIP device: this engine has no CPU path\n"); return 3; }
  sl2_engine* eng = nullptr;
  const int max_features = 128;
  CHECK(sl2_create(&cam, &prm, 1, max_features, 0, nullptr, &eng));

  double xv[13] = {num(kv, "state.rw_x"), num(kv, "state.rw_y"), num(kv, "state.rw_z"),
                   num(kv, "state.qwr_w"), num(kv, "state.qwr_x"), num(kv, "state.qwr_y"), num(kv, "state.qwr_z"),
                   num(kv, "state.vw_x"), num(kv, "state.vw_y"), num(kv, "state.vw_z"),
                   num(kv, "state.ww_x"), num(kv, "state.ww_y"), num(kv, "state.ww_z")};
  double Pxx[169];
  for (int r = 0; r < 13; ++r)
    for (int c = 0; c < 13; ++c) Pxx[r * 13 + c] = num(kv, "state.pxx" + std::to_string(r) + "_" + std::to_string(c));
  CHECK(sl2_set_vehicle_state(eng, 0, 1, xv, Pxx));
  int n_known = 0;
  for (int k = 1; kv.count("f" + std::to_string(k) + ".yi_x"); ++k) {      // AddNewKnownFeature, monoslam.cpp:1941-1957
    const std::string p = "f" + std::to_string(k) + ".";
    const double y[3] = {num(kv, p + "yi_x"), num(kv, p + "yi_y"), num(kv, p + "yi_z")};
    double xp[7];
    for (int j = 0; j < 7; ++j) xp[j] = num(kv, p + "xp_org_" + std::to_string(j));
    uint8_t patch[121];
    int w = 0, h = 0;
    auto it = kv.find(p + "identifier");
    const std::string ident = base + "/" + (it == kv.end() ? std::string("empty") : it->second);
    CHECK(sl2_read_pgm(ident.c_str(), patch, sizeof(patch), &w, &h));
    if (w != 11 || h != 11) { fprintf(stderr, "%s is not an 11x11 template\n", ident.c_str()); return 4; }
    CHECK(sl2_add_known_features(eng, 0, 1, 1, y, xp, patch));
    ++n_known;
  }

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
    CHECK(sl2_ingest_next(grab, nullptr, &d_frame, &stride));
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
