// The scene file of the examples: the keys of the reference's data/SceneLib2.cfg ("name = value;", '#' comments) - cam.*,
// params.*, state.* (rw_*, qwr_*, vw_*, ww_*, pxxR_C), fK.yi_*, fK.xp_org_J, fK.identifier (an 11x11 binary PGM, looked up beside
// the cfg).  Keys that are absent read as 0, like pangolin::Var<T>(key, 0).  What MonoSLAM::Init reads (monoslam.cpp:1574-1969).
#pragma once
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


struct Scene {
  sl2_camera cam;
  sl2_params prm;
  double xv[13], Pxx[169];
  std::vector<double> y, xp;          // [n][3], [n][7]
  std::vector<uint8_t> patches;       // [n][121]
  int n_known = 0;
};

// 0 = ok; otherwise the message has been printed
static int load_scene(const std::string& cfg, Scene& sc) {
  const auto kv = parse_vars(cfg);
  const std::string base = cfg.find('/') == std::string::npos ? "." : cfg.substr(0, cfg.rfind('/'));
  sl2_camera& cam = sc.cam;
  cam.width = (int)num(kv, "cam.width"); cam.height = (int)num(kv, "cam.height");
  cam.fku = (int)num(kv, "cam.fku"); cam.fkv = (int)num(kv, "cam.fkv");          // read as Var<int> (monoslam.cpp:1597-1602)
  cam.u0 = (int)num(kv, "cam.u0"); cam.v0 = (int)num(kv, "cam.v0");
  cam.kd1 = num(kv, "cam.kd1"); cam.sd = (int)num(kv, "cam.sd");
  sl2_params& prm = sc.prm;
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
  const char* names[13] = {"rw_x", "rw_y", "rw_z", "qwr_w", "qwr_x", "qwr_y", "qwr_z", "vw_x", "vw_y", "vw_z", "ww_x", "ww_y", "ww_z"};
  for (int i = 0; i < 13; ++i) sc.xv[i] = num(kv, std::string("state.") + names[i]);
  for (int r = 0; r < 13; ++r)
    for (int c = 0; c < 13; ++c) sc.Pxx[r * 13 + c] = num(kv, "state.pxx" + std::to_string(r) + "_" + std::to_string(c));
  for (int k = 1; kv.count("f" + std::to_string(k) + ".yi_x"); ++k) {      // AddNewKnownFeature, monoslam.cpp:1941-1957
    const std::string p = "f" + std::to_string(k) + ".";
    sc.y.push_back(num(kv, p + "yi_x")); sc.y.push_back(num(kv, p + "yi_y")); sc.y.push_back(num(kv, p + "yi_z"));
    for (int j = 0; j < 7; ++j) sc.xp.push_back(num(kv, p + "xp_org_" + std::to_string(j)));
    uint8_t patch[121];
    int w = 0, h = 0;
    auto it = kv.find(p + "identifier");
    const std::string ident = base + "/" + (it == kv.end() ? std::string("empty") : it->second);
    if (sl2_read_pgm(ident.c_str(), patch, sizeof(patch), &w, &h) != SL2_OK) { fprintf(stderr, "%s: %s\n", ident.c_str(), sl2_last_error()); return 1; }
    if (w != 11 || h != 11) { fprintf(stderr, "%s is not an 11x11 template\n", ident.c_str()); return 4; }
    sc.patches.insert(sc.patches.end(), patch, patch + 121);
    ++sc.n_known;
  }
  return 0;
}
