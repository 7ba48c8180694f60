// MonoSLAM-shaped C++ adapter over the C ABI.
//
// Header-only.  It carries the public surface of SceneLib2::MonoSLAM that the reference's example and its GraphicTool use
// (monoslam.h:69-219, examples/MonoSlamSceneLib1.cpp:55-142, graphic/graphictool.cpp:130-358) under the reference's member
// names, so that a driver written against the reference reads the same here:
//
//     SceneLib2Amd::MonoSLAM slam;
//     slam.Init("SceneLib2.cfg");                                       // monoslam.cpp:1574-1969
//     while (grabber.GetFrame(k++, &frame))                             // examples/MonoSlamSceneLib1.cpp:132-142
//       slam.GoOneStep(frame, save_trajectory, enable_mapping);         // monoslam.cpp:108-180
//     slam.xv_, slam.Pxx_, slam.feature_list_[i]->y_, ->h_, ->S_, ...   // what GraphicTool draws
//
// Differences, all forced by the absence of Eigen / OpenCV / Pangolin in this build environment:
//   * vectors and matrices are std::array / std::vector<double> (row-major) instead of Eigen types;
//   * a frame is a `Frame` view (pointer, width, height) instead of cv::Mat — `Frame{mat.data, mat.cols, mat.rows}` at an
//     integrator's site; frames must be 8-bit, single channel, continuous;
//   * Init() parses the cfg itself ("name = value;", '#' comments: the pangolin::Var file format), absent keys read 0.
// The per-frame arithmetic happens on the GPU behind sl2_go_one_step; after every call the public members are refreshed
// from the engine by ONE sl2_snapshot call (refresh_public_members: one kernel, one synchronisation), so reads between
// frames see what the reference's members would hold.
#ifndef SCENELIB2_AMD_MONOSLAM_HPP
#define SCENELIB2_AMD_MONOSLAM_HPP

#include <scenelib2_amd.h>

#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace SceneLib2Amd {

struct Frame {                 // stands in for cv::Mat (8-bit, 1 channel, continuous)
  const uint8_t* data = nullptr;
  int cols = 0, rows = 0;
  bool on_device = false;      // true: `data` is device memory (e.g. from sl2_ingest_next)
};

struct Camera {                // camera.h:61-77
  int width_ = 0, height_ = 0;
  double fku_ = 0, fkv_ = 0, kd1_ = 0;
  std::array<double, 2> centre_{{0, 0}};
  int measurement_sd_ = 0;
};

struct Feature {               // feature.h:56-143 (the members read from outside)
  int label_ = 0;
  bool fully_initialised_flag_ = true;
  bool selected_flag_ = false;
  bool successful_measurement_flag_ = false;
  int attempted_measurements_of_feature_ = 0;
  int successful_measurements_of_feature_ = 0;
  int position_in_total_state_vector_ = 0;
  int state_size_ = 3;                         // FeatureModel::kFeatureStateSize_: 3 full, 6 partial
  std::vector<double> y_;                      // 3, or 6 while partially initialised
  std::array<double, 7> xp_org_{};
  std::array<double, 2> h_{}, z_{}, nu_{};
  std::array<double, 4> S_{}, R_{};            // 2x2 row-major
  std::array<double, 26> dh_by_dxv_{};         // 2x13 (columns 7..12 are zero, monoslam.cpp:298-300)
  std::array<double, 6> dh_by_dy_{};           // 2x3
  std::vector<double> Pxy_;                    // 13 x state_size_
  std::vector<double> Pyy_;                    // state_size_ x state_size_
  std::array<uint8_t, SL2_PATCH_BYTES> patch_{};
};

struct Particle {              // feature_init_info.h:46-75
  double lambda_ = 0, probability_ = 0, cumulative_probability_ = 0;
  std::array<double, 2> m_h_{}, m_z_{};
  std::array<double, 4> m_SInv_{};
  double m_detS_ = 0;
  bool m_successful_measurement_flag_ = false;
};

struct FeatureInitInfo {       // feature_init_info.h:77-118
  Feature* fp_ = nullptr;
  std::vector<Particle> particle_vector_;
  double mean_ = 0, covariance_ = 0;
  int number_of_match_attempts_ = 0;
  bool making_measurement_on_this_step_flag_ = false;
};

class MonoSLAM {
 public:
  explicit MonoSLAM(int max_features = 128, int device = 0) : max_features_(max_features), device_(device) {}
  ~MonoSLAM() { if (eng_) sl2_destroy(eng_); }
  MonoSLAM(const MonoSLAM&) = delete;
  MonoSLAM& operator=(const MonoSLAM&) = delete;

  // MonoSLAM::Init (monoslam.cpp:1574-1969).  Template files (fK.identifier, 11x11 binary PGM or PNG) are looked up
  // beside the cfg.  Throws std::runtime_error like FileGrabber does for a missing directory (filegrabber.cpp:81).
  void Init(const std::string& config_path) {
    const auto kv = parse_vars(config_path);
    const std::string base = config_path.find('/') == std::string::npos ? "." : config_path.substr(0, config_path.rfind('/'));
    camera_.reset(new Camera());
    camera_->width_ = (int)num(kv, "cam.width"); camera_->height_ = (int)num(kv, "cam.height");
    camera_->fku_ = (int)num(kv, "cam.fku"); camera_->fkv_ = (int)num(kv, "cam.fkv");       // Var<int>, monoslam.cpp:1597-1602
    camera_->centre_ = {{(double)(int)num(kv, "cam.u0"), (double)(int)num(kv, "cam.v0")}};
    camera_->kd1_ = num(kv, "cam.kd1"); camera_->measurement_sd_ = (int)num(kv, "cam.sd");
    kDeltaT_ = num(kv, "params.delta_t");
    kNumberOfFeaturesToSelect_ = (int)num(kv, "params.number_of_features_to_select");
    kNumberOfFeaturesToKeepVisible_ = (int)num(kv, "params.number_of_features_to_keep_visible");
    kMaxFeaturesToInitAtOnce_ = (int)num(kv, "params.max_features_to_init_at_once");
    kMinLambda_ = num(kv, "params.min_lambda"); kMaxLambda_ = num(kv, "params.max_lambda");
    kNumberOfParticles_ = (int)num(kv, "params.number_of_particles");
    kStandardDeviationDepthRatio_ = num(kv, "params.standard_deviation_depth_ratio");
    kMinNumberOfParticles_ = (int)num(kv, "params.min_number_of_particles");
    kPruneProbabilityThreshold_ = num(kv, "params.prune_probability_threshold");
    kErasePartiallyInitFeatureAfterThisManyAttempts_ = (int)num(kv, "params.erase_partially_init_feature_after_this_many_attempts");
    minimum_attempted_measurements_of_feature_ = 10;       // monoslam.cpp:1875-1876
    successful_match_fraction_ = 0.5;

    sl2_camera cam;
    cam.width = camera_->width_; cam.height = camera_->height_; cam.fku = camera_->fku_; cam.fkv = camera_->fkv_;
    cam.u0 = camera_->centre_[0]; cam.v0 = camera_->centre_[1]; cam.kd1 = camera_->kd1_; cam.sd = camera_->measurement_sd_;
    sl2_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.delta_t = kDeltaT_;
    prm.number_of_features_to_select = kNumberOfFeaturesToSelect_;
    prm.number_of_features_to_keep_visible = kNumberOfFeaturesToKeepVisible_;
    prm.max_features_to_init_at_once = kMaxFeaturesToInitAtOnce_;
    prm.min_lambda = kMinLambda_; prm.max_lambda = kMaxLambda_;
    prm.number_of_particles = kNumberOfParticles_;
    prm.standard_deviation_depth_ratio = kStandardDeviationDepthRatio_;
    prm.min_number_of_particles = kMinNumberOfParticles_;
    prm.prune_probability_threshold = kPruneProbabilityThreshold_;
    prm.erase_partially_init_feature_after_this_many_attempts = kErasePartiallyInitFeatureAfterThisManyAttempts_;
    prm.minimum_attempted_measurements_of_feature = minimum_attempted_measurements_of_feature_;
    prm.successful_match_fraction = successful_match_fraction_;
    if (sl2_device_count() < 1) throw std::runtime_error("MonoSLAM::Init: no HIP device (the engine has no CPU path)");
    if (eng_) { sl2_destroy(eng_); eng_ = nullptr; }
    traj_seen_ = 0; patch_seen_ = 0; patch_cache_.clear(); feature_list_.clear(); trajectory_store_.clear();
    check(sl2_create(&cam, &prm, 1, max_features_, device_, nullptr, &eng_), "sl2_create");

    static const char* xv_keys[13] = {"state.rw_x", "state.rw_y", "state.rw_z", "state.qwr_w", "state.qwr_x", "state.qwr_y", "state.qwr_z",
                                      "state.vw_x", "state.vw_y", "state.vw_z", "state.ww_x", "state.ww_y", "state.ww_z"};
    for (int i = 0; i < 13; ++i) xv_[i] = num(kv, xv_keys[i]);
    for (int r = 0; r < 13; ++r)
      for (int c = 0; c < 13; ++c) Pxx_[r * 13 + c] = num(kv, "state.pxx" + std::to_string(r) + "_" + std::to_string(c));
    check(sl2_set_vehicle_state(eng_, 0, 1, xv_.data(), Pxx_.data()), "sl2_set_vehicle_state");
    for (int k = 1; kv.count("f" + std::to_string(k) + ".yi_x"); ++k) {        // monoslam.cpp:1941-1957
      const std::string p = "f" + std::to_string(k) + ".";
      const std::array<double, 3> y{{num(kv, p + "yi_x"), num(kv, p + "yi_y"), num(kv, p + "yi_z")}};
      std::array<double, 7> xp;
      for (int j = 0; j < 7; ++j) xp[j] = num(kv, p + "xp_org_" + std::to_string(j));
      auto it = kv.find(p + "identifier");
      AddNewKnownFeature(y, xp, base + "/" + (it == kv.end() ? std::string("empty") : it->second));
    }
    refresh_public_members();
  }

  // MonoSLAM::AddNewKnownFeature (monoslam.cpp:1278-1291; Feature ctor feature.cpp:106-142): identifier = template file.
  void AddNewKnownFeature(const std::array<double, 3>& y_new, const std::array<double, 7>& xp_o, const std::string& identifier) {
    std::array<uint8_t, SL2_PATCH_BYTES> patch;
    int w = 0, h = 0;
    check(sl2_read_image(identifier.c_str(), patch.data(), patch.size(), &w, &h), "sl2_read_image");
    if (w != SL2_PATCH_SIZE || h != SL2_PATCH_SIZE) throw std::runtime_error(identifier + " is not an 11x11 template");
    check(sl2_add_known_features(eng_, 0, 1, 1, y_new.data(), xp_o.data(), patch.data()), "sl2_add_known_features");
  }

  // The stream the engine steps on: what sl2_ingest_next wants, so that the upload of the next frame runs under this one's step.
  void* stream() const { return eng_ ? sl2_get_stream(eng_) : nullptr; }

  // MonoSLAM::GoOneStep (monoslam.cpp:108-180).  Always true, like the reference (:179).
  bool GoOneStep(const Frame& frame, bool save_trajectory, bool enable_mapping) {
    if (!eng_ || !frame.data || frame.cols != camera_->width_ || frame.rows != camera_->height_)
      throw std::runtime_error("MonoSLAM::GoOneStep: frame does not match the camera");
    const auto t0 = std::chrono::steady_clock::now();
    check(sl2_go_one_step(eng_, frame.data, (size_t)frame.cols * frame.rows, frame.on_device ? 1 : 0, save_trajectory ? 1 : 0,
                          enable_mapping ? 1 : 0),
          "sl2_go_one_step");
    if (measure_timing_) check(sl2_synchronize(eng_), "sl2_synchronize");   // only to SPLIT the time; costs a second wait
    const auto t1 = std::chrono::steady_clock::now();
    refresh_public_members();
    const auto t2 = std::chrono::steady_clock::now();
    last_step_us_ = std::chrono::duration<double, std::micro>(t1 - t0).count();
    last_refresh_us_ = std::chrono::duration<double, std::micro>(t2 - t1).count();
    if (enable_mapping) {   // feature_list_ is unbounded in the reference; here at most max_features LIVE features: never silently
      const int32_t flags = status_flags_;     // (came with the snapshot)
      const bool full = (flags & SL2_STATUS_LABELS_EXHAUSTED) != 0, rose = full && !map_full_;
      map_full_ = full;
      if (rose)             // once per episode: the bit clears again when a deletion has made room
        throw std::runtime_error("MonoSLAM::GoOneStep: all max_features feature slots hold live features: mapping can "
                                 "initialise no features until one is deleted; construct MonoSLAM with a larger max_features");
    }
    return true;
  }

  // The seams of GoOneStep, callable one by one like the reference's public members (monoslam.cpp:118-177; kalman.h:51-52
  // through the Kalman struct below).  With enable_mapping = false, predict -> auto_select_n_features ->
  // make_measurements -> update -> finish_step is GoOneStep.
  int auto_select_n_features(int n) {                                   // monoslam.cpp:187-254 (+ the measurement predictions)
    check(sl2_auto_select_n_features(eng_, n), "sl2_auto_select_n_features");
    refresh_public_members();
    return (int)selected_feature_list_.size();
  }
  int make_measurements(const Frame& image) {                           // monoslam.cpp:336-359
    if (!image.data || image.cols != camera_->width_ || image.rows != camera_->height_)
      throw std::runtime_error("MonoSLAM::make_measurements: frame does not match the camera");
    check(sl2_make_measurements(eng_, image.data, (size_t)image.cols * image.rows, image.on_device ? 1 : 0), "sl2_make_measurements");
    refresh_public_members();
    return successful_measurement_vector_size_;
  }
  // normalise_state + delete_bad_features + the symmetrisation + the trajectory_store_ push (monoslam.cpp:137-177) in one
  // call: the engine does them in one kernel
  void finish_step(bool save_trajectory) {
    check(sl2_finish_step(eng_, save_trajectory ? 1 : 0), "sl2_finish_step");
    refresh_public_members();
  }
  void kalman_filter_predict() { check(sl2_kalman_filter_predict(eng_), "sl2_kalman_filter_predict"); refresh_public_members(); }
  void kalman_filter_update() { check(sl2_kalman_filter_update(eng_), "sl2_kalman_filter_update"); refresh_public_members(); }
  Feature* find_feature_lab(int lab) {                                  // monoslam.cpp:719-741
    for (const auto& f : feature_list_) if (f->label_ == lab) return f.get();
    return nullptr;
  }

  // construct_total_state / construct_total_covariance (monoslam.cpp:501-546)
  void construct_total_state(std::vector<double>& V) {
    V.assign(total_state_size_, 0.0);
    check(sl2_get_total_state(eng_, 0, V.data(), total_state_size_), "sl2_get_total_state");
  }
  void construct_total_covariance(std::vector<double>& M) {
    M.assign((size_t)total_state_size_ * total_state_size_, 0.0);
    check(sl2_get_total_covariance(eng_, 0, M.data(), total_state_size_), "sl2_get_total_covariance");
  }

  // MonoSLAM::mark_feature_by_lab + delete_feature (monoslam.cpp:743-812): removes the feature marked_feature_label_
  bool delete_feature() {
    if (marked_feature_label_ == -1) return false;
    const int32_t lab = marked_feature_label_;
    int32_t done = 0;
    check(sl2_delete_features(eng_, 0, 1, &lab, &done), "sl2_delete_features");
    if (done) { marked_feature_label_ = -1; refresh_public_members(); }
    return done != 0;
  }

  // MonoSLAM::InitialiseFeature(frame) (monoslam.cpp:1211-1235): a partially initialised feature at the selected pixel
  // (uu_, vv_).  False where the reference would have created one but this engine cannot (see sl2_initialise_feature).
  bool InitialiseFeature(const Frame& frame) {
    const int32_t uv[2] = {uu_, vv_};
    int32_t created = 0;
    check(sl2_initialise_feature(eng_, frame.data, (size_t)frame.cols * frame.rows, frame.on_device ? 1 : 0, uv, &created),
          "sl2_initialise_feature");
    refresh_public_members();
    return created != 0;
  }
  // MonoSLAM::InitialiseAutoFeature(frame) (monoslam.cpp:1535-1541)
  bool InitialiseAutoFeature(const Frame& frame) {
    int32_t created = 0;
    check(sl2_initialise_auto_feature(eng_, frame.data, (size_t)frame.cols * frame.rows, frame.on_device ? 1 : 0, &created),
          "sl2_initialise_auto_feature");
    refresh_public_members();
    return created != 0;
  }
  // MonoSLAM::mark_feature_by_lab (monoslam.cpp:743-768)
  void mark_feature_by_lab(int lab) {
    if (lab != -1) {
      bool found = false;
      for (const auto& f : feature_list_) found = found || f->label_ == lab;
      if (!found) return;
    }
    marked_feature_label_ = lab;
  }
  // MonoSLAM::SavePatch (monoslam.cpp:1551-1572): the marked feature's template to "patch.png" in the working directory
  bool SavePatch(const char* path = "patch.png") {
    if (marked_feature_label_ == -1) return false;
    bool found = false;
    for (const auto& f : feature_list_) found = found || f->label_ == marked_feature_label_;
    if (!found) return false;
    check(sl2_save_patch(eng_, 0, marked_feature_label_, path), "sl2_save_patch");
    return true;
  }

  // MonoSLAM::print_robot_state (monoslam.cpp: "[Robot state]" xv_, "[Robot covariance]" Pxx_)
  void print_robot_state(FILE* out = stdout) const {
    std::fprintf(out, "[Robot state]\n");
    for (double v : xv_) std::fprintf(out, "%g\n", v);
    std::fprintf(out, "[Robot covariance]\n");
    for (int r = 0; r < 13; ++r) {
      for (int c = 0; c < 13; ++c) std::fprintf(out, c ? " %g" : "%g", Pxx_[r * 13 + c]);
      std::fprintf(out, "\n");
    }
  }

  sl2_engine* engine() { return eng_; }
  // Wall time of the last GoOneStep: the sl2_go_one_step call and the read-back of the public members (one sl2_snapshot).
  // The step is asynchronous, so without measure_timing_ the first figure is its enqueue time and the second one holds the
  // wait for the GPU; with measure_timing_ = true a synchronisation is inserted between the two and they are the step and
  // the read-back proper (examples/monoslam_adapter --latency).
  bool measure_timing_ = false;
  double last_step_us_ = 0.0, last_refresh_us_ = 0.0, last_snapshot_us_ = 0.0;   // (the last: the sl2_snapshot call inside the read-back)

  // ---- public data members, names as in monoslam.h:158-218 ----
  std::unique_ptr<Camera> camera_;
  std::array<double, 13> xv_{};
  std::array<double, 169> Pxx_{};                             // 13x13 row-major
  std::vector<std::unique_ptr<Feature>> feature_list_;
  std::vector<Feature*> selected_feature_list_;
  std::vector<FeatureInitInfo> feature_init_info_vector_;
  std::vector<std::array<double, 3>> trajectory_store_;       // keeps the reference's stale-scratch entries
  int number_of_visible_features_ = 0;
  int next_free_label_ = 0;
  int marked_feature_label_ = -1;                             // GUI selection (graphictool.cpp); delete_feature() acts on it
  int total_state_size_ = 13;
  int successful_measurement_vector_size_ = 0;
  double kDeltaT_ = 0;
  int kNumberOfFeaturesToSelect_ = 0, kNumberOfFeaturesToKeepVisible_ = 0, kMaxFeaturesToInitAtOnce_ = 0;
  double kMinLambda_ = 0, kMaxLambda_ = 0;
  int kNumberOfParticles_ = 0;
  double kStandardDeviationDepthRatio_ = 0;
  int kMinNumberOfParticles_ = 0;
  double kPruneProbabilityThreshold_ = 0;
  int kErasePartiallyInitFeatureAfterThisManyAttempts_ = 0;
  int init_feature_search_ustart_ = 0, init_feature_search_vstart_ = 0, init_feature_search_ufinish_ = 0, init_feature_search_vfinish_ = 0;
  bool init_feature_search_region_defined_flag_ = false;
  int minimum_attempted_measurements_of_feature_ = 10;
  double successful_match_fraction_ = 0.5;
  int uu_ = 0, vv_ = 0;
  bool location_selected_flag_ = false;
  const int kBoxSize_ = SL2_PATCH_SIZE;                       // monoslam.cpp:48
  const double kNoSigma_ = 3.0, kCorrThresh2_ = 0.40, kCorrelationSigmaThreshold_ = 10.0;

 private:
  static void check(int rc, const char* what) {
    if (rc != SL2_OK) throw std::runtime_error(std::string(what) + ": " + sl2_last_error());
  }
  static std::map<std::string, std::string> parse_vars(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("MonoSLAM::Init: cannot read " + path);
    std::map<std::string, std::string> kv;
    std::string line;
    auto trim = [](std::string s) {
      const char* ws = " \t\r\n;";
      const size_t a = s.find_first_not_of(ws);
      if (a == std::string::npos) return std::string();
      return s.substr(a, s.find_last_not_of(ws) - a + 1);
    };
    while (std::getline(in, line)) {
      const size_t hash = line.find('#');
      if (hash != std::string::npos) line.erase(hash);
      const size_t eq = line.find('=');
      if (eq != std::string::npos) kv[trim(line.substr(0, eq))] = trim(line.substr(eq + 1));
    }
    return kv;
  }
  static double num(const std::map<std::string, std::string>& kv, const std::string& k) {
    auto it = kv.find(k);
    return it == kv.end() ? 0.0 : std::atof(it->second.c_str());
  }

  // What the reference's members hold after a step: ONE sl2_snapshot call (one kernel, one stream synchronisation, no
  // allocation inside the library) instead of a dozen blocking accessors.  Feature objects are kept across frames and
  // matched by label_ (a label is handed out once), templates are requested only for labels not seen before, and only the
  // trajectory_store_ entries pushed since the last call cross the bus.
  void refresh_public_members() {
    const void* blob = nullptr;
    size_t nbytes = 0;
    const auto ts0 = std::chrono::steady_clock::now();
    check(sl2_snapshot(eng_, 0, traj_seen_, patch_seen_, &blob, &nbytes), "sl2_snapshot");
    last_snapshot_us_ = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ts0).count();
    const unsigned char* base = static_cast<const unsigned char*>(blob);
    const sl2_snapshot_header& hd = *reinterpret_cast<const sl2_snapshot_header*>(base);
    if (hd.api_version != SL2_API_VERSION) throw std::runtime_error("scenelib2_amd: library / header version mismatch");
    std::memcpy(xv_.data(), base + hd.off_xv, sizeof(double) * 13);
    std::memcpy(Pxx_.data(), base + hd.off_Pxx, sizeof(double) * 169);
    total_state_size_ = hd.total_state_size;
    number_of_visible_features_ = hd.number_of_visible_features;
    successful_measurement_vector_size_ = hd.successful_measurement_vector_size;
    next_free_label_ = hd.next_free_label;
    status_flags_ = hd.status_flags;
    // templates of the labels this object has not seen yet
    for (int k = 0; k < hd.n_patches; ++k) {
      const unsigned char* rec = base + hd.off_patches + (size_t)k * 128;
      int32_t lab;
      std::memcpy(&lab, rec, 4);
      std::memcpy(patch_cache_[lab].data(), rec + 4, SL2_PATCH_BYTES);
    }
    patch_seen_ = hd.next_free_label;
    // feature_list_: reuse the objects of the features that are still there (list order never changes: features are only
    // appended and erased)
    std::vector<std::unique_ptr<Feature>> old;
    old.swap(feature_list_);
    size_t oi = 0;
    const sl2_feature_info* fi = reinterpret_cast<const sl2_feature_info*>(base + hd.off_features);
    const double* cov = reinterpret_cast<const double*>(base + hd.off_cov);
    feature_list_.reserve(hd.n_features);
    for (int i = 0; i < hd.n_features; ++i) {
      const sl2_feature_info& s = fi[i];
      std::unique_ptr<Feature> f;
      while (oi < old.size() && old[oi]->label_ != s.label) ++oi;      // erased features drop out here
      if (oi < old.size()) f = std::move(old[oi++]);
      else {
        f.reset(new Feature());
        f->label_ = s.label;
        auto it = patch_cache_.find(s.label);
        if (it != patch_cache_.end()) f->patch_ = it->second;
        else check(sl2_get_feature_patch(eng_, 0, s.label, f->patch_.data()), "sl2_get_feature_patch");   // (not reached)
      }
      f->fully_initialised_flag_ = s.fully_initialised_flag != 0;
      f->selected_flag_ = s.selected_flag != 0;
      f->successful_measurement_flag_ = s.successful_measurement_flag != 0;
      f->attempted_measurements_of_feature_ = s.attempted_measurements_of_feature;
      f->successful_measurements_of_feature_ = s.successful_measurements_of_feature;
      f->position_in_total_state_vector_ = s.position_in_total_state_vector;
      f->state_size_ = s.state_size;
      f->y_.assign(s.y, s.y + 3);
      if (s.state_size == 6) f->y_.insert(f->y_.end(), s.y_direction, s.y_direction + 3);
      std::memcpy(f->xp_org_.data(), s.xp_org, sizeof(s.xp_org));
      for (int k = 0; k < 2; ++k) { f->h_[k] = s.h[k]; f->z_[k] = s.z[k]; f->nu_[k] = s.nu[k]; }
      for (int k = 0; k < 4; ++k) f->S_[k] = s.S[k];
      f->R_ = {{s.R, 0.0, 0.0, s.R}};
      for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 7; ++c) f->dh_by_dxv_[r * 13 + c] = s.dh_by_dxp[r * 7 + c];
      for (int k = 0; k < 6; ++k) f->dh_by_dy_[k] = s.dh_by_dy[k];
      const int d = s.state_size;
      f->Pxy_.assign(cov, cov + 13 * d);
      f->Pyy_.assign(cov + 13 * d, cov + 13 * d + d * d);
      cov += 13 * d + d * d;
      feature_list_.push_back(std::move(f));
    }
    patch_cache_.clear();      // only needed while the Feature objects of new labels are created above (labels churn with mapping on)
    const int32_t* sel = reinterpret_cast<const int32_t*>(base + hd.off_selection);
    selected_feature_list_.clear();
    for (int k = 0; k < hd.n_selected; ++k) {          // feature_list_ is in label order (labels are handed out in creation order)
      size_t lo = 0, hi = feature_list_.size();
      while (lo < hi) { const size_t mid = (lo + hi) / 2; if (feature_list_[mid]->label_ < sel[k]) lo = mid + 1; else hi = mid; }
      if (lo < feature_list_.size() && feature_list_[lo]->label_ == sel[k]) selected_feature_list_.push_back(feature_list_[lo].get());
    }
    // trajectory_store_: append what is new, keep at most 1000 entries (monoslam.cpp:172-177)
    const double* tr = reinterpret_cast<const double*>(base + hd.off_traj);
    if (hd.traj_first > traj_seen_) trajectory_store_.clear();     // (more than 1000 pushes between two calls)
    for (int k = 0; k < hd.traj_count; ++k) trajectory_store_.push_back({{tr[3 * k], tr[3 * k + 1], tr[3 * k + 2]}});
    if (trajectory_store_.size() > 1000) trajectory_store_.erase(trajectory_store_.begin(), trajectory_store_.end() - 1000);
    traj_seen_ = hd.traj_total;
    // feature_init_info_vector_
    uu_ = hd.uu; vv_ = hd.vv;
    init_feature_search_region_defined_flag_ = hd.init_feature_search_region_defined_flag != 0;
    init_feature_search_ustart_ = hd.init_feature_search_region[0]; init_feature_search_vstart_ = hd.init_feature_search_region[1];
    init_feature_search_ufinish_ = hd.init_feature_search_region[2]; init_feature_search_vfinish_ = hd.init_feature_search_region[3];
    location_selected_flag_ = hd.location_selected_flag != 0;
    feature_init_info_vector_.clear();
    const unsigned char* pp = base + hd.off_partial;
    for (int j = 0; j < hd.n_partial; ++j) {
      const sl2_partial_info& pinfo = *reinterpret_cast<const sl2_partial_info*>(pp);
      const double* parts = reinterpret_cast<const double*>(pp + sizeof(sl2_partial_info));
      FeatureInitInfo info;
      for (auto& f : feature_list_)
        if (f->label_ == pinfo.label) info.fp_ = f.get();
      info.number_of_match_attempts_ = pinfo.number_of_match_attempts;
      info.making_measurement_on_this_step_flag_ = pinfo.making_measurement_on_this_step_flag != 0;
      info.mean_ = pinfo.mean; info.covariance_ = pinfo.covariance;
      info.particle_vector_.resize(pinfo.n_particles);
      for (int k = 0; k < pinfo.n_particles; ++k) {
        const double* o = parts + (size_t)k * 12;
        Particle& p = info.particle_vector_[k];
        p.lambda_ = o[0]; p.probability_ = o[1]; p.cumulative_probability_ = o[2];
        p.m_h_ = {{o[3], o[4]}}; p.m_z_ = {{o[5], o[6]}};
        p.m_SInv_ = {{o[7], o[8], o[8], o[9]}};
        p.m_detS_ = o[10];
        p.m_successful_measurement_flag_ = o[11] != 0.0;
      }
      feature_init_info_vector_.push_back(std::move(info));
      pp += sizeof(sl2_partial_info) + (size_t)pinfo.n_particles * 12 * sizeof(double);
    }
  }

  int traj_seen_ = 0, patch_seen_ = 0, status_flags_ = 0;
  std::map<int, std::array<uint8_t, SL2_PATCH_BYTES>> patch_cache_;
  int max_features_, device_;
  bool map_full_ = false;
  sl2_engine* eng_ = nullptr;
};

// Kalman (kalman.h:51-52): the reference's filter object works on the MonoSLAM it is handed; so does this one.
struct Kalman {
  void KalmanFilterPredict(MonoSLAM* monoslam, const std::array<double, 3>& /*u: the constant-velocity model takes no control*/ = {}) {
    monoslam->kalman_filter_predict();                                  // kalman.cpp:50-69
  }
  void KalmanFilterUpdate(MonoSLAM* monoslam) { monoslam->kalman_filter_update(); }   // kalman.cpp:72-119
};

}  // namespace SceneLib2Amd

#endif  // SCENELIB2_AMD_MONOSLAM_HPP
