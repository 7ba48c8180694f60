// The loop of the reference's examples/MonoSlamSceneLib1.cpp (:55, :132-142) written against the MonoSLAM-shaped
// adapter (include/scenelib2_amd_monoslam.hpp) instead of SceneLib2::MonoSLAM: Init(cfg), GetFrame, GoOneStep, then the
// members GraphicTool would draw.  No Pangolin window: the per-frame read-out is printed / dumped instead.
//
//   monoslam_adapter --cfg scene.cfg --frames frame_dir [--mapping | --seams] [--dump out.txt] [--latency out.json]
//
// --latency: end-to-end wall time per frame of the drop-in loop (GetFrame + GoOneStep + every public member refreshed), its
// split into the step and the read-back (a second pass with a synchronisation between the two), as one JSON object.
//
// The dump lists, one value per line: total_state_size_, the total state (construct_total_state), then per feature in
// feature_list_ order: label_, fully_initialised_flag_, attempted_, successful_, position_in_total_state_vector_, Pyy_
// (row-major), patch_ (121 bytes); then trajectory_store_.  tests/test_gpu_headless_example.py compares it with the oracle.
#include <scenelib2_amd_monoslam.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>

int main(int argc, char** argv) {
  std::string cfg, frames_dir, dump, latency;
  bool enable_mapping = false, seams = false;    // --seams: the step through the reference's individual members instead of GoOneStep
  bool zero_copy = true;                         // --copy-frames: upload every frame instead of letting the device read the pinned batch (A/B)
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--cfg" && i + 1 < argc) cfg = argv[++i];
    else if (a == "--frames" && i + 1 < argc) frames_dir = argv[++i];
    else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
    else if (a == "--latency" && i + 1 < argc) latency = argv[++i];
    else if (a == "--mapping") enable_mapping = true;
    else if (a == "--seams") seams = true;
    else if (a == "--copy-frames") zero_copy = false;
    else { fprintf(stderr, "usage: %s --cfg scene.cfg --frames dir [--mapping | --seams] [--dump file]\n", argv[0]); return 2; }
  }
  if (cfg.empty() || frames_dir.empty()) { fprintf(stderr, "need --cfg and --frames\n"); return 2; }
  if (!latency.empty()) {
    // two passes over the same frames with fresh objects: (A) the loop as a user runs it, timed end to end per frame;
    // (B) the same with a synchronisation between the step and the read-back, for the split
    try {
      auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
      auto mean = [](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return v.empty() ? 0.0 : s / v.size(); };
      std::vector<double> loop_us, call_us, step_us, refresh_us, snap_us;
      std::vector<double> loop_free_us, loop_partial_us;      // frames that start without / with a partially initialised feature
      std::string partial_seq;                                // their number at the start of every frame of pass 0, one digit each
      size_t n_features = 0, n_frames = 0, snap_cap = 0;
      for (int pass = 0; pass < 2; ++pass) {
        SceneLib2Amd::MonoSLAM slam;
        slam.Init(cfg);
        slam.measure_timing_ = pass == 1;
        const char* dirs[1] = {frames_dir.c_str()};
        sl2_ingest* grab = nullptr;
        if (sl2_ingest_open(dirs, 1, slam.camera_->width_, slam.camera_->height_, 0, 8, &grab) != SL2_OK) { fprintf(stderr, "%s\n", sl2_last_error()); return 1; }
        if (!zero_copy) sl2_ingest_set_zero_copy(grab, 0);
        const int n = sl2_ingest_frame_count(grab);
        for (int frame_id = 0; frame_id < n; ++frame_id) {
          const bool partial_before = !slam.feature_init_info_vector_.empty();
          if (pass == 0) partial_seq.push_back((char)('0' + std::min<size_t>(slam.feature_init_info_vector_.size(), 9)));
          const auto t0 = std::chrono::steady_clock::now();
          SceneLib2Amd::Frame frame;
          size_t stride = 0;
          if (sl2_ingest_next(grab, slam.stream(), &frame.data, &stride) != SL2_OK) { fprintf(stderr, "%s\n", sl2_last_error()); return 1; }
          frame.cols = slam.camera_->width_; frame.rows = slam.camera_->height_; frame.on_device = true;
          const auto t1 = std::chrono::steady_clock::now();
          slam.GoOneStep(frame, true, enable_mapping);
          const auto t2 = std::chrono::steady_clock::now();
          if (frame_id < 5) continue;                                   // warm-up: first launches, lazy allocations
          if (pass == 0) {
            loop_us.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
            (partial_before ? loop_partial_us : loop_free_us).push_back(loop_us.back());
            call_us.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
          } else {
            step_us.push_back(slam.last_step_us_);
            refresh_us.push_back(slam.last_refresh_us_);
            snap_us.push_back(slam.last_snapshot_us_);
          }
        }
        n_features = slam.feature_list_.size();
        snap_cap = sl2_snapshot_capacity(slam.engine());
        n_frames = (size_t)n;
        sl2_ingest_close(grab);
      }
      FILE* f = fopen(latency.c_str(), "w");
      if (!f) { fprintf(stderr, "cannot write %s\n", latency.c_str()); return 5; }
      fprintf(f, "{\"frames\": %zu, \"timed_frames\": %zu, \"features_at_end\": %zu, \"mapping\": %s, "
                 "\"frame_us_median\": %.2f, \"frame_us_mean\": %.2f, \"go_one_step_us_median\": %.2f, \"go_one_step_us_mean\": %.2f, "
                 "\"step_us_median\": %.2f, \"step_us_mean\": %.2f, \"readback_us_median\": %.2f, \"readback_us_mean\": %.2f, \"snapshot_call_us_median\": %.2f, "
                 "\"frames_starting_without_partial_feature\": %zu, \"frame_us_median_without_partial_feature\": %.2f, "
                 "\"frames_starting_with_partial_feature\": %zu, \"frame_us_median_with_partial_feature\": %.2f, "
                 "\"partial_features_at_frame_start\": \"%s\", \"blocking_copies_per_frame\": 0, \"synchronisations_per_frame\": 1, \"snapshot_capacity_bytes\": %zu}\n",
              n_frames, loop_us.size(), n_features, enable_mapping ? "true" : "false", median(loop_us), mean(loop_us), median(call_us),
              mean(call_us), median(step_us), mean(step_us), median(refresh_us), mean(refresh_us), median(snap_us), loop_free_us.size(),
              median(loop_free_us), loop_partial_us.size(), median(loop_partial_us), partial_seq.c_str(), snap_cap);
      fclose(f);
    } catch (const std::exception& e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
    }
    return 0;
  }
  try {
    SceneLib2Amd::MonoSLAM slam;
    slam.Init(cfg);                                                     // MonoSlamSceneLib1.cpp:55
    printf("camera %dx%d, %zu known features\n", slam.camera_->width_, slam.camera_->height_, slam.feature_list_.size());

    // frame_grabber_->GetFrame(frame_id, &frame): the batched grabber of the library, with one sequence
    const char* dirs[1] = {frames_dir.c_str()};
    sl2_ingest* grab = nullptr;
    if (sl2_ingest_open(dirs, 1, slam.camera_->width_, slam.camera_->height_, 0, 8, &grab) != SL2_OK) {
      fprintf(stderr, "%s\n", sl2_last_error());
      return 1;
    }
    if (!zero_copy) sl2_ingest_set_zero_copy(grab, 0);
    const int n = sl2_ingest_frame_count(grab);
    const bool save_trajectory = true;
    for (int frame_id = 0; frame_id < n; ++frame_id) {                  // MonoSlamSceneLib1.cpp:132-142
      SceneLib2Amd::Frame frame;
      size_t stride = 0;
      if (sl2_ingest_next(grab, slam.stream(), &frame.data, &stride) != SL2_OK) { fprintf(stderr, "%s\n", sl2_last_error()); return 1; }
      frame.cols = slam.camera_->width_; frame.rows = slam.camera_->height_; frame.on_device = true;
      if (!seams) {
        slam.GoOneStep(frame, save_trajectory, enable_mapping);
      } else {                                                          // monoslam.cpp:118-177 call by call (no mapping tail)
        SceneLib2Amd::Kalman kalman;
        kalman.KalmanFilterPredict(&slam);
        slam.auto_select_n_features(slam.kNumberOfFeaturesToSelect_);
        if (slam.make_measurements(frame) > 0) kalman.KalmanFilterUpdate(&slam);
        slam.finish_step(save_trajectory);
      }
      if (frame_id % 10 == 9 || frame_id + 1 == n) {
        int measured = 0;
        for (const SceneLib2Amd::Feature* f : slam.selected_feature_list_) measured += f->successful_measurement_flag_ ? 1 : 0;
        printf("frame %4d  r = (% .4f % .4f % .4f)  features %zu  visible %d  selected %zu  measured %d  partial %zu\n", frame_id,
               slam.xv_[0], slam.xv_[1], slam.xv_[2], slam.feature_list_.size(), slam.number_of_visible_features_,
               slam.selected_feature_list_.size(), measured, slam.feature_init_info_vector_.size());
      }
    }
    sl2_ingest_close(grab);
    slam.print_robot_state();                                          // the example's "Print Robot State" button (:196-197)
    if (!dump.empty()) {
      FILE* f = fopen(dump.c_str(), "w");
      if (!f) { fprintf(stderr, "cannot write %s\n", dump.c_str()); return 5; }
      std::vector<double> V;
      slam.construct_total_state(V);
      fprintf(f, "%d\n", slam.total_state_size_);
      for (double v : V) fprintf(f, "%.17g\n", v);
      for (const auto& ft : slam.feature_list_) {
        fprintf(f, "%d\n%d\n%d\n%d\n%d\n", ft->label_, ft->fully_initialised_flag_ ? 1 : 0, ft->attempted_measurements_of_feature_,
                ft->successful_measurements_of_feature_, ft->position_in_total_state_vector_);
        for (double v : ft->Pyy_) fprintf(f, "%.17g\n", v);
        for (uint8_t v : ft->patch_) fprintf(f, "%d\n", (int)v);
      }
      for (const auto& r : slam.trajectory_store_) fprintf(f, "%.17g\n%.17g\n%.17g\n", r[0], r[1], r[2]);
      fclose(f);
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
