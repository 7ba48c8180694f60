// The loop of the reference's examples/MonoSlamSceneLib1.cpp (:55, :132-142) written against the MonoSLAM-shaped
// adapter (include/scenelib2_amd_monoslam.hpp) instead of SceneLib2::MonoSLAM: Init(cfg), GetFrame, GoOneStep, then the
// members GraphicTool would draw.  No Pangolin window: the per-frame read-out is printed / dumped instead.
//
//   monoslam_adapter --cfg scene.cfg --frames frame_dir [--mapping | --seams] [--dump out.txt]
//
// The dump lists, one value per line: total_state_size_, the total state (construct_total_state), then per feature in
// feature_list_ order: label_, fully_initialised_flag_, attempted_, successful_, position_in_total_state_vector_, Pyy_
// (row-major), patch_ (121 bytes); then trajectory_store_.  tests/test_gpu_headless_example.py compares it with the oracle.
#include <scenelib2_amd_monoslam.hpp>

#include <cstdio>

int main(int argc, char** argv) {
  std::string cfg, frames_dir, dump;
  bool enable_mapping = false, seams = false;    // --seams: the step through the reference's individual members instead of GoOneStep
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--cfg" && i + 1 < argc) cfg = argv[++i];
    else if (a == "--frames" && i + 1 < argc) frames_dir = argv[++i];
    else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
    else if (a == "--mapping") enable_mapping = true;
    else if (a == "--seams") seams = true;
    else { fprintf(stderr, "usage: %s --cfg scene.cfg --frames dir [--mapping | --seams] [--dump file]\n", argv[0]); return 2; }
  }
  if (cfg.empty() || frames_dir.empty()) { fprintf(stderr, "need --cfg and --frames\n"); return 2; }
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
    const int n = sl2_ingest_frame_count(grab);
    const bool save_trajectory = true;
    for (int frame_id = 0; frame_id < n; ++frame_id) {                  // MonoSlamSceneLib1.cpp:132-142
      SceneLib2Amd::Frame frame;
      size_t stride = 0;
      if (sl2_ingest_next(grab, nullptr, &frame.data, &stride) != SL2_OK) { fprintf(stderr, "%s\n", sl2_last_error()); return 1; }
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
