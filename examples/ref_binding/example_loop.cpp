// The loop and the button handlers of the reference's only caller, examples/MonoSlamSceneLib1.cpp:132-142 and 190-204,
// headless, written against SceneLib2::MonoSLAM with the reference's own types (examples/ref_binding/monoslam_amd.h; needs
// the real Eigen / OpenCV headers - not built in this repository's image).  The GUI's inputs are scripted:
//   example_loop <cfg> <frame.pgm> <frames> <dump> [--seams]
//     frame 2: click (uu_, vv_) + "Initialise Manual Feature"   frame 14: "Initialise Auto Feature"
//     frame 6: "Print Robot State"   frame 7: mark label 2 + "Delete Feature"   frame 8: mark label 1 + "Save Patch"
// After every frame the members GraphicTool reads are appended to <dump>.
#include "monoslam_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace SceneLib2;

static MonoSLAM* g_monoslam = nullptr;

static void dump_members(FILE* out) {
  MonoSLAM* m = g_monoslam;
  Eigen::VectorXd x;
  Eigen::MatrixXd P;
  m->construct_total_state(x);
  m->construct_total_covariance(P);
  fprintf(out, "%d %d %d %d %d\n", m->total_state_size_, (int)m->feature_list_.size(), (int)m->selected_feature_list_.size(),
          m->successful_measurement_vector_size_, (int)m->feature_init_info_vector_.size());
  for (int i = 0; i < m->total_state_size_; ++i) fprintf(out, "%.17g\n", x(i));
  for (int r = 0; r < m->total_state_size_; ++r)
    for (int c = 0; c < m->total_state_size_; ++c) fprintf(out, "%.17g\n", P(r, c));
  for (Feature* f : m->feature_list_)
    fprintf(out, "%d %d %d %d %d %d %.17g %.17g\n", f->label_, f->fully_initialised_flag_ ? 1 : 0, f->selected_flag_ ? 1 : 0,
            f->successful_measurement_flag_ ? 1 : 0, f->attempted_measurements_of_feature_, f->successful_measurements_of_feature_,
            f->z_(0), f->z_(1));
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s cfg frame.pgm nframes dump [--seams]\n", argv[0]); return 2; }
  const bool seams = argc > 5 && !strcmp(argv[5], "--seams");
  g_monoslam = new MonoSLAM();
  g_monoslam->Init(argv[1]);                                               // MonoSlamSceneLib1.cpp:55
  const int nframes = atoi(argv[3]);
  FILE* out = fopen(argv[4], "w");
  if (!out) return 3;
  const bool chk_display_trajectory = true, chk_enable_mapping = false, chk_toggle_tracking = true;
  for (int g_frame_id = 0; g_frame_id < nframes; ++g_frame_id) {
    cv::Mat frame = cv::imread(argv[2], 0);                                // frame_grabber_->GetFrame(g_frame_id, &frame)
    if (frame.empty()) { fprintf(stderr, "cannot read %s\n", argv[2]); return 4; }
    if (chk_toggle_tracking) {
      if (!seams) {
        g_monoslam->GoOneStep(frame, chk_display_trajectory, chk_enable_mapping);        // :135-139
      } else {                                                             // GoOneStep's body, seam by seam (monoslam.cpp:118-150)
        Eigen::Vector3d u(0.0, 0.0, 0.0);
        g_monoslam->kalman_->KalmanFilterPredict(g_monoslam, u);
        g_monoslam->number_of_visible_features_ = g_monoslam->auto_select_n_features(g_monoslam->kNumberOfFeaturesToSelect_);
        if (g_monoslam->selected_feature_list_.size() != 0) {
          g_monoslam->make_measurements(frame);
          if (g_monoslam->successful_measurement_vector_size_ != 0) g_monoslam->kalman_->KalmanFilterUpdate(g_monoslam);
        }
        g_monoslam->pending_save_trajectory_ = chk_display_trajectory;
        g_monoslam->normalise_state();
        g_monoslam->delete_bad_features();
      }
    }
    // Buttons handling (:190-204), scripted
    if (g_frame_id == 2 && !seams) { g_monoslam->uu_ = 60; g_monoslam->vv_ = 200; g_monoslam->location_selected_flag_ = true; g_monoslam->InitialiseFeature(frame); }
    if (g_frame_id == 14 && !seams) g_monoslam->InitialiseAutoFeature(frame);   // (once the manual feature is gone: one partial feature at a time)
    if (g_frame_id == 6) g_monoslam->print_robot_state();
    if (g_frame_id == 7) { g_monoslam->mark_feature_by_lab(2); g_monoslam->delete_feature(); }
    if (g_frame_id == 8) { g_monoslam->mark_feature_by_lab(1); if (!g_monoslam->SavePatch()) return 5; }
    dump_members(out);
  }
  fclose(out);
  printf("%d frames, %d features\n", nframes, (int)g_monoslam->feature_list_.size());
  delete g_monoslam;
  return 0;
}
