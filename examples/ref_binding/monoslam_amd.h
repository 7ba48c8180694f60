// The binding of INTEGRATION.md section 2(b): SceneLib2::MonoSLAM / Kalman / Feature with the REFERENCE'S OWN
// signatures and member types - cv::Mat frames, Eigen::VectorXd / MatrixXd members,
// Kalman::KalmanFilterPredict(MonoSLAM*, Eigen::Vector3d&) - over the C ABI of include/scenelib2_amd.h
// (monoslam.h:69-219, kalman.h:44-53, feature.h:56-143, feature_init_info.h:46-118).
//
// Eigen and OpenCV are not in this build environment, so this header is NOT compiled here (rounds 3-5 compiled and ran it
// against stand-in headers, retired in round 6 with the stand-in reference build - DESIGN.md section 2); it is the text a
// maintainer adds at a site that has the real libraries, where it compiles as it stands: it uses only
// VectorXd / MatrixXd / Vector2d / Vector3d element access, resize, and cv::Mat::{data, rows, cols} / cv::imread.
//
// It is a thin layer: the C-ABI calls and the read-back live in include/scenelib2_amd_monoslam.hpp (SceneLib2Amd::MonoSLAM,
// plain arrays); here its members are re-typed after every call, so that code written against the reference's headers -
// the example's loop and buttons (examples/MonoSlamSceneLib1.cpp:132-142, 190-204), GraphicTool's reads - compiles and
// reads the same values.
#ifndef TESTS_REF_BINDING_MONOSLAM_AMD_H
#define TESTS_REF_BINDING_MONOSLAM_AMD_H

#include <Eigen/Eigen>
#include <opencv2/opencv.hpp>

#include <scenelib2_amd_monoslam.hpp>

#include <iostream>
#include <string>
#include <vector>

namespace SceneLib2 {

using std::string;
using std::vector;

class Camera {               // camera.h:61-77 (the members the example reads: MonoSlamSceneLib1.cpp:65-69)
 public:
  int width_ = 0, height_ = 0;
  double fku_ = 0, fkv_ = 0, kd1_ = 0;
  Eigen::Vector2d centre_;
  int measurement_sd_ = 0;
};

class Feature {              // feature.h:56-143
 public:
  Eigen::VectorXd y_, xp_org_, h_, z_, nu_;
  Eigen::MatrixXd Pxy_, Pyy_, dh_by_dxv_, dh_by_dy_, R_, S_;
  cv::Mat patch_;
  int label_ = 0, position_in_total_state_vector_ = 0;
  int attempted_measurements_of_feature_ = 0, successful_measurements_of_feature_ = 0;
  bool selected_flag_ = false, successful_measurement_flag_ = false, fully_initialised_flag_ = true;
};

class Particle {             // feature_init_info.h:46-75
 public:
  double lambda_ = 0, probability_ = 0, cumulative_probability_ = 0;
  Eigen::VectorXd m_h_, m_z_;
  Eigen::MatrixXd m_SInv_;
  double m_detS_ = 0;
  bool m_successful_measurement_flag_ = false;
};

class FeatureInitInfo {      // feature_init_info.h:77-118
 public:
  Feature* fp_ = nullptr;
  vector<Particle> particle_vector_;
  double mean_ = 0, covariance_ = 0;
  int number_of_match_attempts_ = 0;
  bool making_measurement_on_this_step_flag_ = false;
};

class MonoSLAM;

class Kalman {               // kalman.h:44-53
 public:
  void KalmanFilterPredict(MonoSLAM* monoslam, Eigen::Vector3d& u);
  void KalmanFilterUpdate(MonoSLAM* monoslam);
};

class MonoSLAM {             // monoslam.h:69-219
 public:
  MonoSLAM() : kBoxSize_(11), kNoSigma_(3.0), kCorrThresh2_(0.40), kCorrelationSigmaThreshold_(10.0) {
    camera_ = new Camera();
    kalman_ = new Kalman();
  }
  ~MonoSLAM() {
    clear_features();
    delete camera_;
    delete kalman_;
  }
  MonoSLAM(const MonoSLAM&) = delete;
  MonoSLAM& operator=(const MonoSLAM&) = delete;

  void Init(const string& config_path) {                                   // monoslam.cpp:1574-1969
    impl_.Init(config_path);
    refresh();
  }
  bool GoOneStep(cv::Mat frame, bool save_trajectory, bool enable_mapping) {   // monoslam.cpp:108-180
    const bool r = impl_.GoOneStep(view(frame), save_trajectory, enable_mapping);
    refresh();
    return r;
  }
  void InitialiseFeature(cv::Mat frame) {                                  // monoslam.cpp:1211-1235: at the clicked (uu_, vv_)
    impl_.uu_ = uu_; impl_.vv_ = vv_;
    impl_.InitialiseFeature(view(frame));
    refresh();
  }
  void InitialiseAutoFeature(cv::Mat frame) {                              // monoslam.cpp:1535-1541
    impl_.InitialiseAutoFeature(view(frame));
    refresh();
  }
  void print_robot_state() {                                               // monoslam.cpp:1543-1549
    std::cout << "[Robot state]" << std::endl;
    for (int i = 0; i < 13; ++i) std::cout << xv_(i) << std::endl;
    std::cout << "[Robot covariance]" << std::endl;
    for (int r = 0; r < 13; ++r) {
      for (int c = 0; c < 13; ++c) std::cout << (c ? " " : "") << Pxx_(r, c);
      std::cout << std::endl;
    }
  }
  bool SavePatch() {                                                       // monoslam.cpp:1551-1572 ("patch.png")
    impl_.marked_feature_label_ = marked_feature_label_;
    return impl_.SavePatch();
  }
  void AddNewKnownFeature(const Eigen::VectorXd& y, const Eigen::VectorXd& xp, const string& identifier) {   // monoslam.cpp:1278-1291
    std::array<double, 3> yy{{y(0), y(1), y(2)}};
    std::array<double, 7> xo;
    for (int i = 0; i < 7; ++i) xo[i] = xp(i);
    impl_.AddNewKnownFeature(yy, xo, identifier);
    refresh();
  }
  int auto_select_n_features(int n) { const int r = impl_.auto_select_n_features(n); refresh(); return r; }        // :187-254
  int make_measurements(cv::Mat image) { const int r = impl_.make_measurements(view(image)); refresh(); return r; }   // :336-359
  // normalise_state (:616-637) and delete_bad_features (:644-660) are one device launch together with the symmetrisation
  // and the trajectory push (sl2_finish_step): the first of the two calls the reference makes does the work
  void normalise_state() { impl_.finish_step(pending_save_trajectory_); finished_ = true; refresh(); }
  void delete_bad_features() { if (!finished_) normalise_state(); finished_ = false; }
  void construct_total_state(Eigen::VectorXd& V) {                         // :501-512
    std::vector<double> v;
    impl_.construct_total_state(v);
    V.resize((int)v.size());
    for (size_t i = 0; i < v.size(); ++i) V((int)i) = v[i];
  }
  void construct_total_covariance(Eigen::MatrixXd& M) {                    // :518-546
    std::vector<double> m;
    impl_.construct_total_covariance(m);
    const int n = impl_.total_state_size_;
    M.resize(n, n);
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) M(r, c) = m[(size_t)r * n + c];
  }
  Feature* find_feature_lab(int lab) {                                     // :719-741
    for (Feature* f : feature_list_) if (f->label_ == lab) return f;
    return nullptr;
  }
  void mark_feature_by_lab(int lab) {                                      // :743-768
    if (lab > 0 && !find_feature_lab(lab)) return;                         // (label 0 is always accepted: SURVEY Q21)
    marked_feature_label_ = lab;
  }
  bool delete_feature() {                                                  // :770-812
    impl_.marked_feature_label_ = marked_feature_label_;
    const bool r = impl_.delete_feature();
    refresh();
    marked_feature_label_ = impl_.marked_feature_label_;
    return r;
  }

  // used by Kalman (the reference's Kalman is a friend-like function bag working on the MonoSLAM it is handed)
  void kalman_predict_() { impl_.kalman_filter_predict(); refresh(); }
  void kalman_update_() { impl_.kalman_filter_update(); refresh(); }

  Camera* camera_;
  Kalman* kalman_;

  Eigen::VectorXd xv_;
  Eigen::MatrixXd Pxx_;
  vector<Feature*> feature_list_;
  vector<Feature*> selected_feature_list_;
  vector<FeatureInitInfo> feature_init_info_vector_;
  vector<Eigen::Vector3d> trajectory_store_;

  int number_of_visible_features_ = 0;
  int next_free_label_ = 0;
  int marked_feature_label_ = -1;
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
  bool pending_save_trajectory_ = false;      // (what the seam-wise caller wants pushed by the step it is assembling)

  const int kBoxSize_;
  const double kNoSigma_, kCorrThresh2_, kCorrelationSigmaThreshold_;

 private:
  static SceneLib2Amd::Frame view(const cv::Mat& m) {
    SceneLib2Amd::Frame f;
    f.data = m.data; f.cols = m.cols; f.rows = m.rows;
    return f;
  }
  void clear_features() {
    for (Feature* f : feature_list_) delete f;
    feature_list_.clear();
    selected_feature_list_.clear();
    feature_init_info_vector_.clear();
  }
  static Eigen::VectorXd vec(const double* p, int n) {
    Eigen::VectorXd v(n);
    for (int i = 0; i < n; ++i) v(i) = p[i];
    return v;
  }
  static Eigen::MatrixXd mat(const double* p, int r, int c) {          // row-major source (the ABI) -> Eigen
    Eigen::MatrixXd m(r, c);
    for (int i = 0; i < r; ++i)
      for (int j = 0; j < c; ++j) m(i, j) = p[(size_t)i * c + j];
    return m;
  }
  // the adapter's plain members, re-typed
  void refresh() {
    const SceneLib2Amd::MonoSLAM& a = impl_;
    camera_->width_ = a.camera_->width_; camera_->height_ = a.camera_->height_;
    camera_->fku_ = a.camera_->fku_; camera_->fkv_ = a.camera_->fkv_; camera_->kd1_ = a.camera_->kd1_;
    camera_->centre_ = Eigen::Vector2d(a.camera_->centre_[0], a.camera_->centre_[1]);
    camera_->measurement_sd_ = a.camera_->measurement_sd_;
    xv_ = vec(a.xv_.data(), 13);
    Pxx_ = mat(a.Pxx_.data(), 13, 13);
    clear_features();
    for (const auto& s : a.feature_list_) {
      Feature* f = new Feature();
      const int d = s->state_size_;
      f->y_ = vec(s->y_.data(), (int)s->y_.size());
      f->xp_org_ = vec(s->xp_org_.data(), 7);
      f->h_ = vec(s->h_.data(), 2); f->z_ = vec(s->z_.data(), 2); f->nu_ = vec(s->nu_.data(), 2);
      f->Pxy_ = mat(s->Pxy_.data(), 13, d);
      f->Pyy_ = mat(s->Pyy_.data(), d, d);
      f->dh_by_dxv_ = mat(s->dh_by_dxv_.data(), 2, 13);
      f->dh_by_dy_ = mat(s->dh_by_dy_.data(), 2, 3);
      f->R_ = mat(s->R_.data(), 2, 2);
      f->S_ = mat(s->S_.data(), 2, 2);
      f->patch_ = cv::Mat(11, 11, CV_8UC1);
      std::memcpy(f->patch_.data, s->patch_.data(), 121);
      f->label_ = s->label_;
      f->position_in_total_state_vector_ = s->position_in_total_state_vector_;
      f->attempted_measurements_of_feature_ = s->attempted_measurements_of_feature_;
      f->successful_measurements_of_feature_ = s->successful_measurements_of_feature_;
      f->selected_flag_ = s->selected_flag_;
      f->successful_measurement_flag_ = s->successful_measurement_flag_;
      f->fully_initialised_flag_ = s->fully_initialised_flag_;
      feature_list_.push_back(f);
    }
    for (const SceneLib2Amd::Feature* s : a.selected_feature_list_) selected_feature_list_.push_back(find_feature_lab(s->label_));
    for (const auto& s : a.feature_init_info_vector_) {
      FeatureInitInfo info;
      info.fp_ = s.fp_ ? find_feature_lab(s.fp_->label_) : nullptr;
      info.mean_ = s.mean_; info.covariance_ = s.covariance_;
      info.number_of_match_attempts_ = s.number_of_match_attempts_;
      info.making_measurement_on_this_step_flag_ = s.making_measurement_on_this_step_flag_;
      for (const auto& p : s.particle_vector_) {
        Particle q;
        q.lambda_ = p.lambda_; q.probability_ = p.probability_; q.cumulative_probability_ = p.cumulative_probability_;
        q.m_h_ = vec(p.m_h_.data(), 2); q.m_z_ = vec(p.m_z_.data(), 2);
        q.m_SInv_ = mat(p.m_SInv_.data(), 2, 2);
        q.m_detS_ = p.m_detS_;
        q.m_successful_measurement_flag_ = p.m_successful_measurement_flag_;
        info.particle_vector_.push_back(q);
      }
      feature_init_info_vector_.push_back(info);
    }
    trajectory_store_.clear();
    for (const auto& t : a.trajectory_store_) trajectory_store_.push_back(Eigen::Vector3d(t[0], t[1], t[2]));
    number_of_visible_features_ = a.number_of_visible_features_;
    next_free_label_ = a.next_free_label_;
    total_state_size_ = a.total_state_size_;
    successful_measurement_vector_size_ = a.successful_measurement_vector_size_;
    kDeltaT_ = a.kDeltaT_; kNumberOfFeaturesToSelect_ = a.kNumberOfFeaturesToSelect_;
    kNumberOfFeaturesToKeepVisible_ = a.kNumberOfFeaturesToKeepVisible_; kMaxFeaturesToInitAtOnce_ = a.kMaxFeaturesToInitAtOnce_;
    kMinLambda_ = a.kMinLambda_; kMaxLambda_ = a.kMaxLambda_; kNumberOfParticles_ = a.kNumberOfParticles_;
    kStandardDeviationDepthRatio_ = a.kStandardDeviationDepthRatio_; kMinNumberOfParticles_ = a.kMinNumberOfParticles_;
    kPruneProbabilityThreshold_ = a.kPruneProbabilityThreshold_;
    kErasePartiallyInitFeatureAfterThisManyAttempts_ = a.kErasePartiallyInitFeatureAfterThisManyAttempts_;
    init_feature_search_ustart_ = a.init_feature_search_ustart_; init_feature_search_vstart_ = a.init_feature_search_vstart_;
    init_feature_search_ufinish_ = a.init_feature_search_ufinish_; init_feature_search_vfinish_ = a.init_feature_search_vfinish_;
    init_feature_search_region_defined_flag_ = a.init_feature_search_region_defined_flag_;
    uu_ = a.uu_; vv_ = a.vv_; location_selected_flag_ = a.location_selected_flag_;
  }

  SceneLib2Amd::MonoSLAM impl_{128, 0};
  bool finished_ = false;
};

inline void Kalman::KalmanFilterPredict(MonoSLAM* monoslam, Eigen::Vector3d& /*u: the constant-velocity model takes no control*/) {
  monoslam->kalman_predict_();                                             // kalman.cpp:50-69
}
inline void Kalman::KalmanFilterUpdate(MonoSLAM* monoslam) { monoslam->kalman_update_(); }   // kalman.cpp:72-119

}  // namespace SceneLib2

#endif  // TESTS_REF_BINDING_MONOSLAM_AMD_H
