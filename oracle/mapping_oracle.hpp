// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// Feature-initialisation path of MonoSLAM, restated from the behaviour of
// /root/reference/scenelib2/monoslam.cpp:823-1533, feature.cpp:45-104 / 204-269 and
// feature_init_info.cpp.  Included at the end of slam_oracle.hpp (member definitions of
// oracle::MonoSLAM).  Parity status: see slam_oracle.hpp / feature_init_oracle.hpp (unpinned by
// reference outputs; pinned by properties in tests/test_oracle_mapping.py).
//
// Reference quirks kept on purpose:
//   Q28  convert_from_partially_to_fully_initialised shifts the later features'
//        position_in_total_state_vector_ by the PARTIAL size 6 instead of 6 - 3 (feature.cpp:254);
//        dormant with the shipped max_features_to_init_at_once = 1 (the partial feature is the
//        last one in the list), reproduced here as written.
//   Q29  a partially initialised feature is never matched in the frame it was created in
//        (number_of_match_attempts_++ != 0, monoslam.cpp:1366).
//   Q30  the search centre of a particle's ellipse is truncated, not rounded (improc/search_
//        multiple_overlapping_ellipses.cpp:127-128), unlike elliptical_search.
#pragma once
#include "feature_init_oracle.hpp"

namespace oracle {

inline double MonoSLAM::drand48_() {
  Rand48 r;
  r.x = rand48_state;
  const double v = r.next();
  rand48_state = r.x;
  return v;
}

// monoslam.cpp:823-865
inline bool MonoSLAM::AutoInitialiseFeature(const uint8_t* frame) {
  const int kStepsToPredict = 10;
  const double kDepthHypothesis = 2.5;
  const double kSuitablePatchScoreThreshold = 20000;
  if (FindNonOverlappingRegion(init_feature_search_ustart, init_feature_search_vstart, init_feature_search_ufinish,
                               init_feature_search_vfinish, kStepsToPredict, kDepthHypothesis)) {
    init_feature_search_region_defined_flag = true;
    if (set_image_selection_automatically(frame, init_feature_search_ustart, init_feature_search_vstart,
                                          init_feature_search_ufinish, init_feature_search_vfinish) > kSuitablePatchScoreThreshold) {
      InitialiseFeature(frame);
    } else {
      return false;
    }
  } else {
    return false;
  }
  return true;
}

// monoslam.cpp:867-943
inline bool MonoSLAM::FindNonOverlappingRegion(int& ustart, int& vstart, int& ufinish, int& vfinish, int steps_to_predict,
                                               double depth_hypothesis) {
  Vec local_xv = xv;
  const double u0[3] = {0, 0, 0};
  for (int i = 0; i < steps_to_predict; ++i) {
    motion_model.func_fv_and_dfv_by_dxv(local_xv, u0, kDeltaT);
    local_xv = motion_model.fvRES;
  }
  motion_model.rRES[0] = local_xv(0); motion_model.rRES[1] = local_xv(1); motion_model.rRES[2] = local_xv(2);  // func_r
  const Quat qWR(local_xv(3), local_xv(4), local_xv(5), local_xv(6));
  Mat hR(3, 1);
  hR(2) = depth_hypothesis;
  const Mat rot = mul(qrot(qWR), hR);
  double yW[3];
  for (int i = 0; i < 3; ++i) yW[i] = motion_model.rRES[i] + rot(i);
  double xp[7];
  for (int i = 0; i < 7; ++i) xp[i] = xv(i);  // func_xp(xv_)
  full_feature_model.func_hi_and_jacobians(yW, xp);
  const double predicted_motion_u = camera.width / 2.0 - full_feature_model.hi[0];
  const double predicted_motion_v = camera.height / 2.0 - full_feature_model.hi[1];
  int safe_ustart = (int)(-predicted_motion_u);
  int safe_vstart = (int)(-predicted_motion_v);
  int safe_ufinish = (int)(camera.width - predicted_motion_u);
  int safe_vfinish = (int)(camera.height - predicted_motion_v);
  const int kBox = 11;
  if (safe_ustart < ((int)((kBox - 1) / 2) + 1)) safe_ustart = (kBox - 1) / 2 + 1;
  if (safe_ufinish > (int)camera.width - ((int)((kBox - 1) / 2) + 1)) safe_ufinish = (int)camera.width - (kBox - 1) / 2 - 1;
  if (safe_vstart < ((int)((kBox - 1) / 2) + 1)) safe_vstart = (kBox - 1) / 2 + 1;
  if (safe_vfinish > (int)camera.height - ((int)((kBox - 1) / 2) + 1)) safe_vfinish = camera.height - (kBox - 1) / 2 - 1;
  return FindNonOverlappingRegionNoPredict(safe_ustart, safe_vstart, safe_ufinish, safe_vfinish, ustart, vstart, ufinish, vfinish);
}

// monoslam.cpp:945-1032
inline bool MonoSLAM::FindNonOverlappingRegionNoPredict(int safe_ustart, int safe_vstart, int safe_ufinish, int safe_vfinish,
                                                        int& ustart, int& vstart, int& ufinish, int& vfinish) {
  const int kSearchWidth = 80, kSearchHeight = 60;
  if (safe_ufinish - safe_ustart > kSearchWidth && safe_vfinish - safe_vstart > kSearchHeight) {
    const int kTries = 5, kSeparationMinimum = 10;
    std::vector<double> u_array, v_array;
    double xp[7];
    for (int i = 0; i < 7; ++i) xp[i] = xv(i);  // motion_model_->xpRES_ holds the current position state here
    for (const Feature* f : feature_list) {
      if (!f->fully_initialised_flag) continue;
      full_feature_model.func_hi_and_jacobians(f->y, xp);
      const double hu = full_feature_model.hi[0], hv = full_feature_model.hi[1];
      full_feature_model.func_zeroedyi(f->y, xp);  // func_zeroedyigraphics_and_Pzeroedyigraphics: only its z is read
      if (full_feature_model.zeroedyi[2] > 0) { u_array.push_back(hu); v_array.push_back(hv); }
    }
    int i = 0;
    while (i < kTries) {
      const int u_offset = int((safe_ufinish - safe_ustart - kSearchWidth) * drand48_());
      const int v_offset = int((safe_vfinish - safe_vstart - kSearchHeight) * drand48_());
      ustart = safe_ustart + u_offset;
      ufinish = ustart + kSearchWidth;
      vstart = safe_vstart + v_offset;
      vfinish = vstart + kSearchHeight;
      bool found = false;
      for (size_t k = 0; k < u_array.size(); ++k)
        if (u_array[k] >= ustart - kSeparationMinimum && u_array[k] < ufinish + kSeparationMinimum &&
            v_array[k] >= vstart - kSeparationMinimum && v_array[k] < vfinish + kSeparationMinimum) {
          found = true;
          break;
        }
      if (!found) break;
      ++i;
    }
    if (i == kTries) return false;
  } else {
    return false;
  }
  return true;
}

// monoslam.cpp:1043-1055
inline double MonoSLAM::set_image_selection_automatically(const uint8_t* frame, int ustart, int vstart, int ufinish, int vfinish) {
  double evbest = 0;
  find_best_patch_inside_region(frame, camera.width, camera.height, &uu, &vv, &evbest, 11, ustart, vstart, ufinish, vfinish);
  location_selected_flag = true;
  return evbest;
}

// monoslam.cpp:1211-1251
inline void MonoSLAM::InitialiseFeature(const uint8_t* frame) {
  const double z[2] = {(double)uu, (double)vv};
  uint8_t patch[121];
  for (int r = 0; r < 11; ++r)
    for (int c = 0; c < 11; ++c) patch[r * 11 + c] = frame[(size_t)(r + vv - 5) * camera.width + c + uu - 5];  // copy_into_patch
  add_new_partially_initialised_feature(patch, z);
  const double lambda_step = (1.0 / double(kNumberOfParticles)) * (kMaxLambda - kMinLambda);
  const double uniform_probability = 1.0 / double(kNumberOfParticles);
  double lambda = kMinLambda;
  for (int i = 0; i < kNumberOfParticles; ++i) {
    feature_init_info_vector.back().add_particle(lambda, uniform_probability);
    lambda += lambda_step;
  }
  ++features_initialised;
}

// monoslam.cpp:1262-1276 + the partially-initialised Feature constructor, feature.cpp:45-104
inline void MonoSLAM::add_new_partially_initialised_feature(const uint8_t patch[121], const double h[2]) {
  Feature* nf = new Feature(6);
  nf->fully_initialised_flag = false;
  std::memcpy(nf->patch, patch, 121);
  nf->label = next_free_label;
  nf->position_in_list = (int)feature_list.size();
  nf->position_in_total_state_vector = total_state_size;
  double xp[7];
  for (int i = 0; i < 7; ++i) { xp[i] = xv(i); nf->xp_org[i] = xv(i); }
  part_feature_model.func_ypi_and_jacobians_and_Ri(h, xp);
  for (int i = 0; i < 6; ++i) nf->y[i] = part_feature_model.ypi[i];
  Mat dxp_by_dxv(7, 13);
  for (int i = 0; i < 7; ++i) dxp_by_dxv(i, i) = 1.0;
  const Mat T = mul(part_feature_model.dypi_by_dxp, dxp_by_dxv);  // dypi_by_dxv, 6x13
  nf->Pxy = mul(Pxx, transpose(T));
  Mat Ri(2, 2);
  Ri(0, 0) = part_feature_model.Ri; Ri(1, 1) = part_feature_model.Ri;
  nf->Pyy = add(mul(mul(T, Pxx), transpose(T)),
                mul(mul(part_feature_model.dypi_by_dhi, Ri), transpose(part_feature_model.dypi_by_dhi)));
  for (int j = 0; j < nf->position_in_list; ++j) nf->matrix_block_list.push_back(transpose(mul(T, feature_list[j]->Pxy)));
  feature_list.push_back(nf);
  total_state_size += 6;
  ++next_free_label;
  FeatureInitInfo info;
  info.fp = nf;
  feature_init_info_vector.push_back(info);
}

// monoslam.cpp:1299-1342
inline void MonoSLAM::MatchPartiallyInitialisedFeatures(const uint8_t* frame) {
  predict_partially_initialised_feature_measurements();
  for (FeatureInitInfo& feat : feature_init_info_vector)
    if (feat.making_measurement_on_this_step_flag) measure_feature_with_multiple_priors(frame, feat.fp->patch, feat.particle_vector);
  update_partially_initialised_feature_probabilities(kPruneProbabilityThreshold);
  for (size_t k = 0; k < feature_init_info_vector.size(); ++k) {
    FeatureInitInfo& feat = feature_init_info_vector[k];
    if (!feat.making_measurement_on_this_step_flag) continue;
    const double mean_sd_ratio = std::sqrt(feat.covariance) / feat.mean;
    if (mean_sd_ratio < kStandardDeviationDepthRatio && feat.particle_vector.size() > (unsigned int)kMinNumberOfParticles) {
      convert_from_partially_to_fully_initialised(feat.fp, feat.mean, feat.covariance);
      feature_init_info_vector.erase(feature_init_info_vector.begin() + k);  // erase(feat--) then ++feat: the next element is examined
      --k;
      ++features_converted;
    }
  }
  delete_partially_initialised_features_past_sell_by_date(kErasePartiallyInitFeatureAfterThisManyAttempts, kMinNumberOfParticles);
}

// monoslam.cpp:1349-1401
inline void MonoSLAM::predict_partially_initialised_feature_measurements() {
  double xp[7];
  for (int i = 0; i < 7; ++i) xp[i] = xv(i);
  Mat dxp_by_dxv(7, 13);
  for (int i = 0; i < 7; ++i) dxp_by_dxv(i, i) = 1.0;
  for (FeatureInitInfo& feat : feature_init_info_vector) {
    Feature* fp = feat.fp;
    if (feat.number_of_match_attempts++ != 0) {  // Q29
      feat.making_measurement_on_this_step_flag = true;
      for (Particle& part : feat.particle_vector) {
        part_feature_model.func_hpi_and_jacobians(fp->y, xp, part.lambda);
        part.m_h[0] = part_feature_model.hpi[0]; part.m_h[1] = part_feature_model.hpi[1];
        const double Ri = camera.MeasurementNoise(part.m_h);
        part_feature_model.func_Si(Pxx, fp->Pxy, fp->Pyy, mul(part_feature_model.dhpi_by_dxp, dxp_by_dxv),
                                   part_feature_model.dhpi_by_dyi, Ri);
        part.set_S(part_feature_model.Si);
      }
    } else {
      feat.making_measurement_on_this_step_flag = false;
    }
  }
}

// monoslam.cpp:1411-1439
inline void MonoSLAM::measure_feature_with_multiple_priors(const uint8_t* frame, const uint8_t* patch, std::vector<Particle>& particles) {
  MultiEllipseSearch search(frame, camera.width, camera.height, patch, 11);
  for (const Particle& p : particles) search.add_ellipse(p.SInv[0], p.SInv[1], p.SInv[2], p.m_h[0], p.m_h[1]);
  search.search();
  for (size_t i = 0; i < particles.size(); ++i) {
    if (search.data[i].result_flag) {
      particles[i].m_z[0] = search.data[i].result_u;
      particles[i].m_z[1] = search.data[i].result_v;
      particles[i].m_successful_measurement_flag = true;
    } else {
      particles[i].m_successful_measurement_flag = false;
    }
  }
}

// monoslam.cpp:1449-1497
inline void MonoSLAM::update_partially_initialised_feature_probabilities(double prune_probability_threshold) {
  for (size_t k = 0; k < feature_init_info_vector.size(); ++k) {
    FeatureInitInfo& feat = feature_init_info_vector[k];
    if (!feat.making_measurement_on_this_step_flag) continue;
    for (Particle& p : feat.particle_vector) {
      double likelihood;
      if (p.m_successful_measurement_flag) {
        const double nu0 = p.m_z[0] - p.m_h[0], nu1 = p.m_z[1] - p.m_h[1];
        const double t0 = p.SInv[0] * nu0 + p.SInv[1] * nu1;
        const double t1 = p.SInv[1] * nu0 + p.SInv[2] * nu1;
        const double nuT_Sinv_nu = nu0 * t0 + nu1 * t1;
        likelihood = (1.0 / (std::sqrt(2.0 * M_PI * p.detS))) * std::exp(-0.5 * nuT_Sinv_nu);
      } else {
        likelihood = 0.0;
      }
      p.probability = p.probability * likelihood;
    }
    if (feat.normalise_particle_vector_and_calculate_cumulative()) {
      feat.prune_particle_vector(prune_probability_threshold);
      feat.calculate_mean_and_covariance();
    } else {
      // all matches failed: the feature goes.  The reference erases inside a for(; feat < end; ++feat) loop, so the
      // element that slides into this slot is skipped in this pass.
      delete_partially_initialised_feature(k);
    }
  }
}

// monoslam.cpp:1506-1521
inline void MonoSLAM::delete_partially_initialised_features_past_sell_by_date(int erase_after_attempts, int min_number_of_particles) {
  for (size_t k = 0; k < feature_init_info_vector.size();) {
    const FeatureInitInfo& feat = feature_init_info_vector[k];
    if (feat.number_of_match_attempts > erase_after_attempts || feat.particle_vector.size() <= (unsigned int)min_number_of_particles)
      delete_partially_initialised_feature(k);
    else
      ++k;
  }
}

// monoslam.cpp:1523-1538
inline void MonoSLAM::delete_partially_initialised_feature(size_t index) {
  const int currently_marked_feature = marked_feature_label;
  mark_feature_by_lab(feature_init_info_vector[index].fp->label);
  delete_feature();
  feature_init_info_vector.erase(feature_init_info_vector.begin() + index);
  if (currently_marked_feature != -1) mark_feature_by_lab(currently_marked_feature);
  ++partial_features_deleted;
}

// feature.cpp:204-269
inline void MonoSLAM::convert_from_partially_to_fully_initialised(Feature* f, double lambda, double Plambda) {
  part_feature_model.func_yfi_and_jacobians(f->y, lambda);
  const Mat& J = part_feature_model.dyfi_by_dypi;
  const Mat JT = transpose(J);
  const Mat dT = transpose(part_feature_model.dyfi_by_dlambda);
  for (int i = 0; i < 3; ++i) f->y[i] = part_feature_model.yfi[i];
  for (int i = 3; i < 6; ++i) f->y[i] = 0.0;
  f->Pxy = mul(f->Pxy, JT);
  const Mat P1 = mul(mul(J, f->Pyy), JT);
  const Mat P2 = mul(scaled(part_feature_model.dyfi_by_dlambda, Plambda), dT);
  f->Pyy = add(P1, P2);
  const int i = f->position_in_list;
  for (int k = 0; k < i; ++k) f->matrix_block_list[k] = mul(f->matrix_block_list[k], JT);
  size_t pos = 0;
  while (feature_list[pos] != f) ++pos;
  for (++pos; pos < feature_list.size(); ++pos) {
    feature_list[pos]->matrix_block_list[i] = mul(J, feature_list[pos]->matrix_block_list[i]);
    feature_list[pos]->position_in_total_state_vector -= 6;  // Q28: the partial model's size, not 6 - 3
  }
  total_state_size += (3 - 6);
  f->state_size = 3;
  f->dh_by_dy = Mat(2, 3);
  f->fully_initialised_flag = true;
}

}  // namespace oracle
