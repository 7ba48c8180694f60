/*
 * scenelib2_amd — C ABI of the MI355X-native MonoSLAM per-frame engine.
 *
 * The reference (hanmekim/SceneLib2) has no plugin/FFI layer: its boundary is the
 * public C++ class SceneLib2::MonoSLAM (scenelib2/monoslam.h:69-219) used by
 * examples/MonoSlamSceneLib1.cpp:132-142.  This header is the C-ABI a maintainer
 * binds instead: plain pointers and sizes, int status codes,
 * no exceptions, no torch/Eigen/OpenCV types.  Each entry point cites the
 * reference interface it replaces (paths relative to the reference's scenelib2/).
 *
 * One engine = B independent image sequences ("batch"), each an independent
 * MonoSLAM instance with its own state vector, covariance, map and templates,
 * stepped together on one HIP stream of one GPU.  All *_dev pointers are device
 * pointers valid on the engine's device; host pointers are plain host memory.
 * All matrices crossing the ABI are row-major FP64 unless stated.
 *
 * Threading: one engine per host thread / stream; calls are asynchronous on the
 * engine's stream unless they return data to host memory (those synchronise).
 */
#ifndef SCENELIB2_AMD_H
#define SCENELIB2_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Version of this interface.  Bumped whenever an entry point changes meaning or a buffer it fills changes size; the
 * library reports the version it was built from through sl2_api_version(), so a caller compiled against an older header can
 * tell.  History: 1 = round 1; 2 = sl2_get_step_work fills 11 doubles, search variants 0-3; 3 = sl2_get_step_work fills 12
 * doubles (out[11] = candidate tiles), search variants renumbered (1 = int8 matrix-core walk, 0 = exact kernel; 2 and 3 are
 * gone and return SL2_ERR_INVALID), sl2_set_update_variant accepts only (1, 1) outside the TEST build; 4 = sl2_get_step_work
 * takes the capacity of the caller's array, sl2_snapshot / sl2_snapshot_capacity, several partially initialised features per
 * sequence (max_features_to_init_at_once > 1), sl2_get_partial_feature takes the index of the partial feature; 5 = sl2_ingest_next
 * uploads on a stream of its own, one frame ahead (or hands a small batch out in place): its `stream` argument is the stream the
 * frames are CONSUMED on, not the one the copy is queued on; sl2_set_update_variant no longer takes chol_variant 2; new:
 * sl2_set_step_fusion, sl2_get_stream, sl2_ingest_set_zero_copy; scenelib2_amd_comm.h. */
#define SL2_API_VERSION 5

#define SL2_OK 0
#define SL2_ERR_INVALID 1   /* bad argument */
#define SL2_ERR_HIP 2       /* HIP runtime error (sl2_last_error() has the text) */
#define SL2_ERR_CAPACITY 3  /* feature / batch capacity exceeded */
#define SL2_ERR_NO_DEVICE 4 /* no HIP device: the engine has NO CPU fallback */

#define SL2_STATE_SIZE 13    /* MotionModel::kStateSize_, motion_model.cpp:44 */
#define SL2_POSITION_SIZE 7  /* MotionModel::kPositionStateSize_ */
#define SL2_PATCH_SIZE 11    /* MonoSLAM::kBoxSize_, monoslam.cpp:48 */
#define SL2_PATCH_BYTES 121

typedef struct sl2_engine sl2_engine;

/* Camera::SetCameraParameters(width,height,fku,fkv,u0,v0,kd1,sd) — camera.cpp:49-62.
 * cfg keys cam.* (data/SceneLib2.cfg:24-31); fku,fkv,u0,v0,sd are read as ints by
 * the reference (monoslam.cpp:1597-1602) but stored as doubles/ints as below. */
typedef struct sl2_camera {
  int32_t width, height;
  double fku, fkv, u0, v0, kd1;
  int32_t sd;
} sl2_camera;

/* cfg keys params.* (data/SceneLib2.cfg:59-69, monoslam.cpp:1583-1593) plus the two
 * hard-wired deletion constants (monoslam.cpp:1875-1876). */
typedef struct sl2_params {
  double delta_t;
  int32_t number_of_features_to_select;
  int32_t number_of_features_to_keep_visible;
  int32_t max_features_to_init_at_once;
  double min_lambda, max_lambda;
  int32_t number_of_particles;
  double standard_deviation_depth_ratio;
  int32_t min_number_of_particles;
  double prune_probability_threshold;
  int32_t erase_partially_init_feature_after_this_many_attempts;
  int32_t minimum_attempted_measurements_of_feature; /* 10 */
  double successful_match_fraction;                  /* 0.5 */
} sl2_params;

/* Per-feature record returned by sl2_get_features — the public members of
 * SceneLib2::Feature that GraphicTool / the example read (feature.h:78-142). */
typedef struct sl2_feature_info {
  int32_t label;                               /* Feature::label_ */
  int32_t active;                              /* 0 once delete_feature() removed it */
  int32_t selected_flag;                       /* Feature::selected_flag_ */
  int32_t successful_measurement_flag;         /* Feature::successful_measurement_flag_ */
  int32_t attempted_measurements_of_feature;   /* Feature::attempted_measurements_of_feature_ */
  int32_t successful_measurements_of_feature;  /* Feature::successful_measurements_of_feature_ */
  int32_t position_in_total_state_vector;      /* as the reference would report it (deleted features removed) */
  int32_t visible;                             /* visibility_test()==0 in the last auto_select_n_features */
  double y[3];                                 /* Feature::y_ */
  double h[2], z[2], nu[2];                    /* h_, z_, nu_ */
  double R;                                    /* R_ = R * I2 */
  double S[4];                                 /* S_ (2x2) */
  double dh_by_dxp[14];                        /* first 7 columns of dh_by_dxv_ (rest are zero) */
  double dh_by_dy[6];                          /* dh_by_dy_ (2x3) */
  double xp_org[7];                            /* xp_org_ */
  int32_t fully_initialised_flag;              /* Feature::fully_initialised_flag_ (0: the six-state ray of a partially initialised feature) */
  int32_t state_size;                          /* 3, or 6 while partially initialised */
  double y_direction[3];                       /* partially initialised: y_(3..5), the unit ray direction; else 0 */
} sl2_feature_info;

/* ------------------------------------------------------------------ lifecycle */

/* SL2_API_VERSION of the library that is loaded (compare with the header's). */
int sl2_api_version(void);
/* Number of HIP devices visible (0 => every other call fails with SL2_ERR_NO_DEVICE). */
int sl2_device_count(void);

/* Replaces `new MonoSLAM` + the object wiring of MonoSLAM::Init (monoslam.cpp:1852-1885)
 * for `batch` sequences with room for `max_features` features each.  `stream` is a
 * hipStream_t (NULL => the engine creates its own). */
/* Limits (SL2_ERR_CAPACITY): at most 676 feature slots per sequence (2048 state columns: one k_build_AS workgroup holds a
 * sequence's row of A^T) and 512 features measured per frame (number_of_features_to_select); the reference's feature_list_ is
 * unbounded.  BASELINE's largest configuration (500 features, 1513 states) is inside. */
int sl2_create(const sl2_camera* cam, const sl2_params* params, int batch, int max_features, int device,
               void* stream, sl2_engine** out);
void sl2_destroy(sl2_engine* e);
const char* sl2_last_error(void);
int sl2_synchronize(sl2_engine* e);
/* The hipStream_t the engine's steps are queued on (the one given to sl2_create, or the engine's own): what sl2_ingest_next and
 * the scenelib2_amd_comm.h calls want as their `stream`. */
void* sl2_get_stream(sl2_engine* e);
int sl2_batch(const sl2_engine* e);
int sl2_max_features(const sl2_engine* e);

/* xv_ and Pxx_ as MonoSLAM::Init sets them (monoslam.cpp:1881-1938): xv [nseq][13] in
 * the order r(3) q(w,x,y,z) v(3) omega(3); Pxx [nseq][13][13]. */
int sl2_set_vehicle_state(sl2_engine* e, int seq0, int nseq, const double* xv, const double* Pxx);
int sl2_get_vehicle_state(sl2_engine* e, int seq0, int nseq, double* xv, double* Pxx);

/* MonoSLAM::AddNewKnownFeature(y, xp, identifier) (monoslam.cpp:1278-1291 ->
 * Feature ctor feature.cpp:108-149), batched: `nfeat` features appended to each of
 * sequences [seq0, seq0+nseq).  y [nseq][nfeat][3], xp_org [nseq][nfeat][7],
 * patches [nseq][nfeat][121] = the 11x11 8-bit template the reference would
 * cv::imread from `identifier`.  Labels continue from next_free_label_; a sequence that lacks free slots first gets the slots
 * of its deleted features back (see SL2_STATUS_LABELS_EXHAUSTED below); SL2_ERR_CAPACITY if it still holds more than
 * max_features - nfeat live features. */
int sl2_add_known_features(sl2_engine* e, int seq0, int nseq, int nfeat, const double* y, const double* xp_org,
                           const uint8_t* patches);

/* Feature::Pyy_ of the first `nfeat` features of each sequence (feature.h:84): Pyy [nseq][nfeat][3][3].
 * AddNewKnownFeature leaves it zero (feature.cpp:134-135); a map whose features carry a prior
 * uncertainty (what feature initialisation produces, feature.cpp:241-244) is loaded with this. */
int sl2_set_feature_covariances(sl2_engine* e, int seq0, int nseq, int nfeat, const double* Pyy);

/* ------------------------------------------------------------------- stepping */

/* MonoSLAM::GoOneStep(frame, save_trajectory, enable_mapping) (monoslam.cpp:108-180)
 * for every sequence.  frames: [batch] images of width*height bytes, 8-bit single
 * channel, row pitch == width (Q25); seq_stride = bytes between consecutive
 * sequences' frames.  frames_on_device != 0 => `frames` is a device pointer
 * (zero-copy); otherwise host memory, copied H2D on the engine's stream.
 * enable_mapping != 0 runs the feature-initialisation tail (monoslam.cpp:152-170: AutoInitialiseFeature behind the
 * 0.2 m/s speed gate, MatchPartiallyInitialisedFeatures) with up to params.max_features_to_init_at_once <= 4 partially
 * initialised features in flight per sequence (the shipped value is 1; the six state columns of each are reserved at
 * sl2_create) and up to 1024 depth particles (params.number_of_particles; the shipped value is 100); other settings are
 * rejected with SL2_ERR_INVALID.  With more than one feature in flight the reference's own arithmetic is kept, including the
 * position it records for later features after a conversion (feature.cpp:254).  SL2_STATUS_LABELS_EXHAUSTED = a sequence could
 * not take a new feature because all max_features slots hold live features. */
int sl2_go_one_step(sl2_engine* e, const uint8_t* frames, size_t seq_stride, int frames_on_device,
                    int save_trajectory, int enable_mapping);

/* The seams of GoOneStep, individually callable (same order as the reference):
 *   Kalman::KalmanFilterPredict(monoslam,u=0)            kalman.cpp:50-69
 *   MonoSLAM::auto_select_n_features(n)                  monoslam.cpp:187-254
 *   MonoSLAM::make_measurements(image)                   monoslam.cpp:336-359
 *   Kalman::KalmanFilterUpdate(monoslam)+normalise_state kalman.cpp:72-119, monoslam.cpp:616-637
 *   delete_bad_features + symmetrise                     monoslam.cpp:141-150
 */
/* MonoSLAM::InitialiseFeature(frame) (monoslam.cpp:1211-1235; the "initialise manual feature" button of
 * examples/MonoSlamSceneLib1.cpp:191-192, after a mouse click set uu_ / vv_): for every sequence s with uv[2 s] >= 0 a
 * partially initialised feature is created at pixel (uu_, vv_) = (uv[2 s], uv[2 s + 1]) of that sequence's frame - the
 * 11 x 11 patch copied from the frame, the semi-infinite line from the current pose, number_of_particles depth hypotheses
 * - exactly as the feature-initialisation tail of sl2_go_one_step creates one.  Later steps match and convert it
 * (MatchPartiallyInitialisedFeatures runs in every step from now on, like monoslam.cpp:167).  uv: host [batch][2], consumed
 * before the call returns (pageable, pinned or registered memory alike).
 * created: host [batch], may be NULL (the kernels then stay asynchronous); 1 = a feature was created.  It is NOT created -
 * and the reference would have created it - when the sequence already has a partially initialised feature (this engine
 * carries one at a time, the shipped max_features_to_init_at_once = 1), when the patch would leave the frame (the
 * reference reads out of bounds), or when the label slots are exhausted (SL2_STATUS_LABELS_EXHAUSTED). */
int sl2_initialise_feature(sl2_engine* e, const uint8_t* frames, size_t seq_stride, int frames_on_device, const int32_t* uv,
                           int32_t* created);
/* MonoSLAM::InitialiseAutoFeature(frame) (monoslam.cpp:1535-1541, the "initialise auto feature" button): AutoInitialiseFeature
 * with zero control input for every sequence - FindNonOverlappingRegion (consumes the sequence's drand48 stream), the
 * Shi-Tomasi detector, the 20000 score threshold, then the creation above - without the speed and visible-feature gates
 * of GoOneStep.  created as above. */
int sl2_initialise_auto_feature(sl2_engine* e, const uint8_t* frames, size_t seq_stride, int frames_on_device, int32_t* created);
/* MonoSLAM::SavePatch (monoslam.cpp:1551-1572): writes Feature::patch_ of the feature with this label (the reference saves
 * the marked feature to "patch.png").  ".pgm" writes a binary PGM, any other name an 8-bit greyscale PNG.  Synchronises. */
int sl2_save_patch(sl2_engine* e, int seq, int label, const char* path);

/* Split the batch into `groups` contiguous sequence groups, each stepped on its own HIP stream
 * (latency-bound kernels of one group overlap throughput-bound kernels of another).  Default 1
 * (measured on MI355X: no gain from 2, a loss from 4+ at batch 1024).  The library never reads the environment: the
 * SL2_* development switches exist in the TEST build (libscenelib2_amd_test.so) only. */
int sl2_set_groups(sl2_engine* e, int groups);
/* Whole-step HIP graphs: when enabled, sl2_go_one_step with device-resident frames captures its launches once per
 * (frame buffer, flags) and replays the graph (at batch 1 the step is launch-bound: ~12 kernels).  Off by default;
 * ignored while per-kernel profiling is on or with more than one sequence group. */
int sl2_set_graph_mode(sl2_engine* e, int enabled);
/* Which search kernel sl2_make_measurements / sl2_go_one_step use: 1 = int8 matrix-core walk (k_search_mfma, default),
 * 0 = the exact kernel with one candidate per lane (k_search_exact, the cross-check).  Identical results, bit for bit. */
int sl2_set_search_variant(sl2_engine* e, int variant);
/* Small maps (the reference's own workload: data/SceneLib2.cfg:60-62 keeps a dozen features and measures ten per frame):
 * when the whole state fits 128 columns (max_features <= 35 with one partially initialised feature) and at most 16 features
 * are measured per frame, sl2_go_one_step issues THREE launches instead of ten - predict + measurement prediction + selection,
 * the patch search, scoring + EKF update + normalise / delete / symmetrise (monoslam.cpp:108-180 unchanged in meaning;
 * search results, selection and every counter identical bit for bit, state and covariance equal to rounding).
 * The choice follows the LIVE maps, not the capacity: the engine keeps an upper bound on the number of feature slots in use
 * (exact wherever a call synchronises anyway, from a device-to-host mailbox in between; never a synchronisation of its own)
 * and takes the fused step while 13 + 3 slots + 7 <= 128 and either the batch (per sequence group) is at most 256 sequences
 * or the capacity is large (state columns >= 256: the one-stage kernels work on the whole capacity, the fused ones on the
 * live part); small capacities at large batches fuse the stages behind the search only (six launches).
 * enabled = 1 (default) / 0 = always the one-stage-per-launch kernels / 2 = fused whatever the batch size (measurements).
 * The seam entry points below always use the one-stage kernels. */
int sl2_set_step_fusion(sl2_engine* e, int enabled);
/* Search windows of at least `min_bands` bands (a band = 32 x 16 candidate positions; a window of nu x nv positions has
 * ceil(ceil(nu / 16) / 2) * ceil(nv / 16) of them) are not walked by one wavefront but cut into units of four bands that
 * extra workgroups at the end of the search launch (a quarter of it, at most 2048) work off - the window of a poorly constrained feature can be the
 * whole frame (150 bands at 320 x 240), and one wavefront walking it alone was the tail of the search.  Default: a twentieth
 * of the frame's bands (8 at 320 x 240, 96 at 1280 x 720: what is shared out should be the rare oversized window); 0 =
 * never (and no extra workgroups: they cost the headline step 0.06 %).  Results are identical either way: the parts' best
 * and second-best candidates are combined into the decision a single wavefront takes.  (An addition within SL2_API_VERSION 4 at the time: no existing entry point
 * changed; sl2_get_step_work has a 13th value.) */
int sl2_set_search_split(sl2_engine* e, int min_bands);
/* Kernel choice inside sl2_kalman_filter_update (identical algebra, results equal to rounding):
 * chol_variant 1 = one-launch left-looking Cholesky (k_chol_left; default), 0 = three launches per block column;
 * fwd_variant  1 = forward substitution with L streamed through LDS and the solved rows in registers (k_fwdsub_lds;
 *              default), 0 = operands re-read from memory (k_fwdsub).
 * Only the defaults (1, 1) are compiled into this library; the alternatives live in the TEST build
 * (libscenelib2_amd_test.so, include/scenelib2_amd_testing.h), where this call selects them - here anything else
 * returns SL2_ERR_INVALID.  (chol_variant 2, the right-looking one-launch kernel of rounds 1-2, was retired in round 6 and is
 * SL2_ERR_INVALID everywhere.)  Systems of more than 16 blocks are factored panel-wise (k_chol_left + k_fwdsub_lds +
 * k_chol_syrk per 256 columns) and systems of more than 13 blocks substituted in groups of eight block rows (k_fwd_gemm +
 * k_fwdsub_lds) either way. */
int sl2_set_update_variant(sl2_engine* e, int chol_variant, int fwd_variant);
int sl2_kalman_filter_predict(sl2_engine* e);
int sl2_auto_select_n_features(sl2_engine* e, int n);
/* Consumes the selection of the sl2_auto_select_n_features call before it (as make_measurements consumes
 * selected_feature_list_, monoslam.cpp:336-352): ONE call per selection - the list of large search windows that the selection
 * left for the search (sl2_set_search_split) is worked off and cleared by this call. */
int sl2_make_measurements(sl2_engine* e, const uint8_t* frames, size_t seq_stride, int frames_on_device);
int sl2_kalman_filter_update(sl2_engine* e);
int sl2_finish_step(sl2_engine* e, int save_trajectory);

/* MonoSLAM::elliptical_search (monoslam.cpp:401-477) as a stateless batch: `count`
 * independent searches.  image_index[i] selects one of `nimages` images (all
 * width x height).  patches [count][121]; centre [count][2]; puinv [count][3] =
 * (PuInv(0,0), PuInv(0,1), PuInv(1,1)).  Outputs: ok[count] (the bool return),
 * uv[count][2] (left untouched where no candidate qualified, Q4), score[count]
 * (corrmax).  All pointers are HOST pointers; variant 1 = production search core (int8 matrix-core walk, the one
 * k_search_mfma runs), variant 0 = the exact kernel kept for cross-checking (identical results); anything else is
 * SL2_ERR_INVALID. */
int sl2_elliptical_search_batch(int device, const uint8_t* images, int nimages, int width, int height,
                                const int32_t* image_index, const uint8_t* patches, const double* centre,
                                const double* puinv, int count, int32_t* ok, int32_t* uv, double* score,
                                int variant);

/* ----------------------------------------------------- feature initialisation */

/* MonoSLAM::find_best_patch_inside_region (monoslam.cpp:1070-1192) + find_eigenvalues (:1194-1205) as a
 * stateless batch: the Shi-Tomasi detector over `njobs` regions.  region [njobs][4] = (ustart, vstart,
 * ufinish, vfinish), clamped inside like the reference (6 px from every border for the 11x11 box);
 * uv [njobs][2] = (*ubest, *vbest) IN/OUT: left untouched when no position has a positive smaller
 * eigenvalue (evbest = 0), set to the clamped (ustart, vstart) for an empty region; evbest [njobs] = the
 * smaller eigenvalue of the winner (first maximum in scan order v outer / u inner).  Bit-exact with the
 * reference's FP64 arithmetic.  All pointers are HOST pointers; kernel_ms (optional) receives the device
 * time of the kernels alone (HIP events). */
int sl2_find_best_patch_batch(int device, const uint8_t* images, int nimages, int width, int height, int njobs,
                              const int32_t* image_index, const int32_t* region, int32_t* uv, double* evbest,
                              double* kernel_ms);

/* SearchMultipleOverlappingEllipses (improc/search_multiple_overlapping_ellipses.{h,cpp}) as a stateless
 * batch: job j = one (image, 11x11 patch) with ellipse_count[j] ellipses (the particles of one partially
 * initialised feature); ellipses of all jobs are concatenated in puinv [total][3] = (PuInv(0,0), PuInv(0,1),
 * PuInv(1,1)) and centre [total][2] (add_ellipse order).  result [total][3] = (result_flag_, result_u_,
 * result_v_); corrmax [total] (optional) = the best score of each ellipse (diagnostic).  Semantics kept:
 * centre truncated without rounding (cpp:127-128), +5.0 penalty for image sigma < 10 instead of a
 * rejection (cpp:173-175), no patch-sigma test, "<=" => last minimum in scan order wins, flag = score <= 0.40.
 * Each position of the union is scored once (the reference's cache), by the first ellipse that visits it. */
int sl2_search_multiple_overlapping_ellipses_batch(int device, const uint8_t* images, int nimages, int width, int height,
                                                   int njobs, const int32_t* image_index, const uint8_t* patches,
                                                   const int32_t* ellipse_count, const double* puinv, const double* centre,
                                                   int32_t* result, double* corrmax, double* kernel_ms);

/* ------------------------------------------------------------- frame ingest */

/* FileGrabber::ProcessFiles (framegrabber/filegrabber.cpp:63-83): every regular file below `dir`, recursively,
 * sorted by full path.  buf receives the paths separated by '\n' (may be NULL to query *count only).  Host only. */
int sl2_list_frames(const char* dir, char* buf, size_t capacity, int* count);
/* One binary PGM (P5, maxval <= 255) -> 8-bit grey, row-major, step == width (the frame contract of GoOneStep;
 * the reference decodes with cv::imread(path, 0), filegrabber.cpp:105-108).  out may be NULL to query the size.  Host only. */
int sl2_read_pgm(const char* path, uint8_t* out, size_t capacity, int* width, int* height);
/* FileGrabber::GetImageFile (filegrabber.cpp:106-109: cv::imread(path, 0)) for the three containers this library decodes,
 * chosen by the file's magic bytes: binary PGM as above; PNG (plain or Adam7-interlaced; 8-bit grey / grey+alpha / RGB / RGBA /
 * palette, 1-2-4-bit grey / palette) - grey PNGs are byte-exact, colour is reduced like libpng's rgb_to_gray, which is what
 * imread(.., 0) uses: (9797 R + 19234 G + 3737 B + 16384) >> 15; JPEG (sequential and progressive DCT, Huffman, 8 bits, one or
 * three components) - the luminance component through libjpeg's integer inverse DCT (jpeg_idct_islow), which is what
 * imread(.., 0) gets from libjpeg with out_color_space = JCS_GRAYSCALE; arithmetic-coded / lossless / 12-bit / four-component
 * files are refused with an error.  out may be NULL to query the size.  Host only. */
int sl2_read_image(const char* path, uint8_t* out, size_t capacity, int* width, int* height);
/* FileGrabber + FrameGrabber for a batch: dirs[s] is the frame directory of sequence s.  A producer thread decodes
 * (sl2_read_image: PGM, PNG or JPEG) ahead into `depth` (2..50, framegrabber.cpp:93-104) pinned host batches; sl2_ingest_next hands out
 * the next frame of every sequence in one of two device buffers for sl2_go_one_step(frames_on_device = 1).  The returned pointer
 * stays valid until the next-but-one call.
 * The upload runs on a copy stream of the grabber's own, ONE FRAME AHEAD: the call that hands out frame k also starts the copy of
 * frame k + 1 (if it is decoded), which then runs under the caller's work on frame k.
 * Stream contract: `stream` is the stream on which the caller CONSUMES the frames (the one the engine steps on: the stream given to
 * sl2_create, or sl2_get_stream of an engine-owned one).  The call makes that stream wait for frame k's copy, and takes the point
 * that stream has reached as the moment the buffer of frame k - 1 may be overwritten - so the consumer of a frame must be queued
 * on that stream BEFORE the next call.  NULL = the legacy default stream, which is ordered with every blocking stream (correct
 * with an engine-owned stream, at the price of the default stream's implicit synchronisations).
 * A decode failure is reported by the call whose frame could not be produced, not by earlier ones.
 * SL2_ERR_CAPACITY = the shortest sequence is exhausted (sl2_ingest_frame_count). */
typedef struct sl2_ingest sl2_ingest;
int sl2_ingest_open(const char* const* dirs, int nseq, int width, int height, int device, int depth, sl2_ingest** out);
int sl2_ingest_frame_count(const sl2_ingest* g);
int sl2_ingest_next(sl2_ingest* g, void* stream, const uint8_t** d_frames, size_t* seq_stride);
/* Batches of at most max_batch_bytes (nseq * width * height; default 512 KB, 0 = never) are not uploaded at all: sl2_ingest_next
 * hands out the pinned host batch itself, which the device reads in place (depth >= 4; the same stream contract - the batch goes
 * back to the decoder once the caller's stream is past the point it had reached at the NEXT call).  A single 320 x 240 sequence
 * saves the ~10 us per frame that queueing a copy costs the host.  Before the first sl2_ingest_next only. */
int sl2_ingest_set_zero_copy(sl2_ingest* g, size_t max_batch_bytes);
void sl2_ingest_close(sl2_ingest* g);

/* ----------------------------------------------------------------- state access */

/* total_state_size_ per sequence (13 + 3 * live features). */
int sl2_get_total_state_sizes(sl2_engine* e, int seq0, int nseq, int32_t* sizes);
/* construct_total_state / construct_total_covariance (monoslam.cpp:501-546) for one
 * sequence, deleted features removed: x [n], P [n][n], n = total state size. */
int sl2_get_total_state(sl2_engine* e, int seq, double* x, int capacity);
int sl2_get_total_covariance(sl2_engine* e, int seq, double* P, int capacity_n);
/* feature_list_ of one sequence in list order (deleted features skipped unless
 * include_deleted).  Returns the number written through *count. */
int sl2_get_features(sl2_engine* e, int seq, sl2_feature_info* out, int capacity, int include_deleted, int* count);
/* Entry `index` of feature_init_info_vector_ of one sequence (FeatureInitInfo + its particles, feature_init_info.h:46-118;
 * up to params.max_features_to_init_at_once <= 4 entries, in the vector's order).
 * ints [16] = feature_init_info_vector_.size(), label, number_of_match_attempts_, #particles,
 * making_measurement_on_this_step_flag_, uu_, vv_, region_defined (this step), init_feature_search_{ustart,vstart,ufinish,vfinish}_,
 * #initialised, #converted, #deleted (totals for this sequence), created (this step); dbl [9] = mean_, covariance_, y_(0..5),
 * evbest of the last detection; particles [capacity][12] = lambda_, probability_, cumulative_probability_, m_h_(2), m_z_(2),
 * m_SInv_(00,01,11), m_detS_, m_successful_measurement_flag_ (may be NULL).  ints [0] and the sequence-level entries (5..15,
 * dbl [8]) are filled for any index; the feature's own entries are zero when index >= ints [0]. */
int sl2_get_partial_feature(sl2_engine* e, int seq, int index, int32_t* ints, double* dbl, double* particles, int capacity);
/* Feature::patch_ (feature.h:118; 11x11, row-major) of the feature with this label: the template given to
 * sl2_add_known_features, or the one copy_into_patch cut from the frame when the feature was initialised
 * (monoslam.cpp:1236-1250).  Labels of deleted features keep their last template. */
int sl2_get_feature_patch(sl2_engine* e, int seq, int label, uint8_t* patch121);
/* ---- one-call read-back of everything the reference exposes as public members (monoslam.h:158-218, feature.h:78-142) ----
 *
 * The reference's caller reads xv_, Pxx_, feature_list_[i]->{y_, Pxy_, Pyy_, h_, z_, S_, ...}, selected_feature_list_,
 * trajectory_store_ and feature_init_info_vector_ straight out of the object after every GoOneStep
 * (examples/MonoSlamSceneLib1.cpp:132-151, graphic/graphictool.cpp:130-167, 290-347).  sl2_snapshot is that read for one
 * sequence: ONE kernel packs the blob below on the device and streams it into an engine-owned pinned host buffer, ONE
 * stream synchronisation follows, and *blob points into that buffer (valid until the next sl2_snapshot / sl2_destroy on
 * this engine; no allocation per call).
 *
 *   traj_cursor       number of trajectory_store_ pushes the caller has already seen (0 the first time): the blob carries
 *                     the entries [max(traj_cursor, total - 1000), total) - the caller appends them and drops from the
 *                     front beyond 1000 entries like monoslam.cpp:172-177;
 *   patch_from_label  the 11 x 11 templates (Feature::patch_) of the live features whose label_ is >= this value are
 *                     included (labels are handed out once and a feature's template never changes, so a caller that caches
 *                     templates by label passes its next_free_label_ of the previous call; 0 = all, INT32_MAX = none).
 *
 * Blob layout (every section starts on an 8-byte boundary; offsets in bytes from the start of the blob):
 *   sl2_snapshot_header
 *   off_xv         double[13]                      xv_
 *   off_Pxx        double[13][13]                  Pxx_
 *   off_features   sl2_feature_info[n_features]    feature_list_ order, deleted features removed
 *   off_cov        per feature, in the same order: Pxy_ (13 x d, row-major) then Pyy_ (d x d), d = state_size (3 or 6)
 *   off_selection  int32[n_selected]               selected_feature_list_ as labels, selection order
 *   off_traj       double[traj_count][3]           trajectory_store_ entries traj_first .. traj_first + traj_count - 1
 *   off_partial    n_partial records, feature_init_info_vector_ order: sl2_partial_info, then double[n_particles][12]
 *                  (lambda_, probability_, cumulative_probability_, m_h_(2), m_z_(2), m_SInv_(00, 01, 11), m_detS_,
 *                  m_successful_measurement_flag_)
 *   off_patches    n_patches records of 128 bytes: int32 label, uint8[121] patch (row-major), 3 bytes of padding */
typedef struct sl2_snapshot_header {
  int32_t magic;                               /* 0x53324c53 ("SL2S") */
  int32_t api_version;                         /* SL2_API_VERSION of the library */
  int32_t bytes;                               /* size of the whole blob */
  int32_t seq;
  int32_t n_features;                          /* feature_list_.size() */
  int32_t total_state_size;                    /* total_state_size_ */
  int32_t number_of_visible_features;          /* number_of_visible_features_ */
  int32_t n_selected;                          /* selected_feature_list_.size() */
  int32_t successful_measurement_vector_size;  /* successful_measurement_vector_size_ */
  int32_t next_free_label;                     /* next_free_label_ */
  int32_t status_flags;                        /* SL2_STATUS_* */
  int32_t traj_total, traj_first, traj_count;
  int32_t n_partial;                           /* feature_init_info_vector_.size() */
  int32_t n_patches;
  int32_t uu, vv;                              /* uu_, vv_ */
  int32_t location_selected_flag;              /* location_selected_flag_ */
  int32_t init_feature_search_region_defined_flag;
  int32_t init_feature_search_region[4];       /* ustart, vstart, ufinish, vfinish */
  int32_t off_xv, off_Pxx, off_features, off_cov, off_selection, off_traj, off_partial, off_patches;
  int32_t steps_done;                          /* GoOneStep calls so far (low 31 bits) */
  int32_t reserved[31];
} sl2_snapshot_header;                         /* 256 bytes */
typedef struct sl2_partial_info {              /* FeatureInitInfo, feature_init_info.h:77-118 */
  int32_t label;                               /* fp_->label_ */
  int32_t number_of_match_attempts;
  int32_t n_particles;                         /* particle_vector_.size() */
  int32_t making_measurement_on_this_step_flag;
  double mean, covariance;                     /* mean_, covariance_ of lambda */
} sl2_partial_info;                            /* 32 bytes, followed by the particles */
/* Upper bound of a blob of this engine in bytes. */
size_t sl2_snapshot_capacity(const sl2_engine* e);
int sl2_snapshot(sl2_engine* e, int seq, int traj_cursor, int patch_from_label, const void** blob, size_t* bytes);

/* selected_feature_list_ (labels, selection order) and per-step counters:
 * counters[0] = number_of_visible_features_, [1] = #selected,
 * [2] = successful_measurement_vector_size_. */
int sl2_get_selection(sl2_engine* e, int seq, int32_t* labels, int capacity, int32_t counters[3]);
/* trajectory_store_ (monoslam.cpp:172-177; keeps the reference's stale-scratch
 * behaviour): up to `capacity` most recent entries of 3 doubles. */
int sl2_get_trajectory(sl2_engine* e, int seq, double* out, int capacity, int* count);
/* xv_(0..2) after each of the last `count` steps (the actual camera trajectory; the
 * reference's trajectory_store_ holds a stale scratch value instead):
 * out [nseq][count][3], count = min(capacity, steps done, 1000). */
int sl2_get_position_log(sl2_engine* e, int seq0, int nseq, double* out, int capacity, int* count);
/* MonoSLAM::mark_feature_by_lab(label) + delete_feature() (monoslam.cpp:743-812) for one feature per sequence:
 * labels [nseq], -1 = leave that sequence alone.  deleted [nseq] (may be NULL) receives the reference's bool: 1 if a live,
 * fully initialised feature with that label existed and was removed (partially initialised ones are removed by the engine's
 * own sell-by / conversion logic only).  The label is never reused (its slot may be).  Synchronises. */
int sl2_delete_features(sl2_engine* e, int seq0, int nseq, const int32_t* labels, int32_t* deleted);
/* Per-sequence status bits.  SL2_STATUS_NONFINITE (sticky): NaN / Inf seen in the state (e.g. the omega == 0 hazard, Q10).
 * SL2_STATUS_LABELS_EXHAUSTED: the LAST feature initialisation that was called for (speed gate and visible-feature count of
 * GoOneStep, or one of the two initialise-feature entry points) could not take place because every one of the sequence's
 * max_features slots holds a LIVE feature (the reference's feature_list_ is unbounded).  Deleted features do not count:
 * their slots are squeezed out, in feature_list_ order, when a sequence runs out of slots, and labels (Feature::label_ =
 * next_free_label_++) are kept apart from slots and never reused - a sequence may hand out any number of labels over its
 * lifetime.  The bit is cleared again by the next initialisation attempt that finds room (after a feature has been deleted),
 * so a temporarily full map is not a permanent condition.  While it is set the sequence keeps tracking its map but
 * initialises no further features.  Callers that run with enable_mapping must poll this (the MonoSLAM adapters do, and
 * raise when the bit RISES). */
#define SL2_STATUS_NONFINITE 1
#define SL2_STATUS_LABELS_EXHAUSTED 2
/* SL2_STATUS_REFERENCE_OUT_OF_BOUNDS (sticky; only with max_features_to_init_at_once > 1): the position the reference records
 * for a feature after several conversions and deletions (feature.cpp:254, Q28) has gone NEGATIVE; the reference then writes
 * its dh_by_dy block outside dh_by_dx_tot (monoslam.cpp:564, undefined behaviour).  The engine holds the block at column 0
 * and flags the sequence: from here on its filter is not the reference's. */
#define SL2_STATUS_REFERENCE_OUT_OF_BOUNDS 4
int sl2_get_status_flags(sl2_engine* e, int seq0, int nseq, int32_t* flags);

/* ------------------------------------------------------------------- profiling */

/* enabled = 1: the roofline kernels (k_search_mfma, k_build_AS, k_chol_left, k_fwdsub_lds, k_syrk and their large-map
 * forms; narrowed by sl2_set_profile_focus) are bracketed by HIP events on the engine's stream; enabled = 2: every kernel
 * launch is; sl2_get_kernel_time returns accumulated milliseconds and launch counts per kernel symbol since the last reset. */
int sl2_set_profiling(sl2_engine* e, int enabled);
/* Level 1 brackets only the kernels named here (comma separated scope names, e.g. "k_syrk,k_search_mfma"; NULL or "" = the
 * four large ones): every bracket is two event markers on the stream, and four of them cost 1-3 % of a step. */
int sl2_set_profile_focus(sl2_engine* e, const char* names);
int sl2_reset_kernel_times(sl2_engine* e);
int sl2_kernel_count(sl2_engine* e);
int sl2_get_kernel_time(sl2_engine* e, int idx, const char** name, double* total_ms, int64_t* launches);
/* Algorithmic work of the last completed step summed over the batch:
 * out[0] = search window bytes sum_f (2hw+11)(2hh+11) capped per sequence at W*H,
 * out[1] = number of searched features, out[2] = in-ellipse candidates,
 * out[3] = sum over sequences of m (measurement rows), out[4] = sum of m^2,
 * out[5] = sum of m^3, out[6] = sum of n (state size) , out[7] = sum n*m, out[8] = sum n*n*m, out[9] = sum n*m*m,
 * out[10] = searches that took the exact fallback kernel path,
 * out[11] = 16 x 16 candidate tiles of the search windows, sum_f ceil(nu / 16) ceil(nv / 16): the matrix-core work of
 * k_search_mfma (24 v_mfma_i32_16x16x64_i8 per tile),
 * out[12] = search windows that were shared out over the wavefronts of the launch (sl2_set_search_split) */
#define SL2_STEP_WORK_COUNT 13
/* capacity = length of the caller's array: min(capacity, SL2_STEP_WORK_COUNT) values are written (a caller built against a
 * header with fewer entries is never overrun; one built against more sees the extra entries untouched). */
int sl2_get_step_work(sl2_engine* e, double* out, int capacity);

/* How sl2_create placed the large matrices (an addition within SL2_API_VERSION 5).  The speed of the update's kernels depends on
 * the physical memory behind P, A^T, V^T and S - up to 8 % for k_build_AS at 1024 sequences x 100 features, fixed for the life of
 * an allocation, not visible in its address - so sl2_create times a streaming probe on several candidate allocations and keeps
 * the fastest of each kind (engines whose covariance is smaller than 256 MB: not tried, all zeros here).
 * out[0] = candidate allocations of P probed, out[1..4] = probe time in ms of the kept P, V^T, A^T, S, out[5..7] = of the slowest candidate
 * seen for P, for A^T / V^T, for S, out[8..9] = k_syrk itself on the kept pair (P, V^T) and on the slowest pair tried (the kernel
 * is its own probe: on an all-zero engine it does its full work and changes nothing).  min(capacity, SL2_PLACEMENT_COUNT) values
 * are written. */
#define SL2_PLACEMENT_COUNT 10
int sl2_get_placement(sl2_engine* e, double* out, int capacity);

/* ------------------------------------------------------------- synthetic input */

/* Render `count` frames of the synthetic textured plane z = 0:
 * poses [count][7] = camera r(3), q(w,x,y,z); tex = tex_size^2 8-bit texture
 * covering tex_extent metres (torus-wrapped), shifted by tex_origin[count][2].
 * out [count][height][width].  The device and host versions run the SAME source
 * (scenelib2_amd/csrc/sl2_synth.hpp) and produce identical bytes. */
int sl2_synth_render_host(const sl2_camera* cam, const uint8_t* tex, int tex_size, double tex_extent,
                          const double* tex_origin, const double* poses, int count, uint8_t* out);
int sl2_synth_render_device(int device, void* stream, const sl2_camera* cam, const uint8_t* tex_dev, int tex_size,
                            double tex_extent, const double* tex_origin_dev, const double* poses_dev, int count,
                            uint8_t* out_dev);

/* ------------------------------------------------------ device-memory helpers */
/* For hosts without their own HIP binding (the Python tests use these; a torch
 * caller passes tensor.data_ptr() instead). */
int sl2_dev_malloc(int device, size_t bytes, void** out);
int sl2_dev_free(int device, void* p);
int sl2_dev_upload(int device, void* dst_dev, const void* src_host, size_t bytes);
int sl2_dev_download(int device, void* dst_host, const void* src_dev, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* SCENELIB2_AMD_H */
