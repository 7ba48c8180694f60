/* scenelib2_amd_testing.h - TEST HOOKS AND MICRO-BENCHMARKS.  NOT part of the product ABI.
 *
 * These entry points exist only in scenelib2_amd/libscenelib2_amd_test.so (the same sources built with -DSL2_TESTING);
 * the product library libscenelib2_amd.so does not export them (tests/test_capi_symbols.py checks both).  The test
 * library is a full build: a hook that takes an engine works on an engine created by either library (same structures,
 * one shared HIP runtime).
 */
#ifndef SCENELIB2_AMD_TESTING_H
#define SCENELIB2_AMD_TESTING_H
#include "scenelib2_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Feature::attempted_/successful_measurements_of_feature_ (feature.h:95-96) written directly: lets a test put a feature
 * in front of delete_bad_features without ten frames of failed matches.  Synchronises. */
int sl2_set_feature_counters(sl2_engine* e, int seq, int label, int attempted, int successful);

/* The error of a feature's recorded position_in_total_state_vector_ (how far it lies below the true one: a multiple of three,
 * what feature.cpp:254 - Q28 - accumulates over conversions), written directly.  Lets a test drive the dh_by_dy block of H into
 * the vehicle state (monoslam.cpp:562-565: it OVERWRITES dh_by_dxv there) and below column 0
 * (SL2_STATUS_REFERENCE_OUT_OF_BOUNDS) without the dozen conversions that would take.  Synchronises. */
int sl2_debug_set_position_error(sl2_engine* e, int seq, int label, int err);

/* FP64 epilogue of correlate2_warning (improc.cpp:99-133) evaluated ON THE DEVICE
 * for `count` tuples of the five integer sums: checks IEEE div/sqrt parity. */
int sl2_debug_ncc_score(int device, const int32_t* sums5, int count, double* score, double* sd0, double* sd1);
/* C[M][N] = sum_k XT[k][m] * YT[k][n] on the FP64 MFMA tile path used by the EKF
 * kernels (k-major operands): XT [K][ldx], YT [K][ldy], C [M][ldc].  Host pointers. */
int sl2_debug_gemm_kt(int device, const double* XT, int ldx, const double* YT, int ldy, int M, int N, int K,
                      double* C, int ldc);

/* Micro-benchmarks that calibrate the roofline peaks on the box: which = 0 FP64 MFMA
 * TFLOP/s (4 independent accumulators), 1 = dependent chain, 2 = streaming copy GB/s. */
int sl2_debug_microbench(int device, int which, double* result);

#ifdef __cplusplus
}
#endif
#endif /* SCENELIB2_AMD_TESTING_H */
