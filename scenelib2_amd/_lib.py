"""ctypes binding of the C ABI declared in include/scenelib2_amd.h.

The shared library is built in-tree (scenelib2_amd/libscenelib2_amd.so) by
`__graft_entry__.build()` / `make -C scenelib2_amd/csrc`.  There is NO CPU
fallback: if the library is missing, or no HIP device is visible, calls fail.

If the host process also uses PyTorch-ROCm, import torch BEFORE this module so
both share one HIP runtime (torch bundles libamdhip64.so under the same SONAME).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SL2_LIB_PATH") or os.path.join(_HERE, "libscenelib2_amd.so")   # override: development builds only

SL2_OK = 0
SL2_ERR_INVALID = 1
SL2_ERR_HIP = 2
SL2_ERR_CAPACITY = 3
SL2_ERR_NO_DEVICE = 4


class Sl2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("scenelib2_amd error %d: %s" % (code, msg))
        self.code = code


class sl2_camera(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fku", C.c_double), ("fkv", C.c_double),
                ("u0", C.c_double), ("v0", C.c_double), ("kd1", C.c_double), ("sd", C.c_int32)]


class sl2_params(C.Structure):
    _fields_ = [("delta_t", C.c_double),
                ("number_of_features_to_select", C.c_int32),
                ("number_of_features_to_keep_visible", C.c_int32),
                ("max_features_to_init_at_once", C.c_int32),
                ("min_lambda", C.c_double), ("max_lambda", C.c_double),
                ("number_of_particles", C.c_int32),
                ("standard_deviation_depth_ratio", C.c_double),
                ("min_number_of_particles", C.c_int32),
                ("prune_probability_threshold", C.c_double),
                ("erase_partially_init_feature_after_this_many_attempts", C.c_int32),
                ("minimum_attempted_measurements_of_feature", C.c_int32),
                ("successful_match_fraction", C.c_double)]


class sl2_feature_info(C.Structure):
    _fields_ = [("label", C.c_int32), ("active", C.c_int32), ("selected_flag", C.c_int32),
                ("successful_measurement_flag", C.c_int32),
                ("attempted_measurements_of_feature", C.c_int32),
                ("successful_measurements_of_feature", C.c_int32),
                ("position_in_total_state_vector", C.c_int32), ("visible", C.c_int32),
                ("y", C.c_double * 3), ("h", C.c_double * 2), ("z", C.c_double * 2), ("nu", C.c_double * 2),
                ("R", C.c_double), ("S", C.c_double * 4), ("dh_by_dxp", C.c_double * 14),
                ("dh_by_dy", C.c_double * 6), ("xp_org", C.c_double * 7),
                ("fully_initialised_flag", C.c_int32), ("state_size", C.c_int32), ("y_direction", C.c_double * 3)]


class sl2_snapshot_header(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "magic", "api_version", "bytes", "seq", "n_features", "total_state_size", "number_of_visible_features", "n_selected",
        "successful_measurement_vector_size", "next_free_label", "status_flags", "traj_total", "traj_first", "traj_count",
        "n_partial", "n_patches", "uu", "vv", "location_selected_flag", "init_feature_search_region_defined_flag")] + \
        [("init_feature_search_region", C.c_int32 * 4)] + \
        [(n, C.c_int32) for n in ("off_xv", "off_Pxx", "off_features", "off_cov", "off_selection", "off_traj", "off_partial",
                                  "off_patches", "steps_done")] + [("reserved", C.c_int32 * 31)]


class sl2_partial_info(C.Structure):
    _fields_ = [("label", C.c_int32), ("number_of_match_attempts", C.c_int32), ("n_particles", C.c_int32),
                ("making_measurement_on_this_step_flag", C.c_int32), ("mean", C.c_double), ("covariance", C.c_double)]


assert C.sizeof(sl2_snapshot_header) == 256 and C.sizeof(sl2_partial_info) == 32

# every symbol include/scenelib2_amd.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "sl2_api_version", "sl2_device_count", "sl2_create", "sl2_destroy", "sl2_last_error", "sl2_synchronize", "sl2_batch",
    "sl2_max_features", "sl2_set_vehicle_state", "sl2_get_vehicle_state", "sl2_add_known_features", "sl2_set_feature_covariances",
    "sl2_go_one_step", "sl2_initialise_feature", "sl2_initialise_auto_feature", "sl2_save_patch", "sl2_set_groups", "sl2_set_search_variant", "sl2_set_step_fusion", "sl2_get_stream", "sl2_set_search_split", "sl2_set_update_variant", "sl2_set_graph_mode", "sl2_kalman_filter_predict", "sl2_auto_select_n_features", "sl2_make_measurements",
    "sl2_kalman_filter_update", "sl2_finish_step", "sl2_elliptical_search_batch", "sl2_find_best_patch_batch",
    "sl2_search_multiple_overlapping_ellipses_batch", "sl2_list_frames", "sl2_read_pgm", "sl2_read_image", "sl2_ingest_open",
    "sl2_ingest_frame_count", "sl2_ingest_next", "sl2_ingest_set_zero_copy", "sl2_ingest_close", "sl2_get_total_state_sizes",
    "sl2_get_total_state", "sl2_get_total_covariance", "sl2_get_features", "sl2_get_partial_feature", "sl2_get_selection",
    "sl2_snapshot_capacity", "sl2_snapshot", "sl2_get_trajectory", "sl2_get_feature_patch", "sl2_get_position_log", "sl2_delete_features", "sl2_get_status_flags", "sl2_set_profiling", "sl2_set_profile_focus",
    "sl2_reset_kernel_times", "sl2_kernel_count", "sl2_get_kernel_time", "sl2_get_step_work", "sl2_get_placement",
    "sl2_synth_render_host", "sl2_synth_render_device", "sl2_dev_malloc", "sl2_dev_free", "sl2_dev_upload",
    "sl2_dev_download",
]
# test hooks and micro-benchmarks (include/scenelib2_amd_testing.h): exported by libscenelib2_amd_test.so ONLY
TEST_SYMBOLS = ["sl2_set_feature_counters", "sl2_debug_set_position_error", "sl2_debug_ncc_score", "sl2_debug_gemm_kt", "sl2_debug_microbench"]
TEST_LIB_PATH = os.path.join(_HERE, "libscenelib2_amd_test.so")
_testlib = None

_lib = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


def dp(a):
    return a.ctypes.data_as(c_dp)


def ip(a):
    return a.ctypes.data_as(c_ip)


def u8p(a):
    return a.ctypes.data_as(c_u8p)


def _bind(L):
    """argtypes / restypes of every entry point of include/scenelib2_amd.h."""
    L.sl2_last_error.restype = C.c_char_p
    L.sl2_device_count.restype = C.c_int
    L.sl2_create.argtypes = [C.POINTER(sl2_camera), C.POINTER(sl2_params), C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    L.sl2_destroy.argtypes = [vp]
    L.sl2_destroy.restype = None
    L.sl2_synchronize.argtypes = [vp]
    L.sl2_batch.argtypes = [vp]
    L.sl2_max_features.argtypes = [vp]
    L.sl2_set_vehicle_state.argtypes = [vp, C.c_int, C.c_int, c_dp, c_dp]
    L.sl2_get_vehicle_state.argtypes = [vp, C.c_int, C.c_int, c_dp, c_dp]
    L.sl2_add_known_features.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_u8p]
    L.sl2_set_feature_covariances.argtypes = [vp, C.c_int, C.c_int, C.c_int, c_dp]
    L.sl2_go_one_step.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int]
    L.sl2_set_groups.argtypes = [vp, C.c_int]
    L.sl2_set_search_variant.argtypes = [vp, C.c_int]
    if "SL2_LIB_PATH" not in os.environ or hasattr(L, "sl2_set_search_split"):      # (an older build under test, scripts/ab_libs.sh)
        L.sl2_set_search_split.argtypes = [vp, C.c_int]
    L.sl2_set_update_variant.argtypes = [vp, C.c_int, C.c_int]
    L.sl2_set_graph_mode.argtypes = [vp, C.c_int]
    L.sl2_set_step_fusion.argtypes = [vp, C.c_int]
    L.sl2_get_stream.argtypes = [vp]
    L.sl2_get_stream.restype = vp
    L.sl2_kalman_filter_predict.argtypes = [vp]
    L.sl2_auto_select_n_features.argtypes = [vp, C.c_int]
    L.sl2_make_measurements.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.sl2_kalman_filter_update.argtypes = [vp]
    L.sl2_finish_step.argtypes = [vp, C.c_int]
    L.sl2_elliptical_search_batch.argtypes = [C.c_int, c_u8p, C.c_int, C.c_int, C.c_int, c_ip, c_u8p, c_dp, c_dp,
                                              C.c_int, c_ip, c_ip, c_dp, C.c_int]
    L.sl2_find_best_patch_batch.argtypes = [C.c_int, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, c_ip, c_ip, c_ip, c_dp, c_dp]
    L.sl2_search_multiple_overlapping_ellipses_batch.argtypes = [C.c_int, c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, c_ip, c_u8p,
                                                                 c_ip, c_dp, c_dp, c_ip, c_dp, c_dp]
    L.sl2_get_partial_feature.argtypes = [vp, C.c_int, C.c_int, c_ip, c_dp, c_dp, C.c_int]
    L.sl2_list_frames.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t, c_ip]
    L.sl2_read_pgm.argtypes = [C.c_char_p, c_u8p, C.c_size_t, c_ip, c_ip]
    L.sl2_read_image.argtypes = [C.c_char_p, c_u8p, C.c_size_t, c_ip, c_ip]
    L.sl2_ingest_open.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.sl2_ingest_frame_count.argtypes = [vp]
    L.sl2_ingest_next.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.sl2_ingest_set_zero_copy.argtypes = [vp, C.c_size_t]
    L.sl2_ingest_close.argtypes = [vp]
    L.sl2_ingest_close.restype = None
    L.sl2_get_total_state_sizes.argtypes = [vp, C.c_int, C.c_int, c_ip]
    L.sl2_get_total_state.argtypes = [vp, C.c_int, c_dp, C.c_int]
    L.sl2_get_total_covariance.argtypes = [vp, C.c_int, c_dp, C.c_int]
    L.sl2_get_features.argtypes = [vp, C.c_int, C.POINTER(sl2_feature_info), C.c_int, C.c_int, C.POINTER(C.c_int)]
    L.sl2_get_selection.argtypes = [vp, C.c_int, c_ip, C.c_int, c_ip]
    L.sl2_get_trajectory.argtypes = [vp, C.c_int, c_dp, C.c_int, C.POINTER(C.c_int)]
    L.sl2_get_feature_patch.argtypes = [vp, C.c_int, C.c_int, c_u8p]
    L.sl2_initialise_feature.argtypes = [vp, vp, C.c_size_t, C.c_int, c_ip, c_ip]
    L.sl2_initialise_auto_feature.argtypes = [vp, vp, C.c_size_t, C.c_int, c_ip]
    L.sl2_save_patch.argtypes = [vp, C.c_int, C.c_int, C.c_char_p]
    L.sl2_delete_features.argtypes = [vp, C.c_int, C.c_int, c_ip, c_ip]
    L.sl2_get_position_log.argtypes = [vp, C.c_int, C.c_int, c_dp, C.c_int, C.POINTER(C.c_int)]
    L.sl2_get_status_flags.argtypes = [vp, C.c_int, C.c_int, c_ip]
    L.sl2_set_profiling.argtypes = [vp, C.c_int]
    L.sl2_set_profile_focus.argtypes = [vp, C.c_char_p]
    L.sl2_reset_kernel_times.argtypes = [vp]
    L.sl2_kernel_count.argtypes = [vp]
    L.sl2_get_kernel_time.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), c_dp, C.POINTER(C.c_int64)]
    L.sl2_get_step_work.argtypes = [vp, c_dp, C.c_int]
    L.sl2_get_placement.argtypes = [vp, c_dp, C.c_int]
    L.sl2_api_version.restype = C.c_int
    L.sl2_snapshot_capacity.argtypes = [vp]
    L.sl2_snapshot_capacity.restype = C.c_size_t
    L.sl2_snapshot.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.sl2_synth_render_host.argtypes = [C.POINTER(sl2_camera), c_u8p, C.c_int, C.c_double, c_dp, c_dp, C.c_int, c_u8p]
    L.sl2_synth_render_device.argtypes = [C.c_int, vp, C.POINTER(sl2_camera), vp, C.c_int, C.c_double, vp, vp,
                                          C.c_int, vp]
    L.sl2_dev_malloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp)]
    L.sl2_dev_free.argtypes = [C.c_int, vp]
    L.sl2_dev_upload.argtypes = [C.c_int, vp, vp, C.c_size_t]
    L.sl2_dev_download.argtypes = [C.c_int, vp, vp, C.c_size_t]
    return L


def load_testing():
    """The TEST build of the library (same sources + -DSL2_TESTING): every entry point of the product library plus the
    hooks of include/scenelib2_amd_testing.h and the superseded / experimental kernel variants (sl2_set_update_variant,
    sl2_set_search_variant beyond the defaults, the SL2_* environment switches).  An Engine may be created on it
    (Engine(..., lib=load_testing())); a hook that takes an engine also accepts engines created through load() (same
    structures, one HIP runtime)."""
    global _testlib
    if _testlib is not None:
        return _testlib
    load()      # the product library first: it pins the HIP runtime both share
    if not os.path.exists(TEST_LIB_PATH):
        raise ImportError("scenelib2_amd: %s not built (make -C scenelib2_amd/csrc)" % TEST_LIB_PATH)
    T = _bind(C.CDLL(TEST_LIB_PATH))
    T.sl2_set_feature_counters.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int]
    T.sl2_debug_set_position_error.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    T.sl2_debug_ncc_score.argtypes = [C.c_int, c_ip, C.c_int, c_dp, c_dp, c_dp]
    T.sl2_debug_gemm_kt.argtypes = [C.c_int, c_dp, C.c_int, c_dp, C.c_int, C.c_int, C.c_int, C.c_int, c_dp, C.c_int]
    T.sl2_debug_microbench.argtypes = [C.c_int, C.c_int, c_dp]
    _testlib = T
    return T


def load():
    """Load the native library (fails loudly if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("scenelib2_amd: native library %s not built (run __graft_entry__.build() or "
                          "`make -C scenelib2_amd/csrc`); there is no CPU fallback" % LIB_PATH)
    _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def check(rc, L=None):
    """Raise on a non-zero status; L = the library whose call returned rc (its sl2_last_error holds the message)."""
    if rc != SL2_OK:
        raise Sl2Error(rc, (L or load()).sl2_last_error().decode("utf-8", "replace"))


def device_count():
    return load().sl2_device_count()


def make_camera(cam):
    return sl2_camera(int(cam["width"]), int(cam["height"]), float(cam["fku"]), float(cam["fkv"]),
                      float(cam["u0"]), float(cam["v0"]), float(cam["kd1"]), int(cam["sd"]))


def make_params(p):
    d = dict(delta_t=0.0, number_of_features_to_select=10, number_of_features_to_keep_visible=12,
             max_features_to_init_at_once=1, min_lambda=0.5, max_lambda=5.0, number_of_particles=100,
             standard_deviation_depth_ratio=0.3, min_number_of_particles=20, prune_probability_threshold=0.05,
             erase_partially_init_feature_after_this_many_attempts=10,
             minimum_attempted_measurements_of_feature=10, successful_match_fraction=0.5)
    d.update({k: v for k, v in p.items() if k in d})
    return sl2_params(float(d["delta_t"]), int(d["number_of_features_to_select"]),
                      int(d["number_of_features_to_keep_visible"]), int(d["max_features_to_init_at_once"]),
                      float(d["min_lambda"]), float(d["max_lambda"]), int(d["number_of_particles"]),
                      float(d["standard_deviation_depth_ratio"]), int(d["min_number_of_particles"]),
                      float(d["prune_probability_threshold"]),
                      int(d["erase_partially_init_feature_after_this_many_attempts"]),
                      int(d["minimum_attempted_measurements_of_feature"]), float(d["successful_match_fraction"]))


class DeviceBuffer:
    """A raw HIP allocation owned by Python (used by tests/bench when torch is not)."""

    def __init__(self, nbytes, device=0):
        self.device = device
        self.nbytes = int(nbytes)
        p = vp()
        check(load().sl2_dev_malloc(device, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        check(load().sl2_dev_upload(self.device, vp(self.ptr + offset), a.ctypes.data_as(vp), a.nbytes))

    def download(self, shape, dtype, offset=0):
        out = np.empty(shape, dtype=dtype)
        assert offset + out.nbytes <= self.nbytes
        check(load().sl2_dev_download(self.device, out.ctypes.data_as(vp), vp(self.ptr + offset), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load().sl2_dev_free(self.device, vp(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
