"""ctypes binding for the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)
c_u8p = C.POINTER(C.c_uint8)


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _ip(a):
    return a.ctypes.data_as(c_ip)


def _u8(a):
    return a.ctypes.data_as(c_u8p)


def build():
    subprocess.check_call(["make", "-s", "-C", _ORACLE_DIR])


def _declare(L):
    """argtypes / restypes of the flat C interface (oracle/slam_oracle.cpp)."""
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                             C.c_double, C.c_int, C.c_double, C.c_int]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_set_state.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_get_state.argtypes = [C.c_void_p, c_dp, c_dp]
    L.orc_add_known_feature.argtypes = [C.c_void_p, c_dp, c_dp, c_u8p]
    L.orc_go_one_step.argtypes = [C.c_void_p, c_u8p, C.c_int, C.c_int]
    for f in ("orc_num_features", "orc_num_selected", "orc_total_state_size", "orc_num_visible",
              "orc_measurement_size"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = C.c_int
    L.orc_get_total_state.argtypes = [C.c_void_p, c_dp]
    L.orc_get_total_covariance.argtypes = [C.c_void_p, c_dp]
    L.orc_get_feature.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp]
    L.orc_get_selected_labels.argtypes = [C.c_void_p, c_ip]
    L.orc_trajectory.argtypes = [C.c_void_p, c_dp, C.c_int]
    L.orc_trajectory.restype = C.c_int
    L.orc_kalman_filter_predict.argtypes = [C.c_void_p]
    L.orc_auto_select_n_features.argtypes = [C.c_void_p, C.c_int]
    L.orc_auto_select_n_features.restype = C.c_int
    L.orc_make_measurements.argtypes = [C.c_void_p, c_u8p]
    L.orc_make_measurements.restype = C.c_int
    L.orc_kalman_filter_update.argtypes = [C.c_void_p]
    L.orc_normalise_state.argtypes = [C.c_void_p]
    L.orc_delete_bad_features.argtypes = [C.c_void_p]
    L.orc_delete_feature.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_feature_Pyy.argtypes = [C.c_void_p, C.c_int, c_dp]
    L.orc_set_feature_counters.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_set_feature_position.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_correlate2_warning.restype = C.c_double
    L.orc_correlate2_warning.argtypes = [C.c_int] * 6 + [c_u8p, C.c_int, c_u8p, C.c_int, c_dp, c_dp]
    L.orc_elliptical_search.restype = C.c_int
    L.orc_elliptical_search.argtypes = [c_u8p, C.c_int, C.c_int, c_u8p, c_dp, C.c_double, C.c_double,
                                        C.c_double, c_ip, c_dp]
    L.orc_sinv_from_S.argtypes = [c_dp, c_dp]
    L.orc_set_mapping_params.argtypes = [C.c_void_p, c_ip, c_dp]
    L.orc_get_mapping_info.argtypes = [C.c_void_p, c_ip]
    L.orc_get_partial_feature.argtypes = [C.c_void_p, C.c_int, c_ip, c_dp, c_dp, C.c_int]
    L.orc_get_partial_feature.restype = C.c_int
    L.orc_get_feature_kinds.argtypes = [C.c_void_p, c_ip]
    L.orc_get_feature_patch.argtypes = [C.c_void_p, C.c_int, c_u8p]
    L.orc_find_best_patch.argtypes = [c_u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ip, c_dp]
    L.orc_search_multiple_ellipses.restype = C.c_longlong
    L.orc_search_multiple_ellipses.argtypes = [c_u8p, C.c_int, C.c_int, c_u8p, C.c_int, c_dp, c_dp, c_ip, c_dp]
    L.orc_drand48_sequence.argtypes = [C.c_long, C.c_int, c_dp]
    L.orc_motion_model.argtypes = [c_dp, C.c_double, c_dp, c_dp, c_dp]
    L.orc_dqnorm_by_dq.argtypes = [c_dp, c_dp]
    L.orc_measurement_model.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp]
    L.orc_initialise_feature.argtypes = [C.c_void_p, c_u8p, C.c_int, C.c_int]
    L.orc_initialise_auto_feature.argtypes = [C.c_void_p, c_u8p]
    L.orc_dense_llt.restype = C.c_int
    L.orc_dense_llt.argtypes = [C.c_int, c_dp, c_dp]
    L.orc_dense_inverse.argtypes = [C.c_int, c_dp, c_dp]
    L.orc_dense_mul.argtypes = [C.c_int, C.c_int, C.c_int, c_dp, c_dp, c_dp]
    L.orc_quat_ops.argtypes = [c_dp, c_dp, c_dp, c_dp, c_dp]
    L.orc_run_sequences.restype = C.c_double
    L.orc_run_sequences.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(c_u8p), C.c_int, C.c_size_t,
                                    C.c_int, c_dp]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_ORACLE_DIR, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        _declare(L)
        L.orc_get_diag.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), c_dp]
        _LIB = L
    return _LIB


class OracleSLAM:
    """One MonoSLAM instance of the oracle (single sequence, like the reference)."""

    def __init__(self, cam, delta_t, n_select, L=None):
        self.L = L if L is not None else lib()
        self.cam = dict(cam)
        self.h = self.L.orc_create(cam["width"], cam["height"], cam["fku"], cam["fkv"], cam["u0"], cam["v0"],
                                   cam["kd1"], cam["sd"], delta_t, n_select)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def set_state(self, xv, Pxx):
        xv = np.ascontiguousarray(xv, dtype=np.float64)
        Pxx = np.ascontiguousarray(Pxx, dtype=np.float64)
        self.L.orc_set_state(self.h, _dp(xv), _dp(Pxx))

    def get_state(self):
        xv = np.zeros(13)
        Pxx = np.zeros((13, 13))
        self.L.orc_get_state(self.h, _dp(xv), _dp(Pxx))
        return xv, Pxx

    def add_known_feature(self, y, xp_org, patch):
        y = np.ascontiguousarray(y, dtype=np.float64)
        xp = np.ascontiguousarray(xp_org, dtype=np.float64)
        p = np.ascontiguousarray(patch, dtype=np.uint8).reshape(121)
        self.L.orc_add_known_feature(self.h, _dp(y), _dp(xp), _u8(p))

    def go_one_step(self, frame, save_trajectory=False, enable_mapping=False):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        return self.L.orc_go_one_step(self.h, _u8(f), int(save_trajectory), int(enable_mapping))

    @property
    def num_features(self):
        return self.L.orc_num_features(self.h)

    @property
    def num_selected(self):
        return self.L.orc_num_selected(self.h)

    @property
    def num_visible(self):
        return self.L.orc_num_visible(self.h)

    @property
    def measurement_size(self):
        return self.L.orc_measurement_size(self.h)

    @property
    def total_state_size(self):
        return self.L.orc_total_state_size(self.h)

    def total_state(self):
        x = np.zeros(self.total_state_size)
        self.L.orc_get_total_state(self.h, _dp(x))
        return x

    def total_covariance(self):
        n = self.total_state_size
        P = np.zeros((n, n))
        self.L.orc_get_total_covariance(self.h, _dp(P))
        return P

    def feature(self, idx):
        ints = np.zeros(6, dtype=np.int32)
        d = np.zeros(46)
        self.L.orc_get_feature(self.h, idx, _ip(ints), _dp(d))
        return dict(label=int(ints[0]), selected=bool(ints[1]), success=bool(ints[2]), attempted=int(ints[3]),
                    successful=int(ints[4]), pos=int(ints[5]), y=d[0:3].copy(), h=d[3:5].copy(), z=d[5:7].copy(),
                    nu=d[7:9].copy(), R=float(d[9]), S=d[10:14].reshape(2, 2).copy(),
                    dh_by_dxv=d[14:40].reshape(2, 13).copy(), dh_by_dy=d[40:46].reshape(2, 3).copy())

    def selected_labels(self):
        n = self.num_selected
        a = np.zeros(max(n, 1), dtype=np.int32)
        self.L.orc_get_selected_labels(self.h, _ip(a))
        return a[:n].copy()

    def trajectory(self, max_entries=1000):
        out = np.zeros((max_entries, 3))
        n = self.L.orc_trajectory(self.h, _dp(out), max_entries)
        return out[:n].copy()

    def diag(self):
        cand = C.c_longlong(0)
        wb = C.c_longlong(0)
        t = np.zeros(5)
        self.L.orc_get_diag(self.h, C.byref(cand), C.byref(wb), _dp(t))
        return dict(candidates=cand.value, window_bytes=wb.value,
                    times=dict(predict=t[0], select=t[1], search=t[2], update=t[3], rest=t[4]))

    # seams
    def kalman_filter_predict(self):
        self.L.orc_kalman_filter_predict(self.h)

    def auto_select_n_features(self, n):
        return self.L.orc_auto_select_n_features(self.h, n)

    def make_measurements(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        return self.L.orc_make_measurements(self.h, _u8(f))

    def kalman_filter_update(self):
        self.L.orc_kalman_filter_update(self.h)

    def normalise_state(self):
        self.L.orc_normalise_state(self.h)

    def delete_bad_features(self):
        self.L.orc_delete_bad_features(self.h)

    def delete_feature(self, label):
        return bool(self.L.orc_delete_feature(self.h, int(label)))

    def set_feature_Pyy(self, idx, Pyy):
        P = np.ascontiguousarray(Pyy, dtype=np.float64).reshape(9)
        self.L.orc_set_feature_Pyy(self.h, idx, _dp(P))

    def set_mapping_params(self, params):
        ip = np.array([params["number_of_features_to_keep_visible"], params["max_features_to_init_at_once"],
                       params["number_of_particles"], params["min_number_of_particles"],
                       params["erase_partially_init_feature_after_this_many_attempts"]], dtype=np.int32)
        dp = np.array([params["min_lambda"], params["max_lambda"], params["standard_deviation_depth_ratio"],
                       params["prune_probability_threshold"]], dtype=np.float64)
        self.L.orc_set_mapping_params(self.h, _ip(ip), _dp(dp))

    def mapping_info(self):
        o = np.zeros(12, dtype=np.int32)
        self.L.orc_get_mapping_info(self.h, _ip(o))
        keys = ["n_partial", "initialised", "converted", "deleted", "uu", "vv", "location_selected", "region_defined",
                "ustart", "vstart", "ufinish", "vfinish"]
        return dict(zip(keys, [int(x) for x in o]))

    def partial_feature(self, k, max_particles=256):
        ints = np.zeros(4, dtype=np.int32)
        dbl = np.zeros(8)
        parts = np.zeros((max_particles, 12))
        if not self.L.orc_get_partial_feature(self.h, k, _ip(ints), _dp(dbl), _dp(parts), max_particles):
            return None
        n = int(ints[1])
        return dict(label=int(ints[0]), n_particles=n, attempts=int(ints[2]), making=bool(ints[3]), mean=dbl[0],
                    covariance=dbl[1], y=dbl[2:8].copy(), particles=parts[:n].copy())

    def feature_kinds(self):
        n = self.num_features
        o = np.zeros((max(n, 1), 3), dtype=np.int32)
        self.L.orc_get_feature_kinds(self.h, _ip(o))
        return o[:n]

    def feature_patch(self, idx):
        p = np.zeros(121, dtype=np.uint8)
        self.L.orc_get_feature_patch(self.h, idx, _u8(p))
        return p.reshape(11, 11)

    def set_feature_counters(self, idx, attempted, successful):
        self.L.orc_set_feature_counters(self.h, idx, attempted, successful)

    def set_feature_position(self, idx, pos):
        """position_in_total_state_vector_ of feature idx, as recorded (Q28 test hook)."""
        self.L.orc_set_feature_position(self.h, idx, int(pos))

    # MonoSLAM::InitialiseFeature at (uu_, vv_) = (u, v) / InitialiseAutoFeature (monoslam.cpp:1211-1235, 1535-1541)
    def initialise_feature(self, frame, u, v):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        self.L.orc_initialise_feature(self.h, _u8(f), int(u), int(v))

    def initialise_auto_feature(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        self.L.orc_initialise_auto_feature(self.h, _u8(f))


def correlate2_warning(patch, image, x1, y1, x0=0, y0=0, x0lim=11, y0lim=11, L=None):
    L = L if L is not None else lib()
    p0 = np.ascontiguousarray(patch, dtype=np.uint8)
    p1 = np.ascontiguousarray(image, dtype=np.uint8)
    sd0 = C.c_double(0)
    sd1 = C.c_double(0)
    c = L.orc_correlate2_warning(x0, y0, x0lim, y0lim, x1, y1, _u8(p0), p0.shape[1], _u8(p1), p1.shape[1],
                                 C.byref(sd0), C.byref(sd1))
    return c, sd0.value, sd1.value


def elliptical_search(image, patch, centre, a, b, c, L=None):
    """Returns dict(ok,u,v,ncand,hw,hh,corr) — monoslam.cpp:401-477."""
    L = L if L is not None else lib()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    p = np.ascontiguousarray(patch, dtype=np.uint8).reshape(121)
    ce = np.ascontiguousarray(centre, dtype=np.float64)
    oi = np.zeros(5, dtype=np.int32)
    corr = C.c_double(0)
    ok = L.orc_elliptical_search(_u8(img), img.shape[1], img.shape[0], _u8(p), _dp(ce), a, b, c, _ip(oi),
                                 C.byref(corr))
    return dict(ok=bool(ok), u=int(oi[0]), v=int(oi[1]), ncand=int(oi[2]), hw=int(oi[3]), hh=int(oi[4]),
                corr=corr.value)


def find_best_patch(image, region, uv_in=(-1, -1), L=None):
    """monoslam.cpp:1070-1192.  region = (ustart, vstart, ufinish, vfinish).  Returns (u, v, evbest); (u, v) keep
    uv_in when no position scores."""
    L = L if L is not None else lib()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    uv = np.array(uv_in, dtype=np.int32)
    ev = C.c_double(0)
    L.orc_find_best_patch(_u8(img), img.shape[1], img.shape[0], int(region[0]), int(region[1]), int(region[2]),
                          int(region[3]), _ip(uv), C.byref(ev))
    return int(uv[0]), int(uv[1]), ev.value


def search_multiple_ellipses(image, patch, puinv, centre, L=None):
    """SearchMultipleOverlappingEllipses::search over the given ellipses.  Returns (result [n][3] = flag, u, v;
    corrmax [n]; number of positions correlated)."""
    L = L if L is not None else lib()
    img = np.ascontiguousarray(image, dtype=np.uint8)
    p = np.ascontiguousarray(patch, dtype=np.uint8).reshape(121)
    pu = np.ascontiguousarray(puinv, dtype=np.float64).reshape(-1, 3)
    ce = np.ascontiguousarray(centre, dtype=np.float64).reshape(-1, 2)
    n = pu.shape[0]
    out = np.zeros((n, 3), dtype=np.int32)
    corr = np.zeros(n)
    ncorr = L.orc_search_multiple_ellipses(_u8(img), img.shape[1], img.shape[0], _u8(p), n, _dp(pu), _dp(ce), _ip(out),
                                           _dp(corr))
    return out, corr, int(ncorr)


def drand48_sequence(seed, n, L=None):
    out = np.zeros(n)
    (L if L is not None else lib()).orc_drand48_sequence(int(seed), int(n), _dp(out))
    return out


def sinv_from_S(S, L=None):
    L = L if L is not None else lib()
    S4 = np.ascontiguousarray(S, dtype=np.float64).reshape(4)
    abc = np.zeros(3)
    L.orc_sinv_from_S(_dp(S4), _dp(abc))
    return abc


def motion_model(xv, dt, L=None):
    L = L if L is not None else lib()
    xv = np.ascontiguousarray(xv, dtype=np.float64)
    f = np.zeros(13)
    F = np.zeros((13, 13))
    Q = np.zeros((13, 13))
    L.orc_motion_model(_dp(xv), dt, _dp(f), _dp(F), _dp(Q))
    return f, F, Q


def dqnorm_by_dq(q, L=None):
    L = L if L is not None else lib()
    q = np.ascontiguousarray(q, dtype=np.float64)
    J = np.zeros((4, 4))
    L.orc_dqnorm_by_dq(_dp(q), _dp(J))
    return J


def cam8(cam):
    return np.array([cam["width"], cam["height"], cam["fku"], cam["fkv"], cam["u0"], cam["v0"], cam["kd1"],
                     cam["sd"]], dtype=np.float64)


def measurement_model(cam, xp, y, xp_org=None, L=None):
    L = L if L is not None else lib()
    xp = np.ascontiguousarray(xp, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    xo = xp if xp_org is None else np.ascontiguousarray(xp_org, dtype=np.float64)
    out = np.zeros(24)
    c8 = cam8(cam)
    L.orc_measurement_model(_dp(c8), _dp(xp), _dp(y), _dp(xo), _dp(out))
    return dict(h=out[0:2].copy(), dh_by_dxp=out[2:16].reshape(2, 7).copy(), dh_by_dy=out[16:22].reshape(2, 3).copy(),
                R=float(out[22]), vis=int(out[23]))


def run_sequences(slams, frames_list, nthreads=1, want_traj=True, L=None):
    """frames_list[s]: uint8 array [nframes][H][W]. Returns (seconds, traj[nseq][nframes][3])."""
    L = L if L is not None else lib()
    nseq = len(slams)
    nframes = frames_list[0].shape[0]
    fb = int(np.prod(frames_list[0].shape[1:]))
    hs = (C.c_void_p * nseq)(*[s.h for s in slams])
    keep = [np.ascontiguousarray(f, dtype=np.uint8) for f in frames_list]
    fr = (c_u8p * nseq)(*[_u8(f) for f in keep])
    traj = np.zeros((nseq, nframes, 3)) if want_traj else None
    secs = L.orc_run_sequences(hs, nseq, fr, nframes, fb, nthreads, _dp(traj) if want_traj else None)
    return secs, traj


def dense_llt(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    out = np.zeros_like(A)
    ok = lib().orc_dense_llt(A.shape[0], _dp(A), _dp(out))
    return bool(ok), out


def dense_inverse(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    out = np.zeros_like(A)
    lib().orc_dense_inverse(A.shape[0], _dp(A), _dp(out))
    return out


def dense_mul(A, B):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    out = np.zeros((A.shape[0], B.shape[1]))
    lib().orc_dense_mul(A.shape[0], A.shape[1], B.shape[1], _dp(A), _dp(B), _dp(out))
    return out


def quat_ops(qa, qb):
    """(qa * qb, qa.inverse(), qa.toRotationMatrix()) with q = (w, x, y, z) - Eigen::Quaterniond semantics of oracle/dense.hpp."""
    qa = np.ascontiguousarray(qa, dtype=np.float64)
    qb = np.ascontiguousarray(qb, dtype=np.float64)
    prod, inv, R = np.zeros(4), np.zeros(4), np.zeros((3, 3))
    lib().orc_quat_ops(_dp(qa), _dp(qb), _dp(prod), _dp(inv), _dp(R))
    return prod, inv, R
