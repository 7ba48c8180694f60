"""Host-side mirror of the reference's MonoSLAM / Kalman / Feature interface over the
C ABI (include/scenelib2_amd.h).

* `Engine` — the batched engine: B independent sequences stepped together.
* `MonoSLAM` — a single-sequence object with the reference's member names
  (scenelib2/monoslam.h:69-219): Init(cfg), GoOneStep(frame, save_trajectory,
  enable_mapping), xv_, Pxx_, feature_list_, selected_feature_list_,
  trajectory_store_, AddNewKnownFeature, print_robot_state ... so that tests read
  like the reference's own usage (examples/MonoSlamSceneLib1.cpp:132-142).

All numerical work happens in the HIP library; nothing here computes SLAM math.
"""
import ctypes as C

import os

import numpy as np

from . import _lib
from .config import load_config, read_pgm, resolve_identifier


class Engine:
    """B independent MonoSLAM instances on one GPU (one HIP stream)."""

    def __init__(self, cam, params, batch, max_features, device=0, stream=None, lib=None):
        self.L = lib or _lib.load()          # lib = _lib.load_testing(): the TEST build (kernel variants, hooks)
        self.cam = dict(cam)
        self.params = dict(params)
        self.batch = int(batch)
        self.max_features = int(max_features)
        self.device = int(device)
        self.frame_bytes = int(cam["width"]) * int(cam["height"])
        self._cam = _lib.make_camera(cam)
        self._prm = _lib.make_params(params)
        h = _lib.vp()
        self._ck(self.L.sl2_create(C.byref(self._cam), C.byref(self._prm), self.batch, self.max_features,
                                     self.device, _lib.vp(stream) if stream else None, C.byref(h)))
        self.h = h

    def _ck(self, rc):
        _lib.check(rc, self.L)

    def close(self):
        if getattr(self, "h", None):
            self.L.sl2_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -------------------------------------------------------------
    def set_vehicle_state(self, xv, Pxx, seq0=0):
        xv = np.ascontiguousarray(xv, dtype=np.float64).reshape(-1, 13)
        Pxx = np.ascontiguousarray(Pxx, dtype=np.float64).reshape(-1, 13, 13)
        assert xv.shape[0] == Pxx.shape[0]
        self._ck(self.L.sl2_set_vehicle_state(self.h, seq0, xv.shape[0], _lib.dp(xv), _lib.dp(Pxx)))

    def get_vehicle_state(self, seq0=0, nseq=None):
        nseq = self.batch - seq0 if nseq is None else nseq
        xv = np.zeros((nseq, 13))
        Pxx = np.zeros((nseq, 13, 13))
        self._ck(self.L.sl2_get_vehicle_state(self.h, seq0, nseq, _lib.dp(xv), _lib.dp(Pxx)))
        return xv, Pxx

    def add_known_features(self, y, xp_org, patches, seq0=0):
        """y [nseq][nfeat][3], xp_org [nseq][nfeat][7], patches [nseq][nfeat][11][11] uint8."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        nseq, nfeat = y.shape[0], y.shape[1]
        xp = np.ascontiguousarray(xp_org, dtype=np.float64).reshape(nseq, nfeat, 7)
        p = np.ascontiguousarray(patches, dtype=np.uint8).reshape(nseq, nfeat, 121)
        self._ck(self.L.sl2_add_known_features(self.h, seq0, nseq, nfeat, _lib.dp(y), _lib.dp(xp), _lib.u8p(p)))

    def set_feature_covariances(self, Pyy, seq0=0):
        """Pyy [nseq][nfeat][3][3]: prior covariance of the first nfeat features of each sequence."""
        P = np.ascontiguousarray(Pyy, dtype=np.float64)
        nseq, nfeat = P.shape[0], P.shape[1]
        self._ck(self.L.sl2_set_feature_covariances(self.h, seq0, nseq, nfeat, _lib.dp(P.reshape(nseq, nfeat, 9))))

    # ---- stepping ----------------------------------------------------------
    def _frames_arg(self, frames, seq_stride, on_device):
        if on_device:
            return _lib.vp(int(frames)), int(seq_stride if seq_stride else self.frame_bytes), 1, None
        f = np.ascontiguousarray(frames, dtype=np.uint8).reshape(self.batch, self.frame_bytes)
        return f.ctypes.data_as(_lib.vp), self.frame_bytes, 0, f

    def go_one_step(self, frames, save_trajectory=False, enable_mapping=False, on_device=False, seq_stride=0):
        ptr, stride, dev, keep = self._frames_arg(frames, seq_stride, on_device)
        self._ck(self.L.sl2_go_one_step(self.h, ptr, stride, dev, int(save_trajectory), int(enable_mapping)))
        if keep is not None:
            self.synchronize()  # host buffer must outlive the async H2D copy

    def set_groups(self, groups):
        self._ck(self.L.sl2_set_groups(self.h, int(groups)))

    @property
    def stream(self):
        """The hipStream_t (as an integer) the engine's steps are queued on: what Ingest.next wants."""
        return self.L.sl2_get_stream(self.h)

    def set_step_fusion(self, enabled=True):
        """Small maps step in three launches (default) / always one stage per launch (sl2_set_step_fusion)."""
        self._ck(self.L.sl2_set_step_fusion(self.h, int(enabled)))

    def set_graph_mode(self, enabled=True):
        self._ck(self.L.sl2_set_graph_mode(self.h, int(bool(enabled))))

    def set_search_variant(self, variant):
        self._ck(self.L.sl2_set_search_variant(self.h, int(variant)))

    def set_search_split(self, min_bands):
        """Windows of at least `min_bands` 32 x 16 bands are shared out over the wavefronts of the search launch (0 = never)."""
        self._ck(self.L.sl2_set_search_split(self.h, int(min_bands)))

    def set_update_variant(self, chol_variant=1, fwd_variant=1):
        self._ck(self.L.sl2_set_update_variant(self.h, int(chol_variant), int(fwd_variant)))

    def kalman_filter_predict(self):
        self._ck(self.L.sl2_kalman_filter_predict(self.h))

    def auto_select_n_features(self, n):
        self._ck(self.L.sl2_auto_select_n_features(self.h, int(n)))

    def make_measurements(self, frames, on_device=False, seq_stride=0):
        ptr, stride, dev, keep = self._frames_arg(frames, seq_stride, on_device)
        self._ck(self.L.sl2_make_measurements(self.h, ptr, stride, dev))
        if keep is not None:
            self.synchronize()

    def kalman_filter_update(self):
        self._ck(self.L.sl2_kalman_filter_update(self.h))

    def finish_step(self, save_trajectory=False):
        self._ck(self.L.sl2_finish_step(self.h, int(save_trajectory)))

    def synchronize(self):
        self._ck(self.L.sl2_synchronize(self.h))

    # ---- state access ------------------------------------------------------
    def total_state_sizes(self, seq0=0, nseq=None):
        nseq = self.batch - seq0 if nseq is None else nseq
        out = np.zeros(nseq, dtype=np.int32)
        self._ck(self.L.sl2_get_total_state_sizes(self.h, seq0, nseq, _lib.ip(out)))
        return out

    def total_state(self, seq):
        n = int(self.total_state_sizes(seq, 1)[0])
        x = np.zeros(n)
        self._ck(self.L.sl2_get_total_state(self.h, seq, _lib.dp(x), n))
        return x

    def total_covariance(self, seq):
        n = int(self.total_state_sizes(seq, 1)[0])
        P = np.zeros((n, n))
        self._ck(self.L.sl2_get_total_covariance(self.h, seq, _lib.dp(P), n))
        return P

    def snapshot(self, seq, traj_cursor=0, patch_from_label=0):
        """sl2_snapshot: everything the reference exposes as public members of one sequence, in ONE call (one kernel, one
        synchronisation).  Returns a dict of numpy arrays / python values decoded from the blob."""
        blob, nbytes = _lib.vp(), C.c_size_t(0)
        self._ck(self.L.sl2_snapshot(self.h, int(seq), int(traj_cursor), int(patch_from_label), C.byref(blob), C.byref(nbytes)))
        raw = C.string_at(blob.value, nbytes.value)      # the engine's pinned buffer is reused by the next call: copy out
        h = _lib.sl2_snapshot_header.from_buffer_copy(raw[:256])
        assert h.magic == 0x53324C53 and h.bytes == nbytes.value
        f64 = lambda off, n: np.frombuffer(raw, dtype=np.float64, count=n, offset=off).copy()
        out = dict(header=h, xv=f64(h.off_xv, 13), Pxx=f64(h.off_Pxx, 169).reshape(13, 13), features=[], partial=[], patches={})
        fsz = C.sizeof(_lib.sl2_feature_info)
        cov = h.off_cov
        for i in range(h.n_features):
            f = _lib.sl2_feature_info.from_buffer_copy(raw[h.off_features + i * fsz: h.off_features + (i + 1) * fsz])
            d = f.state_size
            Pxy = f64(cov, 13 * d).reshape(13, d)
            Pyy = f64(cov + 13 * d * 8, d * d).reshape(d, d)
            cov += (13 * d + d * d) * 8
            out["features"].append(dict(info=f, label=f.label, Pxy=Pxy, Pyy=Pyy))
        out["selection"] = np.frombuffer(raw, dtype=np.int32, count=h.n_selected, offset=h.off_selection).copy()
        out["trajectory"] = f64(h.off_traj, 3 * h.traj_count).reshape(-1, 3)
        off = h.off_partial
        for _ in range(h.n_partial):
            pi = _lib.sl2_partial_info.from_buffer_copy(raw[off: off + 32])
            parts = f64(off + 32, 12 * pi.n_particles).reshape(-1, 12)
            out["partial"].append(dict(info=pi, particles=parts))
            off += 32 + 96 * pi.n_particles
        for k in range(h.n_patches):
            o = h.off_patches + 128 * k
            lab = int(np.frombuffer(raw, dtype=np.int32, count=1, offset=o)[0])
            out["patches"][lab] = np.frombuffer(raw, dtype=np.uint8, count=121, offset=o + 4).reshape(11, 11).copy()
        return out

    def features(self, seq, include_deleted=False):
        arr = (_lib.sl2_feature_info * self.max_features)()
        cnt = C.c_int(0)
        self._ck(self.L.sl2_get_features(self.h, seq, arr, self.max_features, int(include_deleted), C.byref(cnt)))
        out = []
        for i in range(cnt.value):
            f = arr[i]
            out.append(dict(label=f.label, active=bool(f.active), selected=bool(f.selected_flag),
                            success=bool(f.successful_measurement_flag),
                            attempted=f.attempted_measurements_of_feature,
                            successful=f.successful_measurements_of_feature,
                            pos=f.position_in_total_state_vector, visible=bool(f.visible),
                            y=np.array(f.y[:]), h=np.array(f.h[:]), z=np.array(f.z[:]), nu=np.array(f.nu[:]),
                            R=float(f.R), S=np.array(f.S[:]).reshape(2, 2),
                            dh_by_dxp=np.array(f.dh_by_dxp[:]).reshape(2, 7),
                            dh_by_dy=np.array(f.dh_by_dy[:]).reshape(2, 3), xp_org=np.array(f.xp_org[:]),
                            fully_initialised=bool(f.fully_initialised_flag), state_size=int(f.state_size),
                            y_direction=np.array(f.y_direction[:])))
        return out

    def partial_feature(self, seq, index=0, capacity=1024):
        """Entry `index` of feature_init_info_vector_ of a sequence (FeatureInitInfo + particles), or None; plus the mapping
        counters of the sequence under key 'info' either way (info['n_partial'] = the vector's size)."""
        ints = np.zeros(16, dtype=np.int32)
        dbl = np.zeros(9)
        parts = np.zeros((capacity, 12))
        self._ck(self.L.sl2_get_partial_feature(self.h, seq, int(index), _lib.ip(ints), _lib.dp(dbl), _lib.dp(parts), capacity))
        info = dict(n_partial=int(ints[0]), initialised=int(ints[12]), converted=int(ints[13]), deleted=int(ints[14]),
                    uu=int(ints[5]), vv=int(ints[6]), region_defined=int(ints[7]), ustart=int(ints[8]), vstart=int(ints[9]),
                    ufinish=int(ints[10]), vfinish=int(ints[11]), created=int(ints[15]), evbest=float(dbl[8]))
        if index >= ints[0]:
            return dict(info=info, pf=None)
        n = int(ints[3])
        return dict(info=info, pf=dict(label=int(ints[1]), n_particles=n, attempts=int(ints[2]), making=bool(ints[4]),
                                       mean=float(dbl[0]), covariance=float(dbl[1]), y=dbl[2:8].copy(),
                                       particles=parts[:n].copy()))

    def partial_features(self, seq, capacity=1024):
        """All entries of feature_init_info_vector_, in the vector's order."""
        first = self.partial_feature(seq, 0, capacity)
        out = [first["pf"]] if first["pf"] is not None else []
        for k in range(1, first["info"]["n_partial"]):
            out.append(self.partial_feature(seq, k, capacity)["pf"])
        return out

    def selection(self, seq):
        labels = np.zeros(self.max_features, dtype=np.int32)
        counters = np.zeros(3, dtype=np.int32)
        self._ck(self.L.sl2_get_selection(self.h, seq, _lib.ip(labels), self.max_features, _lib.ip(counters)))
        return labels[:counters[1]].copy(), dict(visible=int(counters[0]), selected=int(counters[1]),
                                                 measurement_size=int(counters[2]))

    def trajectory(self, seq, capacity=1000):
        out = np.zeros((capacity, 3))
        cnt = C.c_int(0)
        self._ck(self.L.sl2_get_trajectory(self.h, seq, _lib.dp(out), capacity, C.byref(cnt)))
        return out[:cnt.value].copy()

    def position_log(self, seq0=0, nseq=None, capacity=1000):
        """xv[0:3] after each of the last steps: [nseq][count][3]."""
        nseq = self.batch - seq0 if nseq is None else nseq
        out = np.zeros((nseq, capacity, 3))
        cnt = C.c_int(0)
        self._ck(self.L.sl2_get_position_log(self.h, seq0, nseq, _lib.dp(out), capacity, C.byref(cnt)))
        return out.reshape(-1)[: nseq * cnt.value * 3].reshape(nseq, cnt.value, 3).copy()

    def set_feature_counters(self, seq, label, attempted, successful):
        # test hook: lives in the TEST build of the library only (include/scenelib2_amd_testing.h)
        T = _lib.load_testing()
        _lib.check(T.sl2_set_feature_counters(self.h, seq, label, attempted, successful), T)

    def debug_set_position_error(self, seq, label, err):
        # test hook (TEST build only): Q28's error of the recorded position_in_total_state_vector_, written directly
        T = _lib.load_testing()
        _lib.check(T.sl2_debug_set_position_error(self.h, seq, label, err), T)

    def delete_features(self, labels, seq0=0):
        """mark_feature_by_lab + delete_feature, one label per sequence (-1: none); returns the per-sequence bool."""
        lab = np.ascontiguousarray(labels, dtype=np.int32)
        done = np.zeros(lab.size, dtype=np.int32)
        self._ck(self.L.sl2_delete_features(self.h, int(seq0), lab.size, _lib.ip(lab), _lib.ip(done)))
        return done.astype(bool)

    def initialise_feature(self, frames, uv, on_device=False, seq_stride=0):
        """MonoSLAM::InitialiseFeature at (uu_, vv_) = uv[s] for every sequence (u < 0: skip); returns created [batch]."""
        ptr, stride, dev, keep = self._frames_arg(frames, seq_stride, on_device)
        sel = np.ascontiguousarray(uv, dtype=np.int32).reshape(self.batch, 2)
        created = np.zeros(self.batch, dtype=np.int32)
        self._ck(self.L.sl2_initialise_feature(self.h, ptr, stride, dev, _lib.ip(sel), _lib.ip(created)))
        return created.astype(bool)

    def initialise_auto_feature(self, frames, on_device=False, seq_stride=0):
        """MonoSLAM::InitialiseAutoFeature for every sequence; returns created [batch]."""
        ptr, stride, dev, keep = self._frames_arg(frames, seq_stride, on_device)
        created = np.zeros(self.batch, dtype=np.int32)
        self._ck(self.L.sl2_initialise_auto_feature(self.h, ptr, stride, dev, _lib.ip(created)))
        return created.astype(bool)

    def save_patch(self, seq, label, path):
        """MonoSLAM::SavePatch: Feature::patch_ of the feature with this label as PNG (or PGM by extension)."""
        self._ck(self.L.sl2_save_patch(self.h, int(seq), int(label), os.fsencode(path)))

    def feature_patch(self, seq, label):
        """Feature::patch_ of the feature with this label (11x11 uint8)."""
        out = np.zeros((11, 11), dtype=np.uint8)
        self._ck(self.L.sl2_get_feature_patch(self.h, int(seq), int(label), _lib.u8p(out)))
        return out

    def status_flags(self):
        out = np.zeros(self.batch, dtype=np.int32)
        self._ck(self.L.sl2_get_status_flags(self.h, 0, self.batch, _lib.ip(out)))
        return out

    # ---- profiling -----------------------------------------------------------
    def set_profile_focus(self, names=""):
        self._ck(self.L.sl2_set_profile_focus(self.h, names.encode() if names else None))

    def set_profiling(self, level):
        """0 off, 1 roofline kernels only, 2 every launch."""
        self._ck(self.L.sl2_set_profiling(self.h, int(level)))

    def reset_kernel_times(self):
        self._ck(self.L.sl2_reset_kernel_times(self.h))

    def kernel_times(self):
        n = self.L.sl2_kernel_count(self.h)
        out = {}
        for i in range(n):
            name = C.c_char_p()
            ms = C.c_double(0)
            cnt = C.c_int64(0)
            self._ck(self.L.sl2_get_kernel_time(self.h, i, C.byref(name), C.byref(ms), C.byref(cnt)))
            out[name.value.decode()] = dict(total_ms=ms.value, launches=cnt.value)
        return out

    def placement(self):
        """sl2_get_placement: how sl2_create placed the large matrices (candidates probed, probe ms of the kept P, V^T, A^T, S and
        of the slowest candidate of each size); zeros for an engine too small for it to matter."""
        w = np.zeros(10)
        self._ck(self.L.sl2_get_placement(self.h, _lib.dp(w), w.size))
        keys = ["candidates_of_P", "kept_P_ms", "kept_Vt_ms", "kept_At_ms", "kept_S_ms", "slowest_P_ms", "slowest_A_ms", "slowest_S_ms",
                "k_syrk_on_kept_pair_ms", "k_syrk_on_slowest_pair_ms"]
        return dict(zip(keys, w.tolist()))

    def step_work(self):
        w = np.zeros(13)
        self._ck(self.L.sl2_get_step_work(self.h, _lib.dp(w), w.size))
        keys = ["window_bytes", "searched", "candidates", "sum_m", "sum_m2", "sum_m3", "sum_n", "sum_nm", "sum_nnm",
                "sum_nmm", "search_fallbacks", "search_tiles", "search_shared"]
        return dict(zip(keys, w.tolist()))


class Feature:
    """Read-only view with the member names of SceneLib2::Feature (feature.h:78-142)."""

    def __init__(self, d, patch=None):
        self.label_ = d["label"]
        self.y_ = d["y"]
        self.xp_org_ = d["xp_org"]
        self.h_ = d["h"]
        self.z_ = d["z"]
        self.nu_ = d["nu"]
        self.S_ = d["S"]
        self.R_ = np.eye(2) * d["R"]
        dh = np.zeros((2, 13))
        dh[:, :7] = d["dh_by_dxp"]
        self.dh_by_dxv_ = dh
        self.dh_by_dy_ = d["dh_by_dy"]
        self.selected_flag_ = d["selected"]
        self.successful_measurement_flag_ = d["success"]
        self.attempted_measurements_of_feature_ = d["attempted"]
        self.successful_measurements_of_feature_ = d["successful"]
        self.position_in_total_state_vector_ = d["pos"]
        self.fully_initialised_flag_ = d.get("fully_initialised", True)
        self.patch_ = patch


class MonoSLAM:
    """Single-sequence MonoSLAM with the reference's public surface (monoslam.h:69-219)."""

    kBoxSize_ = 11
    kNoSigma_ = 3.0
    kCorrThresh2_ = 0.40
    kCorrelationSigmaThreshold_ = 10.0

    def __init__(self, max_features=64, device=0):
        self._max_features = max_features
        self._device = device
        self._engine = None
        self._patches = {}
        self.camera_ = None
        self.uu_, self.vv_ = 0, 0                   # image selection (set by the GUI's mouse handler in the reference)
        self.location_selected_flag_ = False
        self.marked_feature_label_ = -1             # monoslam.h:171

    # MonoSLAM::Init(config_path) — monoslam.cpp:1574-1969
    def Init(self, config_path, template_dirs=()):
        cfg = load_config(config_path)
        self.InitFromValues(cfg["cam"], cfg["params"], cfg["xv"], cfg["Pxx"])
        for f in cfg["features"]:
            self.AddNewKnownFeature(f["y"], f["xp_org"], resolve_identifier(f, template_dirs))
        return self

    def InitFromValues(self, cam, params, xv, Pxx):
        self.camera_ = dict(cam)
        self.kDeltaT_ = params["delta_t"]
        self.kNumberOfFeaturesToSelect_ = params["number_of_features_to_select"]
        self._engine = Engine(cam, params, 1, self._max_features, self._device)
        self._engine.set_vehicle_state(np.asarray(xv).reshape(1, 13), np.asarray(Pxx).reshape(1, 13, 13))
        return self

    # MonoSLAM::AddNewKnownFeature(y, xp, identifier) — monoslam.cpp:1278-1291
    def AddNewKnownFeature(self, y, xp, identifier):
        patch = read_pgm(identifier) if isinstance(identifier, str) else np.asarray(identifier, dtype=np.uint8)
        if patch.shape != (11, 11):
            raise ValueError("template must be 11x11")
        label = len(self._patches)
        self._engine.add_known_features(np.asarray(y).reshape(1, 1, 3), np.asarray(xp).reshape(1, 1, 7),
                                        patch.reshape(1, 1, 11, 11))
        self._patches[label] = patch

    # MonoSLAM::GoOneStep(frame, save_trajectory, enable_mapping) — monoslam.cpp:108-180
    def GoOneStep(self, frame, save_trajectory=False, enable_mapping=False):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        if f.size != self._engine.frame_bytes:
            raise ValueError("frame must be %d x %d 8-bit single channel" % (self.camera_["width"], self.camera_["height"]))
        self._engine.go_one_step(f.reshape(1, -1), save_trajectory, enable_mapping)
        # the reference's feature_list_ is unbounded; the engine holds at most max_features LIVE features per sequence
        # (deleted ones give their slots back): a full map must not pass silently (SL2_STATUS_LABELS_EXHAUSTED)
        if enable_mapping:
            full = bool(int(self._engine.status_flags()[0]) & 2)
            rose, self._map_full = full and not getattr(self, "_map_full", False), full
            if rose:          # (once per episode: the step itself has been applied; the bit clears when a slot frees up)
                raise RuntimeError("MonoSLAM: all %d feature slots hold live features (max_features); mapping cannot "
                                   "initialise features until one is deleted - create the engine with a larger max_features"
                                   % self._max_features)
        return True  # the reference always returns true (monoslam.cpp:179)

    def _frame(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.uint8)
        if f.size != self._engine.frame_bytes:
            raise ValueError("frame must be %d x %d 8-bit single channel" % (self.camera_["width"], self.camera_["height"]))
        return f.reshape(1, -1)

    # MonoSLAM::InitialiseFeature(frame) at (uu_, vv_) — monoslam.cpp:1211-1235
    def InitialiseFeature(self, frame):
        return bool(self._engine.initialise_feature(self._frame(frame), [[self.uu_, self.vv_]])[0])

    # MonoSLAM::InitialiseAutoFeature(frame) — monoslam.cpp:1535-1541
    def InitialiseAutoFeature(self, frame):
        created = bool(self._engine.initialise_auto_feature(self._frame(frame))[0])
        info = self._engine.partial_feature(0)["info"]
        if created or info["region_defined"]:   # (a refused call - a partially initialised feature is still in flight: no region
            self.uu_, self.vv_ = info["uu"], info["vv"]          # was searched - leaves the selection alone)
            self.location_selected_flag_ = True
        return created

    # MonoSLAM::mark_feature_by_lab — monoslam.cpp:743-768
    def mark_feature_by_lab(self, lab):
        if lab == -1 or any(f.label_ == lab for f in self.feature_list_):
            self.marked_feature_label_ = lab

    # MonoSLAM::delete_feature — monoslam.cpp:770-812
    def delete_feature(self):
        if self.marked_feature_label_ == -1:
            return False
        done = bool(self._engine.delete_features([self.marked_feature_label_])[0])
        if done:
            self.marked_feature_label_ = -1
        return done

    # MonoSLAM::SavePatch — monoslam.cpp:1551-1572 (writes "patch.png" in the working directory)
    def SavePatch(self, path="patch.png"):
        if self.marked_feature_label_ == -1 or not any(f.label_ == self.marked_feature_label_ for f in self.feature_list_):
            return False
        self._engine.save_patch(0, self.marked_feature_label_, path)
        return True

    @property
    def xv_(self):
        return self._engine.get_vehicle_state(0, 1)[0][0]

    @property
    def Pxx_(self):
        return self._engine.get_vehicle_state(0, 1)[1][0]

    @property
    def feature_list_(self):
        return [Feature(d, self._patches.get(d["label"])) for d in self._engine.features(0)]

    @property
    def selected_feature_list_(self):
        labels, _ = self._engine.selection(0)
        by_label = {f.label_: f for f in self.feature_list_}
        return [by_label[l] for l in labels if l in by_label]

    @property
    def number_of_visible_features_(self):
        return self._engine.selection(0)[1]["visible"]

    @property
    def successful_measurement_vector_size_(self):
        return self._engine.selection(0)[1]["measurement_size"]

    @property
    def total_state_size_(self):
        return int(self._engine.total_state_sizes(0, 1)[0])

    @property
    def trajectory_store_(self):
        return self._engine.trajectory(0)

    def construct_total_state(self):
        return self._engine.total_state(0)

    def construct_total_covariance(self):
        return self._engine.total_covariance(0)

    # MonoSLAM::print_robot_state — monoslam.cpp:1543-1549
    def print_robot_state(self):
        print("[Robot state]")
        print(self.xv_)
        print("[Robot covariance]")
        print(self.Pxx_)
