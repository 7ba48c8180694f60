"""Shared set-up for the parity tests: the same synthetic sequences fed to the CPU
oracle (one OracleSLAM per sequence) and to the HIP engine (one batch)."""
import numpy as np

import oracle_api as oa
from scenelib2_amd import Engine, synth

_TEX = {}


def texture(size=synth.TEX_SIZE):
    if size not in _TEX:
        _TEX[size] = synth.make_texture(size=size)
    return _TEX[size]


class Pair:
    """B synthetic sequences: frames, oracle instances and one engine."""

    def __init__(self, n_features, n_frames, batch=1, cam=None, n_select=None, seq0=0, max_features=None,
                 feature_counts=None, make_engine=True, feature_sigma=0.0, lib=None, **spec_kw):
        """The checker is the CPU restatement (oracle/liboracle.so)."""
        self.cam = cam or synth.default_camera()
        self.N = n_features
        self.B = batch
        self.n_frames = n_frames
        n_select = n_features if n_select is None else n_select
        self.params = synth.default_params(n_select)
        tex = texture()
        self.specs, self.templates, self.frames = [], [], []
        for b in range(batch):
            nf = n_features if feature_counts is None else feature_counts[b]
            spec, tpl, frames, _ = synth.make_sequence(self.cam, max(nf, 1), n_frames, seq_index=seq0 + b, tex=tex, **spec_kw)
            if nf == 0:
                spec.n_features = 0
                spec.feat_y = spec.feat_y[:0]
                tpl = tpl[:0]
            self.specs.append(spec)
            self.templates.append(tpl)
            self.frames.append(frames)
        self.oracles = []
        for b in range(batch):
            s = oa.OracleSLAM(self.cam, self.params["delta_t"], n_select)
            s.set_state(self.specs[b].xv0, self.specs[b].Pxx0)
            xo = self.specs[b].poses[0]
            for i in range(self.specs[b].feat_y.shape[0]):
                s.add_known_feature(self.specs[b].feat_y[i], xo, self.templates[b][i])
            if feature_sigma > 0.0:
                for i in range(self.specs[b].feat_y.shape[0]):
                    s.set_feature_Pyy(i, np.eye(3) * feature_sigma ** 2)
            self.oracles.append(s)
        self.engine = None
        if make_engine:
            # lib = _lib.load_testing(): the TEST build of the library (superseded kernel variants, SL2_* switches)
            self.engine = Engine(self.cam, self.params, batch, max_features or max(n_features, 1), lib=lib)
            self.engine.set_vehicle_state(np.stack([s.xv0 for s in self.specs]), np.stack([s.Pxx0 for s in self.specs]))
            for b in range(batch):
                nf = self.specs[b].feat_y.shape[0]
                if nf:
                    self.engine.add_known_features(self.specs[b].feat_y[None], np.tile(self.specs[b].poses[0], (1, nf, 1)),
                                                   self.templates[b][None], seq0=b)
                    if feature_sigma > 0.0:
                        self.engine.set_feature_covariances(np.tile(np.eye(3) * feature_sigma ** 2, (1, nf, 1, 1)), seq0=b)

    def frame_batch(self, k):
        return np.stack([f[k] for f in self.frames])

    def step_both(self, k, save_trajectory=False, threads=1):
        """One GoOneStep of every oracle and of the engine.  threads > 1: the oracles (independent objects, ctypes calls that
        release the GIL) step on a thread pool - the 1513-state shapes cost 0.8 s per sequence-frame on one core."""
        if threads > 1 and self.B > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(threads, self.B)) as ex:
                list(ex.map(lambda b: self.oracles[b].go_one_step(self.frames[b][k], save_trajectory), range(self.B)))
        else:
            for b in range(self.B):
                self.oracles[b].go_one_step(self.frames[b][k], save_trajectory)
        self.engine.go_one_step(self.frame_batch(k), save_trajectory)

    def compare_state(self, tol_x=1e-9, tol_P=1e-9, exact_z=True):
        """Returns worst deviations; asserts structure equality."""
        worst = dict(x=0.0, P=0.0)
        for b in range(self.B):
            o = self.oracles[b]
            n = o.total_state_size
            assert int(self.engine.total_state_sizes(b, 1)[0]) == n, "state size differs (seq %d)" % b
            xo, xe = o.total_state(), self.engine.total_state(b)
            Po, Pe = o.total_covariance(), self.engine.total_covariance(b)
            dx = np.abs(xo - xe).max() if n else 0.0
            dP = np.linalg.norm(Po - Pe) / max(np.linalg.norm(Po), 1e-300)
            worst["x"] = max(worst["x"], float(dx))
            worst["P"] = max(worst["P"], float(dP))
            assert dx <= tol_x, "state differs by %g (seq %d)" % (dx, b)
            assert dP <= tol_P, "covariance rel-Frobenius differs by %g (seq %d)" % (dP, b)
            feats = self.engine.features(b)
            assert len(feats) == o.num_features
            sel, counters = self.engine.selection(b)
            assert counters["visible"] == o.num_visible
            assert list(sel) == list(o.selected_labels()), "selection order differs (seq %d)" % b
            assert counters["measurement_size"] == (o.measurement_size if o.num_selected else counters["measurement_size"])
            for i, fe in enumerate(feats):
                fo = o.feature(i)
                assert fe["label"] == fo["label"]
                assert fe["attempted"] == fo["attempted"] and fe["successful"] == fo["successful"], \
                    "counters differ (seq %d feature %d)" % (b, fe["label"])
                assert fe["selected"] == fo["selected"]
                if fe["selected"]:
                    assert fe["success"] == fo["success"]
                    if exact_z and fo["success"]:
                        assert np.array_equal(fe["z"], fo["z"]), "measurement differs (seq %d feature %d): %s vs %s" % (
                            b, fe["label"], fe["z"], fo["z"])
        return worst
