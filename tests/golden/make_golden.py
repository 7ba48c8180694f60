"""Regenerates the committed regression fixtures (run from the repo root: `python tests/golden/make_golden.py`).

PROVENANCE - read this first.  The reference ships no golden vectors and cannot be built in this image
(Eigen3 / OpenCV / Pangolin are absent), so these are NOT reference outputs: they are outputs of the CPU oracle
(oracle/liboracle.so, a restatement of the reference's algorithm - "parity unpinned") on fixed
inputs.  They pin the oracle against accidental change and give the GPU tests committed numbers to compare with; they say
nothing about the reference beyond what the oracle's own anchors say (tests/test_oracle_kat.py, tests/test_oracle_numpy.py).
(Rounds 2-5 generated the same files from a build of the reference's sources over stand-in Eigen / OpenCV headers; that
build was retired in round 6 - it is not a reference build - and the files were regenerated from the oracle.  The two
generations are identical bit for bit in every array: the retired build's linear algebra was the builder's own
fixed-order loops, i.e. the same arithmetic as the oracle's - which is why it never was an independent pin.)

* synth_seq5_sha256.txt - SHA-256 of three rendered synthetic frames (pins the input generator).
* oracle_shipped.npz  - the shipped scene (tests/golden/scenelib2_shipped.cfg = the values of the reference's
                        data/SceneLib2.cfg, known_patch*.pgm) + three GoOneStep calls on a deterministic frame: total
                        state, total covariance, measurements per step.
* oracle_mapping.npz  - 40-frame run with enable_mapping (tests/mapping_helpers.py, seed 7): per frame the camera position,
                        the selected pixel, partial-feature count, event counters and total state size; final total state
                        and covariance.
* oracle_seq100.npz   - 12 frames of a 100-feature sequence (n = 313, the BASELINE headline shape, 5 mm feature prior):
                        per frame xv, the measured pixels and match flags; final total state; of the final 313 x 313
                        covariance the vehicle block, the diagonal, the Frobenius norm and 256 sampled entries.
* oracle_seq200.npz   - the same for 6 frames of a 640x480 / 200-feature sequence (n = 613, BASELINE configs[3]'s shape).
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HERE = os.path.dirname(os.path.abspath(__file__))


def shipped_scene_frame(oa, cfg, patches):
    probe = oa.OracleSLAM(cfg["cam"], cfg["params"]["delta_t"], 10)
    probe.set_state(cfg["xv"], cfg["Pxx"])
    for f, p in zip(cfg["features"], patches):
        probe.add_known_feature(f["y"], f["xp_org"], p)
    probe.kalman_filter_predict()
    probe.auto_select_n_features(10)
    rng = np.random.default_rng(7)
    frame = rng.integers(90, 110, (240, 320)).astype(np.uint8)
    for i, p in enumerate(patches):
        h = probe.feature(i)["h"]
        u, v = int(round(h[0])) + (i - 1), int(round(h[1])) + (2 - i)
        frame[v - 5:v + 6, u - 5:u + 6] = p
    return frame


SEQ100 = dict(n_features=100, n_frames=12, seq_index=3, feature_sigma=0.005, width=320, height=240)
SEQ200 = dict(n_features=200, n_frames=6, seq_index=11, feature_sigma=0.005, width=640, height=480)     # BASELINE configs[3] shape


def seq_inputs(S):
    from scenelib2_amd import synth
    cam = synth.default_camera(S["width"], S["height"])
    params = synth.default_params(S["n_features"])
    spec, tpl, frames, _ = synth.make_sequence(cam, S["n_features"], S["n_frames"], seq_index=S["seq_index"],
                                               tex=synth.make_texture())
    return cam, params, spec, tpl, frames


def seq100_inputs():
    return seq_inputs(SEQ100)


def seq100_sample_index(n=313, count=256):
    rng = np.random.default_rng(313)
    return rng.integers(0, n, count), rng.integers(0, n, count)


def main():
    import oracle_api as oa
    from scenelib2_amd import synth
    from scenelib2_amd.config import load_config, read_pgm
    tex = synth.make_texture(size=512)
    cam = synth.default_camera()
    _, _, frames, _ = synth.make_sequence(cam, 24, 3, seq_index=5, tex=tex)
    open(os.path.join(HERE, "synth_seq5_sha256.txt"), "w").write(hashlib.sha256(frames.tobytes()).hexdigest() + "\n")

    # ---- shipped scene
    cfg = load_config(os.path.join(HERE, "scenelib2_shipped.cfg"))
    patches = [read_pgm(os.path.join(HERE, "known_patch%d.pgm" % i)) for i in range(4)]
    frame = shipped_scene_frame(oa, cfg, patches)
    r = oa.OracleSLAM(cfg["cam"], cfg["params"]["delta_t"], 10)
    r.set_state(cfg["xv"], cfg["Pxx"])
    for f, p in zip(cfg["features"], patches):
        r.add_known_feature(f["y"], f["xp_org"], p)
    xs, Ps, zs = [], [], []
    for _ in range(3):
        r.go_one_step(frame, True)
        xs.append(r.total_state())
        Ps.append(r.total_covariance())
        zs.append(np.array([r.feature(i)["z"] for i in range(r.num_features)]))
    np.savez_compressed(os.path.join(HERE, "oracle_shipped.npz"), frame=frame, x=np.array(xs), P=np.array(Ps), z=np.array(zs))

    # ---- mapping run
    from mapping_helpers import make_mapping_sequence, oracle_for
    cam_m, params_m, spec_m, frames_m, templates_m = make_mapping_sequence(n_frames=40)
    s = oracle_for(cam_m, params_m, spec_m, templates_m, oa)
    events, pos = [], []
    for k in range(1, 41):
        s.go_one_step(frames_m[k], True, True)
        info = s.mapping_info()
        events.append([info["n_partial"], info["initialised"], info["converted"], info["deleted"], info["uu"], info["vv"],
                       s.total_state_size])
        pos.append(s.get_state()[0][:3])
    np.savez_compressed(os.path.join(HERE, "oracle_mapping.npz"), events=np.array(events, np.int32), pos=np.array(pos),
                        x=s.total_state(), P=s.total_covariance(), frames_sha256=hashlib.sha256(frames_m.tobytes()).hexdigest())

    # ---- the headline shape (n = 313) and the configs[3] shape (n = 613)
    for S, name in ((SEQ100, "oracle_seq100.npz"), (SEQ200, "oracle_seq200.npz")):
        cam, params, spec, tpl, frames = seq_inputs(S)
        N = S["n_features"]
        r = oa.OracleSLAM(cam, params["delta_t"], N)
        r.set_state(spec.xv0, spec.Pxx0)
        for i in range(N):
            r.add_known_feature(spec.feat_y[i], spec.poses[0], tpl[i])
        for i in range(N):
            r.set_feature_Pyy(i, np.eye(3) * S["feature_sigma"] ** 2)
        xv, z, ok = [], [], []
        for k in range(S["n_frames"]):
            r.go_one_step(frames[k], False)
            xv.append(r.get_state()[0])
            f = [r.feature(i) for i in range(N)]
            ok.append(np.array([q["selected"] and q["success"] for q in f]))
            z.append(np.array([q["z"] for q in f]))
        P = r.total_covariance()
        ii, jj = seq100_sample_index(P.shape[0])
        np.savez_compressed(os.path.join(HERE, name), xv=np.array(xv), z=np.array(z), ok=np.array(ok),
                            x=r.total_state(), Pxx=P[:13, :13], Pdiag=np.diag(P).copy(), Pfro=np.linalg.norm(P),
                            Psample=P[ii, jj], frames_sha256=hashlib.sha256(frames.tobytes()).hexdigest())
    print("golden fixtures written")


if __name__ == "__main__":
    main()
