"""Regenerates the committed golden fixtures (run from the repo root, in the container that has /root/reference:
`python tests/golden/make_golden.py`).

The numeric fixtures are OUTPUTS OF THE REFERENCE ITSELF: oracle/_ref/libref.so is the reference's own translation units
compiled unmodified from /root/reference (`make -C oracle ref`, see oracle/ref_glue.cpp; Eigen / OpenCV / Pangolin are
stand-in headers under oracle/ref_shim, so "reference arithmetic" means the reference's code over the shim's fixed-order
products).  /root/reference does not exist on the GPU box, hence the committed vectors.

* synth_seq5_sha256.txt — SHA-256 of three rendered synthetic frames (pins the input generator).
* ref_shipped.npz   — MonoSLAM::Init on the shipped cfg (data/SceneLib2.cfg values, known_patch*.pgm) + three GoOneStep
                      calls on a deterministic frame: total state, total covariance, measurements per step.
* ref_mapping.npz   — 40-frame run with enable_mapping (tests/mapping_helpers.py, seed 7): per frame the camera position,
                      the selected pixel, partial-feature count and total state size from the reference; final total state
                      and covariance from the reference.  The three event counters (initialised / converted / deleted)
                      are bookkeeping the reference does not keep: they come from the oracle run on the same frames,
                      which this script first checks against the reference frame by frame.
* ref_seq100.npz    — 12 frames of a 100-feature sequence (n = 313, the BASELINE headline shape, 5 mm feature prior):
                      per frame xv, the measured pixels and match flags; final total state; of the final 313 x 313
                      covariance the vehicle block, the diagonal, the Frobenius norm and 256 sampled entries.
* ref_seq200.npz    — the same for 6 frames of a 640x480 / 200-feature sequence (n = 613, BASELINE configs[3]'s shape).
"""
import hashlib
import os
import sys
import tempfile

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


def shipped_cfg_with_absolute_identifiers(dst_dir):
    text = open(os.path.join(HERE, "scenelib2_shipped.cfg")).read()
    for i in range(4):   # identifiers are relative to the reference's working directory
        text = text.replace("= known_patch%d.pgm" % i, "= " + os.path.join(HERE, "known_patch%d.pgm" % i))
    path = os.path.join(dst_dir, "shipped_abs.cfg")
    open(path, "w").write(text)
    return path


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

    # ---- shipped scene through the reference's own Init
    cfg = load_config(os.path.join(HERE, "scenelib2_shipped.cfg"))
    patches = [read_pgm(os.path.join(HERE, "known_patch%d.pgm" % i)) for i in range(4)]
    frame = shipped_scene_frame(oa, cfg, patches)
    with tempfile.TemporaryDirectory() as td:
        r = oa.RefSLAM(cfg["cam"], cfg["params"]["delta_t"], 10, cfg_path=shipped_cfg_with_absolute_identifiers(td))
    xs, Ps, zs = [], [], []
    for _ in range(3):
        r.go_one_step(frame, True)
        xs.append(r.total_state())
        Ps.append(r.total_covariance())
        zs.append(np.array([r.feature(i)["z"] for i in range(r.num_features)]))
    np.savez_compressed(os.path.join(HERE, "ref_shipped.npz"), frame=frame, x=np.array(xs), P=np.array(Ps), z=np.array(zs))

    # ---- mapping run: reference numbers, oracle counters (cross-checked)
    from mapping_helpers import make_mapping_sequence, oracle_for
    cam_m, params_m, spec_m, frames_m, templates_m = make_mapping_sequence(n_frames=40)
    s = oracle_for(cam_m, params_m, spec_m, templates_m, oa)
    r = oa.RefSLAM(cam_m, params_m["delta_t"], params_m["number_of_features_to_select"])
    r.set_mapping_params(params_m)
    r.set_state(spec_m.xv0, spec_m.Pxx0)
    for i in range(spec_m.n_features):
        r.add_known_feature(spec_m.feat_y[i], spec_m.xp_org()[i], templates_m[i])
    events, pos = [], []
    for k in range(1, 41):
        s.go_one_step(frames_m[k], True, True)
        r.go_one_step(frames_m[k], True, True)
        info, iref = s.mapping_info(), r.mapping_info()
        assert info["n_partial"] == iref["n_partial"] and s.total_state_size == r.total_state_size, k
        assert np.array_equal(s.feature_kinds(), r.feature_kinds()), k
        if iref["location_selected"]:
            assert (info["uu"], info["vv"]) == (iref["uu"], iref["vv"]), k
        events.append([iref["n_partial"], info["initialised"], info["converted"], info["deleted"], info["uu"], info["vv"],
                       r.total_state_size])
        pos.append(r.get_state()[0][:3])
    np.savez_compressed(os.path.join(HERE, "ref_mapping.npz"), events=np.array(events, np.int32), pos=np.array(pos),
                        x=r.total_state(), P=r.total_covariance(), frames_sha256=hashlib.sha256(frames_m.tobytes()).hexdigest())

    # ---- the headline shape (n = 313) and the configs[3] shape (n = 613)
    for S, name in ((SEQ100, "ref_seq100.npz"), (SEQ200, "ref_seq200.npz")):
        cam, params, spec, tpl, frames = seq_inputs(S)
        N = S["n_features"]
        r = oa.RefSLAM(cam, params["delta_t"], N)
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
