"""Regenerates the committed golden fixtures (run from the repo root: python tests/golden/make_golden.py).

* synth_seq5_sha256.txt   — SHA-256 of three rendered synthetic frames (pins the input generator).
* oracle_shipped.npz      — the ORACLE's outputs on the reference's shipped scene (cfg values +
                            known_patch*.pgm) for three GoOneStep calls on a deterministic frame.
                            The reference itself cannot be run here (no Eigen/OpenCV/Pangolin), so this
                            pins the oracle against regressions; it is not a reference output.
* oracle_mapping.npz      — the ORACLE's event log of a 40-frame mapping run (tests/mapping_helpers.py, seed 7): per frame
                            the counters (partial features, initialised, converted, deleted), the selected pixel, the
                            total state size and the camera position; final total state and covariance.  The frames come
                            from the synthetic generator (pinned by its own checksum); same status as above.
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


def main():
    import oracle_api as oa
    from scenelib2_amd import synth
    from scenelib2_amd.config import load_config, read_pgm
    tex = synth.make_texture(size=512)
    cam = synth.default_camera()
    _, _, frames, _ = synth.make_sequence(cam, 24, 3, seq_index=5, tex=tex)
    open(os.path.join(HERE, "synth_seq5_sha256.txt"), "w").write(hashlib.sha256(frames.tobytes()).hexdigest() + "\n")
    cfg = load_config(os.path.join(HERE, "scenelib2_shipped.cfg"))
    patches = [read_pgm(os.path.join(HERE, "known_patch%d.pgm" % i)) for i in range(4)]
    frame = shipped_scene_frame(oa, cfg, patches)
    o = oa.OracleSLAM(cfg["cam"], cfg["params"]["delta_t"], 10)
    o.set_state(cfg["xv"], cfg["Pxx"])
    for f, p in zip(cfg["features"], patches):
        o.add_known_feature(f["y"], f["xp_org"], p)
    xs, Ps, zs = [], [], []
    for _ in range(3):
        o.go_one_step(frame, True)
        xs.append(o.total_state())
        Ps.append(o.total_covariance())
        zs.append(np.array([o.feature(i)["z"] for i in range(o.num_features)]))
    np.savez_compressed(os.path.join(HERE, "oracle_shipped.npz"), frame=frame, x=np.array(xs), P=np.array(Ps), z=np.array(zs))
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
    print("golden fixtures written")


if __name__ == "__main__":
    main()
