"""The headless C++ example (examples/headless_monoslam.cpp) driven end to end on a GPU: cfg file +
PGM frame directory in, total state out; must equal the oracle fed the same bytes."""
import os
import subprocess

import numpy as np
import pytest

import oracle_api as oa
from mapping_helpers import make_mapping_sequence, oracle_for
from scenelib2_amd import ingest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write_scene(d, cam, params, spec, frames, templates):
    lines = ["cam.%s = %.17g;" % (k, cam[k]) for k in ("width", "height", "fku", "fkv", "u0", "v0", "kd1", "sd")]
    for k in ("delta_t", "number_of_features_to_select", "number_of_features_to_keep_visible", "max_features_to_init_at_once",
              "min_lambda", "max_lambda", "number_of_particles", "standard_deviation_depth_ratio", "min_number_of_particles",
              "prune_probability_threshold", "erase_partially_init_feature_after_this_many_attempts"):
        lines.append("params.%s = %.17g;" % (k, params[k]))
    names = ["rw_x", "rw_y", "rw_z", "qwr_w", "qwr_x", "qwr_y", "qwr_z", "vw_x", "vw_y", "vw_z", "ww_x", "ww_y", "ww_z"]
    lines += ["state.%s = %.17g;" % (n, v) for n, v in zip(names, spec.xv0)]
    for r in range(13):
        for c in range(13):
            if spec.Pxx0[r, c] != 0.0:
                lines.append("state.pxx%d_%d = %.17g;" % (r, c, spec.Pxx0[r, c]))
    xo = spec.xp_org()
    for i in range(spec.n_features):
        p = "f%d." % (i + 1)
        lines += ["%syi_%s = %.17g;" % (p, ax, spec.feat_y[i, j]) for j, ax in enumerate("xyz")]
        lines += ["%sxp_org_%d = %.17g;" % (p, j, xo[i, j]) for j in range(7)]
        lines.append("%sidentifier = patch%d.pgm;   # 11x11 template" % (p, i))
        ingest.write_pgm(os.path.join(d, "patch%d.pgm" % i), templates[i])
    with open(os.path.join(d, "scene.cfg"), "w") as f:
        f.write("# written by the test\n" + "\n".join(lines) + "\n")
    fd = os.path.join(d, "frames")
    os.makedirs(fd)
    for k in range(1, frames.shape[0]):
        ingest.write_pgm(os.path.join(fd, "%05d.pgm" % k), frames[k])
    return os.path.join(d, "scene.cfg"), fd


@pytest.mark.parametrize("mapping", [False, True])
def test_headless_example_matches_oracle(tmp_path, mapping):
    exe = os.path.join(ROOT, "examples", "headless_monoslam")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=20)
    cfg, fd = _write_scene(str(tmp_path), cam, params, spec, frames, templates)
    dump = os.path.join(str(tmp_path), "state.txt")
    cmd = [exe, "--cfg", cfg, "--frames", fd, "--dump", dump] + (["--mapping"] if mapping else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "%d known features, 20 frames" % spec.n_features in out.stdout
    x = np.loadtxt(dump)
    s = oracle_for(cam, params, spec, templates, oa)
    for k in range(1, 21):
        s.go_one_step(frames[k], True, mapping)
    x0 = s.total_state()
    assert x.size == x0.size and np.abs(x - x0).max() < 1e-9
    if mapping:
        assert s.mapping_info()["initialised"] >= 2


@pytest.mark.parametrize("mapping", [False, True, "seams"])
def test_monoslam_adapter_example_exposes_the_reference_members(tmp_path, mapping):
    """examples/monoslam_adapter.cpp: the reference example's loop written against include/scenelib2_amd_monoslam.hpp
    (Init / GoOneStep / xv_, feature_list_, selected_feature_list_, trajectory_store_ ...).  Its read-out must be what the
    oracle's members hold after the same frames."""
    exe = os.path.join(ROOT, "examples", "monoslam_adapter")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=20)
    cfg, fd = _write_scene(str(tmp_path), cam, params, spec, frames, templates)
    dump = os.path.join(str(tmp_path), "members.txt")
    seams = mapping == "seams"      # the step through Kalman::KalmanFilterPredict / auto_select_n_features / make_measurements / ...
    mapping = bool(mapping) and not seams
    cmd = [exe, "--cfg", cfg, "--frames", fd, "--dump", dump] + (["--mapping"] if mapping else []) + (["--seams"] if seams else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "%d known features" % spec.n_features in out.stdout
    v = np.loadtxt(dump)
    s = oracle_for(cam, params, spec, templates, oa)
    for k in range(1, 21):
        s.go_one_step(frames[k], True, mapping)
    n = int(v[0])
    assert n == s.total_state_size
    x0, P0 = s.total_state(), s.total_covariance()
    assert np.abs(v[1:1 + n] - x0).max() < 1e-9
    at = 1 + n
    kinds = s.feature_kinds()                      # per feature: state size, fully initialised, label
    assert s.num_features >= spec.n_features
    for i in range(s.num_features):
        fo = s.feature(i)
        label, fully, attempted, successful, pos = (int(t) for t in v[at:at + 5])
        at += 5
        d = int(kinds[i][0])
        assert (label, fully, d) == (fo["label"], int(kinds[i][1]), 3 if fully else 6)
        assert (attempted, successful, pos) == (fo["attempted"], fo["successful"], fo["pos"])
        Pyy = v[at:at + d * d].reshape(d, d)
        at += d * d
        ref = P0[pos:pos + d, pos:pos + d]
        assert np.abs(Pyy - ref).max() <= 1e-8 * max(np.abs(ref).max(), 1e-12)
        patch = v[at:at + 121].astype(np.uint8).reshape(11, 11)
        at += 121
        assert np.array_equal(patch, s.feature_patch(i)), "patch_ of feature %d" % label
    traj = v[at:].reshape(-1, 3)
    t0 = s.trajectory()
    assert traj.shape == t0.shape and np.abs(traj - t0).max() < 1e-9


def test_adapter_loop_of_120_frames_holds_the_oracles_partial_features_frame_by_frame(tmp_path):
    """The adapter's timed loop (--latency: what scripts/adapter_latency.py runs) on the feature-initialisation sequence: frames
    read in place from the grabber's pinned ring, one snapshot per frame, the engine choosing the feature-initialisation launches
    of each step from the previous step's report (sl2_engine.hip: parts_state_for_step).  feature_init_info_vector_.size() at the
    start of every frame must be the oracle's - a feature made, matched, converted or dropped one frame late would show here."""
    import json
    exe = os.path.join(ROOT, "examples", "monoslam_adapter")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=120)
    cfg, fd = _write_scene(str(tmp_path), cam, params, spec, frames, templates)
    out = os.path.join(str(tmp_path), "latency.json")
    subprocess.run([exe, "--cfg", cfg, "--frames", fd, "--latency", out, "--mapping"], check=True, timeout=300)
    got = json.load(open(out))
    s = oracle_for(cam, params, spec, templates, oa)
    seq = ""
    for k in range(1, 121):
        seq += str(min(s.mapping_info()["n_partial"], 9))
        s.go_one_step(frames[k], True, True)
    assert got["partial_features_at_frame_start"] == seq
    assert got["frames_starting_without_partial_feature"] + got["frames_starting_with_partial_feature"] == got["timed_frames"]
    assert got["features_at_end"] == s.num_features
