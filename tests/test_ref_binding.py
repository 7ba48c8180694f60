"""INTEGRATION.md section 2(b), compiled and run: tests/ref_binding/monoslam_amd.h is SceneLib2::MonoSLAM / Kalman / Feature with
the reference's own signatures (cv::Mat frames, Eigen members, Kalman::KalmanFilterPredict(MonoSLAM*, Eigen::Vector3d&)) over
the C ABI, compiled against the Eigen / OpenCV stand-ins of oracle/ref_shim (a test-only include path).
tests/ref_binding/example_loop.cpp is the loop and the five buttons of the reference's only caller
(examples/MonoSlamSceneLib1.cpp:132-142, 190-204), headless.  On a GPU its members are compared, frame by frame, with the
reference's own code (oracle/_ref/libref.so) running MonoSLAM::Init on the shipped cfg and the same calls."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.join(ROOT, "tests", "ref_binding")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _build(out_dir):
    exe = os.path.join(out_dir, "example_loop")
    cmd = ["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "oracle", "ref_shim"),
           "-I" + os.path.join(ROOT, "include"), "-I" + HERE, "-o", exe, os.path.join(HERE, "example_loop.cpp"),
           "-L" + os.path.join(ROOT, "scenelib2_amd"), "-lscenelib2_amd", "-Wl,-rpath," + os.path.join(ROOT, "scenelib2_amd")]
    subprocess.check_call(cmd)
    return exe


def test_reference_typed_binding_compiles(tmp_path):
    """No GPU needed: the binding with the reference's signatures compiles (warnings are errors) and links against the
    product library."""
    assert os.path.exists(_build(str(tmp_path)))


def _scene_frame():
    sys.path.insert(0, GOLDEN)
    import make_golden as mg
    import oracle_api as oa
    from scenelib2_amd.config import load_config, read_pgm
    cfg = load_config(os.path.join(GOLDEN, "scenelib2_shipped.cfg"))
    patches = [read_pgm(os.path.join(GOLDEN, "known_patch%d.pgm" % i)) for i in range(4)]
    frame = mg.shipped_scene_frame(oa, cfg, patches)
    rng = np.random.default_rng(11)              # some texture for the detector of InitialiseAutoFeature
    for _ in range(40):
        x, y = int(rng.integers(10, 300)), int(rng.integers(10, 220))
        if all(abs(x - 160) > 60 or abs(y - 125) > 45 for _ in (0,)):
            frame[y:y + 6, x:x + 6] = rng.integers(0, 2) * 190 + 30
    return cfg, frame, mg


def _parse_dump(path):
    v = open(path).read().split()
    at, frames = 0, []
    while at < len(v):
        n, nf, nsel, msize, npart = (int(t) for t in v[at:at + 5])
        at += 5
        x = np.array(v[at:at + n], dtype=np.float64); at += n
        P = np.array(v[at:at + n * n], dtype=np.float64).reshape(n, n); at += n * n
        feats = []
        for _ in range(nf):
            feats.append(([int(t) for t in v[at:at + 6]], [float(t) for t in v[at + 6:at + 8]]))
            at += 8
        frames.append(dict(n=n, nsel=nsel, msize=msize, npart=npart, x=x, P=P, feats=feats))
    return frames


@pytest.mark.gpu
@pytest.mark.parametrize("seams", [False, True])
def test_reference_typed_binding_runs_the_example_loop_like_the_reference(tmp_path, seams):
    import oracle_api as oa
    from scenelib2_amd import ingest
    if not oa.ref_available():
        pytest.skip("oracle/_ref/libref.so not built")
    exe = _build(str(tmp_path))
    cfg, frame, mg = _scene_frame()
    fpath = os.path.join(str(tmp_path), "frame.pgm")
    ingest.write_pgm(fpath, frame)
    dump = os.path.join(str(tmp_path), "members.txt")
    nframes = 16
    out = subprocess.run([exe, os.path.join(GOLDEN, "scenelib2_shipped.cfg"), fpath, str(nframes), dump] + (["--seams"] if seams else []),
                         capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "[Robot state]" in out.stdout and "[Robot covariance]" in out.stdout        # print_robot_state
    got = _parse_dump(dump)
    assert len(got) == nframes

    # the reference's own Init + GoOneStep + the same button calls
    r = oa.RefSLAM(cfg["cam"], cfg["params"]["delta_t"], 10, cfg_path=mg.shipped_cfg_with_absolute_identifiers(str(tmp_path)))
    for k in range(nframes):
        r.go_one_step(frame, True, False)
        if k == 2 and not seams:
            r.initialise_feature(frame, 60, 200)
        if k == 14 and not seams:
            r.initialise_auto_feature(frame)
        if k == 7:
            assert r.delete_feature(2)
        g = got[k]
        assert g["n"] == r.total_state_size, k
        assert np.abs(g["x"] - r.total_state()).max() <= 1e-9, k
        P = r.total_covariance()
        assert np.linalg.norm(g["P"] - P) <= 1e-8 * np.linalg.norm(P), k
        assert len(g["feats"]) == r.num_features, k
        assert g["nsel"] == r.num_selected and (g["msize"] == r.measurement_size or r.num_selected == 0), k
        for i, (ints, z) in enumerate(g["feats"]):
            fo = r.feature(i)
            kinds = r.feature_kinds()
            assert ints[0] == fo["label"] and ints[1] == int(kinds[i][1]), (k, i)
            assert ints[2] == int(fo["selected"]) and ints[4:6] == [fo["attempted"], fo["successful"]], (k, i)
            if fo["selected"]:
                assert ints[3] == int(fo["success"]), (k, i)
                if fo["success"]:
                    assert z == list(fo["z"]), (k, i)
    # "Save Patch" at frame 8 wrote the marked feature's template (label 1) to patch.png in the working directory
    saved = ingest.read_image(os.path.join(str(tmp_path), "patch.png"))
    from scenelib2_amd.config import read_pgm
    assert np.array_equal(saved, read_pgm(os.path.join(GOLDEN, "known_patch1.pgm")))
    if not seams:
        assert any(g["npart"] for g in got), "the manual initialisation created no partially initialised feature"
