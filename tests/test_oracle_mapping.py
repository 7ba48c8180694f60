"""Feature-initialisation path of the oracle (monoslam.cpp:823-1533 restated in oracle/mapping_oracle.hpp): behaviour
checks on a synthetic sequence.  CPU only (the frames come from the native HOST renderer)."""
import numpy as np
import pytest

import oracle_api as oa
from mapping_helpers import make_mapping_sequence, oracle_for


@pytest.fixture(scope="module")
def run():
    cam, params, spec, frames, templates = make_mapping_sequence()
    s = oracle_for(cam, params, spec, templates, oa)
    log = []
    for k in range(1, spec.n_frames + 1):
        s.go_one_step(frames[k], False, True)
        info = s.mapping_info()
        pf = s.partial_feature(0) if info["n_partial"] else None
        log.append(dict(info=info, pf=pf, n=s.num_features, size=s.total_state_size, kinds=s.feature_kinds().copy(),
                        visible=s.num_visible, xv=s.get_state()[0].copy()))
    return cam, params, spec, frames, s, log


def test_features_are_initialised_matched_and_converted(run):
    cam, params, spec, frames, s, log = run
    last = log[-1]["info"]
    assert last["initialised"] >= 2 and last["converted"] >= 1, last
    # state size bookkeeping: 13 + 3 per full + 6 per partial feature, at every frame
    for e in log:
        assert e["size"] == 13 + int((e["kinds"][:, 0]).sum())
        assert e["info"]["n_partial"] == int((e["kinds"][:, 1] == 0).sum()) <= params["max_features_to_init_at_once"]


def test_partial_feature_is_not_matched_in_its_first_frame_and_particles_collapse(run):
    cam, params, spec, frames, s, log = run
    first = next(i for i, e in enumerate(log) if e["pf"] is not None)
    pf0 = log[first]["pf"]
    assert pf0["attempts"] == 1 and not pf0["making"] and pf0["n_particles"] == params["number_of_particles"]
    lam = pf0["particles"][:, 0]
    assert lam[0] == params["min_lambda"] and np.allclose(np.diff(lam), (params["max_lambda"] - params["min_lambda"]) / 100)
    assert np.allclose(pf0["particles"][:, 1], 0.01)
    pf1 = log[first + 1]["pf"]
    if pf1 is not None:
        assert pf1["making"] and pf1["n_particles"] <= 100 and abs(pf1["particles"][:, 1].sum() - 1) < 1e-12
        assert np.all(np.diff(pf1["particles"][:, 2]) >= 0) and abs(pf1["particles"][-1, 2] - 1) < 1e-12


def test_converted_features_lie_on_the_scene_plane(run):
    cam, params, spec, frames, s, log = run
    n_known = spec.n_features
    x = s.total_state()
    kinds = s.feature_kinds()
    pos = 13
    n_checked = 0
    for i in range(kinds.shape[0]):
        if kinds[i, 1] and kinds[i, 2] >= n_known:      # a feature that came out of the initialisation path
            y = x[pos:pos + 3]
            assert abs(y[2]) < 0.2, (i, y)                # plane z = 0, camera at z = -0.6 (sd/mean < 0.3 at conversion)
            n_checked += 1
        pos += kinds[i, 0]
    assert n_checked >= 1
    P = s.total_covariance()
    assert np.allclose(P, P.T, atol=1e-12) and np.linalg.eigvalsh(P).min() > -1e-9


def test_camera_still_tracks_with_mapping_on(run):
    cam, params, spec, frames, s, log = run
    err = np.array([np.linalg.norm(e["xv"][:3] - spec.poses[k + 1, :3]) for k, e in enumerate(log)])
    assert err[:20].max() < 0.03, err[:20].max()          # later the fast camera has lost most of the known map


def test_oracle_matches_committed_mapping_golden():
    """Regression: the feature-initialisation oracle against its own committed outputs - the per-frame log and the final
    state of the 40-frame mapping run in tests/golden/oracle_mapping.npz (tests/golden/make_golden.py)."""
    import hashlib
    from conftest import golden_path
    g = np.load(golden_path("oracle_mapping.npz"))
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=40)
    assert hashlib.sha256(frames.tobytes()).hexdigest() == str(g["frames_sha256"])     # same input bytes
    s = oracle_for(cam, params, spec, templates, oa)
    for k in range(1, 41):
        s.go_one_step(frames[k], True, True)
        info = s.mapping_info()
        got = [info["n_partial"], info["initialised"], info["converted"], info["deleted"], info["uu"], info["vv"], s.total_state_size]
        assert got == list(g["events"][k - 1]), k
        assert np.allclose(s.get_state()[0][:3], g["pos"][k - 1], rtol=0, atol=1e-13)
    assert np.allclose(s.total_state(), g["x"], rtol=0, atol=1e-12)
    assert np.allclose(s.total_covariance(), g["P"], rtol=1e-9, atol=1e-16)


def test_particle_update_equals_numpy_bayes_rule():
    """update_partially_initialised_feature_probabilities + prune + calculate_mean_and_covariance (monoslam.cpp:1449-1497,
    feature_init_info.cpp:131-174) re-evaluated in numpy from the particle set of the frame before and the measurements
    (z, h, S^-1, |S|) stored with the surviving particles: posterior = prior x N(z; h, S), normalised, pruned at
    threshold / n, re-normalised; mean and variance of lambda."""
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=30)
    s = oracle_for(cam, params, spec, templates, oa)
    checked = 0
    prev = None
    for k in range(1, 31):
        s.go_one_step(frames[k], False, True)
        cur = s.partial_feature(0)
        if cur is not None and prev is not None and cur["label"] == prev["label"] and cur["making"]:
            prior = {float(p[0]): float(p[1]) for p in prev["particles"]}            # lambda -> probability
            n_prev = prev["n_particles"]
            post = {}
            for p in cur["particles"]:                                              # survivors carry this frame's measurement
                lam, z, h = float(p[0]), p[5:7], p[3:5]
                Sinv = np.array([[p[7], p[8]], [p[8], p[9]]])
                nu = z - h
                like = np.exp(-0.5 * nu @ Sinv @ nu) / np.sqrt(2.0 * np.pi * p[10]) if p[11] != 0.0 else 0.0
                post[lam] = prior[lam] * like
            # the pruned particles' mass is unknown from the survivors alone, but pruning keeps exactly those with
            # normalised weight >= threshold / n: ratios between survivors are preserved by both normalisations
            w = np.array([post[float(p[0])] for p in cur["particles"]])
            got = cur["particles"][:, 1]
            assert np.all(w > 0)
            assert np.allclose(w / w.sum(), got, rtol=1e-10, atol=0), k
            assert np.allclose(np.cumsum(got), cur["particles"][:, 2], rtol=1e-12, atol=1e-15)
            lam = cur["particles"][:, 0]
            mean = float((got * lam).sum())
            assert abs(mean - cur["mean"]) <= 1e-12 * abs(mean)
            assert abs(float((got * lam * lam).sum()) - mean * mean - cur["covariance"]) <= 1e-10 * mean * mean
            # nobody below the pruning threshold survived (re-normalising after the pruning only raises the weights)
            assert got.min() >= params["prune_probability_threshold"] / n_prev * 0.999999
            assert cur["n_particles"] <= n_prev
            checked += 1
        prev = cur
    assert checked >= 8


def test_two_features_initialised_at_once_what_the_oracle_does():
    """params.max_features_to_init_at_once = 2 with 200 depth particles: two partially initialised features in flight (twelve
    extra states).  When the FIRST converts while the second is still partial, the reference's
    convert_from_partially_to_fully_initialised subtracts the partial size (6) instead of the difference (3) from every later
    feature's position_in_total_state_vector_ (feature.cpp:254, quirk Q28) - so that feature's recorded position ends up 3
    below where construct_total_state puts it (monoslam.cpp:501-512 sums the state sizes in list order).  Pinned here from
    the reference's TEXT, not from a run of it: every recorded position is either the running sum or lies a positive multiple
    of 3 below it, a deficit appears only in a frame in which a conversion happened, and it does appear."""
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=60, v_amp=0.5)
    params = dict(params)
    params["max_features_to_init_at_once"] = 2
    params["number_of_particles"] = 200
    params["number_of_features_to_keep_visible"] = 14
    o = oracle_for(cam, params, spec, templates, oa)
    max_partial, q28_frames, prev_deficits, prev_converted = 0, 0, {}, 0
    for k in range(1, 61):
        o.go_one_step(frames[k], False, True)
        info = o.mapping_info()
        max_partial = max(max_partial, info["n_partial"])
        kinds = o.feature_kinds()
        pos, deficits = 13, {}
        for i in range(o.num_features):
            f = o.feature(i)
            d = pos - f["pos"]
            assert d >= 0 and d % 3 == 0, (k, i, pos, f["pos"])
            if d:
                deficits[int(f["label"])] = d
            pos += int(kinds[i][0])
        assert pos == o.total_state_size
        grew = {lab: d for lab, d in deficits.items() if d > prev_deficits.get(lab, 0)}
        if grew:
            assert info["converted"] > prev_converted, (k, grew)     # only a conversion moves recorded positions
            q28_frames += 1
        prev_deficits, prev_converted = deficits, info["converted"]
    assert max_partial == 2, "the scene never had two partially initialised features in flight"
    assert q28_frames > 0, "Q28 (position moved by 6 instead of 3) never showed: no conversion happened next to a second partial feature"
