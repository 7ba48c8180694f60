"""sl2_snapshot (include/scenelib2_amd.h): the one-call read-back of a sequence's public members must carry exactly what the
individual accessors return - which are themselves compared with the reference's members elsewhere - in the reference's
layout: feature_list_ order with deleted features removed, Pxy_ / Pyy_ blocks, selected_feature_list_, the trajectory_store_
entries pushed since the caller's cursor, feature_init_info_vector_ with its particles, and the templates of labels the
caller has not seen (monoslam.h:158-218, feature.h:78-142; graphic/graphictool.cpp:130-167, 290-347 read these every frame)."""
import numpy as np
import pytest

from mapping_helpers import make_mapping_sequence
from slam_helpers import Pair
from scenelib2_amd import Engine

pytestmark = pytest.mark.gpu


def _check(e, seq, snap, seen_traj, patch_from):
    h = snap["header"]
    feats = e.features(seq)
    assert h.n_features == len(feats) == len(snap["features"])
    assert h.total_state_size == int(e.total_state_sizes(seq, 1)[0])
    xv, Pxx = e.get_vehicle_state(seq, 1)
    assert np.array_equal(snap["xv"], xv[0]) and np.array_equal(snap["Pxx"], Pxx[0])
    P = e.total_covariance(seq)
    x = e.total_state(seq)
    for fs, fa in zip(snap["features"], feats):
        i = fs["info"]
        assert (i.label, i.selected_flag, i.successful_measurement_flag) == (fa["label"], int(fa["selected"]), int(fa["success"]))
        assert (i.attempted_measurements_of_feature, i.successful_measurements_of_feature) == (fa["attempted"], fa["successful"])
        assert i.position_in_total_state_vector == fa["pos"] and i.visible == int(fa["visible"])
        d, pos = i.state_size, i.position_in_total_state_vector
        assert d == (3 if i.fully_initialised_flag else 6)
        y = np.array(list(i.y[:]) + (list(i.y_direction[:]) if d == 6 else []))
        assert np.array_equal(y, x[pos:pos + d])
        assert np.array_equal(np.array(i.h[:]), fa["h"]) and np.array_equal(np.array(i.z[:]), fa["z"])
        assert np.array_equal(np.array(i.nu[:]), fa["nu"]) and i.R == fa["R"]
        assert np.array_equal(np.array(i.S[:]).reshape(2, 2), fa["S"])
        assert np.array_equal(np.array(i.dh_by_dxp[:]).reshape(2, 7), fa["dh_by_dxp"])
        assert np.array_equal(np.array(i.dh_by_dy[:]).reshape(2, 3), fa["dh_by_dy"])
        assert np.array_equal(np.array(i.xp_org[:]), fa["xp_org"])
        assert np.array_equal(fs["Pxy"], P[:13, pos:pos + d]) and np.array_equal(fs["Pyy"], P[pos:pos + d, pos:pos + d])
    sel, cnt = e.selection(seq)
    assert list(snap["selection"]) == list(sel)
    assert (h.number_of_visible_features, h.n_selected, h.successful_measurement_vector_size) == \
           (cnt["visible"], len(sel), cnt["measurement_size"])
    traj = e.trajectory(seq)
    assert h.traj_total >= len(traj) and h.traj_first == max(seen_traj, h.traj_total - 1000) and h.traj_count == h.traj_total - h.traj_first
    if h.traj_count:
        assert np.array_equal(snap["trajectory"], traj[len(traj) - h.traj_count:])
    assert h.status_flags == int(e.status_flags()[seq])
    labels = [f["label"] for f in feats]
    assert sorted(snap["patches"]) == sorted(l for l in labels if l >= patch_from)
    for lab, p in snap["patches"].items():
        assert np.array_equal(p, e.feature_patch(seq, lab))
    assert h.next_free_label > max(labels + [-1])


def test_snapshot_equals_the_accessors_with_mapping_on():
    """Three copies of a mapping sequence (features are initialised, converted and deleted along the way): after every frame
    the blob of each sequence against the accessors; cursors as the adapter uses them."""
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=40)
    B = 3
    e = Engine(cam, params, B, 24)
    e.set_vehicle_state(np.tile(spec.xv0, (B, 1)), np.tile(spec.Pxx0, (B, 1, 1)))
    e.add_known_features(np.tile(spec.feat_y, (B, 1, 1)), np.tile(spec.xp_org(), (B, 1, 1)), np.tile(templates, (B, 1, 1, 1)))
    seen_traj, seen_label = [0] * B, [0] * B
    n_partial_seen, n_converted = 0, 0
    for k in range(1, 41):
        e.go_one_step(np.tile(frames[k], (B, 1, 1)), save_trajectory=(k % 3 != 0), enable_mapping=True)
        for b in range(B):
            snap = e.snapshot(b, seen_traj[b], seen_label[b])
            _check(e, b, snap, seen_traj[b], seen_label[b])
            h = snap["header"]
            assert h.seq == b and h.api_version == e.L.sl2_api_version() and h.steps_done == k
            rec = e.partial_feature(b)
            info, pf = rec["info"], rec["pf"]
            assert h.n_partial == info["n_partial"] == (0 if pf is None else 1)
            if pf is not None:
                pi = snap["partial"][0]["info"]
                assert (pi.label, pi.number_of_match_attempts, pi.n_particles, pi.making_measurement_on_this_step_flag) == \
                       (pf["label"], pf["attempts"], pf["n_particles"], int(pf["making"]))
                assert pi.mean == pf["mean"] and pi.covariance == pf["covariance"]
                assert np.array_equal(snap["partial"][0]["particles"], pf["particles"])
                n_partial_seen += 1
            n_converted = max(n_converted, info["converted"])
            assert (h.uu, h.vv, h.location_selected_flag) == (info["uu"], info["vv"], info["created"])
            seen_traj[b], seen_label[b] = h.traj_total, h.next_free_label
    assert n_partial_seen > 10 and n_converted >= 1
    # a second reader that starts late gets everything: all templates, the whole (bounded) trajectory
    snap = e.snapshot(1)
    assert snap["header"].traj_first == 0 and len(snap["patches"]) == snap["header"].n_features
    # ... and one that wants no templates gets none
    assert e.snapshot(1, 0, 2 ** 31 - 1)["header"].n_patches == 0
    assert e.L.sl2_snapshot_capacity(e.h) >= snap["header"].bytes


def test_snapshot_at_the_headline_shape_after_deletions():
    """100 features, dense covariance, a feature deleted by hand: positions and blocks follow the reference's compacted layout."""
    pr = Pair(100, 3, batch=2, feature_sigma=0.005)
    for k in range(3):
        pr.step_both(k, save_trajectory=True)
    assert pr.engine.delete_features([17, -1])[0]
    for b in range(2):
        _check(pr.engine, b, pr.engine.snapshot(b), 0, 0)
    assert pr.engine.snapshot(0)["header"].n_features == 99 and pr.engine.snapshot(0)["header"].total_state_size == 13 + 3 * 99
    bad = pr.engine.L.sl2_snapshot(pr.engine.h, 5, 0, 0, None, None)
    assert bad != 0
