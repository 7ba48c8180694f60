"""oracle/ref_mapping_worker.py (the bench's mapping-on CPU leg): sequences stepped through the reference build in a
process of its own, srand48(0) in front of each, equal the oracle restatement's per-object generator frame by frame."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_api as oa
from mapping_helpers import make_mapping_sequence, oracle_for

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not oa.ref_available(), reason="oracle/_ref/libref.so absent and /root/reference not here")
def test_worker_runs_sequences_one_after_the_other_like_fresh_reference_processes(tmp_path):
    seqs = [make_mapping_sequence(seed=7 + k, n_frames=30) for k in range(3)]
    cam, params = seqs[0][0], seqs[0][1]
    job = dict(xv0=np.stack([s[2].xv0 for s in seqs]), Pxx0=np.stack([s[2].Pxx0 for s in seqs]),
               feat_y=np.stack([s[2].feat_y for s in seqs]), xp_org=np.stack([s[2].xp_org() for s in seqs]),
               templates=np.stack([s[4] for s in seqs]), n_select=params["number_of_features_to_select"])
    job.update({"cam_" + k: v for k, v in cam.items()})
    job.update({"params_" + k: v for k, v in params.items()})
    np.savez(tmp_path / "job.npz", **job)
    frames = np.stack([s[3] for s in seqs], axis=1)          # [frames + 1][S][H][W]
    np.save(tmp_path / "frames.npy", frames)
    outs = []
    for w, (lo, hi) in enumerate([(0, 2), (2, 3)]):          # two workers: one with two sequences in a row, one with the third
        out = tmp_path / ("out%d.npz" % w)
        subprocess.check_call([sys.executable, os.path.join(ROOT, "oracle", "ref_mapping_worker.py"), str(tmp_path / "job.npz"),
                               str(tmp_path / "frames.npy"), str(out), str(lo), str(hi)])
        outs.append(np.load(out, allow_pickle=True))
    traj = np.concatenate([o["traj"] for o in outs])
    finals = [f for o in outs for f in o["final_state"]]
    infos = np.concatenate([o["info"] for o in outs])
    assert infos[:, 0].min() >= 1                             # every sequence initialised at least one feature
    for b, (cam_b, params_b, spec, fr, tpl) in enumerate(seqs):
        o = oracle_for(cam_b, params_b, spec, tpl, oa)
        for k in range(fr.shape[0] - 1):
            o.go_one_step(fr[k + 1], False, True)
            assert np.abs(o.get_state()[0][:3] - traj[b, k]).max() < 1e-11, (b, k)
        xo = o.total_state()
        assert xo.shape == finals[b].shape
        assert np.abs(xo - finals[b]).max() < 1e-10
        io = o.mapping_info()
        assert (io["initialised"], o.num_features, io["n_partial"]) == tuple(infos[b])
