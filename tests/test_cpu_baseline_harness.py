"""bench.py's CPU-baseline harness (oracle/cpu_baseline.py): what it reads of the host, and that worker processes return
the same numbers as the oracle stepped in this process."""
import os
import sys

import numpy as np

import oracle_api as oa
from scenelib2_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cpu_baseline as cb  # noqa: E402


def test_host_topology_reads_affinity_and_quota():
    t = cb.host_topology()
    aff = len(os.sched_getaffinity(0))
    assert t["affinity_cpus"] == aff and 1 <= t["physical_cores_in_affinity"] <= aff
    assert 1 <= t["cores_usable"] <= t["physical_cores_in_affinity"]
    assert t["cgroup_cpu_quota"] is None or t["cgroup_cpu_quota"] > 0
    assert len(set(t["one_cpu_per_core"])) == len(t["one_cpu_per_core"]) and set(t["one_cpu_per_core"]) <= os.sched_getaffinity(0)


def test_workers_reproduce_the_in_process_oracle():
    cam = synth.default_camera()
    N, F, S = 12, 4, 3
    params = synth.default_params(N)
    tex = synth.make_texture(size=512)
    specs, tpls, frames = [], [], []
    for b in range(S):
        spec, tpl, fr, _ = synth.make_sequence(cam, N, F, seq_index=40 + b, tex=tex)
        specs.append(spec); tpls.append(tpl); frames.append(fr)
    # make_sequence renders frames 1..F (the templates come from its own frame 0): prepend a dummy frame 0
    allf = np.stack([np.concatenate([fr[:1], fr]) for fr in frames], axis=1)
    rec, traj, finals, infos = cb.run(cam, params, N, specs, tpls, allf, feature_sigma=0.004, workers=2)
    assert rec["kind"] == "port" and rec["cores"] == 2 and rec["value"] > 0 and 0 < rec["cpu_time_fraction"] <= 1.5
    for b in range(S):
        s = oa.OracleSLAM(cam, params["delta_t"], N)
        s.set_state(specs[b].xv0, specs[b].Pxx0)
        for i in range(N):
            s.add_known_feature(specs[b].feat_y[i], specs[b].xp_org()[i], tpls[b][i])
        for i in range(N):
            s.set_feature_Pyy(i, np.eye(3) * 0.004 ** 2)
        for k in range(F):
            s.go_one_step(frames[b][k], False)
            assert np.array_equal(traj[b, k], s.get_state()[0][:3])
        assert np.array_equal(finals[b], s.total_state())
