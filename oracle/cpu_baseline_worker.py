"""CPU baseline worker - TEST INFRASTRUCTURE (started by oracle/cpu_baseline.py; never on the product path).

Steps its share of the sample's sequences through the oracle (oracle/liboracle.so), ONE AFTER THE OTHER, pinned to one CPU,
in a process of its own.  Start-up and set-up are not timed: the worker builds its objects, says so (file ready<w>), waits
for the common start flag (file go) and then times only its GoOneStep loop.

usage: python cpu_baseline_worker.py <dir> <worker> <first> <last> <cpu or -1>
  <dir>/job.npz    cam_* / params_* scalars, xv0 [S][13], Pxx0 [S][13][13], feat_y [S][N][3], xp_org [S][N][7],
                   templates [S][N][11][11] u8, n_select, feature_sigma, mapping
  <dir>/frames.npy [frames + 1][S][H][W] u8 (frame 0 = the t = 0 view, not stepped)
  <dir>/out<w>.npz traj [n][frames][3], final_state (object array), info [n][3] = features initialised / features in the map
                   at the end / of which partially initialised, seconds (wall, inside GoOneStep), cpu_seconds (process CPU
                   time over the same region)
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    td, w, first, last, cpu = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    if cpu >= 0:
        try:
            os.sched_setaffinity(0, {cpu})
        except (AttributeError, OSError):
            pass
    import oracle_api as oa
    job = np.load(os.path.join(td, "job.npz"), allow_pickle=True)
    cam = {k[4:]: job[k].item() for k in job.files if k.startswith("cam_")}
    params = {k[7:]: job[k].item() for k in job.files if k.startswith("params_")}
    mapping = bool(int(job["mapping"]))
    sigma = float(job["feature_sigma"])
    frames = np.load(os.path.join(td, "frames.npy"), mmap_mode="r")
    nfr = frames.shape[0] - 1
    N = job["feat_y"].shape[1]
    slams, seqs = [], []
    for b in range(first, last):
        s = oa.OracleSLAM(cam, params["delta_t"], int(job["n_select"]))
        if mapping:
            s.set_mapping_params(params)
        s.set_state(job["xv0"][b], job["Pxx0"][b])
        for i in range(N):
            s.add_known_feature(job["feat_y"][b, i], job["xp_org"][b, i], job["templates"][b, i])
        if sigma > 0.0:
            for i in range(N):
                s.set_feature_Pyy(i, np.eye(3) * sigma ** 2)
        slams.append(s)
        seqs.append(np.ascontiguousarray(frames[1:, b]))
    traj = np.zeros((last - first, nfr, 3))
    open(os.path.join(td, "ready%d" % w), "w").close()
    go = os.path.join(td, "go")
    while not os.path.exists(go):
        time.sleep(0.002)
    t0, c0 = time.perf_counter(), time.process_time()
    if mapping:
        for j, s in enumerate(slams):
            for k in range(nfr):
                s.go_one_step(seqs[j][k], False, True)
                traj[j, k] = s.get_state()[0][:3]
    else:
        for j, s in enumerate(slams):          # the C loop of the oracle: no Python between the frames
            _, t = oa.run_sequences([s], [seqs[j]], nthreads=1)
            traj[j] = t[0]
    secs, cpu_secs = time.perf_counter() - t0, time.process_time() - c0
    fin = np.empty(len(slams), dtype=object)
    infos = []
    for i, s in enumerate(slams):
        fin[i] = s.total_state()
        if mapping:
            info = s.mapping_info()
            infos.append((info["initialised"], s.num_features, info["n_partial"]))
        else:
            infos.append((0, s.num_features, 0))
    np.savez(os.path.join(td, "out%d.npz" % w), traj=traj, final_state=fin, info=np.array(infos, dtype=np.int64).reshape(-1, 3),
             seconds=secs, cpu_seconds=cpu_secs)


if __name__ == "__main__":
    main()
