"""Test / bench infrastructure (CPU checker, never on the product path): steps sequences through the REFERENCE BUILD
(oracle/_ref/libref.so = /root/reference/scenelib2/*.cpp compiled unmodified) with mapping on, in a process of its own.

Why a process: the reference's feature initialisation draws from the process-global drand48 stream
(monoslam.cpp:986-1021, seeded by srand48(0) in MonoSLAM::Init, :1968), so two MonoSLAM objects cannot step side by
side in one process without sharing it.  Here a process runs its sequences ONE AFTER THE OTHER, srand48(0) in front of
each - every sequence sees the stream the reference's own Init would give it - and bench.py / the tests start one such
process per hardware thread.

usage: python ref_mapping_worker.py <job.npz> <frames.npy> <out.npz> <first> <last>
  job.npz    cam_* / params_* scalars, xv0 [S][13], Pxx0 [S][13][13], feat_y [S][N][3], xp_org [S][N][7],
             templates [S][N][11][11] u8, n_select
  frames.npy [frames + 1][S][H][W] u8 (frame 0 = the t = 0 view, not stepped)
  out.npz    traj [n][frames][3], final_state (object array), info [n][3] = features initialised (labels handed out) / features in the map at the end / of
             which partially initialised,
             seconds = time spent inside GoOneStep (set-up and I/O excluded)
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    job_path, frames_path, out_path, first, last = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
    import oracle_api as oa
    job = np.load(job_path, allow_pickle=True)
    cam = {k[4:]: job[k].item() for k in job.files if k.startswith("cam_")}
    params = {k[7:]: job[k].item() for k in job.files if k.startswith("params_")}
    frames = np.load(frames_path, mmap_mode="r")
    nfr = frames.shape[0] - 1
    N = job["feat_y"].shape[1]
    libc = ctypes.CDLL(None)
    oa.ref_lib()
    traj = np.zeros((last - first, nfr, 3))
    finals, infos = [], []
    secs = 0.0
    for b in range(first, last):
        libc.srand48(0)                                   # MonoSLAM::Init, monoslam.cpp:1968
        s = oa.RefSLAM(cam, params["delta_t"], int(job["n_select"]))
        s.set_mapping_params(params)
        s.set_state(job["xv0"][b], job["Pxx0"][b])
        for i in range(N):
            s.add_known_feature(job["feat_y"][b, i], job["xp_org"][b, i], job["templates"][b, i])
        fr = np.ascontiguousarray(frames[1:, b])
        t0 = time.perf_counter()
        for k in range(nfr):
            s.go_one_step(fr[k], False, True)
            traj[b - first, k] = s.get_state()[0][:3]
        secs += time.perf_counter() - t0
        finals.append(s.total_state())
        info = s.mapping_info()      # (the reference keeps no event counters: "initialised" holds next_free_label_, ref_glue.cpp)
        infos.append((info["initialised"] - N, s.num_features, info["n_partial"]))
    fin = np.empty(len(finals), dtype=object)
    for i, f in enumerate(finals):
        fin[i] = f
    np.savez(out_path, traj=traj, final_state=fin, info=np.array(infos, dtype=np.int64).reshape(-1, 3), seconds=secs)


if __name__ == "__main__":
    main()
