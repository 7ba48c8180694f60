"""CPU baseline harness - TEST INFRASTRUCTURE (only bench.py's cpu_baseline leg, scripts/ and tests/ import it).

Times the oracle (oracle/liboracle.so, a restatement of the reference's algorithm: kind "port") on the host cores THIS JOB
may use: the affinity mask and the cgroup CPU quota are read, not os.cpu_count(); one worker PROCESS per usable physical
core (own heap, own address space: the n x n temporaries of the update are mmap / munmap pairs, which serialise the threads
of one process on its mmap lock), each pinned to its core, all released together by a start flag; every worker reports the
time inside its own GoOneStep loop and its CPU time, so that a quota or a memory-bound plateau is visible in the record.
"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants (float), or None when unlimited / unreadable.  cgroup v2: cpu.max = "<quota>
    <period>" or "max <period>"; v1: cpu.cfs_quota_us / cpu.cfs_period_us (-1 = unlimited)."""
    candidates = ["/sys/fs/cgroup/cpu.max"]
    rel = None
    txt = _read("/proc/self/cgroup")
    if txt:
        for line in txt.splitlines():
            parts = line.split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2]
    if rel and rel != "/":
        candidates.insert(0, "/sys/fs/cgroup" + rel + "/cpu.max")
    for c in candidates:
        v = _read(c)
        if v:
            q = v.split()
            if len(q) == 2 and q[0] != "max":
                try:
                    return float(q[0]) / float(q[1])
                except ValueError:
                    pass
            if len(q) == 2 and q[0] == "max":
                return None
    q, p = _read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _read("/sys/fs/cgroup/cpu/cpu.cfs_period_us")
    try:
        if q is not None and p is not None and float(q) > 0:
            return float(q) / float(p)
    except ValueError:
        pass
    return None


def host_topology():
    """What this job may use: the affinity mask, one CPU id per physical core inside it (SMT siblings folded), the cgroup
    quota, and the machine's totals for comparison."""
    try:
        aff = sorted(os.sched_getaffinity(0))
    except AttributeError:
        aff = list(range(os.cpu_count() or 1))
    cores = {}
    for c in aff:
        sib = _read("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c)
        pkg = _read("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c)
        cid = _read("/sys/devices/system/cpu/cpu%d/topology/core_id" % c)
        key = (pkg, cid) if cid is not None else ("?", sib if sib is not None else str(c))
        cores.setdefault(key, []).append(c)
    one_per_core = sorted(v[0] for v in cores.values())
    quota = cgroup_cpu_quota()
    usable = len(one_per_core)
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    try:
        load1 = os.getloadavg()[0]
    except OSError:
        load1 = None
    return dict(host_hardware_threads=os.cpu_count() or 1, affinity_cpus=len(aff), physical_cores_in_affinity=len(one_per_core),
                cgroup_cpu_quota=quota, cores_usable=usable, one_cpu_per_core=one_per_core[:usable] if quota is None else one_per_core,
                loadavg_1min_before=load1)


def run(cam, params, n_select, specs, templates, frames, feature_sigma=0.0, mapping=False, workers=None, topo=None):
    """frames: uint8 [nframes + 1][nseq][H][W] (frame 0 = the pose the templates were cut at; stepped: 1 ..).  The nseq
    sequences are dealt to `workers` processes (default: one per usable physical core, at most nseq), each steps its
    sequences one after the other.  Returns (record, traj [nseq][nframes][3], finals [nseq] total states, infos [nseq])."""
    topo = topo or host_topology()
    nseq = frames.shape[1]
    nframes = frames.shape[0] - 1
    nproc = max(1, min(workers or topo["cores_usable"], nseq))
    cpus = topo["one_cpu_per_core"]
    with tempfile.TemporaryDirectory(prefix="sl2_cpu_") as td:
        job = dict(xv0=np.stack([specs[b].xv0 for b in range(nseq)]), Pxx0=np.stack([specs[b].Pxx0 for b in range(nseq)]),
                   feat_y=np.stack([specs[b].feat_y for b in range(nseq)]), xp_org=np.stack([specs[b].xp_org() for b in range(nseq)]),
                   templates=np.ascontiguousarray(np.stack([templates[b] for b in range(nseq)])), n_select=n_select,
                   feature_sigma=feature_sigma, mapping=int(bool(mapping)))
        job.update({"cam_" + k: v for k, v in cam.items()})
        job.update({"params_" + k: v for k, v in params.items()})
        np.savez(os.path.join(td, "job.npz"), **job)
        np.save(os.path.join(td, "frames.npy"), np.ascontiguousarray(frames))
        bounds = [(w * nseq // nproc, (w + 1) * nseq // nproc) for w in range(nproc)]
        t0 = time.perf_counter()
        procs = []
        for w, (lo, hi) in enumerate(bounds):
            cpu = cpus[w % len(cpus)] if cpus else -1
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "cpu_baseline_worker.py"), td, str(w), str(lo), str(hi), str(cpu)],
                                          stdout=subprocess.DEVNULL))
        # release them together once every worker has built its objects (start-up and set-up are not timed)
        deadline = time.time() + 600
        while time.time() < deadline:
            if all(os.path.exists(os.path.join(td, "ready%d" % w)) for w in range(nproc)) or any(p.poll() not in (None, 0) for p in procs):
                break
            time.sleep(0.01)
        open(os.path.join(td, "go"), "w").close()
        rcs = [p.wait() for p in procs]
        wall = time.perf_counter() - t0
        if any(rcs):
            raise RuntimeError("oracle/cpu_baseline_worker.py failed: exit codes %s" % rcs)
        traj = np.zeros((nseq, nframes, 3))
        finals, infos = [None] * nseq, [None] * nseq
        secs, cpu_secs, rates = [], [], []
        for w, (lo, hi) in enumerate(bounds):
            o = np.load(os.path.join(td, "out%d.npz" % w), allow_pickle=True)
            traj[lo:hi] = o["traj"]
            for i in range(hi - lo):
                finals[lo + i] = o["final_state"][i]
                infos[lo + i] = tuple(int(v) for v in o["info"][i])
            secs.append(float(o["seconds"]))
            cpu_secs.append(float(o["cpu_seconds"]))
            rates.append((hi - lo) * nframes / float(o["seconds"]))
    slowest = max(secs)
    rec = dict(value=nseq * nframes / slowest, unit="frames/s", cores=nproc, kind="port",
               host_hardware_threads=topo["host_hardware_threads"], affinity_cpus=topo["affinity_cpus"],
               physical_cores_in_affinity=topo["physical_cores_in_affinity"], cgroup_cpu_quota=topo["cgroup_cpu_quota"],
               cores_usable=topo["cores_usable"], loadavg_1min_before=topo["loadavg_1min_before"],
               seconds=slowest, wall_seconds_including_process_startup=wall,
               per_worker_frames_per_s=dict(min=float(min(rates)), median=float(np.median(rates)), max=float(max(rates))),
               # CPU time the workers were GIVEN / wall time they measured: well below 1 = descheduled (quota, oversubscription)
               cpu_time_fraction=float(sum(cpu_secs) / max(sum(secs), 1e-12)))
    return rec, traj, finals, infos
