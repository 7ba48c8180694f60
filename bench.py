#!/usr/bin/env python
"""Headline benchmark: batched MonoSLAM frames/s (320x240, 100 features).

A "step" is one MonoSLAM::GoOneStep (monoslam.cpp:108-180: predict -> select ->
elliptical NCC search -> EKF update -> normalise -> delete -> symmetrise) applied
to every sequence of the batch.  Workload at N = 1 GPU: BASELINE.json configs[2]
(batch 1024 independent 320x240 synthetic sequences, 100 features each); with
--gpus N every rank owns its own 1024 sequences (weak scaling, no collective in
the data path; RCCL only for the barrier / MAX-reduce and the final state gather).

Inputs (synthetic frames rendered on the device, templates, states) are resident
in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6   # MI355X FP64 matrix = FP64 vector peak (256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz); SURVEY §8(d)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: past the filter's start-up transient (the first ~20 frames search wider windows: the initial velocity
    # uncertainty has not been measured away yet, and a step costs up to 10 % more), long enough to average box noise
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1024, help="sequences per GPU")
    ap.add_argument("--features", type=int, default=None, help="known features per sequence (default: 100; 12 with --mapping)")
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--cpu-sample", type=int, default=-1, help="sequences of the CPU-baseline sample (-1: one per hardware thread of the host; 0: skip)")
    ap.add_argument("--cpu-frames", type=int, default=25, help="frames of the CPU-baseline / parity sample")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--feature-sigma", type=float, default=0.005,
                    help="prior std-dev (m) of every map feature; > 0 makes the covariance dense (0: AddNewKnownFeature zeros)")
    ap.add_argument("--graph", action="store_true", help="replay the step as a HIP graph (small batches are launch-bound)")
    ap.add_argument("--groups", type=int, default=0, help="sequence groups / HIP streams per engine (0: engine default)")
    ap.add_argument("--search-split", type=int, default=-1,
                    help="bands from which a search window is shared out over wavefronts (sl2_set_search_split; -1: engine default, 0: never)")
    ap.add_argument("--step-fusion", type=int, default=-1,
                    help="sl2_set_step_fusion: 0 = one stage per launch, 1 = the engine's rule (default), 2 = fused small-map step whatever the batch size")
    ap.add_argument("--no-host-fed", action="store_true",
                    help="skip the host-fed leg (frames starting in pinned host memory, copied under the previous step): an extra "
                         "block of the line, measured after the timed region on an engine of its own; never `value`")
    ap.add_argument("--mapping", action="store_true",
                    help="the reference's DEFAULT workload instead of the headline: --features known features (use 6-12), the shipped "
                         "parameters (select 10, keep 12 visible, 100 depth particles), a camera that translates past the 0.2 m/s "
                         "gate, GoOneStep(enable_mapping = true): features are initialised, converted and deleted along the way")
    a = ap.parse_args()
    if a.features is None:
        a.features = 12 if a.mapping else 100
    if a.mapping and a.feature_sigma == 0.005:
        a.feature_sigma = 0.0          # AddNewKnownFeature's zeros, like the shipped scene
    return a


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_env(rank, n, port, base=None):
    """Environment of rank `rank` of an n-rank single-node job (what torch.distributed.run would export)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def visible_devices():
    """HIP devices visible to this process, without creating a context in the launcher (rocm-smi free): asks a child."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", "import torch;print(torch.cuda.device_count())"], capture_output=True,
                             text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return 0


def spawn_ranks(n, argv=None):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU, RCCL between them.
    Rank 0's stdout (the one JSON line) is passed through; the exit code is non-zero if any rank fails."""
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    ndev = visible_devices()
    if ndev < n and os.environ.get("SL2_BENCH_BACKEND") != "gloo":
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible\n" % (n, ndev))
        return 3
    port = free_port()
    procs = []
    for r in range(n):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=launcher_env(r, n, port),
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for pr in procs:
        rc = pr.wait() or rc
    return rc


def workload_name(B, W, H, N, world):
    """Which BASELINE.json configuration a run corresponds to (configs[1..4]); anything else is named by its shape."""
    shape = "%d independent %dx%d synthetic sequences per GPU, %d features each" % (B, W, H, N)
    if (W, H, N) == (320, 240, 100):
        if B == 1:
            return "BASELINE configs[1]: single 320x240 synthetic sequence, 100 features"
        return ("BASELINE configs[2]: " if B == 1024 else "BASELINE configs[2] shape at batch %d: " % B) + shape
    if (W, H, N) == (640, 480, 200):
        return "BASELINE configs[3] (8192 sequences over 8 GPUs = 1024 per GPU)%s: %s" % ("" if B == 1024 else " at batch %d" % B, shape)
    if (W, H, N) == (1280, 720, 500):
        return "BASELINE configs[4] (4096 sequences over 8 GPUs = 512 per GPU)%s: %s" % ("" if B == 512 else " at batch %d" % B, shape)
    return "custom: " + shape


def per_rank_parity(eng, specs, templates, d_frames, cam, params, args, B, N, W, H, fb, n_render, tdev, world, nseq=2, nframes=8):
    """A rank's leg of the parity check in a multi-GPU run: the first `nseq` sequences of this rank's shard, the first
    `nframes` frames, against the oracle (oracle/liboracle.so), on the same bytes the GPU consumed.  All ranks call this (it ends in two collectives); returns the job-wide summary."""
    from scenelib2_amd import sharding
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_api as oa
    nseq = max(1, min(nseq, B))
    n_state = 13 + 3 * N
    nframes = max(2, min(n_render - 1, nframes if N <= 100 else int(nframes * (313.0 / n_state) ** 3 + 0.5) or 2))
    allf = np.stack([d_frames.download((nseq, H, W), np.uint8, offset=k * B * fb) for k in range(nframes + 1)])
    slams = []
    for b in range(nseq):
        s = oa.OracleSLAM(cam, params["delta_t"], N)
        s.set_state(specs[b].xv0, specs[b].Pxx0)
        xo = specs[b].xp_org()
        for i in range(N):
            s.add_known_feature(specs[b].feat_y[i], xo[i], templates[b][i])
        if args.feature_sigma > 0.0:
            for i in range(N):
                s.set_feature_Pyy(i, np.eye(3) * args.feature_sigma ** 2)
        slams.append(s)
    _, traj = oa.run_sequences(slams, [np.ascontiguousarray(allf[1:, b]) for b in range(nseq)], nthreads=nseq)
    log = eng.position_log(0, nseq, capacity=n_render)[:, :nframes]
    sq = float(((log - traj) ** 2).sum(axis=2).mean())
    worst_rmse = sharding.max_over_ranks(float(np.sqrt(sq)), tdev)
    worst_abs = sharding.max_over_ranks(float(np.abs(log - traj).max()), tdev)
    checked = sharding.sum_over_ranks(1, tdev)
    return dict(traj_rmse_vs_oracle=worst_rmse, position_maxabs=worst_abs, ranks_checked=int(checked), n_gpus=world,
                checker="oracle (CPU restatement; parity unpinned: DESIGN.md section 2)",
                sequences=nseq, frames=nframes,
                note="every rank: the first %d sequences of its own shard x %d frames; worst over ranks" % (nseq, nframes))


def mapping_cpu_and_parity(eng, specs, templates, d_frames, cam, params, args, B, N, W, H, fb, n_render, sample):
    """--mapping: CPU baseline and parity of the mapping-on workload against the oracle (kind "port"), one worker process
    per usable physical core of the host (oracle/cpu_baseline.py), each stepping its sequences one after the other.
    cpu_baseline = sequence-frames / the slowest worker's time inside GoOneStep; parity = every frame the engine stepped,
    every sampled sequence."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_baseline as cb
    nseq = max(1, min(sample, B, 64))
    frames_n = n_render
    allf = np.stack([d_frames.download((nseq, H, W), np.uint8, offset=k * B * fb) for k in range(frames_n + 1)])
    cpu, traj, finals, grown = cb.run(cam, params, params["number_of_features_to_select"], specs, templates, allf,
                                      feature_sigma=0.0, mapping=True)
    cpu["sample"] = ("%d sequences x %d frames (%dx%d, %d known features, mapping on) of this run's input, one worker process per "
                     "usable physical core, its sequences one after the other" % (nseq, frames_n, W, H, N))
    cpu["note"] = ("the oracle restatement (oracle/*.hpp, g++ -O3, naive fixed-order products) - a port, not the reference: the "
                   "reference needs Eigen / OpenCV / Pangolin and is unbuildable in this image")
    checker = "oracle (CPU restatement; parity unpinned: DESIGN.md section 2)"
    log = eng.position_log(0, nseq, capacity=n_render)[:, :frames_n]
    rmse = float(np.sqrt(((log - traj) ** 2).sum(axis=2).mean()))
    parity = dict(traj_rmse_vs_oracle=rmse, checker=checker, sequences=nseq, frames=frames_n, frames_stepped=n_render,
                  covers_every_timed_frame=True, position_maxabs=float(np.abs(log - traj).max()))
    # the maps grew the same way: state size (feature counts and kinds) and the whole state at the end
    dx, same_maps = 0.0, True
    for b in range(nseq):
        xo, xg = finals[b], eng.total_state(b)
        same_maps = same_maps and xo.shape == xg.shape
        if xo.shape == xg.shape:
            dx = max(dx, float(np.abs(xo - xg).max()))
    parity["final_state_maxabs"] = dx if same_maps else float("inf")
    parity["maps_equal"] = bool(same_maps)
    g = np.array(grown).mean(axis=0)
    parity["per_sequence_mean"] = dict(features_initialised=float(g[0]), features_in_map_at_end=float(g[1]), of_which_partial=float(g[2]))
    return cpu, parity


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    # stdout carries exactly ONE line, the JSON record.  Libraries write there too (gloo announces its connections, RCCL prints a
    # version banner through C stdio, buffered until exit): file descriptor 1 is pointed at stderr for the whole run and the
    # record goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch  # first: one shared HIP runtime (scenelib2_amd/_lib.py)
    import torch.distributed as dist
    from scenelib2_amd import Engine, _lib, sharding, synth

    rank, world, local_rank = sharding.env_rank_world()
    if not torch.cuda.is_available() or _lib.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    if world > torch.cuda.device_count() and os.environ.get("SL2_BENCH_BACKEND") != "gloo":
        raise SystemExit("bench.py: %d ranks but %d HIP device(s): one rank per GPU" % (world, torch.cuda.device_count()))
    # the rank's device is chosen BEFORE the process group exists: RCCL binds its communicator to the device it is told
    # (device_id) - left to guess, every rank of a fresh process sits on device 0 and the first collective sees duplicates
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    tdev = torch.device("cuda", dev)
    # SL2_BENCH_FORCE_DIST=1: create the process group even for a single rank (tests: the RCCL path - communicator bound to the
    # device, barrier, reductions and all-gather on device tensors - on a one-GPU box)
    use_dist = world > 1 or os.environ.get("SL2_BENCH_FORCE_DIST") == "1"
    if use_dist:
        # RCCL ("nccl" on ROCm).  SL2_BENCH_BACKEND=gloo is a test hook: it lets two ranks share one GPU on a
        # single-GPU box to exercise this multi-rank code path (collectives then run on host tensors).
        backend = os.environ.get("SL2_BENCH_BACKEND") or "nccl"
        if backend == "nccl":
            dist.init_process_group(backend, device_id=tdev)
            dist.barrier(device_ids=[dev])
        else:
            dist.init_process_group(backend)
            dist.barrier()
        if dist.get_backend() != "nccl":
            tdev = None

    def barrier():
        # RCCL needs to know which device this rank drives (otherwise it guesses from the rank and warns)
        if tdev is not None:
            dist.barrier(device_ids=[dev])
        else:
            dist.barrier()

    B, N, W, H = args.batch, args.features, args.width, args.height
    K, Wm = args.steps, args.warmup
    E = 0 if args.no_profile else 4          # extra untimed steps for the full per-kernel breakdown
    n_frames = K + Wm
    n_render = n_frames + E
    cam = synth.default_camera(W, H)
    params = synth.default_params(min(N, 10) if args.mapping else N)      # data/SceneLib2.cfg:60: number_of_features_to_select = 10
    NCAP = max(32, 2 * N) if args.mapping else N                          # feature slots per sequence (the map grows with mapping on)
    # mapping: a hand-held camera's motion - mostly sideways (templates are not warped: an approaching camera loses its features),
    # 0.15-0.45 m/s, i.e. past the 0.2 m/s gate of monoslam.cpp:159 most of the time; gentle rotation
    spec_kw = dict(v_amp=np.array([0.3, 0.3, 0.03]), w_amp=0.03) if args.mapping else {}
    seq_ids = sharding.global_sequence_ids(B, world, rank)

    # ---- synthetic inputs: specs on the host, frames rendered on the device ----
    t_setup = time.time()
    tex = synth.make_texture()
    specs = [synth.SequenceSpec(cam, N, n_render, synth.BASE_SEED + int(i), **spec_kw) for i in seq_ids]
    fb = W * H
    d_tex = _lib.DeviceBuffer(tex.nbytes, dev); d_tex.upload(tex)
    # pose k of every sequence, k-major: frames[k][b]
    poses = np.ascontiguousarray(np.stack([s.poses for s in specs], axis=1))       # [n_frames+1][B][7]
    origins = np.ascontiguousarray(np.tile(np.stack([s.tex_origin for s in specs])[None], (n_render + 1, 1, 1)))
    d_pose = _lib.DeviceBuffer(poses.nbytes, dev); d_pose.upload(poses)
    d_org = _lib.DeviceBuffer(origins.nbytes, dev); d_org.upload(origins)
    d_frames = _lib.DeviceBuffer((n_render + 1) * B * fb, dev)
    # the t = 0 views now (the templates are cut from them on the host); the frames behind them are rendered at the END of the
    # set-up, so that the device does not sit idle for the seconds of host work between its one heavy set-up kernel and the
    # warm-up steps (with --warmup 5 the first timed steps otherwise run on a device still settling: 1.51 / 1.56 -> 1.48 /
    # 1.53 ms per step in back-to-back pairs, gpurun r04_bc)
    synth.render_device(cam, d_tex.ptr, tex.shape[0], specs[0].tex_extent, d_org.ptr, d_pose.ptr, B, d_frames.ptr, device=dev)
    torch.cuda.synchronize()
    frame0 = d_frames.download((B, H, W), np.uint8)                                  # t = 0 views -> templates
    templates = np.stack([synth.cut_templates(frame0[b], specs[b].feat_px) for b in range(B)])

    eng = Engine(cam, params, B, NCAP, device=dev)
    if args.search_split >= 0:
        eng.set_search_split(args.search_split)
    if args.groups > 0:
        eng.set_groups(args.groups)
    if args.step_fusion >= 0:
        eng.set_step_fusion(args.step_fusion)
    if args.graph:
        eng.set_graph_mode(True)
    eng.set_vehicle_state(np.stack([s.xv0 for s in specs]), np.stack([s.Pxx0 for s in specs]))
    eng.add_known_features(np.stack([s.feat_y for s in specs]), np.stack([s.xp_org() for s in specs]), templates)
    if args.feature_sigma > 0.0:
        eng.set_feature_covariances(np.tile(np.eye(3) * args.feature_sigma ** 2, (B, N, 1, 1)))
    synth.render_device(cam, d_tex.ptr, tex.shape[0], specs[0].tex_extent, d_org.ptr + origins[0].nbytes, d_pose.ptr + poses[0].nbytes,
                        n_render * B, d_frames.ptr + B * fb, device=dev)
    eng.synchronize()
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup

    def step(k):  # frame k (0-based) = pose k+1, resident in HBM
        eng.go_one_step(d_frames.ptr + (k + 1) * B * fb, on_device=True, seq_stride=fb, enable_mapping=args.mapping)

    # ---- warm-up ----  (its last steps, when there are enough, carry a bracket on EVERY launch: they tell which kernel
    # dominates, so that the timed region brackets only that one and the search kernel - each bracket is two event markers
    # on the stream, and bracketing the four large kernels cost 1-3 % of the step)
    # profiling scopes carry the kernel symbols (they join with rocprofv3's kernel_stats.csv on the name)
    MAJOR = ("k_syrk", "k_fwdsub_lds", "k_fwdsub_ksplit", "k_fwd_gemm", "k_build_AS", "k_chol_left", "k_chol_syrk", "k_search_mfma")
    if args.mapping:
        MAJOR = MAJOR + ("k_map_find", "k_map_me_search", "k_me_big", "k_map_particles", "k_map_update", "k_map_finish")
    SEARCH = "k_search_mfma"
    focus = "k_syrk," + SEARCH
    n_probe = 3 if (not args.no_profile and Wm >= 6) else 0
    for k in range(Wm - n_probe):
        step(k)
    if n_probe:
        eng.synchronize()
        eng.set_profiling(2)
        eng.reset_kernel_times()
        for k in range(Wm - n_probe, Wm):
            step(k)
        eng.synchronize()
        probe = eng.kernel_times()
        eng.set_profiling(0)
        cand = {n: probe[n]["total_ms"] for n in MAJOR if n in probe}
        if cand:
            focus = max(cand, key=cand.get) + "," + SEARCH
    eng.synchronize()
    if not args.no_profile:
        eng.set_profile_focus(focus)
        eng.set_profiling(1)          # timed region: only the roofline kernels carry event brackets
        eng.reset_kernel_times()

    # ---- timed region: exactly K steps ----
    if use_dist:
        barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(Wm, Wm + K):
        step(k)
    torch.cuda.synchronize()
    if use_dist:
        barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, tdev if use_dist else None)

    ktimes = eng.kernel_times() if not args.no_profile else {}
    eng.set_profiling(0)
    eng.set_profile_focus("")
    work = eng.step_work()       # algorithmic work of the last timed step, summed over this rank's batch
    xv_final, _ = eng.get_vehicle_state()
    # untimed extra steps with EVERY launch bracketed: the full per-kernel breakdown
    breakdown = {}
    if E > 0:
        eng.set_profiling(2)
        eng.reset_kernel_times()
        for k in range(Wm + K, Wm + K + E):
            step(k)
        eng.synchronize()
        breakdown = eng.kernel_times()
        eng.set_profiling(0)
    total_frames = sharding.sum_over_ranks(B * K, tdev if use_dist else None)
    value = total_frames / elapsed

    # ---- gather of the small results (RCCL all-gather; outside the timed region) ----
    t_g = time.perf_counter()
    all_xv = sharding.gather_states(xv_final, tdev if use_dist else None)
    gather_ms = (time.perf_counter() - t_g) * 1e3
    status_bad = int(eng.status_flags().any())

    # ---- N > 1: every rank checks a small sample of ITS OWN sequences against the reference build (values, not shapes);
    # rank 0 reports the worst deviation and how many ranks took part
    rank_parity = None
    if world > 1 and args.cpu_sample != 0 and not args.mapping:
        rank_parity = per_rank_parity(eng, specs, templates, d_frames, cam, params, args, B, N, W, H, fb, n_render,
                                      tdev if use_dist else None, world)

    out = None
    if rank == 0:
        # ---- roofline of the dominant kernel + the search kernel ----
        roof = None
        roof_search = None
        per_kernel = {}
        if ktimes:
            tot_ms = sum(v["total_ms"] for v in ktimes.values())
            btot = sum(v["total_ms"] for v in breakdown.values())
            for name, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["total_ms"]):
                per_kernel[name] = dict(ms_per_step=v["total_ms"] / E, launches_per_step=v["launches"] / E,
                                        share=v["total_ms"] / btot if btot else 0.0)
            dom = max(ktimes.items(), key=lambda kv: kv[1]["total_ms"])[0]
            # algorithmic FLOPs per launch over the rank's batch (executed formulation)
            flops = {"k_syrk": work["sum_nnm"],                  # P -= V V^T on the symmetric half: n^2 m
                     "k_fwdsub_lds": work["sum_nmm"],            # V = A L^-T: n m^2
                     "k_fwdsub_ksplit": work["sum_nmm"],
                     "k_chol_left": work["sum_m3"] / 3.0}        # S = L L^T (one launch per batch up to 16 blocks)
            # k_build_AS (A = P H^T and S = H A + R in one pass) streams: the measured features' rows of P (1.5 m n
            # doubles) + the 7 pose rows, writes A (m n) and the lower blocks of S (m^2 / 2)
            hbm_bytes = {"k_build_AS": 8.0 * (2.5 * work["sum_nm"] + 0.5 * work["sum_m2"])}
            bsearch = work["window_bytes"] + (121.0 + 64.0) * work["searched"]   # SURVEY 8(d) B_search, summed over the batch
            if SEARCH in ktimes:
                dur = ktimes[SEARCH]["total_ms"] / ktimes[SEARCH]["launches"] * 1e-3
                ach = bsearch / dur / 1e9
                # the formulation's own matrix-core floor: 24 v_mfma_i32_16x16x64_i8 (16 cycles each) per 16 x 16 candidate
                # tile, over the chip's 1024 matrix pipes at the 2.4 GHz spec clock
                mfma_floor_us = work["search_tiles"] * 24.0 * 16.0 / (256 * 4) / 2.4e9 * 1e6
                roof_search = dict(kernel=SEARCH, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                                   frac=ach / HBM_PEAK_GBS, traffic=None, algorithmic_bytes_per_launch=bsearch,
                                   avg_launch_ms=dur * 1e3, candidates_per_launch=work["candidates"],
                                   tiles_per_launch=work["search_tiles"], mfma_floor_us=mfma_floor_us,
                                   valu_insts_per_search=None)
            if dom in flops:
                dur = ktimes[dom]["total_ms"] / ktimes[dom]["launches"] * 1e-3
                ach = flops[dom] / dur / 1e12
                roof = dict(kernel=dom, bound="mfma", achieved=ach, peak=FP64_MFMA_PEAK_TF, unit="TFLOP/s",
                            frac=ach / FP64_MFMA_PEAK_TF, traffic=None, algorithmic_flops_per_launch=flops[dom],
                            avg_launch_ms=dur * 1e3,
                            measured_ceiling={"fp64_mfma_microbench_TFLOPs": 47.8, "fp64_valu_fma_microbench_TFLOPs": 54.1,
                                              "source": "profiles/r01_microbench.txt (scripts/microbench.py on this box type)",
                                              # separate measurement, not of this run: cycle stamps of every k_syrk workgroup
                                              "clock_under_k_syrk_GHz": 1.57, "fp64_mfma_peak_at_that_clock_TFLOPs": 51.5,
                                              "clock_source": "profiles/r02_probes.txt item 17 (scripts/syrk_clock.py, batch 1024 x 100 features); `peak` above is the 2.4 GHz spec"})
            elif dom in hbm_bytes:
                dur = ktimes[dom]["total_ms"] / ktimes[dom]["launches"] * 1e-3
                ach = hbm_bytes[dom] / dur / 1e9
                roof = dict(kernel=dom, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                            traffic=None, algorithmic_bytes_per_launch=hbm_bytes[dom], avg_launch_ms=dur * 1e3)
            elif dom == SEARCH:
                roof = roof_search
            else:
                dur = ktimes[dom]["total_ms"] / ktimes[dom]["launches"] * 1e-3
                roof = dict(kernel=dom, bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None,
                            traffic=None, avg_launch_ms=dur * 1e3)

        # HBM traffic (and the search kernel's instruction count) from the committed PMC passes of this same command
        # (profiles/pmc_traffic.json; rocprofv3 cannot be nested inside this process), corrected as MI355X_MICROARCH.md
        # prescribes; only when the file was measured on this shape
        try:
            pmc_file = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pmc = pmc_file["kernels"]
            pmc_meta = pmc_file.get("meta", {})
            if pmc_meta.get("shape") == [B, W, H, N]:
                for rf in (roof, roof_search):
                    key = rf.get("kernel", "").split("@")[0] if rf else None
                    if key and key in pmc:
                        rf["traffic"] = pmc[key].get("hbm_bytes")
                        rf["traffic_measured_in_run"] = False
                        rf["traffic_source"] = ("profiles/pmc_traffic.json: separate rocprofv3 --pmc passes of this command "
                                                "(FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes, bytes per "
                                                "launch), library git %s" % pmc_meta.get("git", "unknown"))
                if roof_search and SEARCH in pmc and pmc[SEARCH].get("valu_insts") and work["searched"] > 0:
                    roof_search["valu_insts_per_search"] = pmc[SEARCH]["valu_insts"] / pmc[SEARCH].get("searched", work["searched"])
                    roof_search["valu_insts_source"] = "profiles/pmc_traffic.json: SQ_INSTS_VALU per launch (wave instructions, MFMAs included) / searches per launch"
                    # The bound that binds (SURVEY section 7 "hard parts"): the kernel is limited by vector-instruction issue, not
                    # by bytes.  A SIMD issues one wave64 vector instruction per 4 cycles; the launch's SQ_INSTS_VALU spread
                    # evenly over the chip's 1024 SIMDs at the 2.4 GHz spec clock is the time at 100 % of that pipe.
                    issue_us = pmc[SEARCH]["valu_insts"] * 4.0 / (256 * 4) / 2.4e9 * 1e6
                    roof_search["issue_floor_us"] = issue_us
                    roof_search["issue_frac"] = issue_us / (roof_search["avg_launch_ms"] * 1e3)
                    roof_search["issue_pipe"] = "VALU issue (one wave64 instruction per SIMD per 4 cycles, 1024 SIMDs, 2.4 GHz)"
                    if pmc[SEARCH].get("mfma_busy_cycles"):
                        mf_us = pmc[SEARCH]["mfma_busy_cycles"] / (256 * 4) / 2.4e9 * 1e6
                        roof_search["mfma_busy_us"] = mf_us
                        roof_search["mfma_frac"] = mf_us / (roof_search["avg_launch_ms"] * 1e3)
        except Exception:
            pass

        # ---- CPU baseline on the host cores THIS JOB may use, in the same run (rank 0, N = 1 only): a bounded sample of the
        # same workload through the oracle (kind "port": the reference itself needs Eigen / OpenCV / Pangolin and is
        # unbuildable in this image), one worker process per usable physical core (oracle/cpu_baseline.py reads the
        # affinity mask and the cgroup quota).  Reported baseline, not the target.
        cpu = None
        parity = None
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cpu_baseline as cb
        topo = cb.host_topology()
        ncores = topo["cores_usable"]
        sample = args.cpu_sample if args.cpu_sample >= 0 else min(ncores, B)
        if world == 1 and sample > 0 and args.mapping:
            cpu, parity = mapping_cpu_and_parity(eng, specs, templates, d_frames, cam, params, args, B, N, W, H, fb, n_render, sample)
        elif world == 1 and sample > 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_api as oa
            # bounded: about 10-30 s of wall time (cost per sequence-frame grows like n^3), at most 2 GB of frames on the host
            n_state = 13 + 3 * N
            cpu_frames = min(n_frames, args.cpu_frames if N <= 100 else max(2, int(args.cpu_frames * (313.0 / n_state) ** 3 + 0.5)))
            sample = max(1, min(sample, B, int(2e9 // (fb * (cpu_frames + 1)))))
            # the same bytes the GPU consumed: frames[k][b] for the first `sample` sequences, the first frames of the run
            allf = np.stack([d_frames.download((sample, H, W), np.uint8, offset=k * B * fb) for k in range(cpu_frames + 1)])

            def build(b):
                s = oa.OracleSLAM(cam, params["delta_t"], N)
                s.set_state(specs[b].xv0, specs[b].Pxx0)
                xo = specs[b].xp_org()
                for i in range(N):
                    s.add_known_feature(specs[b].feat_y[i], xo[i], templates[b][i])
                if args.feature_sigma > 0.0:
                    for i in range(N):
                        s.set_feature_Pyy(i, np.eye(3) * args.feature_sigma ** 2)
                return s

            frames_list = [np.ascontiguousarray(allf[1:, b]) for b in range(sample)]
            cpu, traj, _, _ = cb.run(cam, params, N, specs, templates, allf, feature_sigma=args.feature_sigma, mapping=False, topo=topo)
            cpu["sample"] = ("%d sequences x %d frames (%dx%d, %d features) of this run's input, one worker process (one MonoSLAM-shaped "
                             "oracle object at a time) per usable physical core" % (sample, cpu_frames, W, H, N))
            cpu["note"] = ("the oracle restatement (oracle/*.hpp, g++ -O3): its dense products are plain fixed-order loops, not Eigen's "
                           "blocked, vectorised GEMM - the reference on a real Eigen would run the EKF update (85-93 % of the CPU time) "
                           "several times faster.  A port, not the reference (which needs Eigen / OpenCV / Pangolin and is unbuildable in "
                           "this image).  A reported baseline, not the target.")
            # 8(d)(i): one sequence on one thread (the reference's own single-threaded design), measured alone
            s1 = build(0)
            secs1, _ = oa.run_sequences([s1], frames_list[:1], nthreads=1)
            cpu["single_thread_frames_per_s"] = cpu_frames / secs1
            cpu["scaling"] = cpu["value"] / (cpu["cores"] * cpu["single_thread_frames_per_s"])
            if cpu["scaling"] < 0.5:
                # name the measured cause: were the workers given the CPU (cpu_time_fraction), or did they run and crawl
                # (memory-bound: the update streams n x n temporaries, 0.8 MB each at n = 313, through shared caches / DRAM)?
                if cpu["cpu_time_fraction"] < 0.8:
                    cpu["scaling_limit"] = ("workers were descheduled: they received %.0f %% of the wall time they measured as CPU time "
                                            "(cgroup quota %s, %d CPUs in the affinity mask, load average %.1f before the run)"
                                            % (100 * cpu["cpu_time_fraction"], cpu["cgroup_cpu_quota"], cpu["affinity_cpus"],
                                               cpu["loadavg_1min_before"] or 0.0))
                else:
                    cpu["scaling_limit"] = ("workers held their cores (CPU time / wall = %.2f) and still ran %.1fx slower than one alone: "
                                            "shared-resource bound (the update's n x n FP64 temporaries stream through the shared caches "
                                            "and DRAM), not a scheduling limit" % (cpu["cpu_time_fraction"],
                                                                                  cpu["single_thread_frames_per_s"] / max(cpu["per_worker_frames_per_s"]["median"], 1e-12)))
            cpu["scaling_note"] = "value / (cores x single_thread); cores = worker processes = usable physical cores (affinity mask and cgroup quota read)"
            # stage split from the oracle's timers (the reference keeps none)
            so = oa.OracleSLAM(cam, params["delta_t"], N)
            so.set_state(specs[0].xv0, specs[0].Pxx0)
            for i in range(N):
                so.add_known_feature(specs[0].feat_y[i], specs[0].xp_org()[i], templates[0][i])
                if args.feature_sigma > 0.0:
                    so.set_feature_Pyy(i, np.eye(3) * args.feature_sigma ** 2)
            oa.run_sequences([so], frames_list[:1], nthreads=1, want_traj=False)
            d = so.diag()
            tt = sum(d["times"].values())
            cpu["stage_split"] = {k: float(v / tt) for k, v in d["times"].items()}
            # parity of the trajectories on the sample (BASELINE metric: traj RMSE vs ref <= 1e-4)
            log = eng.position_log(0, sample, capacity=n_render)[:, :cpu_frames]
            rmse = float(np.sqrt(((log - traj) ** 2).sum(axis=2).mean()))
            parity = dict(traj_rmse_vs_oracle=rmse, checker="oracle (CPU restatement; parity unpinned: DESIGN.md section 2)",
                          sequences=sample, frames=cpu_frames, position_maxabs=float(np.abs(log - traj).max()))
            # ... and over EVERY frame the engine stepped (warm-up, timed region and the bracketed extra steps) on a few
            # sequences: positions after every frame plus the whole state vector / covariance after the last one.  Bounded to
            # about a minute of host time (a sequence-frame of the reference costs ~7 ms x (n / 313)^3).
            full_seq = max(1, min(8, B, ncores))
            cost = 7e-3 * (n_state / 313.0) ** 3
            full_frames = int(max(2, min(n_render, 60.0 / cost)))
            allf2 = np.stack([d_frames.download((full_seq, H, W), np.uint8, offset=k * B * fb) for k in range(full_frames + 1)])
            slams2 = [build(b) for b in range(full_seq)]
            _, traj2 = oa.run_sequences(slams2, [np.ascontiguousarray(allf2[1:, b]) for b in range(full_seq)],
                                        nthreads=min(ncores, full_seq))
            log2 = eng.position_log(0, full_seq, capacity=n_render)[:, :full_frames]
            rmse2 = float(np.sqrt(((log2 - traj2) ** 2).sum(axis=2).mean()))
            # top level: the worst of the two legs; `sequences` x `frames` = the full-length leg (every frame stepped)
            parity.update(traj_rmse_vs_oracle=max(rmse, rmse2), sequences=full_seq, frames=full_frames, frames_stepped=n_render,
                          position_maxabs=max(parity["position_maxabs"], float(np.abs(log2 - traj2).max())),
                          covers_every_timed_frame=bool(full_frames >= Wm + K))
            parity["full_length"] = dict(sequences=full_seq, frames=full_frames, traj_rmse=rmse2,
                                         position_maxabs=float(np.abs(log2 - traj2).max()))
            parity["wide_sample"] = dict(sequences=sample, frames=cpu_frames, traj_rmse=rmse)
            if full_frames == n_render:
                # the engine stands at frame n_render: whole state and covariance against the reference's
                dx, dP = 0.0, 0.0
                for b in range(full_seq):
                    xo, Po = slams2[b].total_state(), slams2[b].total_covariance()
                    xg, Pg = eng.total_state(b), eng.total_covariance(b)
                    if xo.shape == xg.shape:
                        dx = max(dx, float(np.abs(xo - xg).max()))
                        dP = max(dP, float(np.linalg.norm(Po - Pg) / max(np.linalg.norm(Po), 1e-300)))
                    else:
                        dx = dP = float("inf")
                parity["full_length"]["final_state_maxabs"] = dx
                parity["full_length"]["final_covariance_rel_fro"] = dP

        # ---- host-fed leg: the same shape with the frames starting in pinned HOST memory, the copy of frame
        # k + 1 under the step on frame k - what a grabber-fed caller gets.  Outside the timed region, on an engine of its own;
        # the headline stays device-resident.
        host_fed = None
        if world == 1 and not args.mapping and not args.no_host_fed:
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import pcie_inclusive
                hf = pcie_inclusive.host_fed(B, N, W, H, steps=min(K, 20), warm=min(Wm, 5), ring=4)
                host_fed = dict(frames_per_s=hf["overlapped_frames_per_s"], ms_per_step=hf["overlapped_ms_per_step"],
                                serial_copy_ms_per_step=hf["serial_ms_per_step"], resident_ms_per_step_same_harness=hf["resident_ms_per_step"],
                                h2d_GBps=hf["h2d_GBps"], copy_ms_per_step=hf["copy_ms_per_step"], overlap_efficiency=hf["overlap_efficiency"],
                                link_ceiling_frames_per_s=hf["link_ceiling_frames_per_s"], link_bound=hf["link_bound"],
                                note="frames in pinned host memory, copy stream + two device buffers, the copy of frame k + 1 under the step "
                                     "on frame k; measured after the timed region; never `value`")
            except Exception as ex:                       # (no torch streams on this box, out of memory ...): the line still goes out
                host_fed = dict(error=repr(ex))

        out = {
            "metric": "batched MonoSLAM frames/sec (320x240, 100 feat)" if not args.mapping else
                      "batched MonoSLAM frames/sec (%dx%d, mapping on, %d known features: the reference's default workload)" % (W, H, N),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (workload_name(B, W, H, N, world) + ", all features selected, map prior sigma %.3g m (%s covariance)"
                                    % (args.feature_sigma, "dense" if args.feature_sigma > 0 else "block-sparse")) if not args.mapping else
                                   ("the reference's default workload (data/SceneLib2.cfg: select 10, keep 12 visible, 100 depth particles, "
                                    "one feature initialised at a time) on %d independent %dx%d synthetic sequences per GPU, %d known "
                                    "features each, GoOneStep(enable_mapping = true)" % (B, W, H, N)),
                       "enable_mapping": bool(args.mapping),
                       "sequences_per_gpu": B, "features": N, "width": W, "height": H,
                       "state_dim": 13 + 3 * N, "feature_prior_sigma_m": args.feature_sigma, "parallelism": "independent sequences sharded across %d GPU(s), no data-path collective" % world},
            # roofline = the dominant kernel of the step; roofline_search = the NCC search kernel the north star names
            # (HBM roofline + its VALU-issue roofline); the same object is repeated under roofline["search"]
            "roofline": (dict(roof, search=roof_search) if roof is not None and roof_search is not None and roof is not roof_search else roof),
            "roofline_search": roof_search, "cpu_baseline": cpu, "host_fed": host_fed, "parity": parity if parity is not None else rank_parity,
            "kernels": per_kernel,
            "work_per_step": {k: v for k, v in work.items()},
            "gather_ms": gather_ms, "setup_s": setup_s, "status_flags_set": status_bad,
            # how sl2_create placed the large matrices (probe ms of the allocations it kept and of the slowest candidates it saw)
            "placement": eng.placement(),
            "gathered_states": int(all_xv.shape[0]),
        }
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
