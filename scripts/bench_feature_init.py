"""Measurement of the SURVEY 8(f) rank-1 rows (not the headline bench): the two stateless image operators and
GoOneStep(enable_mapping) on a batch, each beside the oracle on the host.  One JSON line per row.
  python scripts/bench_feature_init.py [--batch 256] [--frames 24]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--frames", type=int, default=24)
    args = ap.parse_args()
    import oracle_api as oa
    from mapping_helpers import make_mapping_sequence, oracle_for
    from scenelib2_amd import Engine, improc

    rng = np.random.default_rng(1)
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=args.frames)
    W, H = cam["width"], cam["height"]

    # ---- Shi-Tomasi detector: the reference's 80x60 box, one job per frame copy
    nj = 1024
    imgs = frames[1:9]
    idx = np.arange(nj) % imgs.shape[0]
    regs = np.stack([rng.integers(6, W - 90, nj), rng.integers(6, H - 70, nj)], axis=1)
    regs = np.concatenate([regs, regs + [80, 60]], axis=1)
    improc.find_best_patch_batch(imgs, idx, regs)                       # warm-up
    uv, ev, ms = improc.find_best_patch_batch(imgs, idx, regs, want_ms=True)
    t0 = time.perf_counter()
    for j in range(32):
        oa.find_best_patch(imgs[idx[j]], regs[j])
    cpu = (time.perf_counter() - t0) / 32
    # algorithmic bytes: the (80+12) x (60+12) block of the frame each job must read once
    print(json.dumps({"row": "find_best_patch_inside_region", "jobs": nj, "region": "80x60", "kernel_ms": ms,
                      "regions_per_s": nj / (ms * 1e-3), "oracle_regions_per_s_1thread": 1.0 / cpu,
                      "algorithmic_GBps": nj * 92 * 72 / (ms * 1e-3) / 1e9}))

    # ---- multi-ellipse search: 100 particles along a ray, one job per sequence copy
    nj = 256
    pu, ce, cnt, pats, jidx = [], [], [], [], []
    for j in range(nj):
        cx, cy = rng.integers(60, W - 60), rng.integers(50, H - 50)
        img = imgs[j % imgs.shape[0]]
        pats.append(img[cy - 5:cy + 6, cx - 5:cx + 6].reshape(121)); jidx.append(j % imgs.shape[0]); cnt.append(100)
        ang = rng.uniform(0, np.pi)
        for t in np.linspace(-1, 1, 100):
            s0, s1 = rng.uniform(10, 40), rng.uniform(10, 40)
            r = rng.uniform(-0.5, 0.5) * np.sqrt(s0 * s1)
            pu.append(oa.sinv_from_S(np.array([[s0, r], [r, s1]])))
            ce.append([cx + 40 * t * np.cos(ang), cy + 40 * t * np.sin(ang)])
    pu, ce = np.array(pu), np.array(ce)
    improc.search_multiple_overlapping_ellipses_batch(imgs, jidx, np.stack(pats), cnt, pu, ce)
    res, corr, ms = improc.search_multiple_overlapping_ellipses_batch(imgs, jidx, np.stack(pats), cnt, pu, ce, want_ms=True)
    t0 = time.perf_counter()
    ncorr = 0
    for j in range(8):
        _, _, n = oa.search_multiple_ellipses(imgs[jidx[j]], pats[j], pu[100 * j:100 * j + 100], ce[100 * j:100 * j + 100])
        ncorr += n
    cpu = (time.perf_counter() - t0) / 8
    print(json.dumps({"row": "SearchMultipleOverlappingEllipses", "jobs": nj, "ellipses_per_job": 100, "kernel_ms": ms,
                      "jobs_per_s": nj / (ms * 1e-3), "oracle_jobs_per_s_1thread": 1.0 / cpu,
                      "positions_scored_per_job": ncorr / 8, "found_fraction": float(res[:, 0].mean())}))

    # ---- GoOneStep(enable_mapping) on a batch: BASELINE configs[0]-like (a dozen features, map growing)
    B = args.batch
    eng = Engine(cam, params, B, 32)
    eng.set_vehicle_state(np.tile(spec.xv0, (B, 1)), np.tile(spec.Pxx0, (B, 1, 1)))
    eng.add_known_features(np.tile(spec.feat_y, (B, 1, 1)), np.tile(spec.xp_org(), (B, 1, 1)), np.tile(templates, (B, 1, 1, 1)))
    from scenelib2_amd import _lib
    dev = _lib.DeviceBuffer(frames.shape[0] * B * W * H, 0)
    host = np.ascontiguousarray(np.repeat(frames[:, None], B, axis=1))
    dev.upload(host)
    eng.synchronize()
    t0 = time.perf_counter()
    for k in range(1, args.frames + 1):
        eng.go_one_step(dev.ptr + k * B * W * H, enable_mapping=True, on_device=True, seq_stride=W * H)
    eng.synchronize()
    gpu_s = time.perf_counter() - t0
    s = oracle_for(cam, params, spec, templates, oa)
    t0 = time.perf_counter()
    for k in range(1, args.frames + 1):
        s.go_one_step(frames[k], False, True)
    cpu_s = time.perf_counter() - t0
    info = eng.partial_feature(0)["info"]
    assert np.abs(eng.total_state(0) - s.total_state()).max() < 1e-9 and np.abs(eng.total_state(B - 1) - s.total_state()).max() < 1e-9
    print(json.dumps({"row": "GoOneStep(enable_mapping=1)", "workload": "%d sequences x %d frames, %d known features, map growing "
                      "(%d initialised, %d converted, %d deleted per sequence)" % (B, args.frames, spec.n_features,
                                                                                    info["initialised"], info["converted"], info["deleted"]),
                      "frames_per_s": B * args.frames / gpu_s, "ms_per_step": gpu_s / args.frames * 1e3,
                      "oracle_frames_per_s_1thread": args.frames / cpu_s, "state_maxabs_vs_oracle": float(np.abs(eng.total_state(0) - s.total_state()).max())}))


if __name__ == "__main__":
    main()
