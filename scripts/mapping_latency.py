"""Where a one-sequence step with feature initialisation on spends its time (the reference's default workload behind the
adapter): the synthetic feature-initialisation sequence, every step waited for.
  * wall time of go_one_step + synchronize, split by whether the frame starts with a partially initialised feature;
  * per-kernel durations from the engine's HIP-event brackets (level 2), which add ~5 us of their own to every launch.
Usage: python scripts/mapping_latency.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from scenelib2_amd import Engine, _lib  # noqa: E402
from mapping_helpers import make_mapping_sequence  # noqa: E402


def run(profiling, frames, dev, fb, cam, params, spec, templates):
    eng = Engine(cam, params, 1, 128)
    eng.set_vehicle_state(spec.xv0[None], spec.Pxx0[None])
    eng.add_known_features(spec.feat_y[None], spec.xp_org()[None], templates[None])
    eng.set_profiling(2 if profiling else 0)
    free, part, n_before = [], [], 0
    for k in range(1, frames.shape[0]):
        t0 = time.perf_counter()
        eng.go_one_step(dev.ptr + k * fb, save_trajectory=True, enable_mapping=True, on_device=True, seq_stride=fb)
        eng.synchronize()
        us = (time.perf_counter() - t0) * 1e6
        if k > 5:
            (part if n_before else free).append(us)
        n_before = eng.partial_feature(0)["info"]["n_partial"]
    out = dict(frames_without_partial=len(free), step_us_median_without_partial=float(np.median(free)),
               frames_with_partial=len(part), step_us_median_with_partial=float(np.median(part)))
    if profiling:
        out["kernels_us"] = {n: dict(us=round(v["total_ms"] / max(v["launches"], 1) * 1e3, 2), launches=v["launches"])
                             for n, v in eng.kernel_times().items()}
    return out


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "mapping_latency.json")
    cam, params, spec, frames, templates = make_mapping_sequence(n_frames=120)
    dev = _lib.DeviceBuffer(frames.nbytes, 0)
    dev.upload(frames)
    fb = frames.shape[1] * frames.shape[2]
    res = dict(plain=run(False, frames, dev, fb, cam, params, spec, templates),
               bracketed=run(True, frames, dev, fb, cam, params, spec, templates))
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
