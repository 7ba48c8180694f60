"""Where a small-map step spends its time: one sequence (and a few small batches), a dozen features of which ten are measured
(the reference's shipped workload, data/SceneLib2.cfg:60-62), capacity 128 like the adapter's default.

For each of {one stage per launch, fused three-launch step} x {direct launches, whole-step HIP graph}:
  * wall time of a step that is waited for (sl2_go_one_step + sl2_synchronize): what a caller who needs the result sees;
  * wall time per step when 200 steps are queued and waited for once: the device-side cost of a step;
  * per-kernel durations from the engine's HIP-event brackets (level 2: every launch).
Usage: python scripts/small_latency.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenelib2_amd import Engine, _lib, synth  # noqa: E402


def build(B, N, n_select, capacity, W=320, H=240, n_render=8, sigma=0.004, dev=0):
    cam = synth.default_camera(W, H)
    params = synth.default_params(n_select)
    tex = synth.make_texture()
    specs = [synth.SequenceSpec(cam, N, n_render, synth.BASE_SEED + i) for i in range(B)]
    fb = W * H
    d_tex = _lib.DeviceBuffer(tex.nbytes, dev); d_tex.upload(tex)
    poses = np.ascontiguousarray(np.stack([s.poses for s in specs], axis=1))
    origins = np.ascontiguousarray(np.tile(np.stack([s.tex_origin for s in specs])[None], (n_render + 1, 1, 1)))
    d_pose = _lib.DeviceBuffer(poses.nbytes, dev); d_pose.upload(poses)
    d_org = _lib.DeviceBuffer(origins.nbytes, dev); d_org.upload(origins)
    d_frames = _lib.DeviceBuffer((n_render + 1) * B * fb, dev)
    synth.render_device(cam, d_tex.ptr, tex.shape[0], specs[0].tex_extent, d_org.ptr, d_pose.ptr, (n_render + 1) * B, d_frames.ptr, device=dev)
    frame0 = d_frames.download((B, H, W), np.uint8)
    templates = np.stack([synth.cut_templates(frame0[b], specs[b].feat_px) for b in range(B)])

    def make():
        eng = Engine(cam, params, B, capacity, device=dev)
        eng.set_vehicle_state(np.stack([s.xv0 for s in specs]), np.stack([s.Pxx0 for s in specs]))
        eng.add_known_features(np.stack([s.feat_y for s in specs]), np.stack([s.xp_org() for s in specs]), templates)
        if sigma > 0.0:
            eng.set_feature_covariances(np.tile(np.eye(3) * sigma ** 2, (B, N, 1, 1)))
        eng.synchronize()
        return eng
    return make, d_frames, fb, (d_tex, d_pose, d_org)


def measure(make, d_frames, fb, B, fused, graph, n_render=8):
    eng = make()
    eng.set_step_fusion(2 if fused else 0)       # 2: fused whatever the batch size (the engine's own rule stops at 256)
    eng.set_graph_mode(graph)
    ptr = lambda k: d_frames.ptr + (1 + (k % 2)) * B * fb       # two device buffers in turn (the small camera motion keeps matching)
    for k in range(30):
        eng.go_one_step(ptr(k), on_device=True, seq_stride=fb)
    eng.synchronize()
    waited = []
    for k in range(200):
        t0 = time.perf_counter()
        eng.go_one_step(ptr(k), on_device=True, seq_stride=fb)
        eng.synchronize()
        waited.append((time.perf_counter() - t0) * 1e6)
    t0 = time.perf_counter()
    for k in range(200):
        eng.go_one_step(ptr(k), on_device=True, seq_stride=fb)
    eng.synchronize()
    queued = (time.perf_counter() - t0) / 200 * 1e6
    issue = []
    for k in range(100):
        t0 = time.perf_counter()
        eng.go_one_step(ptr(k), on_device=True, seq_stride=fb)
        issue.append((time.perf_counter() - t0) * 1e6)
        eng.synchronize()
    kernels = {}
    if not graph:
        eng.set_profiling(2)
        eng.reset_kernel_times()
        for k in range(50):
            eng.go_one_step(ptr(k), on_device=True, seq_stride=fb)
        eng.synchronize()
        kernels = {name: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 2) for name, v in eng.kernel_times().items()}
    return dict(step_waited_us_median=float(np.median(waited)), step_queued_us=float(queued), issue_call_us_median=float(np.median(issue)),
                kernels_us=kernels, kernel_sum_us=float(sum(kernels.values())) if kernels else None)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "small_latency.json")
    res = {}
    for B, N, nsel, cap in ((1, 12, 10, 128), (1, 4, 4, 128), (16, 12, 10, 128), (128, 12, 10, 128), (256, 12, 10, 128), (512, 12, 10, 128), (1024, 12, 10, 128)):
        make, d_frames, fb, keep = build(B, N, nsel, cap)
        key = "batch%d_%dfeatures_select%d_capacity%d" % (B, N, nsel, cap)
        res[key] = {}
        for fused in (False, True):
            for graph in (False, True):
                r = measure(make, d_frames, fb, B, fused, graph)
                res[key]["%s_%s" % ("fused" if fused else "ten_launches", "graph" if graph else "direct")] = r
                print(key, "fused" if fused else "ten", "graph" if graph else "direct", "waited %.1f us, queued %.1f us, issue %.1f us" %
                      (r["step_waited_us_median"], r["step_queued_us"], r["issue_call_us_median"]), r["kernels_us"], flush=True)
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
