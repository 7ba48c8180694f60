"""PCIe-inclusive rate of the per-frame step (never the bench's `value`).

The bench keeps the frames resident in HBM.  Here the frames of the same workload start in HOST memory:
  (a) serial:      sl2_go_one_step(frames_on_device = 0) - the H2D copy sits on the engine's stream in front of the step;
  (b) overlapped:  pinned host ring, copy stream + two device buffers; the copy of frame k+1 runs under the step on
                   frame k (what sl2_ingest_next does behind a file reader), events order the two streams.
torch is used for the pinned ring, the streams and the events only.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenelib2_amd import Engine, _lib, synth  # noqa: E402


def host_fed(B=1024, N=100, W=320, H=240, steps=20, warm=5, ring=8, dev=0):
    """The three rates (frames resident / host frames copied on the engine's stream / host frames copied under the step before)
    and the link by itself, for one shape.  Imported by bench.py for the `host_fed` block of its line."""
    torch.cuda.set_device(dev)
    cam = synth.default_camera(W, H)
    params = synth.default_params(N)
    tex = synth.make_texture()
    n_render = ring
    specs = [synth.SequenceSpec(cam, N, n_render, synth.BASE_SEED + i) for i in range(B)]
    fb = W * H
    d_tex = _lib.DeviceBuffer(tex.nbytes, dev); d_tex.upload(tex)
    poses = np.ascontiguousarray(np.stack([s.poses for s in specs], axis=1))
    origins = np.ascontiguousarray(np.tile(np.stack([s.tex_origin for s in specs])[None], (n_render + 1, 1, 1)))
    d_pose = _lib.DeviceBuffer(poses.nbytes, dev); d_pose.upload(poses)
    d_org = _lib.DeviceBuffer(origins.nbytes, dev); d_org.upload(origins)
    d_frames = _lib.DeviceBuffer((n_render + 1) * B * fb, dev)
    synth.render_device(cam, d_tex.ptr, tex.shape[0], specs[0].tex_extent, d_org.ptr, d_pose.ptr, (n_render + 1) * B,
                        d_frames.ptr, device=dev)
    all_frames = d_frames.download((n_render + 1, B, H, W), np.uint8)
    templates = np.stack([synth.cut_templates(all_frames[0, b], specs[b].feat_px) for b in range(B)])

    def make_engine(stream=None):
        eng = Engine(cam, params, B, N, device=dev, stream=stream)
        eng.set_vehicle_state(np.stack([s.xv0 for s in specs]), np.stack([s.Pxx0 for s in specs]))
        eng.add_known_features(np.stack([s.feat_y for s in specs]), np.stack([s.xp_org() for s in specs]), templates)
        eng.set_feature_covariances(np.tile(np.eye(3) * 0.005 ** 2, (B, N, 1, 1)))
        eng.synchronize()
        return eng

    host = torch.from_numpy(all_frames[1:]).pin_memory()          # ring of `ring` frames, pinned
    out = {"batch": B, "features": N, "width": W, "height": H, "bytes_per_step": B * fb, "steps": steps}

    # (0) resident frames, for reference (the bench's timed region)
    eng = make_engine()
    for k in range(warm + steps):
        if k == warm:
            eng.synchronize(); t0 = time.perf_counter()
        eng.go_one_step(d_frames.ptr + (1 + k % ring) * B * fb, on_device=True, seq_stride=fb)
    eng.synchronize()
    out["resident_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    eng.close()

    # (a) serial: host pointer handed to sl2_go_one_step (pinned memory, so the copy itself runs at link speed)
    eng = make_engine()
    L = eng.L
    for k in range(warm + steps):
        if k == warm:
            eng.synchronize(); t0 = time.perf_counter()
        _lib.check(L.sl2_go_one_step(eng.h, _lib.vp(host[k % ring].data_ptr()), fb, 0, 0, 0))
    eng.synchronize()
    out["serial_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    eng.close()

    # (b) overlapped: copy stream + double buffer, engine on its own torch stream
    s_step, s_copy = torch.cuda.Stream(), torch.cuda.Stream()
    eng = make_engine(stream=s_step.cuda_stream)
    dbuf = [torch.empty((B, H, W), dtype=torch.uint8, device="cuda") for _ in range(2)]
    copied = [torch.cuda.Event() for _ in range(2)]
    used = [torch.cuda.Event() for _ in range(2)]

    def issue_copy(k):
        with torch.cuda.stream(s_copy):
            if k >= 2:
                s_copy.wait_event(used[k % 2])             # the step that read this buffer has finished
            dbuf[k % 2].copy_(host[k % ring], non_blocking=True)
            copied[k % 2].record(s_copy)

    issue_copy(0)
    for k in range(warm + steps):
        if k == warm:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        issue_copy(k + 1)
        s_step.wait_event(copied[k % 2])
        eng.go_one_step(dbuf[k % 2].data_ptr(), on_device=True, seq_stride=fb)
        used[k % 2].record(s_step)
    torch.cuda.synchronize()
    out["overlapped_ms_per_step"] = (time.perf_counter() - t0) / steps * 1e3
    eng.close()

    # the link by itself
    with torch.cuda.stream(s_copy):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s_copy)
        for k in range(10):
            dbuf[k % 2].copy_(host[k % ring], non_blocking=True)
        e1.record(s_copy)
    torch.cuda.synchronize()
    out["h2d_GBps"] = 10 * B * fb / (e0.elapsed_time(e1) * 1e-3) / 1e9
    for key in ("resident", "serial", "overlapped"):
        out[key + "_frames_per_s"] = B / (out[key + "_ms_per_step"] * 1e-3)
    out["copy_ms_per_step"] = B * fb / (out["h2d_GBps"] * 1e9) * 1e3
    out["overlap_efficiency"] = out["resident_ms_per_step"] / out["overlapped_ms_per_step"]     # 1 = the copy is entirely hidden
    # what the host link alone allows, whatever the kernels do (one GPU's link; the GPUs of a node each have their own x16)
    out["link_ceiling_frames_per_s"] = out["h2d_GBps"] * 1e9 / fb
    out["link_bound"] = bool(out["copy_ms_per_step"] > out["resident_ms_per_step"])
    return out


def main():
    shapes = {"configs2": (int(os.environ.get("PCIE_B", "1024")), 100, 320, 240, 20, 5), "configs3": (1024, 200, 640, 480, 8, 3)}
    which = sys.argv[1:] or ["configs2"]
    res = {}
    for name in which:
        B, N, W, H, steps, warm = shapes[name]
        res[name] = host_fed(B, N, W, H, steps, warm, ring=8 if name == "configs2" else 4)
        print(name, json.dumps(res[name]), flush=True)
    out_path = os.environ.get("PCIE_OUT")
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
