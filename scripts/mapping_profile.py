import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mapping_helpers import make_mapping_sequence
from scenelib2_amd import Engine, _lib
cam, params, spec, frames, templates = make_mapping_sequence(n_frames=24)
W, H = cam["width"], cam["height"]
for B in (256, 1024):
    eng = Engine(cam, params, B, 32)
    eng.set_vehicle_state(np.tile(spec.xv0, (B, 1)), np.tile(spec.Pxx0, (B, 1, 1)))
    eng.add_known_features(np.tile(spec.feat_y, (B, 1, 1)), np.tile(spec.xp_org(), (B, 1, 1)), np.tile(templates, (B, 1, 1, 1)))
    dev = _lib.DeviceBuffer(frames.shape[0] * B * W * H, 0)
    dev.upload(np.ascontiguousarray(np.repeat(frames[:, None], B, axis=1)))
    eng.set_profiling(2); eng.reset_kernel_times()
    t0 = time.perf_counter()
    for k in range(1, 25):
        eng.go_one_step(dev.ptr + k * B * W * H, enable_mapping=True, on_device=True, seq_stride=W * H)
    eng.synchronize()
    dt = time.perf_counter() - t0
    kt = eng.kernel_times()
    print("B", B, "ms/step", dt / 24 * 1e3, {k: round(v["total_ms"] / 24, 3) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])})
    dev.free()
