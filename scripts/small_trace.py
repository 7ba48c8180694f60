"""Development aid: wall-clock stamps of the phases inside the fused small-map kernels (SL2_FRONT_TRACE build:
   make -C scenelib2_amd/csrc trace ; SL2_LIB_PATH=scenelib2_amd/libscenelib2_amd_trace.so python scripts/small_trace.py)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from small_latency import build  # noqa: E402
from scenelib2_amd import _lib  # noqa: E402

B = int(os.environ.get("TRACE_B", "1"))
make, d_frames, fb, keep = build(B, 12, 10, 128)
eng = make()
buf = _lib.DeviceBuffer(5 * 4096 * 8 * 8, 0)
buf.upload(np.zeros(5 * 4096 * 8, dtype=np.int64))
eng.L.sl2_debug_small_trace.argtypes = [C.c_void_p]
assert eng.L.sl2_debug_small_trace(C.c_void_p(buf.ptr)) == 0
acc = {}
for it in range(40):
    eng.go_one_step(d_frames.ptr + (1 + (it % 2)) * B * fb, on_device=True, seq_stride=fb)
    eng.synchronize()
    if it < 10:
        continue
    tr = buf.download((5, 4096, 8), np.int64)[:, 0].astype(np.float64) * 0.01          # us (100 MHz)
    rows = {"front": (2, ["predict", "feature prediction", "select"]), "back": (3, ["score", "update", "finalize"]),
            "update": (4, ["S", "cholesky", "V", None])}
    for name, (row, phases) in rows.items():
        d = np.diff(tr[row][:len(phases) + 1])
        for p, v in zip(phases, d):
            if p:
                acc.setdefault(name + ": " + p, []).append(v)
    acc.setdefault("update: build A (from its start)", []).append(tr[4][0] - tr[3][1])
    acc.setdefault("update: P -= V^T V, x (to its end)", []).append(tr[3][2] - tr[4][3])
    acc.setdefault("front total", []).append(tr[2][3] - tr[2][0])
    acc.setdefault("back total", []).append(tr[3][3] - tr[3][0])
    acc.setdefault("front end -> back start (search + two launch gaps)", []).append(tr[3][0] - tr[2][3])
for k, v in acc.items():
    print("%-55s %6.2f us" % (k, float(np.median(v))))
