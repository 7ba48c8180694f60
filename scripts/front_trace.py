"""Development aid: phase cycle stamps inside k_predict / k_select (SL2_FRONT_TRACE build, `make trace`)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chol_trace import build_engine  # noqa: E402
from scenelib2_amd import _lib  # noqa: E402

B = int(os.environ.get("TRACE_B", "1024"))
eng, step, keep = build_engine(B, 100, 320, 240)
for it in range(3):
    step(it)
eng.synchronize()
buf = _lib.DeviceBuffer(2 * 4096 * 8 * 8, 0)
buf.upload(np.zeros(2 * 4096 * 8, dtype=np.int64))
eng.L.sl2_debug_front_trace.argtypes = [C.c_void_p]
assert eng.L.sl2_debug_front_trace(C.c_void_p(buf.ptr)) == 0
step(3)
eng.synchronize()
tr = buf.download((2, 4096, 8), np.int64)[:, :B].astype(np.float64)
for k, name, phases in ((0, "k_predict", ["motion model + Pxx load", "T = F Pxx", "Pxx = T F^T + Q", "strip F Pxy"]),
                        (1, "k_select", ["load flags/scores", "rank", "write selection", "pack list"])):
    d = np.diff(tr[k][:, :5], axis=1)
    print(name, "per-workgroup cycles:", {p: int(d[:, i].mean()) for i, p in enumerate(phases)}, "total", int((tr[k][:, 4] - tr[k][:, 0]).mean()),
          " kernel span", int(tr[k][:, 4].max() - tr[k][:, 0].min()))
