"""Development aid: phase cycle stamps inside k_search_mfma (SL2_SEARCH_TRACE build:
   make -C scenelib2_amd/csrc trace ; SL2_LIB_PATH=scenelib2_amd/libscenelib2_amd_trace.so python scripts/search_trace.py)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chol_trace import build_engine  # noqa: E402
from scenelib2_amd import _lib  # noqa: E402


def main():
    B, N = 1024, 100
    warm = int(os.environ.get("SL2_TRACE_WARM", "30"))      # past the start-up transient of the search windows
    eng, step, keep = build_engine(B, N, 320, 240, n_render=warm + 3)
    L = eng.L
    for it in range(warm):
        step(it)
    eng.synchronize()
    nblk = 110000                      # >= xcd_grid(N, B)
    buf = _lib.DeviceBuffer(nblk * 8 * 8, 0)
    buf.upload(np.zeros(nblk * 8, dtype=np.int64))
    L.sl2_debug_search_trace.argtypes = [C.c_void_p]
    assert L.sl2_debug_search_trace(C.c_void_p(buf.ptr)) == 0
    step(warm)
    eng.synchronize()
    tr = buf.download((nblk, 8), np.int64)
    act = tr[:, 6] != 0
    t = tr[act].astype(np.float64)
    print("active waves:", int(act.sum()))
    tot = (t[:, 6] - t[:, 0]).mean()
    # k_search_mfma stamps: 0 = entry, 1 = first loads issued, 6 = exit
    print("entry -> first template / band loads issued  mean %8.0f cycles" % (t[:, 1] - t[:, 0]).mean())
    print("pipelined loop over the wavefront's positions mean %8.0f cycles" % (t[:, 6] - t[:, 1]).mean())
    print("total per wave mean %.0f cycles, p95 %.0f" % (tot, np.percentile(t[:, 6] - t[:, 0], 95)))
    print("launch span %.0f cycles" % (t[:, 6].max() - t[:, 0].min()))


if __name__ == "__main__":
    main()
