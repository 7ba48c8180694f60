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
    buf = _lib.DeviceBuffer(nblk * 16 * 8, 0)
    buf.upload(np.zeros(nblk * 16, dtype=np.int64))
    L.sl2_debug_search_trace.argtypes = [C.c_void_p]
    assert L.sl2_debug_search_trace(C.c_void_p(buf.ptr)) == 0
    step(warm)
    eng.synchronize()
    tr = buf.download((nblk, 16), np.int64)
    act = tr[:, 6] != 0
    t = tr[act].astype(np.float64)
    print("active waves:", int(act.sum()))
    tot = (t[:, 6] - t[:, 0]).mean()
    # k_search_mfma stamps: 0 = entry, 1 = first loads issued, 6 = exit
    print("entry -> first template / band loads issued  mean %8.0f cycles" % (t[:, 1] - t[:, 0]).mean())
    print("pipelined loop over the wavefront's positions mean %8.0f cycles" % (t[:, 6] - t[:, 1]).mean())
    print("total per wave mean %.0f cycles, p95 %.0f" % (tot, np.percentile(t[:, 6] - t[:, 0], 95)))
    print("launch span %.0f cycles" % (t[:, 6].max() - t[:, 0].min()))
    names = ["record (scalar loads) wait", "vmcnt(0): prefetched band / template + previous stores", "template + planes -> LDS, barrier",
             "next position's prefetch issued", "first band: ellipse + matrix cores + scoring", "further bands", "decision + result record", "-"]
    ph = t[:, 8:16]
    print("position loop, cycles per WAVE (sum over its positions), mean / p95:")
    for k in range(7):
        print("  %-58s %8.0f %8.0f  (%4.1f %%)" % (names[k], ph[:, k].mean(), np.percentile(ph[:, k], 95), 100.0 * ph[:, k].mean() / max(ph[:, :7].sum(axis=1).mean(), 1.0)))
    # how many wavefronts are in flight over the launch (start / end stamps)
    t0 = t[:, 0].min()
    span = t[:, 6].max() - t0
    edges = np.linspace(0, span, 21)
    starts, ends = np.sort(t[:, 0] - t0), np.sort(t[:, 6] - t0)
    infl = [int(np.searchsorted(starts, e, side="right") - np.searchsorted(ends, e, side="right")) for e in edges]
    print("wavefronts in flight at 0 %, 5 %, ... 100 % of the launch span:", infl)


if __name__ == "__main__":
    main()
