"""Development aid: entry / exit cycle stamps of every k_syrk workgroup (SL2_CHOL_TRACE build: `make -C scenelib2_amd/csrc trace`,
SL2_LIB_PATH=scenelib2_amd/libscenelib2_amd_trace.so).  With R workgroups resident per CU, sum(lifetimes) / (256 R) is the
launch duration in CYCLES; against the launch's duration in seconds (HIP events) that is the clock the CUs ran at."""
import ctypes as C
import os
os.environ.setdefault("SL2_TRACE_SYRK", "1")   # the trace build stamps k_chol_left by default
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chol_trace import build_engine  # noqa: E402

B = 1024
eng, step, keep = build_engine(B, 100, 320, 240, n_render=40)
L = eng.L
L.sl2_debug_chol_trace.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_size_t]
for it in range(30):
    step(it)
eng.synchronize()
n = B * 4 * 8 * 4 + B * 4
out = np.zeros(n, dtype=np.int64)
L.sl2_debug_chol_trace(eng.h, out.ctypes.data_as(C.POINTER(C.c_longlong)), n)   # allocate
eng.set_profiling(2)
eng.reset_kernel_times()
step(30)
eng.synchronize()
kt = eng.kernel_times()
eng.set_profiling(0)
L.sl2_debug_chol_trace(eng.h, out.ctypes.data_as(C.POINTER(C.c_longlong)), n)
nwg = 15 * 1024
st = out[:2 * nwg].reshape(nwg, 2).astype(np.float64)
life = st[:, 1] - st[:, 0]
ok = (st[:, 0] != 0) & (life > 0)
ms = kt["k_syrk"]["total_ms"] / kt["k_syrk"]["launches"]
print("k_syrk %.4f ms; workgroups %d, lifetime mean %.0f cycles (p5 %.0f, p95 %.0f)" % (ms, ok.sum(), life[ok].mean(), np.percentile(life[ok], 5), np.percentile(life[ok], 95)))
for R in (4, 5):
    cyc = life[ok].sum() / (256 * R)
    print("  if %d workgroups are resident per CU: launch = %.0f cycles -> %.2f GHz" % (R, cyc, cyc / (ms * 1e-3) / 1e9))
