"""Development aid: cycles per phase of me_search_fused_wg in the mapping workload.
Build: cd scenelib2_amd/csrc && hipcc ... -DSL2_ME_TRACE (scripts/build_variant.sh me_trace -DSL2_ME_TRACE), then
    SL2_LIB_PATH=scenelib2_amd/libsl2_var_me_trace.so python scripts/me_trace.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "--mapping", "--cpu-sample", "0", "--no-profile"]
import bench  # noqa: E402
from scenelib2_amd import _lib  # noqa: E402

bench.main()
L = _lib.load()
out = (C.c_ulonglong * 16)()
assert L.sl2_debug_me_trace(out, 0) == 0
n = out[15]
names = ["records + box", "stamps", "image tile", "scores", "arg-min"]
tot = sum(out[k] for k in range(5))
print("workgroups that searched: %d; cycles per workgroup: %s; total %.1f" % (
    n, ", ".join("%s %.1f" % (names[k], out[k] / max(n, 1)) for k in range(5)), tot / max(n, 1)))
# the detector's phases (slots 5-8; slot 15 counts both kinds of jobs, so per-job figures need the detector's own count: slot 14)
print("detector cycles in total: image %d, horizontal sums %d, vertical sums + eigenvalue %d, reduction %d" % (out[5], out[6], out[7], out[8]))
