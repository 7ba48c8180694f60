import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenelib2_amd import _lib
L = _lib.load_testing()
for which, name in ((0, "fp64_mfma_4acc_TFLOPs"), (1, "fp64_mfma_dependent_TFLOPs"), (2, "stream_copy_GBs"), (3, "fp64_mfma_8acc_4blk_TFLOPs"), (4, "fp64_valu_fma_TFLOPs"), (5, "fp64_mixed_mfma_plus_valu_TFLOPs (MFMA share 2048*8 : VALU 128*32 per iter-pair)"), (6, "stream_write_GBs"), (7, "stream_read_GBs")):
    r = C.c_double(0)
    _lib.check(L.sl2_debug_microbench(0, which, C.byref(r)))
    print(name, "%.2f" % r.value)
