"""Frame ingest: thin wrapper over the C ABI's FileGrabber / FrameGrabber replacement (sl2_ingest_*,
include/scenelib2_amd.h; reference: framegrabber/filegrabber.cpp:53-109, framegrabber/framegrabber.cpp:73-104)."""
import ctypes as C

import numpy as np

from . import _lib


def list_frames(directory):
    """FileGrabber::ProcessFiles: every file below `directory`, recursively, sorted by full path."""
    L = _lib.load()
    n = C.c_int(0)
    _lib.check(L.sl2_list_frames(str(directory).encode(), None, 0, C.byref(n)))
    if n.value == 0:
        return []
    cap = 4096 * max(n.value, 1)
    buf = C.create_string_buffer(cap)
    _lib.check(L.sl2_list_frames(str(directory).encode(), buf, cap, C.byref(n)))
    return buf.value.decode().split("\n")[:-1]


def read_pgm(path):
    L = _lib.load()
    w, h = C.c_int(0), C.c_int(0)
    _lib.check(L.sl2_read_pgm(str(path).encode(), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value), dtype=np.uint8)
    _lib.check(L.sl2_read_pgm(str(path).encode(), _lib.u8p(out), out.size, C.byref(w), C.byref(h)))
    return out


def write_pgm(path, image):
    """Test / example helper (binary P5)."""
    img = np.ascontiguousarray(image, dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


class FrameIngest:
    """One frame directory per sequence; next() returns (device pointer, stride) for Engine.go_one_step(on_device=True)."""

    def __init__(self, directories, width, height, device=0, depth=4):
        self.L = _lib.load()
        arr = (C.c_char_p * len(directories))(*[str(d).encode() for d in directories])
        self.h = _lib.vp()
        _lib.check(self.L.sl2_ingest_open(arr, len(directories), width, height, device, depth, C.byref(self.h)))
        self.frame_count = self.L.sl2_ingest_frame_count(self.h)

    def next(self, stream=None):
        ptr = _lib.vp()
        stride = C.c_size_t(0)
        _lib.check(self.L.sl2_ingest_next(self.h, _lib.vp(stream) if stream else None, C.byref(ptr), C.byref(stride)))
        return ptr.value, stride.value

    def close(self):
        if self.h:
            self.L.sl2_ingest_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
