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


def read_image(path):
    """FileGrabber::GetImageFile for the containers the library decodes (binary PGM, PNG) -> 8-bit grey."""
    L = _lib.load()
    w, h = C.c_int(0), C.c_int(0)
    _lib.check(L.sl2_read_image(str(path).encode(), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value), dtype=np.uint8)
    _lib.check(L.sl2_read_image(str(path).encode(), _lib.u8p(out), out.size, C.byref(w), C.byref(h)))
    return out


def write_png(path, image, filters=None, palette=None, bit_depth=8, chunk=None):
    """Test / example helper: a PNG written with nothing but zlib.  image: (H, W) grey or palette indices, (H, W, 2)
    grey + alpha, (H, W, 3) RGB or (H, W, 4) RGBA, uint8.  filters: scan-line filter type per row (default: the row
    number mod 5, so that all five are exercised).  bit_depth 1/2/4 packs grey / palette samples.  chunk: split IDAT."""
    import struct
    import zlib

    img = np.ascontiguousarray(image, dtype=np.uint8)
    H, W = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[ch] if palette is None else 3
    if bit_depth != 8:
        assert ch == 1
        per = 8 // bit_depth
        padded = np.zeros((H, (W + per - 1) // per * per), dtype=np.uint8)
        padded[:, :W] = img
        rows = np.zeros((H, padded.shape[1] // per), dtype=np.uint8)
        for k in range(per):
            rows |= (padded[:, k::per] << (8 - bit_depth * (k + 1))).astype(np.uint8)
        bpp = 1
    else:
        rows = img.reshape(H, W * ch)
        bpp = ch
    stride = rows.shape[1]
    raw = bytearray()
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(H):
        ft = (y % 5) if filters is None else int(filters[y])
        cur = rows[y].astype(np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if stride > bpp else np.zeros(stride, np.int32)
        if ft == 0:
            pred = np.zeros(stride, np.int32)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            pp = left + prev - upleft
            pa, pb, pc = np.abs(pp - left), np.abs(pp - prev), np.abs(pp - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(ft)
        raw += ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk_bytes(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    z = zlib.compress(bytes(raw), 6)
    parts = [z] if not chunk else [z[i:i + chunk] for i in range(0, len(z), chunk)]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk_bytes(b"IHDR", struct.pack(">IIBBBBB", W, H, bit_depth, ctype, 0, 0, 0)))
        if palette is not None:
            f.write(chunk_bytes(b"PLTE", np.ascontiguousarray(palette, dtype=np.uint8).tobytes()))
        f.write(chunk_bytes(b"tEXt", b"Comment\x00written by scenelib2_amd.ingest.write_png"))
        for part in parts:
            f.write(chunk_bytes(b"IDAT", part))
        f.write(chunk_bytes(b"IEND", b""))


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

    def set_zero_copy(self, max_batch_bytes):
        """Batches of at most this many bytes are handed out in place (pinned host memory read by the device); 0 = never."""
        _lib.check(self.L.sl2_ingest_set_zero_copy(self.h, int(max_batch_bytes)))

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
