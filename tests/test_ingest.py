"""Frame ingest: listing order and PGM decode on the host; the batched device path on a GPU."""
import os

import numpy as np
import pytest

from scenelib2_amd import _lib, ingest


def _make_tree(root, rng, n=7, shape=(24, 32)):
    """Frames spread over nested directories so that only the full-path sort gives the right order."""
    names = ["a/0003.pgm", "a/0001.pgm", "b/0000.pgm", "a/sub/0002.pgm", "0009.pgm", "b/0010.pgm", "a/0002.pgm"][:n]
    imgs = {}
    for nm in names:
        p = os.path.join(root, nm)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        imgs[p] = rng.integers(0, 256, shape).astype(np.uint8)
        ingest.write_pgm(p, imgs[p])
    return imgs


def test_listing_is_recursive_and_sorted_by_full_path(tmp_path):
    rng = np.random.default_rng(0)
    imgs = _make_tree(str(tmp_path), rng)
    got = ingest.list_frames(tmp_path)
    assert got == sorted(imgs.keys())
    with pytest.raises(_lib.Sl2Error):
        ingest.list_frames(os.path.join(str(tmp_path), "missing"))


def test_read_pgm_fixtures_and_comments(tmp_path):
    here = os.path.dirname(os.path.abspath(__file__))
    p = ingest.read_pgm(os.path.join(here, "golden", "known_patch0.pgm"))       # the reference's own fixture format
    assert p.shape == (11, 11) and p.dtype == np.uint8
    img = np.arange(6 * 5, dtype=np.uint8).reshape(6, 5)
    path = os.path.join(str(tmp_path), "c.pgm")
    with open(path, "wb") as f:
        f.write(b"P5\n# a comment\n5 6\n# another\n255\n" + img.tobytes())
    assert np.array_equal(ingest.read_pgm(path), img)
    with open(path, "wb") as f:
        f.write(b"P2\n5 6\n255\n")
    with pytest.raises(_lib.Sl2Error):
        ingest.read_pgm(path)


def _grey_like_libpng(rgb):
    r, g, b = (rgb[..., k].astype(np.int64) for k in range(3))
    return ((9797 * r + 19234 * g + 3737 * b + 16384) >> 15).astype(np.uint8)


def test_read_png_grey_is_byte_exact_for_every_filter_type(tmp_path):
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    img[5:20, 7:30] = np.arange(23, dtype=np.uint8)[None, :] * 3          # smooth part: the predictors matter
    for filters in (None, [0] * 37, [1] * 37, [2] * 37, [3] * 37, [4] * 37):
        path = os.path.join(str(tmp_path), "g.png")
        ingest.write_png(path, img, filters=filters, chunk=97)               # IDAT split into many chunks
        assert np.array_equal(ingest.read_image(path), img)
    # the decoder agrees with an independent writer where one is at hand
    try:
        import PIL.Image as Image
    except ImportError:
        Image = None
    if Image is not None:
        path = os.path.join(str(tmp_path), "pil.png")
        Image.fromarray(img).save(path)
        assert np.array_equal(ingest.read_image(path), img)


def test_read_png_colour_alpha_palette_and_packed_depths(tmp_path):
    rng = np.random.default_rng(3)
    d = str(tmp_path)
    rgb = rng.integers(0, 256, (19, 23, 3)).astype(np.uint8)
    ingest.write_png(os.path.join(d, "rgb.png"), rgb)
    assert np.array_equal(ingest.read_image(os.path.join(d, "rgb.png")), _grey_like_libpng(rgb))
    rgba = np.concatenate([rgb, rng.integers(0, 256, (19, 23, 1)).astype(np.uint8)], axis=2)
    ingest.write_png(os.path.join(d, "rgba.png"), rgba)
    assert np.array_equal(ingest.read_image(os.path.join(d, "rgba.png")), _grey_like_libpng(rgb))
    grey = np.repeat(rng.integers(0, 256, (19, 23, 1)).astype(np.uint8), 3, axis=2)
    ingest.write_png(os.path.join(d, "eq.png"), grey)                         # R == G == B must come back unchanged
    assert np.array_equal(ingest.read_image(os.path.join(d, "eq.png")), grey[..., 0])
    ga = rng.integers(0, 256, (19, 23, 2)).astype(np.uint8)
    ingest.write_png(os.path.join(d, "ga.png"), ga)
    assert np.array_equal(ingest.read_image(os.path.join(d, "ga.png")), ga[..., 0])
    pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)
    idx = rng.integers(0, 16, (19, 23)).astype(np.uint8)
    for depth in (8, 4):
        ingest.write_png(os.path.join(d, "pal.png"), idx, palette=pal, bit_depth=depth)
        assert np.array_equal(ingest.read_image(os.path.join(d, "pal.png")), _grey_like_libpng(pal[idx]))
    for depth in (1, 2, 4):
        v = rng.integers(0, 1 << depth, (19, 23)).astype(np.uint8)
        ingest.write_png(os.path.join(d, "packed.png"), v, bit_depth=depth)
        want = (v.astype(np.int32) * 255 // ((1 << depth) - 1)).astype(np.uint8)
        assert np.array_equal(ingest.read_image(os.path.join(d, "packed.png")), want)
    # read_image also takes the PGM container
    ingest.write_pgm(os.path.join(d, "x.pgm"), idx)
    assert np.array_equal(ingest.read_image(os.path.join(d, "x.pgm")), idx)


def _write_adam7_png(path, img, palette=None, bit_depth=8, filter_type=0):
    """An Adam7-interlaced PNG written here (PNG specification 8.2): grey (H, W), grey + alpha (H, W, 2), RGB, RGBA, or palette
    indices; bit_depth < 8 for grey / palette.  Pillow reads such files (it does not write them): the independent check."""
    import struct
    import zlib
    img = np.asarray(img)
    H, W = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = 3 if palette is not None else {1: 0, 2: 4, 3: 2, 4: 6}[ch]

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    raw = bytearray()
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = img[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for row in sub:
            if bit_depth == 8:
                line = row.astype(np.uint8).tobytes()
            else:
                bits = np.zeros(((row.shape[0] * bit_depth + 7) // 8) * 8, dtype=np.uint8)
                for k in range(bit_depth):
                    bits[k:row.shape[0] * bit_depth:bit_depth] = (row >> (bit_depth - 1 - k)) & 1
                line = np.packbits(bits).tobytes()
            cur = np.frombuffer(line, dtype=np.uint8).astype(np.int32)
            bpp = max(1, ch * bit_depth // 8)
            if filter_type == 0:
                out = cur
            elif filter_type == 1:
                out = cur.copy(); out[bpp:] -= cur[:-bpp]
            else:                                              # 2: Up
                out = cur - (prev if prev is not None else 0)
            raw += bytes([filter_type]) + (out & 255).astype(np.uint8).tobytes()
            prev = cur
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, bit_depth, ctype, 0, 0, 1))
    if palette is not None:
        data += chunk(b"PLTE", np.asarray(palette, dtype=np.uint8).tobytes())
    data += chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


def test_read_png_adam7_interlaced(tmp_path):
    """cv::imread reads interlaced PNGs (libpng de-interlaces); sizes below, at and above the 8 x 8 pattern, every colour type,
    packed depths, three filter types.  Pillow's reader, where installed, must see the same image in the file."""
    rng = np.random.default_rng(8)
    d = str(tmp_path)
    try:
        import PIL.Image as Image
    except ImportError:
        Image = None
    n = 0
    for (H, W) in ((1, 1), (3, 5), (8, 8), (9, 17), (37, 53), (240, 320)):
        path = os.path.join(d, "i.png")
        for ft in (0, 1, 2):
            g = rng.integers(0, 256, (H, W)).astype(np.uint8)
            _write_adam7_png(path, g, filter_type=ft)
            if Image is not None:
                assert np.array_equal(np.array(Image.open(path)), g)
            assert np.array_equal(ingest.read_image(path), g)
            n += 1
        rgb = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        _write_adam7_png(path, rgb, filter_type=1)
        if Image is not None:
            assert np.array_equal(np.array(Image.open(path)), rgb)
        assert np.array_equal(ingest.read_image(path), _grey_like_libpng(rgb))
        rgba = np.concatenate([rgb, rng.integers(0, 256, (H, W, 1)).astype(np.uint8)], axis=2)
        _write_adam7_png(path, rgba)
        assert np.array_equal(ingest.read_image(path), _grey_like_libpng(rgb))
        ga = rng.integers(0, 256, (H, W, 2)).astype(np.uint8)
        _write_adam7_png(path, ga, filter_type=2)
        assert np.array_equal(ingest.read_image(path), ga[..., 0])
        pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)
        idx = rng.integers(0, 16, (H, W)).astype(np.uint8)
        for depth in (8, 4):
            _write_adam7_png(path, idx, palette=pal, bit_depth=depth)
            if Image is not None:
                assert np.array_equal(np.array(Image.open(path).convert("RGB")), pal[idx])
            assert np.array_equal(ingest.read_image(path), _grey_like_libpng(pal[idx]))
        for depth in (1, 2, 4):
            v = rng.integers(0, 1 << depth, (H, W)).astype(np.uint8)
            _write_adam7_png(path, v, bit_depth=depth)
            want = (v.astype(np.int32) * 255 // ((1 << depth) - 1)).astype(np.uint8)
            assert np.array_equal(ingest.read_image(path), want)
            n += 1
    assert n >= 30


def test_read_png_rejects_what_it_does_not_decode(tmp_path):
    import struct
    import zlib
    d = str(tmp_path)
    img = np.arange(12, dtype=np.uint8).reshape(3, 4)
    good = os.path.join(d, "good.png")
    ingest.write_png(good, img)
    data = open(good, "rb").read()
    bad = os.path.join(d, "bad.png")
    with open(bad, "wb") as f:                       # truncated file
        f.write(data[:len(data) // 2])
    with pytest.raises(_lib.Sl2Error):
        ingest.read_image(bad)
    ihdr_at = data.index(b"IHDR")
    for off, val in ((12, 1), (8, 16)):              # the interlace flag on a stream that is not interlaced; 16 bits per sample
        hdr = bytearray(data[ihdr_at + 4:ihdr_at + 17])
        hdr[off] = val
        patched = data[:ihdr_at + 4] + bytes(hdr) + struct.pack(">I", zlib.crc32(b"IHDR" + bytes(hdr)) & 0xFFFFFFFF) + data[ihdr_at + 21:]
        with open(bad, "wb") as f:
            f.write(patched)
        with pytest.raises(_lib.Sl2Error):
            ingest.read_image(bad)
    with pytest.raises(_lib.Sl2Error):
        ingest.read_image(os.path.join(d, "missing.png"))


@pytest.mark.gpu
def test_batched_ingest_delivers_every_frame_in_order(tmp_path):
    rng = np.random.default_rng(1)
    H, W, nseq, nfr = 24, 32, 3, 9
    dirs, want = [], []
    for s in range(nseq):
        d = os.path.join(str(tmp_path), "seq%d" % s)
        os.makedirs(d)
        frames = rng.integers(0, 256, (nfr + s, H, W)).astype(np.uint8)      # ragged lengths: the shortest one decides
        for k in range(frames.shape[0]):
            if (k + s) % 2:                                                  # the two containers mixed in one sequence
                ingest.write_png(os.path.join(d, "%04d.png" % k), frames[k])
            else:
                ingest.write_pgm(os.path.join(d, "%04d.pgm" % k), frames[k])
        dirs.append(d); want.append(frames)
    g = ingest.FrameIngest(dirs, W, H, depth=3)
    assert g.frame_count == nfr
    L = _lib.load()
    for k in range(nfr):
        ptr, stride = g.next()
        assert stride == W * H
        host = np.zeros((nseq, H, W), np.uint8)
        _lib.check(L.sl2_dev_download(0, host.ctypes.data_as(_lib.vp), _lib.vp(ptr), host.nbytes))
        for s in range(nseq):
            assert np.array_equal(host[s], want[s][k]), (k, s)
    with pytest.raises(_lib.Sl2Error):
        g.next()
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("zero_copy", [False, True])
def test_ingest_runs_one_frame_ahead_and_never_overwrites_a_frame_in_use(tmp_path, zero_copy):
    """Small batches are handed out in place (zero_copy: the device reads the pinned batch, which goes back to the decoder only
    when the consumer's stream is past it); larger ones are uploaded.  Both under the same test.
    sl2_ingest_next uploads on a stream of its own, one frame ahead of the caller: the call that hands out frame k starts the
    copy of frame k + 1 into the other device buffer, which the caller's work on frame k - 1 read.  The contract (scenelib2_amd.h):
    the consumer of a frame is queued on `stream` before the next call; the copy then waits for it.  Checked with a SLOW
    consumer: on the caller's stream every frame is first held up by a host function (2 ms) and only then copied out - twelve
    frames queued without a single wait; every copy must still deliver its own frame."""
    import ctypes as C
    import time
    rng = np.random.default_rng(5)
    H, W, nseq, nfr = 48, 64, 2, 12
    dirs, want = [], []
    for s in range(nseq):
        d = os.path.join(str(tmp_path), "seq%d" % s)
        os.makedirs(d)
        frames = rng.integers(0, 256, (nfr, H, W)).astype(np.uint8)
        for k in range(nfr):
            ingest.write_pgm(os.path.join(d, "%04d.pgm" % k), frames[k])
        dirs.append(d); want.append(frames)
    _lib.load()
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipLaunchHostFunc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    st, pinned = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    batch = nseq * H * W
    assert hip.hipHostMalloc(C.byref(pinned), nfr * batch, 0) == 0
    hold = C.CFUNCTYPE(None, C.c_void_p)(lambda _arg: time.sleep(0.002))
    g = ingest.FrameIngest(dirs, W, H, depth=4)
    if not zero_copy:
        g.set_zero_copy(0)
    time.sleep(0.2)                                           # let the producer decode ahead, so that the prefetch really is issued
    for k in range(nfr):
        ptr, stride = g.next(stream=st.value)
        assert hip.hipLaunchHostFunc(st, C.cast(hold, C.c_void_p), None) == 0
        assert hip.hipMemcpyAsync(C.c_void_p(pinned.value + k * batch), C.c_void_p(ptr), batch, 2, st) == 0      # hipMemcpyDeviceToHost
    assert hip.hipStreamSynchronize(st) == 0
    got = np.ctypeslib.as_array(C.cast(pinned, C.POINTER(C.c_uint8)), shape=(nfr, nseq, H, W))
    for k in range(nfr):
        for s in range(nseq):
            assert np.array_equal(got[k, s], want[s][k]), (k, s)
    g.close()


def test_decoders_survive_random_corruption(tmp_path):
    """Fuzz: random byte flips / truncations of valid PNG and PGM files must end in a decoded image of the declared size or in
    an error code - never in a crash, a hang or an exception across the ABI (hostile IHDR sizes, broken zlib streams, bad
    filter bytes, short files)."""
    import struct
    import zlib
    rng = np.random.default_rng(11)
    d = str(tmp_path)
    img = rng.integers(0, 256, (23, 31)).astype(np.uint8)
    good_png, good_pgm = os.path.join(d, "g.png"), os.path.join(d, "g.pgm")
    ingest.write_png(good_png, img)
    ingest.write_pgm(good_pgm, img)
    outcomes = {"ok": 0, "err": 0}
    for src in (good_png, good_pgm):
        data = bytearray(open(src, "rb").read())
        for trial in range(150):
            b = bytearray(data)
            kind = trial % 3
            if kind == 0:
                for _ in range(int(rng.integers(1, 6))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            elif kind == 1:
                b = b[:int(rng.integers(0, len(b)))]
            else:
                pos = int(rng.integers(0, len(b)))
                b[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8))
            path = os.path.join(d, "fuzz.bin")
            with open(path, "wb") as f:
                f.write(bytes(b))
            try:
                out = ingest.read_image(path)
                assert out.ndim == 2 and out.dtype == np.uint8
                outcomes["ok"] += 1
            except _lib.Sl2Error:
                outcomes["err"] += 1
    # a header that announces a gigantic image is refused, not allocated
    data = open(good_png, "rb").read()
    at = data.index(b"IHDR")
    hdr = bytearray(data[at + 4:at + 17])
    hdr[0:8] = struct.pack(">II", 60000, 60000)
    huge = data[:at + 4] + bytes(hdr) + struct.pack(">I", zlib.crc32(b"IHDR" + bytes(hdr)) & 0xFFFFFFFF) + data[at + 21:]
    with open(os.path.join(d, "huge.png"), "wb") as f:
        f.write(huge)
    with pytest.raises(_lib.Sl2Error):
        ingest.read_image(os.path.join(d, "huge.png"))
    with open(os.path.join(d, "huge.pgm"), "wb") as f:
        f.write(b"P5\n100000 100000\n255\n" + b"\0" * 64)
    with pytest.raises(_lib.Sl2Error):
        ingest.read_image(os.path.join(d, "huge.pgm"))
    assert outcomes["err"] > 50 and outcomes["ok"] + outcomes["err"] == 300


def _pil():
    try:
        import PIL.Image as Image
        from PIL import features
        return Image if features.check("jpg") else None
    except ImportError:
        return None


@pytest.mark.skipif(_pil() is None, reason="Pillow with libjpeg is the independent checker")
def test_read_jpeg_gives_libjpeg_grey_bytes(tmp_path):
    """cv::imread(path, 0) on a JPEG (filegrabber.cpp:106-109) = libjpeg with out_color_space = JCS_GRAYSCALE: the luminance
    component as the integer inverse DCT leaves it.  An independent libjpeg build (Pillow, draft('L')) must give the same
    bytes as sl2_read_image for grey and colour files, every chroma subsampling, optimised Huffman tables, restart
    intervals, sizes that are not multiples of the MCU, 16-bit-free quantisation at qualities 30 .. 100, sequential and
    progressive files."""
    Image = _pil()
    rng = np.random.default_rng(12)

    def scene(h, w, c):
        yy, xx = np.mgrid[0:h, 0:w]
        base = 128 + 60 * np.sin(xx / 7.0) * np.cos(yy / 5.0) + 40 * ((xx // 16 + yy // 12) % 2)
        img = np.stack([base + rng.normal(0, 12, (h, w)) + 20 * k for k in range(c)], axis=-1)
        return np.clip(img, 0, 255).astype(np.uint8)

    n = 0
    for (h, w) in ((240, 320), (37, 53), (8, 8), (17, 129)):
        for c in (1, 3):
            img = scene(h, w, c)
            pil = Image.fromarray(img[..., 0] if c == 1 else img)
            variants = [dict(quality=q) for q in (30, 75, 92, 100)] + [dict(quality=85, optimize=True)]
            if c == 3:
                variants += [dict(quality=80, subsampling=s) for s in (0, 1, 2)]
            variants += [dict(quality=70, restart_marker_blocks=3), dict(quality=88, restart_marker_rows=1)]
            # progressive files (SOF2): libjpeg's default scan script - DC with successive approximation, spectral bands of the
            # AC coefficients, refinement passes - with and without optimised tables, restart intervals, subsampling
            variants += [dict(quality=q, progressive=True) for q in (40, 90, 100)]
            variants += [dict(quality=80, progressive=True, optimize=True), dict(quality=75, progressive=True, restart_marker_blocks=2)]
            if c == 3:
                variants += [dict(quality=85, progressive=True, subsampling=sub) for sub in (0, 1, 2)]
            for kw in variants:
                path = os.path.join(str(tmp_path), "t.jpg")
                pil.save(path, "JPEG", **kw)
                ref = Image.open(path)
                ref.draft("L", ref.size)                  # libjpeg itself outputs grey: no RGB -> grey conversion by Pillow
                want = np.array(ref.convert("L") if ref.mode != "L" else ref)
                got = ingest.read_image(path)
                assert got.shape == want.shape == (h, w), (kw, got.shape, want.shape)
                assert np.array_equal(got, want), (h, w, c, kw, int(np.abs(got.astype(int) - want.astype(int)).max()))
                n += 1
    assert n >= 100


@pytest.mark.skipif(_pil() is None, reason="Pillow writes the files")
def test_read_jpeg_rejects_what_it_does_not_decode(tmp_path):
    Image = _pil()
    img = (np.arange(64 * 64).reshape(64, 64) % 251).astype(np.uint8)
    path = os.path.join(str(tmp_path), "p.jpg")
    Image.fromarray(np.stack([img] * 4, axis=-1), "CMYK").save(path, "JPEG")
    with pytest.raises(_lib.Sl2Error, match="component"):
        ingest.read_image(path)
    Image.fromarray(img).save(path, "JPEG", quality=80)
    raw = open(path, "rb").read()
    for cut in (len(raw) // 2, 30, 3):
        with open(path, "wb") as f:
            f.write(raw[:cut])
        try:
            out = ingest.read_image(path)                 # a truncated scan decodes to something (libjpeg pads too) or is refused ...
            assert out.shape == (64, 64)
        except _lib.Sl2Error:
            pass                                          # ... but never crashes


@pytest.mark.skipif(_pil() is None, reason="Pillow writes the file that is then doctored")
def test_read_jpeg_bounds_the_work_a_crafted_file_can_ask_for(tmp_path):
    """Advisor, round 4: every SOS segment walks the whole MCU grid, so a small file of thousands of scan headers costs scans x
    pixels in the ingest thread.  The reader stops at 1024 scans (libjpeg's own progressive scripts have about ten)."""
    Image = _pil()
    img = (np.arange(64 * 64).reshape(64, 64) % 251).astype(np.uint8)
    path = os.path.join(str(tmp_path), "p.jpg")
    Image.fromarray(img).save(path, "JPEG", quality=80, progressive=True)
    raw = open(path, "rb").read()
    assert np.array_equal(ingest.read_image(path).shape, (64, 64))
    eoi = raw.rfind(b"\xff\xd9")
    assert eoi > 0
    # a DC refinement scan of component 1 (Ss = Se = 0, Ah = 1, Al = 0) with no data behind it: one bit per block, and a bit reader
    # that has run into a marker feeds zeros - a legal scan that changes nothing and walks the whole grid
    empty_scan = bytes([0xFF, 0xDA, 0x00, 0x08, 0x01, 0x01, 0x00, 0x00, 0x00, 0x10])
    with open(path, "wb") as f:
        f.write(raw[:eoi] + empty_scan * 20 + raw[eoi:])
    assert ingest.read_image(path).shape == (64, 64)         # a few such scans are decoded like any other
    with open(path, "wb") as f:
        f.write(raw[:eoi] + empty_scan * 3000 + raw[eoi:])
    with pytest.raises(_lib.Sl2Error, match="too many scans"):
        ingest.read_image(path)
