"""GPU parity tests of the individual kernels / seams through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as oa
from scenelib2_amd import _lib, synth

pytestmark = pytest.mark.gpu


def test_device_fp64_score_epilogue_is_bit_exact():
    """IEEE division / sqrt on gfx950 vs the host: the score must be bit-identical (improc.cpp:99-133)."""
    rng = np.random.default_rng(100)
    n = 200000
    p = rng.integers(0, 256, (n, 121), dtype=np.int64)
    w = rng.integers(0, 256, (n, 121), dtype=np.int64)
    w[::97] = 13                       # zero image sigma
    p[::89] = 200                      # zero patch sigma
    w[1::101] = p[1::101]              # perfect match
    lo = rng.integers(0, 4, (n // 50, 121))
    w[2::50][: lo.shape[0]] = 100 + lo[: w[2::50].shape[0]]   # sigma near the 10.0 threshold region
    sums = np.stack([p.sum(1), w.sum(1), (p * w).sum(1), (p * p).sum(1), (w * w).sum(1)], axis=1).astype(np.int32)
    sums = np.ascontiguousarray(sums)
    score = np.zeros(n)
    sd0 = np.zeros(n)
    sd1 = np.zeros(n)
    T = _lib.load_testing()
    _lib.check(T.sl2_debug_ncc_score(0, _lib.ip(sums), n, _lib.dp(score), _lib.dp(sd0), _lib.dp(sd1)), T)
    L = oa.lib()
    # host evaluation of the same expression through the oracle's correlate2_warning
    bad = 0
    for i in range(0, n, 7):
        c, a, b = oa.correlate2_warning(p[i].reshape(11, 11).astype(np.uint8), w[i].reshape(11, 11).astype(np.uint8), 0, 0)
        if not (c == score[i] and a == sd0[i] and b == sd1[i]):
            bad += 1
    assert bad == 0


def test_fp64_mfma_fragment_layout():
    """k-major GEMM on v_mfma_f64_16x16x4_f64 with asymmetric operands (transposes would show)."""
    rng = np.random.default_rng(101)
    M, N, K = 64, 96, 52
    XT = np.ascontiguousarray(rng.normal(size=(K, M)))
    YT = np.ascontiguousarray(rng.normal(size=(K, N)) + np.arange(N)[None, :] * 0.01)
    Cm = np.zeros((M, N))
    T = _lib.load_testing()
    _lib.check(T.sl2_debug_gemm_kt(0, _lib.dp(XT), M, _lib.dp(YT), N, M, N, K, _lib.dp(Cm), N), T)
    want = XT.T @ YT
    assert np.allclose(Cm, want, rtol=1e-13, atol=1e-13), np.abs(Cm - want).max()
    # identity check with an asymmetric B
    I = np.zeros((4, 32))
    I[np.arange(4), np.arange(4)] = 1.0
    B = np.arange(4 * 32, dtype=np.float64).reshape(4, 32)
    Cm = np.zeros((32, 32))
    _lib.check(T.sl2_debug_gemm_kt(0, _lib.dp(np.ascontiguousarray(I)), 32, _lib.dp(B), 32, 32, 32, 4, _lib.dp(Cm), 32), T)
    assert np.array_equal(Cm[:4], B) and not Cm[4:].any()


def _search_cases(rng, n, W, H):
    images, idx, patches, centres, puinv = [], [], [], [], []
    for t in range(n):
        kind = t % 8
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        if kind == 1:                                   # exact ties: periodic image
            tile = rng.integers(0, 256, (5, 6), dtype=np.uint8)
            img = np.tile(tile, (H // 5 + 1, W // 6 + 1))[:H, :W].copy()
        if kind == 2:                                   # flat image: every candidate fails the sigma test
            img[:] = 90
        cy, cx = int(rng.integers(6, H - 6)), int(rng.integers(6, W - 6))
        patch = img[cy - 5:cy + 6, cx - 5:cx + 6].copy()
        if kind == 3:
            patch = rng.integers(0, 256, (11, 11), dtype=np.uint8)     # unrelated patch
        if kind == 4:
            patch[:] = 55                                               # flat patch
        s0, s1 = rng.uniform(1.5, 70, 2)
        r = rng.uniform(-0.85, 0.85) * np.sqrt(s0 * s1)
        a, b, c = oa.sinv_from_S(np.array([[s0, r], [r, s1]]))
        ce = np.array([cx + rng.uniform(-5, 5), cy + rng.uniform(-5, 5)])
        if kind == 5:
            ce = np.array([rng.uniform(-4, 9), rng.uniform(H - 9, H + 4)])   # window clamped by the border
        if kind == 6:
            ce = np.array([rng.uniform(W - 8, W + 3), rng.uniform(-3, 8)])
        if kind == 7:
            a, b, c = oa.sinv_from_S(np.array([[300.0, 10.0], [10.0, 250.0]]))  # very large ellipse
        images.append(img); idx.append(t); patches.append(patch.reshape(121)); centres.append(ce); puinv.append([a, b, c])
    return (np.stack(images), np.array(idx, np.int32), np.stack(patches), np.array(centres), np.array(puinv))


@pytest.mark.parametrize("variant", [0, 1])
def test_elliptical_search_batch_matches_oracle_exactly(variant):
    """variant 0 = exact kernel (one candidate per lane), 1 = int8 matrix-core walk (the engine's default search core)."""
    rng = np.random.default_rng(102)
    W, H = 160, 120
    images, idx, patches, centres, puinv = _search_cases(rng, 160, W, H)
    n = len(idx)
    ok = np.zeros(n, np.int32)
    uv = np.full((n, 2), -7, np.int32)
    score = np.zeros(n)
    _lib.check(_lib.load().sl2_elliptical_search_batch(0, _lib.u8p(images), n, W, H, _lib.ip(idx), _lib.u8p(patches),
                                                       _lib.dp(centres), _lib.dp(puinv), n, _lib.ip(ok), _lib.ip(uv),
                                                       _lib.dp(score), variant))
    n_ok = 0
    for t in range(n):
        want = oa.elliptical_search(images[t], patches[t], centres[t], *puinv[t])
        assert bool(ok[t]) == want["ok"], "case %d ok" % t
        assert score[t] == want["corr"], "case %d score %r vs %r" % (t, score[t], want["corr"])
        if want["corr"] < 1e6:
            assert (uv[t, 0], uv[t, 1]) == (want["u"], want["v"]), "case %d uv" % t
        else:
            assert (uv[t, 0], uv[t, 1]) == (-7, -7)       # untouched (Q4)
        n_ok += want["ok"]
    assert n_ok > 20


def test_elliptical_search_windows_larger_than_the_tile_are_walked_in_blocks():
    """3-sigma windows of a few hundred columns / rows (the rule at 1280x720): the matrix-core walk goes through them band
    by band (16 rows x 32 columns of candidates at a time) and must still return the reference's answer bit for bit."""
    rng = np.random.default_rng(7)
    W, H = 512, 384
    tex = synth.make_texture(size=1024)
    images, idx, patches, centres, puinv = [], [], [], [], []
    for t in range(20):
        oy, ox = int(rng.integers(0, 1024 - H)), int(rng.integers(0, 1024 - W))
        img = tex[oy:oy + H, ox:ox + W].copy()
        if t % 5 == 4:
            img = rng.integers(0, 256, (H, W), dtype=np.uint8)            # noise: many near-ties, exercises the fallback
        cy, cx = int(rng.integers(40, H - 40)), int(rng.integers(40, W - 40))
        patch = img[cy - 5:cy + 6, cx - 5:cx + 6].copy()
        s0, s1 = rng.uniform(400, 4000, 2)                                # half-widths 60 .. 190 pixels
        r = rng.uniform(-0.7, 0.7) * np.sqrt(s0 * s1)
        a, b, c = oa.sinv_from_S(np.array([[s0, r], [r, s1]]))
        ce = np.array([cx + rng.uniform(-30, 30), cy + rng.uniform(-30, 30)])
        if t % 5 == 3:
            ce = np.array([rng.uniform(-20, 30), rng.uniform(H - 30, H + 20)])   # clamped by the border
        images.append(img); idx.append(t); patches.append(patch.reshape(121)); centres.append(ce); puinv.append([a, b, c])
    images, idx, patches = np.stack(images), np.array(idx, np.int32), np.stack(patches)
    centres, puinv = np.array(centres), np.array(puinv)
    n = len(idx)
    wants = [oa.elliptical_search(images[t], patches[t], centres[t], *puinv[t]) for t in range(n)]
    for variant in (1, 0):
        ok = np.zeros(n, np.int32)
        uv = np.full((n, 2), -7, np.int32)
        score = np.zeros(n)
        _lib.check(_lib.load().sl2_elliptical_search_batch(0, _lib.u8p(images), n, W, H, _lib.ip(idx), _lib.u8p(patches),
                                                           _lib.dp(centres), _lib.dp(puinv), n, _lib.ip(ok), _lib.ip(uv),
                                                           _lib.dp(score), variant))
        for t in range(n):
            want = wants[t]
            assert bool(ok[t]) == want["ok"], "variant %d case %d ok" % (variant, t)
            assert score[t] == want["corr"], "variant %d case %d score %r vs %r" % (variant, t, score[t], want["corr"])
            if want["corr"] < 1e6:
                assert (uv[t, 0], uv[t, 1]) == (want["u"], want["v"]), "variant %d case %d uv" % (variant, t)
        assert ok.sum() >= 10


@pytest.mark.parametrize("variant", [0, 1])
def test_near_ties_inside_the_fp32_guard_band_are_decided_like_the_reference(variant):
    """Adversarial for the FP32 ranking of the fast search cores: two (or three) copies of the template inside the ellipse
    that differ from it in k and k + 1 pixels by one grey level.  Their reference scores are ~1e-6 .. 4e-6 apart - at or
    below the FP32 ranking error - so the rank alone could invert them; the guard band must send such searches to the
    exact FP64 path and the answer must still be the reference's (position, score and the last-wins tie rule)."""
    rng = np.random.default_rng(2024)
    W, H = 160, 120
    images, idx, patches, centres, puinv = [], [], [], [], []
    for t in range(120):
        img = (128 + 12 * rng.standard_normal((H, W))).clip(0, 255).astype(np.uint8)
        patch = (128 + 45 * rng.standard_normal((11, 11))).clip(1, 254).astype(np.uint8)
        cx, cy = int(rng.integers(40, W - 40)), int(rng.integers(40, H - 40))
        spots = [(cx - 14, cy - 3), (cx + 13, cy + 2)] + ([(cx, cy + 14)] if t % 3 == 0 else [])
        k0 = int(rng.integers(0, 4))
        for si, (x, y) in enumerate(spots):
            q = patch.astype(np.int32).copy()
            nflip = k0 + (si if t % 4 else 0)           # t % 4 == 0: EXACT ties (identical copies): last in scan order wins
            for _ in range(nflip):
                q[rng.integers(0, 11), rng.integers(0, 11)] += int(rng.choice([-1, 1]))
            img[y - 5:y + 6, x - 5:x + 6] = q.clip(0, 255).astype(np.uint8)
        a, b, c = oa.sinv_from_S(np.array([[90.0, 5.0], [5.0, 70.0]]))       # half-widths ~28 x 25: all copies inside
        images.append(img); idx.append(t); patches.append(patch.reshape(121)); centres.append([cx + 0.2, cy - 0.3]); puinv.append([a, b, c])
    images, idx, patches = np.stack(images), np.array(idx, np.int32), np.stack(patches)
    centres, puinv = np.array(centres), np.array(puinv)
    n = len(idx)
    ok = np.zeros(n, np.int32)
    uv = np.full((n, 2), -7, np.int32)
    score = np.zeros(n)
    _lib.check(_lib.load().sl2_elliptical_search_batch(0, _lib.u8p(images), n, W, H, _lib.ip(idx), _lib.u8p(patches),
                                                       _lib.dp(centres), _lib.dp(puinv), n, _lib.ip(ok), _lib.ip(uv),
                                                       _lib.dp(score), variant))
    close = 0
    for t in range(n):
        want = oa.elliptical_search(images[t], patches[t], centres[t], *puinv[t])
        assert want["ok"] and bool(ok[t]), t
        assert (uv[t, 0], uv[t, 1]) == (want["u"], want["v"]), "case %d: %s vs %s" % (t, uv[t], (want["u"], want["v"]))
        assert score[t] == want["corr"], t
        close += want["corr"] < 2e-5
    assert close >= 100          # the constructed copies really are the winners


@pytest.mark.parametrize("variant", [1, 0])
def test_search_windows_in_the_last_rows_of_the_last_image_do_not_read_past_it(variant):
    """Windows clamped into the bottom-right corner: the 16-byte staging pieces of the lean matrix-core walk would reach
    past the end of the frame there (the last image of the batch ends the allocation), so they are re-read byte by byte;
    the sigma == 10 boundary block sits in the corner of some images too.  Results must be the reference's."""
    rng = np.random.default_rng(31)
    W, H = 96, 64
    tex = synth.make_texture(size=512)
    images, idx, patches, centres, puinv = [], [], [], [], []
    for t in range(48):
        oy, ox = int(rng.integers(0, 512 - H)), int(rng.integers(0, 512 - W))
        img = tex[oy:oy + H, ox:ox + W].copy()
        cx, cy = W - 6 - int(rng.integers(0, 6)), H - 6 - int(rng.integers(0, 4))
        patch = img[cy - 5:cy + 6, cx - 5:cx + 6].copy()
        if t % 6 == 5:                                                   # a window whose sigma is exactly 10 in the very corner
            vals = np.array([60] * 21 + [61] * 50 + [81] * 50, dtype=np.uint8)
            img[H - 11:, W - 11:] = rng.permutation(vals).reshape(11, 11)
        s0, s1 = rng.uniform(4, 120, 2)
        r = rng.uniform(-0.6, 0.6) * np.sqrt(s0 * s1)
        a, b, c = oa.sinv_from_S(np.array([[s0, r], [r, s1]]))
        ce = np.array([cx + rng.uniform(-3, 8), cy + rng.uniform(-3, 8)])     # often beyond the border: clamped
        images.append(img); idx.append(t); patches.append(patch.reshape(121)); centres.append(ce); puinv.append([a, b, c])
    images, idx, patches = np.stack(images), np.array(idx, np.int32), np.stack(patches)
    centres, puinv = np.array(centres), np.array(puinv)
    n = len(idx)
    ok = np.zeros(n, np.int32)
    uv = np.full((n, 2), -7, np.int32)
    score = np.zeros(n)
    _lib.check(_lib.load().sl2_elliptical_search_batch(0, _lib.u8p(images), n, W, H, _lib.ip(idx), _lib.u8p(patches),
                                                       _lib.dp(centres), _lib.dp(puinv), n, _lib.ip(ok), _lib.ip(uv),
                                                       _lib.dp(score), variant))
    found = 0
    for t in range(n):
        want = oa.elliptical_search(images[t], patches[t], centres[t], *puinv[t])
        assert bool(ok[t]) == want["ok"], "case %d ok" % t
        assert score[t] == want["corr"], "case %d score %r vs %r" % (t, score[t], want["corr"])
        if want["corr"] < 1e6:
            assert (uv[t, 0], uv[t, 1]) == (want["u"], want["v"]), "case %d uv" % t
            found += 1
    assert found >= 30


def test_search_variant_numbers_outside_the_documented_ones_are_rejected():
    z = np.zeros(4, np.int32)
    img = np.zeros((1, 16, 16), np.uint8)
    rc = _lib.load().sl2_elliptical_search_batch(0, _lib.u8p(img), 1, 16, 16, _lib.ip(z), _lib.u8p(np.zeros(121, np.uint8)),
                                                 _lib.dp(np.zeros(2)), _lib.dp(np.ones(3)), 1, _lib.ip(z), _lib.ip(z),
                                                 _lib.dp(np.zeros(1)), 2)
    assert rc == _lib.SL2_ERR_INVALID


def test_device_renderer_matches_host_bytes():
    tex = synth.make_texture(size=512)
    cam = synth.default_camera()
    spec = synth.SequenceSpec(cam, 16, 5, synth.BASE_SEED + 3)
    host = synth.render_host(cam, tex, spec.tex_extent, spec.tex_origin, spec.poses)
    n = spec.poses.shape[0]
    d_tex = _lib.DeviceBuffer(tex.nbytes); d_tex.upload(tex)
    org = np.ascontiguousarray(np.tile(spec.tex_origin, (n, 1)))
    d_org = _lib.DeviceBuffer(org.nbytes); d_org.upload(org)
    d_pose = _lib.DeviceBuffer(spec.poses.nbytes); d_pose.upload(spec.poses)
    d_out = _lib.DeviceBuffer(host.nbytes)
    synth.render_device(cam, d_tex.ptr, 512, spec.tex_extent, d_org.ptr, d_pose.ptr, n, d_out.ptr)
    dev = d_out.download(host.shape, np.uint8)
    assert np.array_equal(dev, host), "device/host renderer differ in %d bytes" % int((dev != host).sum())
