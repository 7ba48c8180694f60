"""Oracle checks for the feature-initialisation image operators — CPU only.
The reference ships no vectors for them; they are pinned by independent restatements of the same
mathematics and by the libc generator itself (drand48)."""
import ctypes

import numpy as np

import oracle_api as oa


def _texture(rng, H, W):
    img = rng.integers(0, 256, (H, W)).astype(np.float64)
    k = np.array([1, 4, 6, 4, 1.0]); k /= k.sum()
    for ax in (0, 1):
        img = np.apply_along_axis(lambda m: np.convolve(m, k, mode="same"), ax, img)
    img = (img - img.min()) / (img.max() - img.min()) * 255
    return img.astype(np.uint8)


def _detector_direct(img, region):
    """Independent restatement: integer box sums of the doubled gradients, argmax of the smaller eigenvalue."""
    H, W = img.shape
    us, vs, uf, vf = region
    us, uf, vs, vf = max(us, 6), min(uf, W - 6), max(vs, 6), min(vf, H - 6)
    if vs >= vf or us >= uf:
        return us, vs, 0.0, True
    I = img.astype(np.int64)
    gx2 = np.zeros_like(I); gy2 = np.zeros_like(I)
    gx2[:, 1:-1] = I[:, 2:] - I[:, :-2]
    gy2[1:-1, :] = I[2:, :] - I[:-2, :]
    best, bu, bv = 0.0, None, None
    for v in range(vs, vf):
        for u in range(us, uf):
            wx = gx2[v - 5:v + 6, u - 5:u + 6]; wy = gy2[v - 5:v + 6, u - 5:u + 6]
            A, B, Cc = (wx * wx).sum() / 4.0, (wx * wy).sum() / 4.0, (wy * wy).sum() / 4.0
            e2 = (A + Cc - np.sqrt((A + Cc) * (A + Cc) - 4 * (A * Cc - B * B))) / 2.0
            if e2 > best:
                best, bu, bv = e2, u, v
    return bu, bv, best, False


def test_detector_matches_direct_integer_sums():
    rng = np.random.default_rng(5)
    img = _texture(rng, 96, 128)
    for region in [(20, 20, 60, 50), (0, 0, 128, 96), (100, 70, 140, 110), (30, 40, 31, 41), (50, 50, 50, 70), (-5, -5, 12, 12)]:
        u, v, ev = oa.find_best_patch(img, region, (-3, -4))
        wu, wv, wev, empty = _detector_direct(img, region)
        if wu is None:
            assert (u, v, ev) == (-3, -4, 0.0)
        else:
            assert (u, v) == (wu, wv) and ev == wev, (region, (u, v, ev), (wu, wv, wev))


def test_detector_flat_image_leaves_selection_untouched():
    img = np.full((60, 80), 77, np.uint8)
    assert oa.find_best_patch(img, (10, 10, 70, 50), (-9, -8)) == (-9, -8, 0.0)


def test_multi_ellipse_equals_independent_scans_and_caches():
    rng = np.random.default_rng(11)
    W, H = 160, 120
    img = _texture(rng, H, W)
    patch = img[40:51, 70:81].copy()
    # a string of overlapping ellipses along a line through the true location (75, 45)
    pu, ce = [], []
    for t in np.linspace(-1, 1, 12):
        S = np.array([[30.0 + 20 * t * t, 6.0], [6.0, 18.0]])
        pu.append(oa.sinv_from_S(S)); ce.append([75 + 18 * t + 0.7, 45 + 7 * t + 0.2])
    pu, ce = np.array(pu), np.array(ce)
    res, corr, ncorr = oa.search_multiple_ellipses(img, patch, pu, ce)
    visited = set()
    for i in range(len(pu)):
        a, b, c = pu[i]
        hw = int(3.0 / np.sqrt(a - b * b / c)); hh = int(3.0 / np.sqrt(c - b * b / a))
        uc, vc = int(ce[i, 0]), int(ce[i, 1])
        best, bu, bv = 1e6, 0, 0
        for ur in range(max(-hw, 5 - uc), min(hw, W - 11 - uc + 5) + 1):
            for vr in range(max(-hh, 5 - vc), min(hh, H - 11 - vc + 5) + 1):
                if a * ur * ur + 2 * b * ur * vr + c * vr * vr < 9.0:
                    visited.add((uc + ur, vc + vr))
                    sc, sd0, sd1 = oa.correlate2_warning(patch, img, uc + ur - 5, vc + vr - 5)
                    if sd1 < 10.0:
                        sc += 5.0
                    if sc <= best:
                        best, bu, bv = sc, uc + ur, vc + vr
        assert corr[i] == best and (res[i, 1], res[i, 2]) == (bu, bv) and bool(res[i, 0]) == (not best > 0.40), i
    assert ncorr == len(visited)          # every position of the union correlated exactly once
    assert res[:, 0].any() and (res[res[:, 0] == 1, 1:] == [75, 45]).all()


def test_drand48_restatement_equals_libc():
    libc = ctypes.CDLL("libc.so.6")
    libc.drand48.restype = ctypes.c_double
    for seed in (0, 1, 12345):
        libc.srand48(seed)
        want = [libc.drand48() for _ in range(64)]
        assert list(oa.drand48_sequence(seed, 64)) == want
