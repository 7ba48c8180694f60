"""GPU parity of the feature-initialisation image operators against the oracle:
bit-exact (integer positions, FP64 eigenvalues and scores compared with ==)."""
import numpy as np
import pytest

import oracle_api as oa
from test_oracle_feature_init import _texture

pytestmark = pytest.mark.gpu


def test_find_best_patch_batch_matches_oracle_exactly():
    from scenelib2_amd import improc
    rng = np.random.default_rng(21)
    W, H = 320, 240
    images = np.stack([_texture(rng, H, W) for _ in range(4)] + [np.full((H, W), 90, np.uint8)])
    regions, idx = [], []
    for t in range(40):
        us, vs = rng.integers(-10, W - 60), rng.integers(-10, H - 40)
        regions.append([us, vs, us + 80, vs + 60]); idx.append(t % 4)        # the reference's 80x60 search box
    regions += [[0, 0, W, H], [100, 100, 100, 160], [W - 8, H - 8, W + 5, H + 5], [50, 50, 130, 110]]
    idx += [0, 1, 2, 4]                                                     # whole image, empty, clamped-empty, flat image
    uv_in = np.tile(np.array([[-3, -4]], np.int32), (len(regions), 1))
    uv, ev = improc.find_best_patch_batch(images, idx, regions, uv_in)
    for t, reg in enumerate(regions):
        wu, wv, wev = oa.find_best_patch(images[idx[t]], reg, (-3, -4))
        assert (uv[t, 0], uv[t, 1]) == (wu, wv) and ev[t] == wev, (t, reg, uv[t], ev[t], (wu, wv, wev))
    assert (uv[-1] == [-3, -4]).all() and ev[-1] == 0.0                      # flat image: selection untouched


def _ellipse_jobs(rng, W, H, njobs):
    images, patches, counts, pu, ce = [], [], [], [], []
    for j in range(njobs):
        img = _texture(rng, H, W)
        cx, cy = rng.integers(30, W - 30), rng.integers(30, H - 30)
        patch = img[cy - 5:cy + 6, cx - 5:cx + 6].copy()
        kind = j % 5
        if kind == 3:
            patch = rng.integers(0, 256, (11, 11)).astype(np.uint8)          # nothing matches
        if kind == 4:
            img[max(cy - 30, 0):cy + 30, max(cx - 30, 0):cx + 30] = 128      # flat image region: the +5 penalty everywhere
        n = int(rng.integers(1, 40)) if kind != 2 else 100
        ang = rng.uniform(0, np.pi)
        L = rng.uniform(10, 60)
        for t in np.linspace(-1, 1, n):
            s0, s1 = rng.uniform(4, 60), rng.uniform(4, 60)
            r = rng.uniform(-0.8, 0.8) * np.sqrt(s0 * s1)
            pu.append(oa.sinv_from_S(np.array([[s0, r], [r, s1]])))
            ce.append([cx + L * t * np.cos(ang) + rng.uniform(0, 1), cy + L * t * np.sin(ang) + rng.uniform(0, 1)])
        if kind == 1:   # ellipses hanging over the image border
            ce[-1] = [rng.uniform(-6, 4), rng.uniform(-6, 4)]
            ce[-n] = [W + rng.uniform(-4, 6), H + rng.uniform(-4, 6)]
        images.append(img); patches.append(patch.reshape(121)); counts.append(n)
    return np.stack(images), np.stack(patches), np.array(counts, np.int32), np.array(pu), np.array(ce)


def test_multi_ellipse_search_matches_oracle_exactly():
    from scenelib2_amd import improc
    rng = np.random.default_rng(33)
    W, H = 160, 120
    images, patches, counts, pu, ce = _ellipse_jobs(rng, W, H, 15)
    res, corr = improc.search_multiple_overlapping_ellipses_batch(images, np.arange(len(counts)), patches, counts, pu, ce)
    first = np.concatenate([[0], np.cumsum(counts)])
    n_found = 0
    for j in range(len(counts)):
        sl = slice(first[j], first[j + 1])
        want, wcorr, _ = oa.search_multiple_ellipses(images[j], patches[j], pu[sl], ce[sl])
        assert (res[sl] == want).all(), (j, res[sl][(res[sl] != want).any(axis=1)], want[(res[sl] != want).any(axis=1)])
        assert (corr[sl] == wcorr).all(), j
        n_found += int(want[:, 0].sum())
    assert n_found > 20


def test_multi_ellipse_class_mirror_and_empty_job():
    from scenelib2_amd import improc
    rng = np.random.default_rng(4)
    img = _texture(rng, 120, 160)
    patch = img[50:61, 60:71].copy()
    s = improc.SearchMultipleOverlappingEllipses(img, patch, 11)
    for dx in (-6.0, 0.3, 5.5):
        S = np.array([[25.0, 3.0], [3.0, 16.0]])
        a, b, c = oa.sinv_from_S(S)
        s.add_ellipse(np.array([[a, b], [b, c]]), [65 + dx, 55.4])
    s.search()
    want, wcorr, _ = oa.search_multiple_ellipses(img, patch, np.array(s._pu), np.array(s._ce))
    assert [int(f) for f in s.result_flag_] == list(want[:, 0]) and s.result_u_ == list(want[:, 1]) and s.result_v_ == list(want[:, 2])
    assert s.result_flag_[1] and (s.result_u_[1], s.result_v_[1]) == (65, 55)
    # a job without ellipses in the middle of a batch
    res, corr = improc.search_multiple_overlapping_ellipses_batch(img[None], [0, 0, 0], np.stack([patch.reshape(121)] * 3), [1, 0, 2],
                                                                 np.array(s._pu), np.array(s._ce))
    assert (res == want).all()


def test_multi_ellipse_search_small_and_frame_sized_unions():
    """Both forms of the search: a job whose union fits the one-workgroup LDS form (a dozen small ellipses), and jobs whose
    ellipses are as large as the frame (what a freshly created feature under a weak pose estimate produces) - those are
    spread over the k_me_big_* kernels.  Same results, exactly."""
    from scenelib2_amd import improc
    rng = np.random.default_rng(91)
    W, H = 320, 240
    images, patches, counts, pu, ce = [], [], [], [], []
    for j, (n, sig) in enumerate([(12, 5.0), (95, 9000.0), (40, 2500.0), (7, 3.0), (64, 20000.0)]):
        img = _texture(rng, H, W)
        cx, cy = 160 + 10 * j, 120 - 5 * j
        images.append(img); patches.append(img[cy - 5:cy + 6, cx - 5:cx + 6].reshape(121).copy()); counts.append(n)
        for t in range(n):
            r = 0.3 * sig
            pu.append(oa.sinv_from_S(np.array([[sig * (1 + 0.01 * t), r], [r, sig * (1.2 - 0.002 * t)]])))
            ce.append([cx + 0.7 * t + 0.25, cy - 0.4 * t + 0.6])
    images, patches, counts, pu, ce = np.stack(images), np.stack(patches), np.array(counts, np.int32), np.array(pu), np.array(ce)
    res, corr = improc.search_multiple_overlapping_ellipses_batch(images, np.arange(len(counts)), patches, counts, pu, ce)
    first = np.concatenate([[0], np.cumsum(counts)])
    for j in range(len(counts)):
        sl = slice(first[j], first[j + 1])
        want, wcorr, ncorr = oa.search_multiple_ellipses(images[j], patches[j], pu[sl], ce[sl])
        assert (res[sl] == want).all() and (corr[sl] == wcorr).all(), j
        assert (ncorr > 2048) == (j in (1, 2, 4)), (j, ncorr)      # which form each job took (kMeCap positions of bounding box)
    # a second call on the same engine-side maps must start clean (stamps cleared by the big form)
    res2, corr2 = improc.search_multiple_overlapping_ellipses_batch(images, np.arange(len(counts)), patches, counts, pu, ce)
    assert (res2 == res).all() and (corr2 == corr).all()
