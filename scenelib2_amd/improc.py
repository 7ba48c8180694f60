"""Host-side mirror of the reference's feature-initialisation image operators,
thin wrappers over the C ABI (include/scenelib2_amd.h) — no computation happens here.

  find_best_patch_inside_region   MonoSLAM::find_best_patch_inside_region   monoslam.cpp:1070-1192
  SearchMultipleOverlappingEllipses                       improc/search_multiple_overlapping_ellipses.{h,cpp}
"""
import ctypes as C

import numpy as np

from . import _lib


def find_best_patch_batch(images, image_index, regions, uv_in=None, device=0, want_ms=False):
    """Shi-Tomasi detector over regions [n][4] = (ustart, vstart, ufinish, vfinish) of images[image_index[j]].
    Returns (uv [n][2], evbest [n]) (+ kernel milliseconds)."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    if images.ndim == 2:
        images = images[None]
    idx = np.ascontiguousarray(image_index, dtype=np.int32)
    reg = np.ascontiguousarray(regions, dtype=np.int32).reshape(-1, 4)
    n = reg.shape[0]
    uv = np.full((n, 2), -1, np.int32) if uv_in is None else np.ascontiguousarray(uv_in, dtype=np.int32).reshape(n, 2).copy()
    ev = np.zeros(n)
    ms = C.c_double(0)
    _lib.check(_lib.load().sl2_find_best_patch_batch(device, _lib.u8p(images), images.shape[0], images.shape[2], images.shape[1],
                                                     n, _lib.ip(idx), _lib.ip(reg), _lib.ip(uv), _lib.dp(ev), C.byref(ms)))
    return (uv, ev, ms.value) if want_ms else (uv, ev)


def find_best_patch_inside_region(image, ustart, vstart, ufinish, vfinish, ubest=-1, vbest=-1, device=0):
    """One region of one image; returns (ubest, vbest, evbest) like the reference's out-parameters."""
    uv, ev = find_best_patch_batch(np.asarray(image)[None], [0], [[ustart, vstart, ufinish, vfinish]], [[ubest, vbest]], device)
    return int(uv[0, 0]), int(uv[0, 1]), float(ev[0])


class SearchMultipleOverlappingEllipses:
    """Same call sequence as the reference class: construct with (image, patch, BOXSIZE), add_ellipse(PuInv, centre) per
    particle, search(), then read result_flag_ / result_u_ / result_v_ per ellipse."""

    def __init__(self, image, patch, boxsize=11, device=0):
        if boxsize != 11:
            raise ValueError("the engine is built for the reference's 11x11 patches (kBoxSize_)")
        self.image = np.ascontiguousarray(image, dtype=np.uint8)
        self.patch = np.ascontiguousarray(patch, dtype=np.uint8).reshape(121)
        self.device = device
        self._pu, self._ce = [], []
        self.result_flag_, self.result_u_, self.result_v_, self.corrmax = [], [], [], []

    def add_ellipse(self, PuInv, search_centre):
        P = np.asarray(PuInv, dtype=np.float64)
        self._pu.append([P[0, 0], P[0, 1], P[1, 1]])
        self._ce.append([float(search_centre[0]), float(search_centre[1])])

    def size(self):
        return len(self._pu)

    def search(self):
        res, corr = search_multiple_overlapping_ellipses_batch(self.image[None], [0], self.patch[None], [len(self._pu)],
                                                               np.array(self._pu).reshape(-1, 3), np.array(self._ce).reshape(-1, 2),
                                                               device=self.device)
        self.result_flag_ = [bool(r[0]) for r in res]
        self.result_u_ = [int(r[1]) for r in res]
        self.result_v_ = [int(r[2]) for r in res]
        self.corrmax = list(corr)


def search_multiple_overlapping_ellipses_batch(images, image_index, patches, ellipse_count, puinv, centre, device=0, want_ms=False):
    """Batch form: job j = (images[image_index[j]], patches[j]) with ellipse_count[j] ellipses, concatenated in
    puinv [total][3] / centre [total][2].  Returns (result [total][3] = flag, u, v; corrmax [total]) (+ kernel ms)."""
    images = np.ascontiguousarray(images, dtype=np.uint8)
    idx = np.ascontiguousarray(image_index, dtype=np.int32)
    pat = np.ascontiguousarray(patches, dtype=np.uint8).reshape(-1, 121)
    cnt = np.ascontiguousarray(ellipse_count, dtype=np.int32)
    pu = np.ascontiguousarray(puinv, dtype=np.float64).reshape(-1, 3)
    ce = np.ascontiguousarray(centre, dtype=np.float64).reshape(-1, 2)
    total = int(cnt.sum())
    assert pu.shape[0] == total and ce.shape[0] == total and pat.shape[0] == cnt.shape[0] == idx.shape[0]
    res = np.zeros((total, 3), np.int32)
    corr = np.zeros(total)
    ms = C.c_double(0)
    _lib.check(_lib.load().sl2_search_multiple_overlapping_ellipses_batch(
        device, _lib.u8p(images), images.shape[0], images.shape[2], images.shape[1], cnt.shape[0], _lib.ip(idx), _lib.u8p(pat),
        _lib.ip(cnt), _lib.dp(pu), _lib.dp(ce), _lib.ip(res), _lib.dp(corr), C.byref(ms)))
    return (res, corr, ms.value) if want_ms else (res, corr)
