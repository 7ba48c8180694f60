import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from scenelib2_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests never silently pass without a device: they are skipped with a reason.
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle_api
    oracle_api.build()
    oracle_api.lib()
    return oracle_api


@pytest.fixture(scope="session")
def devmath():
    """Device scalar math (sl2_math.hpp) compiled for the host — formula checks only."""
    import ctypes as C
    bdir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(bdir, exist_ok=True)
    so = os.path.join(bdir, "libdevmath_host.so")
    src = os.path.join(ROOT, "tests", "device_math_host.cpp")
    hdr = os.path.join(ROOT, "scenelib2_amd", "csrc", "sl2_math.hpp")
    if (not os.path.exists(so)) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-o", so, src])
    L = C.CDLL(so)
    L.dm_ncc_score.restype = C.c_double
    L.dm_ncc_score.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_double)] * 2
    L.dm_in_ellipse.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    L.dm_search_bounds.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.POINTER(C.c_int)]
    L.dm_search_scan.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_double),
                                 C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                 C.POINTER(C.c_int)]
    L.dm_motion.argtypes = [C.POINTER(C.c_double), C.c_double] + [C.POINTER(C.c_double)] * 3
    L.dm_motion_repeated.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_int] + [C.POINTER(C.c_double)] * 2
    L.dm_predict_cov.argtypes = [C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dm_innovation_cov.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double] + \
        [C.POINTER(C.c_double)] * 4
    return L


SHIPPED_CAM = dict(width=320, height=240, fku=195.0, fkv=195.0, u0=162.0, v0=125.0, kd1=9e-06, sd=1)
SHIPPED_XV = np.array([0, 0, -0.6, 1, 0, 0, 0, 0, 0, -0.1, 0, 0, 0.01], dtype=np.float64)
SHIPPED_Y = np.array([[0.105, 0.07425, 0.0], [-0.105, 0.07425, 0.0], [0.105, -0.07425, 0.0],
                      [-0.105, -0.07425, 0.0]])
SHIPPED_DT = 0.033333333


def shipped_Pxx():
    P = np.zeros((13, 13))
    P[0, 0] = P[1, 1] = P[2, 2] = 0.0004
    return P


def golden_path(name):
    return os.path.join(ROOT, "tests", "golden", name)


def shipped_patches():
    from scenelib2_amd.config import read_pgm
    return [read_pgm(golden_path("known_patch%d.pgm" % i)) for i in range(4)]


def rel_fro(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
