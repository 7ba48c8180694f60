"""The C-ABI library loads on a CPU-only machine and exports exactly what
include/scenelib2_amd.h declares; with no GPU the engine fails loudly (no CPU
fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, SHIPPED_CAM


def _declared():
    text = open(os.path.join(ROOT, "include", "scenelib2_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sl2_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from scenelib2_amd import _lib
    L = _lib.load()
    declared = _declared()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), "missing export %s" % name
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared


def test_product_library_exports_only_the_documented_surface():
    """Test hooks and micro-benchmarks live in libscenelib2_amd_test.so (include/scenelib2_amd_testing.h), not in the
    product library: its exported sl2_* symbols are exactly the header's."""
    import subprocess
    from scenelib2_amd import _lib
    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return sorted(set(re.findall(r"\b(sl2_[a-z0-9_]+)$", out, flags=re.M)))
    prod = exported(_lib.LIB_PATH)
    assert prod == _declared(), sorted(set(prod) ^ set(_declared()))
    assert not [n for n in prod if n.startswith("sl2_debug") or n in _lib.TEST_SYMBOLS]
    text = open(os.path.join(ROOT, "include", "scenelib2_amd_testing.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    hooks = sorted(set(re.findall(r"\b(sl2_[a-z0-9_]+)\s*\(", text)))
    assert hooks == sorted(_lib.TEST_SYMBOLS)
    test = exported(_lib.TEST_LIB_PATH)
    assert set(hooks) <= set(test) and set(prod) <= set(test)


def test_struct_layouts_match_header():
    from scenelib2_amd import _lib
    assert C.sizeof(_lib.sl2_camera) == 56          # 2 x i32, 5 x f64, i32 (+pad)
    assert C.sizeof(_lib.sl2_params) == 88
    assert C.sizeof(_lib.sl2_feature_info) == 8 * 4 + 8 * (3 + 2 + 2 + 2 + 1 + 4 + 14 + 6 + 7) + 2 * 4 + 8 * 3


def test_no_gpu_means_loud_failure():
    from scenelib2_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    from scenelib2_amd import Engine
    with pytest.raises(_lib.Sl2Error) as ei:
        Engine(SHIPPED_CAM, dict(delta_t=1 / 30, number_of_features_to_select=4), 1, 8)
    assert ei.value.code == _lib.SL2_ERR_NO_DEVICE
    ok = np.zeros(1, np.int32)
    uv = np.zeros(2, np.int32)
    sc = np.zeros(1)
    rc = _lib.load().sl2_elliptical_search_batch(0, _lib.u8p(np.zeros((240, 320), np.uint8)), 1, 320, 240,
                                                 _lib.ip(np.zeros(1, np.int32)), _lib.u8p(np.zeros(121, np.uint8)),
                                                 _lib.dp(np.zeros(2)), _lib.dp(np.ones(3)), 1, _lib.ip(ok),
                                                 _lib.ip(uv), _lib.dp(sc), 0)
    assert rc == _lib.SL2_ERR_NO_DEVICE
