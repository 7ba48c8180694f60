"""scenelib2_amd — MI355X-native MonoSLAM per-frame engine (drop-in for the
GoOneStep / Kalman / elliptical-search path of hanmekim/SceneLib2).

The compute path is the HIP library scenelib2_amd/libscenelib2_amd.so (C ABI:
include/scenelib2_amd.h).  This package is the thin host mirror of the
reference's interface plus the synthetic-input generator used by tests and bench.
"""
from . import _lib  # noqa: F401
from .config import load_config, parse_vars_file, read_pgm  # noqa: F401
from .monoslam import Engine, Feature, MonoSLAM  # noqa: F401
from . import improc  # noqa: F401

__all__ = ["Engine", "MonoSLAM", "Feature", "load_config", "parse_vars_file", "read_pgm"]
