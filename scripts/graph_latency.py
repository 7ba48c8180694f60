"""Step latency at small batches with and without whole-step HIP graphs (frames alternate between two device buffers,
as the ingest hands them out)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chol_trace import build_engine  # noqa: E402
from scenelib2_amd import _lib  # noqa: E402

for B in (1, 16, 128):
    for graph in (False, True):
        eng, step, keep = build_engine(B, 100, 320, 240, n_render=3)
        eng.set_graph_mode(graph)
        d_frames = keep[3]
        fb = 320 * 240
        ptrs = [d_frames.ptr + 1 * B * fb, d_frames.ptr + 2 * B * fb]
        for k in range(20):
            eng.go_one_step(ptrs[k & 1], on_device=True, seq_stride=fb)
        eng.synchronize()
        t0 = time.perf_counter()
        n = 400
        for k in range(n):
            eng.go_one_step(ptrs[k & 1], on_device=True, seq_stride=fb)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / n
        print("batch %4d  graph %-5s  %.1f us per step  (%.0f frames/s)" % (B, graph, dt * 1e6, B / dt))
