"""Development aid: per-wave cycle stamps inside k_chol_left (needs the SL2_CHOL_TRACE build:
   make -C scenelib2_amd/csrc trace ; SL2_LIB_PATH=scenelib2_amd/libscenelib2_amd_trace.so python scripts/chol_trace.py)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenelib2_amd import Engine, _lib, synth  # noqa: E402


def build_engine(B, N, W, H, n_render=6, feature_sigma=0.005, dev=0):
    """The bench.py workload: frames rendered on the device, templates cut from the t = 0 views."""
    cam = synth.default_camera(W, H)
    params = synth.default_params(N)
    tex = synth.make_texture()
    specs = [synth.SequenceSpec(cam, N, n_render, synth.BASE_SEED + i) for i in range(B)]
    fb = W * H
    d_tex = _lib.DeviceBuffer(tex.nbytes, dev); d_tex.upload(tex)
    poses = np.ascontiguousarray(np.stack([s.poses for s in specs], axis=1))
    origins = np.ascontiguousarray(np.tile(np.stack([s.tex_origin for s in specs])[None], (n_render + 1, 1, 1)))
    d_pose = _lib.DeviceBuffer(poses.nbytes, dev); d_pose.upload(poses)
    d_org = _lib.DeviceBuffer(origins.nbytes, dev); d_org.upload(origins)
    d_frames = _lib.DeviceBuffer((n_render + 1) * B * fb, dev)
    synth.render_device(cam, d_tex.ptr, tex.shape[0], specs[0].tex_extent, d_org.ptr, d_pose.ptr, (n_render + 1) * B,
                        d_frames.ptr, device=dev)
    frame0 = d_frames.download((B, H, W), np.uint8)
    templates = np.stack([synth.cut_templates(frame0[b], specs[b].feat_px) for b in range(B)])
    eng = Engine(cam, params, B, N, device=dev)
    eng.set_vehicle_state(np.stack([s.xv0 for s in specs]), np.stack([s.Pxx0 for s in specs]))
    eng.add_known_features(np.stack([s.feat_y for s in specs]), np.stack([s.xp_org() for s in specs]), templates)
    if feature_sigma > 0.0:
        eng.set_feature_covariances(np.tile(np.eye(3) * feature_sigma ** 2, (B, N, 1, 1)))
    eng.synchronize()
    keep = (d_tex, d_pose, d_org, d_frames)

    def step(k):
        eng.go_one_step(d_frames.ptr + (k + 1) * B * fb, on_device=True, seq_stride=fb)
    return eng, step, keep


def main():
    B = int(os.environ.get("TRACE_B", "1024"))
    eng, step, keep = build_engine(B, 100, 320, 240)
    L = eng.L
    L.sl2_debug_chol_trace.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_size_t]
    for it in range(3):
        step(it)
    eng.synchronize()
    n = B * 4 * 8 * 4 + B * 4
    out = np.zeros(n, dtype=np.int64)
    L.sl2_debug_chol_trace(eng.h, out.ctypes.data_as(C.POINTER(C.c_longlong)), n)   # allocate
    step(3)
    eng.synchronize()
    L.sl2_debug_chol_trace(eng.h, out.ctypes.data_as(C.POINTER(C.c_longlong)), n)
    tr = out[:B * 128].reshape(B, 4, 8, 4)
    hw = out[B * 128:].reshape(B, 4)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    print("SIMD ids of waves 0..3, first 12 sequences:\n", simd[:12])
    print("histogram of D-wave SIMD:", np.bincount(simd[:, 0], minlength=4), " M0:", np.bincount(simd[:, 1], minlength=4))
    t0 = tr[:, 0, 0, 0][:, None, None, None]
    rel = (tr - t0).astype(np.float64)
    valid = tr[:, 0, :, 1] != 0
    for b in (0, B // 2):
        print("sequence", b, "cu", cu[b], "stamps relative to D start (cycles): rows = J, cols = [loop top, before A, after A, before B]")
        for w in range(4):
            print(" wave", w)
            print(np.array2string(rel[b, w].astype(np.int64), max_line_width=150))
    if os.environ.get("SL2_TRACE_LEFT", "1") == "1":
        # k_chol_left stamps: [loop top, before X1, before X2, before X3]
        np.set_printoptions(precision=0, suppress=True)
        for w, nm in ((0, "D "), (1, "M0"), (2, "M1"), (3, "M2")):
            t = tr[:, w].astype(np.float64)
            ok = tr[:, 0, :, 1] != 0
            p1 = np.where(ok, t[:, :, 1] - t[:, :, 0], np.nan).mean(0)
            p2 = np.where(ok, t[:, :, 2] - t[:, :, 1], np.nan).mean(0)
            p3 = np.where(ok, t[:, :, 3] - t[:, :, 2], np.nan).mean(0)
            print(nm, "P2 (loop top -> before X2):", (p1 + p2)[:7], " wait X2 + P3 (-> before X3):", p3[:7])
        col = np.where(tr[:, 0, 1:, 0] != 0, tr[:, 0, 1:, 0] - tr[:, 0, :-1, 0], 0).astype(np.float64)
        print("D wave, loop top to loop top per column:", col.mean(0)[:7])
        tot = (tr[:, 0, :, 3].max(1) - tr[:, 0, 0, 0]).astype(np.float64)
        print("mean total cycles per sequence:", tot.mean(), "max", tot.max())
        return
    # averages over sequences
    dfac = (tr[:, 0, :, 1] - tr[:, 0, :, 0]).astype(np.float64)          # D: factor duration
    m0 = (tr[:, 1, :, 3] - tr[:, 1, :, 2]).astype(np.float64)            # M0: A..B work
    m1 = (tr[:, 2, :, 3] - tr[:, 2, :, 2]).astype(np.float64)
    nxt = np.zeros_like(dfac)
    nxt[:, :-1] = (tr[:, 1, 1:, 1] - tr[:, 1, :-1, 3]).astype(np.float64)  # M0: B_J .. before A_{J+1} (trailing)
    np.set_printoptions(precision=0, suppress=True)
    print("mean D factor cycles per J       :", np.where(valid, dfac, np.nan).mean(0))
    print("mean M0 A->B cycles per J        :", np.where(valid, m0, np.nan).mean(0))
    print("mean M1 A->B cycles per J        :", np.where(valid, m1, np.nan).mean(0))
    print("mean M0 trailing (B_J..A_J+1)    :", np.where(valid, nxt, np.nan).mean(0))
    tot = (tr[:, 0, :, 1].max(1) - tr[:, 0, 0, 0]).astype(np.float64)
    print("mean total cycles per sequence   :", tot.mean(), " max", tot.max())


if __name__ == "__main__":
    main()
