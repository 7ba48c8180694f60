"""Deterministic synthetic MonoSLAM sequences.

Scene: fronto-parallel textured plane z = 0; camera starts at r = (0, 0, -depth)
looking along +z with q = (1,0,0,0) — the geometry of the reference's shipped
target (data/SceneLib2.cfg:71-83, 267-313).  Known features are points of the
plane on a jittered pixel grid of the t = 0 view; their 11x11 templates are cut
from the rendered t = 0 frame; the camera follows a smooth path whose velocities
are sums of two sinusoids per axis (well inside the filter's sigma_a = 4,
sigma_alpha = 6 noise model, motion_model.cpp:45).

Everything that decides BYTES is either integer arithmetic in numpy (texture) or
runs in the native renderer (sl2_synth.hpp, identical source for host and device),
so the oracle and the engine consume identical frames.  One texture is shared by
all sequences of a run; a sequence's seed (base_seed + index) drives its texture
offset, feature jitter and camera path.
"""
import ctypes as C

import numpy as np

from . import _lib

BASE_SEED = 20240925
TEX_SIZE = 1024


def _splitmix64(seed, n):
    """n 64-bit outputs of splitmix64 started at `seed` (vectorised, wrap-around uint64)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, n):
    return (_splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_texture(seed=BASE_SEED, size=TEX_SIZE):
    """Band-limited noise: uniform bytes, 5x5 box sum twice (torus, exact integers), then an
    affine map to mean 128 / sigma 40, rounded and clamped to u8."""
    raw = (_splitmix64(seed, size * size) >> np.uint64(56)).astype(np.int64).reshape(size, size)

    def box5(a):
        s = np.zeros_like(a)
        for d in (-2, -1, 0, 1, 2):
            s += np.roll(a, d, axis=1)
        t = np.zeros_like(a)
        for d in (-2, -1, 0, 1, 2):
            t += np.roll(s, d, axis=0)
        return t

    b = box5(box5(raw)).astype(np.float64)
    b = (b - b.mean()) * (40.0 / b.std()) + 128.0
    return np.clip(np.floor(b + 0.5), 0, 255).astype(np.uint8)


def default_camera(width=320, height=240):
    """The shipped camera (data/SceneLib2.cfg:24-31) scaled with resolution."""
    s = width / 320.0
    return dict(width=int(width), height=int(height), fku=float(int(195 * s)), fkv=float(int(195 * s)),
                u0=float(int(162 * s)), v0=float(int(125 * height / 240.0)), kd1=9e-06 / (s * s), sd=1)


def default_params(n_select, delta_t=1.0 / 30.0):
    return dict(delta_t=delta_t, number_of_features_to_select=int(n_select), number_of_features_to_keep_visible=12,
                max_features_to_init_at_once=1, min_lambda=0.5, max_lambda=5.0, number_of_particles=100,
                standard_deviation_depth_ratio=0.3, min_number_of_particles=20, prune_probability_threshold=0.05,
                erase_partially_init_feature_after_this_many_attempts=10,
                minimum_attempted_measurements_of_feature=10, successful_match_fraction=0.5)


def _quat_mul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def _quat_from_rotvec(a):
    ang = np.sqrt(a @ a)
    if ang > 0.0:
        s = np.sin(ang / 2.0) / ang
        return np.array([np.cos(ang / 2.0), s * a[0], s * a[1], s * a[2]])
    return np.array([1.0, 0.0, 0.0, 0.0])


class SequenceSpec:
    """Ground truth + filter initialisation of one synthetic sequence."""

    def __init__(self, cam, n_features, n_frames, seed, delta_t=1.0 / 30.0, depth=0.6, tex_extent=2.0,
                 v_amp=0.06, w_amp=0.08, w_bias=(0.0, 0.0, 0.01), v_bias=(0.0, 0.0, 0.0)):
        self.cam = dict(cam)
        self.n_features = n_features
        self.n_frames = n_frames
        self.seed = int(seed)
        self.delta_t = delta_t
        self.tex_extent = tex_extent
        u = uniform01(self.seed * 7919 + 17, 64)
        self.tex_origin = (u[0:2] - 0.5) * tex_extent
        # velocity = sum of two sinusoids per axis: amp * sin(2 pi f t + phase)
        self.v_amp = v_amp * (0.5 + 0.5 * u[2:8].reshape(2, 3))
        self.v_freq = 0.2 + 0.4 * u[8:14].reshape(2, 3)
        self.v_phase = 2 * np.pi * u[14:20].reshape(2, 3)
        self.w_amp = w_amp * (0.5 + 0.5 * u[20:26].reshape(2, 3))
        self.w_freq = 0.2 + 0.4 * u[26:32].reshape(2, 3)
        self.w_phase = 2 * np.pi * u[32:38].reshape(2, 3)
        # omega(0) != 0 (Q10), like data/SceneLib2.cfg:83.  A larger z component rolls the camera about its optical axis:
        # the features stay visible while their (unwarped) 11x11 templates stop matching - the natural way a feature
        # earns delete_bad_features (monoslam.cpp:644-660); v_bias drifts the camera (features leave the view)
        self.w_bias = np.asarray(w_bias, dtype=np.float64)
        self.v_bias = np.asarray(v_bias, dtype=np.float64)
        self.r0 = np.array([0.0, 0.0, -depth])
        self.poses = self._integrate_path()
        self.xv0 = np.concatenate([self.poses[0], self.velocity(0.0), self.omega(0.0)])
        self.Pxx0 = np.zeros((13, 13))
        self.Pxx0[0, 0] = self.Pxx0[1, 1] = self.Pxx0[2, 2] = 0.0004  # data/SceneLib2.cfg:85-113
        self.feat_px, self.feat_y = self._place_features(u[40:42])

    def velocity(self, t):
        return (self.v_amp * np.sin(2 * np.pi * self.v_freq * t + self.v_phase)).sum(axis=0) + self.v_bias

    def omega(self, t):
        return (self.w_amp * np.sin(2 * np.pi * self.w_freq * t + self.w_phase)).sum(axis=0) + self.w_bias

    def _position(self, t):
        w = 2 * np.pi * self.v_freq
        integ = self.v_amp / w * (np.cos(self.v_phase) - np.cos(w * t + self.v_phase))
        return self.r0 + integ.sum(axis=0) + self.v_bias * t

    def _integrate_path(self, substeps=8):
        """poses[k] = (r, q) at t = k * delta_t, k = 0..n_frames (frame k of a run is pose k+1)."""
        poses = np.zeros((self.n_frames + 1, 7))
        q = np.array([1.0, 0.0, 0.0, 0.0])
        h = self.delta_t / substeps
        for k in range(self.n_frames + 1):
            poses[k, :3] = self._position(k * self.delta_t)
            poses[k, 3:] = q
            for s in range(substeps):
                tm = k * self.delta_t + (s + 0.5) * h
                q = _quat_mul(q, _quat_from_rotvec(self.omega(tm) * h))
            q = q / np.sqrt(q @ q)
        return poses

    def _place_features(self, jit_seed):
        cam = self.cam
        W, H, N = cam["width"], cam["height"], self.n_features
        border = 32
        cols = int(np.ceil(np.sqrt(N * (W - 2.0 * border) / (H - 2.0 * border))))
        rows = int(np.ceil(N / cols))
        j = uniform01(self.seed * 104729 + 5, 2 * rows * cols).reshape(rows * cols, 2) - 0.5
        cw, ch = (W - 2.0 * border) / cols, (H - 2.0 * border) / rows
        px = np.zeros((N, 2), dtype=np.int64)
        for i in range(N):
            r, c = divmod(i, cols)
            px[i, 0] = int(border + (c + 0.5 + 0.5 * j[i, 0]) * cw)
            px[i, 1] = int(border + (r + 0.5 + 0.5 * j[i, 1]) * ch)
        # world point of each pixel centre under the t = 0 pose (Unproject, camera.cpp:133-154)
        y = np.zeros((N, 3))
        r0, q0 = self.poses[0, :3], self.poses[0, 3:]
        assert np.allclose(q0, [1, 0, 0, 0])
        for i in range(N):
            c0, c1 = px[i, 0] - cam["u0"], px[i, 1] - cam["v0"]
            factor = np.sqrt(1 - 2 * cam["kd1"] * (c0 * c0 + c1 * c1))
            ray = np.array([(c0 / factor) / -cam["fku"], (c1 / factor) / -cam["fkv"], 1.0])
            t = -r0[2] / ray[2]
            y[i] = [r0[0] + t * ray[0], r0[1] + t * ray[1], 0.0]
        return px, y

    def xp_org(self):
        return np.tile(self.poses[0], (self.n_features, 1))


def render_host(cam, tex, tex_extent, tex_origin, poses):
    """poses [count][7] -> uint8 [count][H][W] with the native host renderer."""
    L = _lib.load()
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
    count = poses.shape[0]
    org = np.ascontiguousarray(np.broadcast_to(np.asarray(tex_origin, dtype=np.float64), (count, 2)))
    out = np.zeros((count, cam["height"], cam["width"]), dtype=np.uint8)
    c = _lib.make_camera(cam)
    t = np.ascontiguousarray(tex, dtype=np.uint8)
    _lib.check(L.sl2_synth_render_host(C.byref(c), _lib.u8p(t), t.shape[0], float(tex_extent), _lib.dp(org),
                                       _lib.dp(poses), count, _lib.u8p(out)))
    return out


def render_device(cam, tex_dev_ptr, tex_size, tex_extent, tex_origin_dev_ptr, poses_dev_ptr, count, out_dev_ptr,
                  device=0, stream=None):
    L = _lib.load()
    c = _lib.make_camera(cam)
    _lib.check(L.sl2_synth_render_device(device, _lib.vp(stream) if stream else None, C.byref(c), _lib.vp(tex_dev_ptr),
                                         tex_size, float(tex_extent), _lib.vp(tex_origin_dev_ptr),
                                         _lib.vp(poses_dev_ptr), count, _lib.vp(out_dev_ptr)))


def cut_templates(frame0, feat_px):
    """11x11 crops of the t = 0 frame centred on each feature pixel."""
    N = feat_px.shape[0]
    out = np.zeros((N, 11, 11), dtype=np.uint8)
    for i in range(N):
        u, v = int(feat_px[i, 0]), int(feat_px[i, 1])
        out[i] = frame0[v - 5:v + 6, u - 5:u + 6]
    return out


def make_sequence(cam, n_features, n_frames, seq_index=0, base_seed=BASE_SEED, tex=None, render_frames=True,
                  **kw):
    """One complete host-side sequence: spec, templates and (optionally) frames [n_frames][H][W]."""
    if tex is None:
        tex = make_texture(base_seed)
    spec = SequenceSpec(cam, n_features, n_frames, base_seed + seq_index, **kw)
    frame0 = render_host(cam, tex, spec.tex_extent, spec.tex_origin, spec.poses[0:1])[0]
    templates = cut_templates(frame0, spec.feat_px)
    frames = None
    if render_frames:
        frames = render_host(cam, tex, spec.tex_extent, spec.tex_origin, spec.poses[1:])
    return spec, templates, frames, frame0
