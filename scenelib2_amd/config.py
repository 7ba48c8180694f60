"""Reader for SceneLib2-style configuration files and P5 PGM templates.

The reference parses `name = value;` lines with `#` comments through
pangolin::ParseVarsFile and reads each scalar with pangolin::Var<T>(key, default)
(monoslam.cpp:1578-1846); keys it does not find default to 0 / "empty".  The
same keys are honoured here (data/SceneLib2.cfg:22-313), so a reference cfg file
can be passed unchanged.
"""
import os

import numpy as np


def parse_vars_file(path):
    """`name = value;` pairs -> dict of strings (later assignments win)."""
    out = {}
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.split("#", 1)[0].strip()
            if not line or "=" not in line:
                continue
            key, val = line.split("=", 1)
            val = val.strip()
            if val.endswith(";"):
                val = val[:-1]
            out[key.strip()] = val.strip()
    return out


def _f(d, k, default=0.0):
    return float(d[k]) if k in d else default


def _i(d, k, default=0):
    # Var<int> in the reference: fku, fkv, u0, v0, sd are truncated to int (Q15)
    return int(float(d[k])) if k in d else default


def load_config(path):
    """Returns dict(cam, params, xv, Pxx, features=[dict(y, xp_org, identifier)], input)."""
    d = parse_vars_file(path)
    cam = dict(width=_i(d, "cam.width"), height=_i(d, "cam.height"), fku=_i(d, "cam.fku"), fkv=_i(d, "cam.fkv"),
               u0=_i(d, "cam.u0"), v0=_i(d, "cam.v0"), kd1=_f(d, "cam.kd1"), sd=_i(d, "cam.sd"))
    params = dict(
        delta_t=_f(d, "params.delta_t"),
        number_of_features_to_select=_i(d, "params.number_of_features_to_select"),
        number_of_features_to_keep_visible=_i(d, "params.number_of_features_to_keep_visible"),
        max_features_to_init_at_once=_i(d, "params.max_features_to_init_at_once"),
        min_lambda=_f(d, "params.min_lambda"), max_lambda=_f(d, "params.max_lambda"),
        number_of_particles=_i(d, "params.number_of_particles"),
        standard_deviation_depth_ratio=_f(d, "params.standard_deviation_depth_ratio"),
        min_number_of_particles=_i(d, "params.min_number_of_particles"),
        prune_probability_threshold=_f(d, "params.prune_probability_threshold"),
        erase_partially_init_feature_after_this_many_attempts=_i(
            d, "params.erase_partially_init_feature_after_this_many_attempts"),
        minimum_attempted_measurements_of_feature=10,  # monoslam.cpp:1875
        successful_match_fraction=0.5,                 # monoslam.cpp:1876
    )
    # xv_ order: r(3), q(w,x,y,z), v(3), omega(3)  (monoslam.cpp:1881-1885)
    xv = np.array([_f(d, "state.rw_x"), _f(d, "state.rw_y"), _f(d, "state.rw_z"),
                   _f(d, "state.qwr_w"), _f(d, "state.qwr_x"), _f(d, "state.qwr_y"), _f(d, "state.qwr_z"),
                   _f(d, "state.vw_x"), _f(d, "state.vw_y"), _f(d, "state.vw_z"),
                   _f(d, "state.ww_x"), _f(d, "state.ww_y"), _f(d, "state.ww_z")])
    Pxx = np.zeros((13, 13))
    for r in range(13):
        for c in range(13):
            Pxx[r, c] = _f(d, "state.pxx%d_%d" % (r, c))
    feats = []
    base = os.path.dirname(os.path.abspath(path))
    k = 1
    while ("f%d.yi_x" % k) in d:  # the reference hard-wires f1..f4 (monoslam.cpp:1800-1846)
        y = np.array([_f(d, "f%d.yi_x" % k), _f(d, "f%d.yi_y" % k), _f(d, "f%d.yi_z" % k)])
        xp = np.array([_f(d, "f%d.xp_org_%d" % (k, j)) for j in range(7)])
        ident = d.get("f%d.identifier" % k, "empty")
        feats.append(dict(y=y, xp_org=xp, identifier=ident, base_dir=base))
        k += 1
    return dict(cam=cam, params=params, xv=xv, Pxx=Pxx, features=feats,
                input=dict(mode=_i(d, "input.mode"), name=d.get("input.name", "empty")))


def read_pgm(path):
    """Binary (P5) 8-bit PGM -> uint8 array [h][w] (what cv::imread(path, 0) yields, feature.cpp:119)."""
    with open(path, "rb") as fh:
        data = fh.read()
    tokens = []
    pos = 0
    while len(tokens) < 4:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] not in (b"\n", b""):
                pos += 1
            continue
        start = pos
        while not data[pos:pos + 1].isspace():
            pos += 1
        tokens.append(data[start:pos])
    pos += 1  # single whitespace after maxval
    if tokens[0] != b"P5":
        raise ValueError("%s: not a binary PGM" % path)
    w, h, maxval = int(tokens[1]), int(tokens[2]), int(tokens[3])
    if maxval > 255:
        raise ValueError("%s: 16-bit PGM not supported" % path)
    return np.frombuffer(data[pos:pos + w * h], dtype=np.uint8).reshape(h, w).copy()


def resolve_identifier(feat, search_dirs=()):
    """Locate a template file named by `fN.identifier` (relative to the cfg, like the reference's cwd)."""
    ident = feat["identifier"]
    cands = [ident, os.path.join(feat.get("base_dir", "."), ident),
             os.path.join(feat.get("base_dir", "."), os.path.basename(ident))]
    for dd in search_dirs:
        cands.append(os.path.join(dd, os.path.basename(ident)))
    for c in cands:
        if os.path.exists(c):
            return c
    raise FileNotFoundError("template %r not found (tried %s)" % (ident, cands))
