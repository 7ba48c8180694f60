"""Synthetic sequence that exercises the feature-initialisation path: few known features (< keep_visible), a camera
that translates faster than the 0.2 m/s gate (monoslam.cpp:159), textured plane at 0.6 m (inside [min_lambda, max_lambda])."""
import numpy as np

from scenelib2_amd import synth


def make_mapping_sequence(seed=7, n_known=6, n_frames=40, width=320, height=240, v_amp=0.45):
    cam = synth.default_camera(width, height)
    params = synth.default_params(n_known)
    spec = synth.SequenceSpec(cam, n_known, n_frames, synth.BASE_SEED + seed, v_amp=v_amp, w_amp=0.05)
    tex = synth.make_texture()
    frames = synth.render_host(cam, tex, spec.tex_extent, spec.tex_origin, spec.poses)     # frames[k] = pose k
    templates = synth.cut_templates(frames[0], spec.feat_px)
    return cam, params, spec, frames, templates


def oracle_for(cam, params, spec, templates, oa):
    s = oa.OracleSLAM(cam, params["delta_t"], params["number_of_features_to_select"])
    s.set_mapping_params(params)
    s.set_state(spec.xv0, spec.Pxx0)
    xo = spec.xp_org()
    for i in range(spec.n_features):
        s.add_known_feature(spec.feat_y[i], xo[i], templates[i])
    return s
