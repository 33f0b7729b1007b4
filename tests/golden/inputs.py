"""Seeded inputs shared by the golden generator and the tests that replay the fixtures.

The fixtures store only seeds for the (large) inputs; both sides rebuild them here with
numpy ``default_rng`` (platform-stable) and compare a checksum kept in the fixture.
"""
import hashlib

import numpy as np


def l2n(x, eps=1e-5):
    return (x / (np.linalg.norm(x, axis=-1, keepdims=True) + eps)).astype(np.float32)


def checksum(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def stage_b_inputs(opt, input_seed, lens_v, lens_q):
    """One ragged, zero-padded batch as prepare_batch_inputs would deliver it."""
    rng = np.random.default_rng(input_seed)
    B = len(lens_v)
    dv, dt = opt.v_appear_feat_dim, opt.t_feat_dim
    vid = np.zeros((B, max(lens_v), dv), np.float32)
    txt = np.zeros((B, max(lens_q), dt), np.float32)
    vmask = np.zeros((B, max(lens_v)), np.float32)
    tmask = np.zeros((B, max(lens_q)), np.float32)
    for b in range(B):
        vid[b, :lens_v[b]] = rng.standard_normal((lens_v[b], dv), dtype=np.float32)
        txt[b, :lens_q[b]] = l2n(rng.standard_normal((lens_q[b], dt), dtype=np.float32))
        vmask[b, :lens_v[b]] = 1
        tmask[b, :lens_q[b]] = 1
    cls = l2n(rng.standard_normal((B, dv), dtype=np.float32))
    return dict(src_vid=vid, src_txt=txt, vid_mask=vmask, txt_mask=tmask, src_cls_txt=cls)


def stage_a_inputs(opt, input_seed, ctx_ls, n_q=3):
    """Raw video features and raw cls text features for a few videos."""
    rng = np.random.default_rng(input_seed)
    dv = opt.v_appear_feat_dim
    out = []
    for ctx_l in ctx_ls:
        raw = rng.standard_normal((int(ctx_l), dv), dtype=np.float32)
        cls = rng.standard_normal((n_q, dv), dtype=np.float32)
        out.append((raw, cls))
    return out
