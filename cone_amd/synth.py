"""Deterministic synthetic weights and EgoVLP/CLIP-shaped features.

There is no network for checkpoints or datasets, so every test, the smoke run and
``bench.py`` use random-initialised weights of the reference architecture and
synthetic features of the reference shapes.
Generation is numpy ``default_rng`` only (platform-stable streams), so the GPU
box regenerates bit-identical inputs from a seed and the committed fixtures only
need the seed plus a checksum.

State-dict keys and shapes follow ``CONE.state_dict()`` of the reference
(``cone/model.py:19-80``; listed in SURVEY.md section 8b).
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np


def state_dict_spec(opt) -> "OrderedDict[str, tuple]":
    d, ff = opt.hidden_dim, opt.dim_feedforward
    spec: "OrderedDict[str, tuple]" = OrderedDict()

    def mha(prefix):
        spec[prefix + ".in_proj_weight"] = (3 * d, d)
        spec[prefix + ".in_proj_bias"] = (3 * d,)
        spec[prefix + ".out_proj.weight"] = (d, d)
        spec[prefix + ".out_proj.bias"] = (d,)

    def ffn(prefix):
        spec[prefix + ".linear1.weight"] = (ff, d)
        spec[prefix + ".linear1.bias"] = (ff,)
        spec[prefix + ".linear2.weight"] = (d, ff)
        spec[prefix + ".linear2.bias"] = (d,)

    def ln(prefix, n=d):
        spec[prefix + ".weight"] = (n,)
        spec[prefix + ".bias"] = (n,)

    for i in range(opt.enc_layers):
        p = f"transformer.encoder.layers.{i}"
        mha(p + ".self_attn"); ffn(p); ln(p + ".norm1"); ln(p + ".norm2")
    for i in range(opt.dec_layers):
        p = f"transformer.decoder.layers.{i}"
        mha(p + ".self_attn"); mha(p + ".multihead_attn"); ffn(p)
        ln(p + ".norm1"); ln(p + ".norm2"); ln(p + ".norm3")
    ln("transformer.decoder.norm")
    spec["txt_position_embed.position_embeddings.weight"] = (opt.max_q_l, d)
    ln("txt_position_embed.LayerNorm")
    spec["span_embed.layers.0.weight"] = (d, d); spec["span_embed.layers.0.bias"] = (d,)
    spec["span_embed.layers.1.weight"] = (d, d); spec["span_embed.layers.1.bias"] = (d,)
    spec["span_embed.layers.2.weight"] = (2, d); spec["span_embed.layers.2.bias"] = (2,)
    spec["class_embed.weight"] = (2, d); spec["class_embed.bias"] = (2,)
    spec["query_embed.weight"] = (opt.num_queries, d)
    for name, din in (("input_txt_proj", opt.t_feat_dim), ("input_vid_proj", opt.v_motion_feat_dim)):
        for i in range(opt.n_input_proj):
            n_in = din if i == 0 else d
            ln(f"{name}.{i}.LayerNorm", n_in)
            spec[f"{name}.{i}.net.1.weight"] = (d, n_in)
            spec[f"{name}.{i}.net.1.bias"] = (d,)
    spec["saliency_proj.weight"] = (1, d); spec["saliency_proj.bias"] = (1,)
    if opt.adapter_module == "linear":
        dv = opt.v_appear_feat_dim
        spec["adapter_layer.layers.0.weight"] = (d, dv); spec["adapter_layer.layers.0.bias"] = (d,)
        spec["adapter_layer.layers.1.weight"] = (dv, d); spec["adapter_layer.layers.1.bias"] = (dv,)
    if getattr(opt, "pre_norm", False):     # the encoder's final LayerNorm exists only then (cone/transformer.py:32)
        ln("transformer.encoder.norm")
    return spec


def make_state_dict(opt, seed: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Xavier-uniform matrices, perturbed LayerNorm affine terms, small biases.
    ``span_embed.layers.2.bias`` is set so that many proposals overrun short
    windows (hazard H3 of SURVEY.md)."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shape in state_dict_spec(opt).items():
        if len(shape) == 2:
            fan_out, fan_in = shape
            a = np.sqrt(6.0 / (fan_in + fan_out))
            if k == "query_embed.weight" or k.startswith("txt_position_embed.position"):
                a = 2.0
            if k in ("span_embed.layers.2.weight", "class_embed.weight"):
                a *= 3.0
            w = (rng.random(shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(a)
            if k.endswith("in_proj_weight"):
                # sharper attention than xavier alone gives (trained checkpoints are peaky);
                # only the W_q / W_k rows
                w[:2 * shape[1]] *= np.float32(2.5)
        elif "norm" in k.lower() and k.endswith(".weight"):
            w = np.float32(0.75) + rng.random(shape, dtype=np.float32) * np.float32(0.5)
        else:
            w = (rng.random(shape, dtype=np.float32) * 2.0 - 1.0) * np.float32(0.1)
        sd[k] = np.ascontiguousarray(w, dtype=np.float32)
    sd["span_embed.layers.2.bias"] = np.array([0.5, 1.0], dtype=np.float32)
    return sd


def state_dict_checksum(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v, dtype=np.float32).tobytes())
    return h.hexdigest()[:16]


def make_motion_feats(opt, video_feats, seed: int = 0):
    """A second visual source for the same clips: clip_id -> (ctx_l, v_motion_feat_dim) fp32 (the reference's
    ``motion_feat_dir`` when it differs from ``appearance_feat_dir``: cone/ego4d_mad_dataloader.py:63-81)."""
    rng = np.random.default_rng(seed + 7919)
    return OrderedDict((c, rng.standard_normal((v.shape[0], opt.v_motion_feat_dim), dtype=np.float32))
                       for c, v in video_feats.items())


def make_dataset(opt, n_queries: int, n_videos: int, seed: int = 0,
                 ctx_range=(850, 950), lq_range=(5, 18)):
    """Synthetic split in the reference's on-disk vocabulary.

    Returns ``(annotations, video_feats, query_feats)``:
      * annotations: list of jsonl rows (``data/README.md:18-25``),
      * video_feats: clip_id -> ``features`` (ctx_l, dv) fp32 (un-normalised, H2),
      * query_feats: query_id -> {``token_features`` (Lq, dt), ``cls_features`` (dv,)}.
    Queries of one video are adjacent, as in the Ego4D annotation files.
    """
    rng = np.random.default_rng(seed)
    dv, dt = opt.v_appear_feat_dim, opt.t_feat_dim
    video_feats, ann, query_feats = OrderedDict(), [], OrderedDict()
    clip_ids = [f"clip{v:05d}" for v in range(n_videos)]
    for cid in clip_ids:
        ctx_l = int(rng.integers(ctx_range[0], ctx_range[1]))
        video_feats[cid] = rng.standard_normal((ctx_l, dv), dtype=np.float32)
    per_video = [n_queries // n_videos + (1 if v < n_queries % n_videos else 0) for v in range(n_videos)]
    qn = 0
    for v, cid in enumerate(clip_ids):
        for j in range(per_video[v]):
            if opt.dset_name == "ego4d":
                qid = f"ann{v:05d}x{qn:06d}_{j}"
            else:
                qid = f"q{qn:06d}"
            lq = int(rng.integers(lq_range[0], lq_range[1]))
            query_feats[qid] = dict(
                token_features=rng.standard_normal((lq, dt), dtype=np.float32),
                cls_features=rng.standard_normal((dv,), dtype=np.float32),
            )
            ann.append(dict(query_id=qid, query=f"synthetic query {qn}", video_id=f"video{v:05d}",
                            clip_id=cid, duration=float(video_feats[cid].shape[0]) * opt.clip_length,
                            timestamps=[0.0, 1.0]))
            qn += 1
    return ann, video_feats, query_feats
