"""CPU ORACLE for the CONE coarse-to-fine inference hot path -- TEST INFRASTRUCTURE ONLY.

This file is a functional restatement (torch-CPU fp32 for tensor math, plain
Python doubles for fusion / NMS exactly as the reference) of the algorithm that
``/root/reference/cone/inference.py`` drives.  It is written from SURVEY.md and
from reading the reference; no reference code is copied.  Every function cites
the reference file:line it follows.

Who may use it: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- as the checker / the timed CPU port, never as a fallback
of the product.  Nothing under ``cone_amd/`` imports this module.

Pinning: the reference holds no golden vectors for this path (SURVEY.md section
4); the oracle is pinned against outputs of the reference itself, imported in the
build container by ``tests/golden/gen_golden.py`` (fixtures committed under
``tests/golden/``; ``tests/test_oracle_golden.py`` replays them).  Third-party
arithmetic (``torch.nn.MultiheadAttention``, ``nn.LayerNorm`` -- torch pinned at
1.12.1 by the reference's INSTALL.md:24, 2.10 in this image) is restated
explicitly below from torch's documented semantics.

Known pinned deviation: window-rank ties use the *stable* descending order
(lower window index first), hazard H6 of SURVEY.md; the golden generator patches
``torch.sort`` to ``stable=True`` and says so.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- utils
def as_torch_sd(sd):
    """numpy / torch state dict -> torch CPU fp32 tensors."""
    out = OrderedDict()
    for k, v in sd.items():
        out[k] = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).float().cpu()
    return out


def l2_normalize_np(x, eps=1e-5):
    """utils/basic_utils.py:97-99 -- eps is added to the norm."""
    return x / (np.linalg.norm(x, axis=-1, keepdims=True) + eps)


def span_cxw_to_xx(cxw):
    """cone/span_utils.py:25-41."""
    x1 = cxw[..., 0] - 0.5 * cxw[..., 1]
    x2 = cxw[..., 0] + 0.5 * cxw[..., 1]
    return torch.stack([x1, x2], dim=-1)


def span_xx_to_cxw(xx):
    """cone/span_utils.py:4-22."""
    return torch.stack([xx.sum(-1) * 0.5, xx[..., 1] - xx[..., 0]], dim=-1)


def temporal_iou(s1, s2):
    """cone/span_utils.py:44-71."""
    a1 = s1[:, 1] - s1[:, 0]
    a2 = s2[:, 1] - s2[:, 0]
    left = torch.max(s1[:, None, 0], s2[:, 0])
    right = torch.min(s1[:, None, 1], s2[:, 1])
    inter = (right - left).clamp(min=0)
    union = a1[:, None] + a2 - inter
    return inter / union, union


def generalized_temporal_iou(s1, s2):
    """cone/span_utils.py:90-122."""
    iou, union = temporal_iou(s1.float(), s2.float())
    left = torch.min(s1[:, None, 0], s2[:, 0])
    right = torch.max(s1[:, None, 1], s2[:, 1])
    enclosing = (right - left).clamp(min=0)
    return iou - (enclosing - union) / enclosing


def window_bounds(i, ctx_l, max_v_l):
    """Window i of a video of ctx_l clips (cone/inference.py:286-292, H1):
    stride S=int(W/2); window 0 is the half window [0, S)."""
    s = int(max_v_l / 2)
    return max((i - 1) * s, 0), min((i - 1) * s + max_v_l, ctx_l)


def num_windows(ctx_l, max_v_l):
    return math.ceil(ctx_l / int(max_v_l / 2)) + 1


# ------------------------------------------------------------------ stage A (A1-A4)
def mlp(x, sd, prefix, n_layers):
    """cone/model.py:428-440."""
    for i in range(n_layers):
        x = F.linear(x, sd[f"{prefix}.layers.{i}.weight"], sd[f"{prefix}.layers.{i}.bias"])
        if i < n_layers - 1:
            x = F.relu(x)
    return x


def adapter_norm(sd, vid, adapter_module="linear"):
    """A2: cone/inference.py:250-260 -- y = adapter(x) + x; y /= ||y|| (no eps)."""
    if adapter_module != "linear":
        return vid
    y = mlp(vid, sd, "adapter_layer", 2) + vid
    return y / y.norm(dim=-1, keepdim=True)


def frame_scores(vid_ctx, cls_txt):
    """A3: cone/inference.py:284 -- einsum('db,b->d')."""
    return torch.einsum("db,b->d", vid_ctx, cls_txt)


def window_scores(fscore, max_v_l):
    """A4: cone/inference.py:285-296 -- max of the frame scores inside each window."""
    ctx_l = fscore.shape[0]
    nw = num_windows(ctx_l, max_v_l)
    out = torch.empty(nw, dtype=torch.float32)
    for i in range(nw):
        s, e = window_bounds(i, ctx_l, max_v_l)
        out[i] = torch.max(fscore[s:e])
    return out


def rank_windows(wscore):
    """A4: cone/inference.py:297-299 with the tie order pinned to stable descending (H6)."""
    _, idx = torch.sort(wscore, descending=True, stable=True)
    return idx.tolist()


# ------------------------------------------------------------------ stage B (A5-A12)
def pad_sequences_1d(seqs):
    """utils/tensor_utils.py:5-53 for a list of 2-D fp32 tensors: zero pad to the
    longest, float mask 1=valid."""
    lengths = [len(s) for s in seqs]
    L = max(lengths)
    out = torch.zeros((len(seqs), L) + tuple(seqs[0].shape[1:]), dtype=torch.float32)
    mask = torch.zeros((len(seqs), L), dtype=torch.float32)
    for i, s in enumerate(seqs):
        out[i, :lengths[i]] = s
        mask[i, :lengths[i]] = 1
    return out, mask


def layer_norm(x, sd, prefix, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def input_proj(x, sd, name, n_input_proj):
    """A6: cone/model.py:58-73, 443-465 -- LN -> Linear -> ReLU (all but the last layer)."""
    for i in range(n_input_proj):
        x = layer_norm(x, sd, f"{name}.{i}.LayerNorm")
        x = F.linear(x, sd[f"{name}.{i}.net.1.weight"], sd[f"{name}.{i}.net.1.bias"])
        if i != n_input_proj - 1:
            x = F.relu(x)
    return x


def sine_position(mask, num_pos_feats, temperature=10000, scale=2 * math.pi):
    """A7: cone/position_encoding.py:51-72 (normalize=True)."""
    x_embed = mask.cumsum(1, dtype=torch.float32)
    x_embed = x_embed / (x_embed[:, -1:] + 1e-6) * scale
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_pos_feats)
    pos = x_embed[:, :, None] / dim_t
    return torch.stack((pos[:, :, 0::2].sin(), pos[:, :, 1::2].cos()), dim=3).flatten(2)


def mha(sd, prefix, q_in, k_in, v_in, nheads, key_pad=None):
    """torch.nn.MultiheadAttention forward (eval, need_weights irrelevant) as used at
    cone/transformer.py:239-240, 304-311, restated batch-first.

    q_in (B,Lq,d), k_in/v_in (B,Lk,d); key_pad (B,Lk) bool, True = padded key.
    Packed in_proj rows [0:d]=W_q, [d:2d]=W_k, [2d:3d]=W_v; q is scaled by
    sqrt(1/head_dim) *after* the projection; padded keys get -inf before softmax."""
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    d = w.shape[1]
    hd = d // nheads
    q = F.linear(q_in, w[:d], b[:d])
    k = F.linear(k_in, w[d:2 * d], b[d:2 * d])
    v = F.linear(v_in, w[2 * d:], b[2 * d:])
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    q = q.view(B, Lq, nheads, hd).transpose(1, 2) * math.sqrt(1.0 / float(hd))
    k = k.view(B, Lk, nheads, hd).transpose(1, 2)
    v = v.view(B, Lk, nheads, hd).transpose(1, 2)
    att = q @ k.transpose(-1, -2)
    if key_pad is not None:
        att = att + torch.zeros(B, 1, 1, Lk).masked_fill(key_pad[:, None, None, :], float("-inf"))
    att = torch.softmax(att, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, Lq, d)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def encoder_layer(sd, p, x, pos, key_pad, nheads, pre_norm=False):
    """A9: cone/transformer.py:233-246 (post-norm); :248-260 (forward_pre, --pre_norm)."""
    if pre_norm:
        x2 = layer_norm(x, sd, p + ".norm1")
        qk = x2 + pos
        x = x + mha(sd, p + ".self_attn", qk, qk, x2, nheads, key_pad)
        x2 = layer_norm(x, sd, p + ".norm2")
        return x + F.linear(F.relu(F.linear(x2, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                            sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    qk = x + pos
    x = layer_norm(x + mha(sd, p + ".self_attn", qk, qk, x, nheads, key_pad), sd, p + ".norm1")
    h = F.linear(F.relu(F.linear(x, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(x + h, sd, p + ".norm2")


def decoder_layer(sd, p, tgt, memory, pos, query_pos, key_pad, nheads, pre_norm=False):
    """A10: cone/transformer.py:296-317 (post-norm); :319-342 (forward_pre, --pre_norm)."""
    if pre_norm:
        t2 = layer_norm(tgt, sd, p + ".norm1")
        qk = t2 + query_pos
        tgt = tgt + mha(sd, p + ".self_attn", qk, qk, t2, nheads)
        t2 = layer_norm(tgt, sd, p + ".norm2")
        tgt = tgt + mha(sd, p + ".multihead_attn", t2 + query_pos, memory + pos, memory, nheads, key_pad)
        t2 = layer_norm(tgt, sd, p + ".norm3")
        return tgt + F.linear(F.relu(F.linear(t2, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                              sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    qk = tgt + query_pos
    tgt = layer_norm(tgt + mha(sd, p + ".self_attn", qk, qk, tgt, nheads), sd, p + ".norm1")
    tgt = layer_norm(tgt + mha(sd, p + ".multihead_attn", tgt + query_pos, memory + pos, memory,
                               nheads, key_pad), sd, p + ".norm2")
    h = F.linear(F.relu(F.linear(tgt, sd[p + ".linear1.weight"], sd[p + ".linear1.bias"])),
                 sd[p + ".linear2.weight"], sd[p + ".linear2.bias"])
    return layer_norm(tgt + h, sd, p + ".norm3")


def cone_forward(sd, opt, src_txt, src_txt_mask, src_vid_motion, src_vid_motion_mask,
                 return_intermediates=False):
    """A6-A11: ``CONE.forward`` (cone/model.py:82-128) + ``Transformer.forward``
    (cone/transformer.py:49-73), batch-first."""
    H = opt.nheads
    src_vid = input_proj(src_vid_motion, sd, "input_vid_proj", opt.n_input_proj)
    src_txt_p = input_proj(src_txt, sd, "input_txt_proj", opt.n_input_proj)
    src = torch.cat([src_vid, src_txt_p], dim=1)
    mask = torch.cat([src_vid_motion_mask, src_txt_mask], dim=1).bool()
    pos_vid = sine_position(src_vid_motion_mask, opt.hidden_dim)
    if getattr(opt, "use_txt_pos", False):
        # cone/model.py:106 with --use_txt_pos: TrainablePositionalEncoding(src_txt) = LayerNorm(src_txt + position_embeddings[t])
        # (cone/position_encoding.py:18-32; eval: the dropout is the identity)
        emb = sd["txt_position_embed.position_embeddings.weight"][: src_txt_p.shape[1]]
        pos_txt = layer_norm(src_txt_p + emb.unsqueeze(0), sd, "txt_position_embed.LayerNorm")
    else:
        pos_txt = torch.zeros_like(src_txt_p)
    pos = torch.cat([pos_vid, pos_txt], dim=1)
    key_pad = ~mask
    x = src
    pre = bool(getattr(opt, "pre_norm", False))        # --pre_norm: normalize_before (cone/transformer.py:19-36)
    for i in range(opt.enc_layers):
        x = encoder_layer(sd, f"transformer.encoder.layers.{i}", x, pos, key_pad, H, pre)
    memory = layer_norm(x, sd, "transformer.encoder.norm") if pre else x      # encoder_norm only when normalize_before (:32)
    B = src.shape[0]
    query_pos = sd["query_embed.weight"][None].expand(B, -1, -1)
    tgt = torch.zeros_like(query_pos)
    hs = []
    for i in range(opt.dec_layers):
        tgt = decoder_layer(sd, f"transformer.decoder.layers.{i}", tgt, memory, pos, query_pos,
                            key_pad, H, pre)
        hs.append(layer_norm(tgt, sd, "transformer.decoder.norm"))
    hs = torch.stack(hs)  # (layers, B, Nq, d)
    logits = F.linear(hs, sd["class_embed.weight"], sd["class_embed.bias"])
    coord = mlp(hs, sd, "span_embed", 3).sigmoid()
    Lv = src_vid.shape[1]
    sal = F.linear(memory[:, :Lv], sd["saliency_proj.weight"], sd["saliency_proj.bias"]).squeeze(-1)
    out = {"pred_logits": logits[-1], "pred_spans": coord[-1], "saliency_scores": sal}
    if getattr(opt, "aux_loss", True):
        out["aux_outputs"] = [{"pred_logits": a, "pred_spans": b}
                              for a, b in zip(logits[:-1], coord[:-1])]
    if return_intermediates:
        out["memory"] = memory
        out["hs"] = hs
        out["src"] = src
        out["pos"] = pos
    return out


def proposal_slices(pred_spans, mask):
    """cone/model.py:186-192: integer [start, end) of every proposal on the padded window."""
    duration = torch.sum(mask, dim=-1)
    prop = torch.einsum("bld,b->bld", span_cxw_to_xx(pred_spans), duration)
    start = F.relu(torch.floor(prop[:, :, 0]).to(torch.int32))
    end = torch.ceil(prop[:, :, 1]).to(torch.int32)
    return start, end, prop


def clip_matching(sd, opt, src_cls_txt, src_vid_appear, src_vid_appear_mask, proposal):
    """A12: ``CONE.forward_clip_matching`` (cone/model.py:130-152, 178-210).  The mean is
    taken over ``feat[s:e]`` of the ZERO-PADDED tensor, so it depends on the padded
    length of the batch (H3)."""
    txt = src_cls_txt / src_cls_txt.norm(dim=1, keepdim=True)
    start, end, _ = proposal_slices(proposal, src_vid_appear_mask)
    B, Nq = start.shape
    feats = []
    for b in range(B):
        for n in range(Nq):
            feats.append(src_vid_appear[b, int(start[b, n]):int(end[b, n])].mean(dim=0))
    pf = torch.stack(feats)
    if opt.adapter_module == "linear":
        pf = mlp(pf, sd, "adapter_layer", 2) + pf
    pf = pf.reshape(B, Nq, -1)
    pf = pf / pf.norm(dim=2, keepdim=True)
    return torch.einsum("bld,bd->bl", pf, txt)


def matching_alternatives(sd, opt, cls_txt, vid_padded, duration, span_cxw, margin=1e-3):
    """Test support for hazard "floor / ceil next to an integer" (SURVEY.md 7): the matching scores the reference's
    ``forward_clip_matching`` (cone/model.py:130-152, 178-210) can produce for ONE proposal when a clip boundary
    ``x * duration`` lies within ``margin`` of an integer -- a 1-ulp difference in the predicted span then moves the slice
    end by one clip.  Returns the list of scores over the poolings ``feat[s:e]`` with s in {floor(x1)} (+ its neighbour when
    x1 is near an integer) and e in {ceil(x2)} (+ neighbour); a proposal away from every boundary has exactly one.
    ``vid_padded`` (Lv_pad, dv) is the zero-padded window of the reference batch (hazard H3), ``cls_txt`` (dv,)."""
    sp = torch.as_tensor(span_cxw, dtype=torch.float32).reshape(1, 1, 2)
    xx = span_cxw_to_xx(sp)[0, 0] * float(duration)                         # fp32, as cone/model.py:188
    x1, x2 = float(xx[0]), float(xx[1])
    starts = {max(int(math.floor(x1)), 0)}
    ends = {int(math.ceil(x2))}
    if abs(x1 - round(x1)) < margin:
        starts |= {max(int(round(x1)) - 1, 0), max(int(round(x1)), 0)}
    if abs(x2 - round(x2)) < margin:
        ends |= {int(round(x2)), int(round(x2)) + 1}
    txt = cls_txt / cls_txt.norm()
    out = []
    for s_ in sorted(starts):
        for e_ in sorted(ends):
            pf = vid_padded[s_:e_].mean(dim=0)[None]
            if opt.adapter_module == "linear":
                pf = mlp(pf, sd, "adapter_layer", 2) + pf
            pf = pf / pf.norm(dim=1, keepdim=True)
            out.append(float((pf[0] * txt).sum()))
    return out


def check_matching_column(sd, opt, cls_norm, vid_raw, vid_row0, vid_len, pad_len, cls_row, pred_spans, matching, tol=1e-4):
    """Every (window, slot) matching score of a device run against ``matching_alternatives`` of ITS OWN predicted span on
    the same zero-padded window: returns (n_checked, n_boundary, worst) where worst = the largest distance to the nearest
    admissible pooling.  Inputs are CPU tensors / arrays: the store's raw clip arena, the window table columns, the
    normalised cls vectors."""
    n_chk = n_bnd = 0
    worst = 0.0
    B, Nq = matching.shape
    for b in range(B):
        r0, L, P = int(vid_row0[b]), int(vid_len[b]), int(pad_len[b])
        win = torch.zeros(max(P, L), vid_raw.shape[1])
        win[:L] = vid_raw[r0:r0 + L]
        win = win[:P] if P >= L else win
        cls = cls_norm[int(cls_row[b])]
        for n in range(Nq):
            alts = matching_alternatives(sd, opt, cls, win, L, pred_spans[b, n])
            got = float(matching[b, n])
            dist = lambda a: 0.0 if (math.isnan(a) and math.isnan(got)) else (
                float("inf") if (math.isnan(a) or math.isnan(got)) else abs(got - a))
            d = min(dist(a) for a in alts)
            worst = max(worst, d)
            n_chk += 1
            n_bnd += len(alts) > 1
    return n_chk, n_bnd, worst


def compose_rows(opt, pred_logits, pred_spans, matching, durations, video_starts):
    """A13: cone/inference.py:47-91 -- per window (Nq,4) fp32 rows [st, ed, prop, match]
    sorted by proposal score (stable, descending) unless --no_sort_results; rounding to
    4 decimals is applied by :func:`round4_rows`."""
    prob = F.softmax(pred_logits, -1)[..., 0]
    out = []
    for b in range(pred_spans.shape[0]):
        spans = (span_cxw_to_xx(pred_spans[b]) * int(durations[b]) + int(video_starts[b])) * opt.clip_length
        rows = torch.cat([spans, prob[b][:, None], matching[b][:, None]], dim=1).tolist()
        if not opt.no_sort_results:
            rows = sorted(rows, key=lambda r: r[2], reverse=True)
        out.append(rows)
    return out


def round4_rows(rows):
    """cone/inference.py:83 -- float(f"{e:.4f}") (H4)."""
    return [[float(f"{e:.4f}") for e in r] for r in rows]


# ------------------------------------------------------------------ stage C (A14-A16)
def normalize_score(vals):
    """utils/basic_utils.py:10-20."""
    amin, amax = min(vals), max(vals)
    if amin == amax:
        return vals
    return [(v - amin) / (amax - amin) for v in vals]


def score_fusion(prediction):
    """A14: cone/inference.py:205-217 -- dict keyed by (st, ed): duplicates collapse,
    last value wins, first position kept (H5)."""
    ret = {}
    a = normalize_score([p[2] for p in prediction])
    m = normalize_score([p[3] for p in prediction])
    for item, fa, fm in zip(prediction, a, m):
        ret[(item[0], item[1])] = [item[2], item[3], sum((fa, fm))]
    return ret


def compute_temporal_iou(pred, gt):
    """utils/temporal_nms.py:6-22 -- pseudo union max(end)-min(start) (H7)."""
    inter = max(0, min(pred[1], gt[1]) - max(pred[0], gt[0]))
    union = max(pred[1], gt[1]) - min(pred[0], gt[0])
    return 0 if union == 0 else 1.0 * inter / union


def temporal_nms(predictions, nms_thd, max_after_nms=100):
    """A15: utils/temporal_nms.py:25-74, restated as the equivalent greedy scan:
    stable sort by score (descending); keep a candidate iff no kept one overlaps it by
    more than nms_thd (strict); stop at max_after_nms."""
    if len(predictions) == 1:
        return predictions
    cand = sorted(predictions, key=lambda x: x[2], reverse=True)
    kept = []
    alive = [True] * len(cand)
    for i, c in enumerate(cand):
        if len(kept) >= max_after_nms:
            break
        if not alive[i]:
            continue
        kept.append(c)
        for j in range(i + 1, len(cand)):
            if alive[j] and compute_temporal_iou(c, cand[j]) > nms_thd:
                alive[j] = False
    return [[c[0], c[1], c[2]] for c in kept]


def post_processing_mr_nms(opt, return_dict, idx):
    """cone/inference.py:103-127."""
    moments = [[k[0], k[1], v[idx]] for k, v in return_dict.items()]
    moments = sorted(moments, key=lambda x: x[2], reverse=True)
    before = [[m[0], m[1]] + return_dict[(m[0], m[1])] for m in moments]
    if opt.nms_thd != -1:
        after = temporal_nms(moments[:opt.max_before_nms], opt.nms_thd, opt.max_after_nms)
        return [[m[0], m[1]] + return_dict[(m[0], m[1])] for m in after]
    return before[:opt.max_after_nms]


def postprocess(submission, opt):
    """A16: cone/inference.py:130-202 -- group window-level rows by query, fuse, 3x NMS."""
    qid2 = OrderedDict()
    for item in submission:
        qid = item["query_id"]
        if qid not in qid2:
            if opt.dset_name == "ego4d":
                parts = qid.split("_")
                assert len(parts) == 2
                qid2[qid] = {"query_idx": int(parts[1]), "annotation_uid": parts[0],
                             "predicted_times": [], "clip_uid": item["clip_id"]}
            else:
                qid2[qid] = {"query_id": qid, "predicted_times": [], "video_id": item["video_id"]}
        qid2[qid]["predicted_times"].extend(item["pred_relevant_windows"])
    fusion, proposal, matching = [], [], []
    for item in qid2.values():
        rd = score_fusion(item["predicted_times"])
        for lst, idx in ((fusion, 2), (proposal, 0), (matching, 1)):
            o = item.copy()
            o["predicted_times"] = post_processing_mr_nms(opt, rd, idx)
            lst.append(o)
    return fusion, proposal, matching


# ------------------------------------------------------- single-video localizer
def localizer_predict(sd, opt, video_feats, text_token_feats, text_cls_feat):
    """``CONELocalizator.predict_moment`` (run_on_video/cone_localizator.py:121-221).  Differences from
    the dataset path that are part of the contract: F.normalize (x / max(||x||, eps)) on clips and tokens,
    the adapted features are NOT re-normalised and the cls vector is used raw for the window ranking,
    every window is padded to (max_v_l, max_q_l), spans scale by max_v_l (not the window's length),
    rows are not sorted per window, NMS runs on the fused score only ([:100], 0.5, keep 5)."""
    sd = as_torch_sd(sd)
    W, K = opt.max_v_l, opt.topk_window
    v = F.normalize(video_feats.float(), dim=-1, eps=1e-5)
    tok = F.normalize(text_token_feats.float(), dim=-1, eps=1e-5)
    av = mlp(v, sd, "adapter_layer", 2) + v if opt.adapter_module == "linear" else v
    ranks = rank_windows(window_scores(frame_scores(av, text_cls_feat.float()), W))
    widx = ranks[:K]
    assert len(widx) == K, "the reference feeds all-padding windows to the model for short videos (NaN)"
    dv, dt = v.shape[1], tok.shape[1]
    vid = torch.zeros(K, W, dv); vmask = torch.zeros(K, W)
    txt = torch.zeros(K, opt.max_q_l, dt); tmask = torch.zeros(K, opt.max_q_l)
    cls = torch.zeros(K, dv)
    starts = []
    for i, w in enumerate(widx):
        s, e = window_bounds(w, v.shape[0], W)
        vid[i, :e - s] = v[s:e]; vmask[i, :e - s] = 1
        txt[i, :tok.shape[0]] = tok; tmask[i, :tok.shape[0]] = 1
        cls[i] = text_cls_feat
        starts.append(s)
    out = cone_forward(sd, opt, txt, tmask, vid, vmask)
    match = clip_matching(sd, opt, cls, vid, vmask, out["pred_spans"])
    prob = F.softmax(out["pred_logits"], -1)[..., 0]
    total = []
    for i in range(K):
        spans = (span_cxw_to_xx(out["pred_spans"][i]) * W + starts[i]) * opt.clip_length
        rows = torch.cat([spans, prob[i][:, None], match[i][:, None]], dim=1).tolist()
        total.extend(round4_rows(rows))
    rd = score_fusion(total)
    moments = sorted([[k[0], k[1], val[2]] for k, val in rd.items()], key=lambda x: x[2], reverse=True)
    return temporal_nms(moments[:100], 0.5, 5)


# ------------------------------------------------------------------- A17 (matcher cost)
def matcher_cost(opt_costs, pred_logits, pred_spans, tgt_spans):
    """cone/matcher.py:61-95 cost matrix C = span*L1 + giou*(-GIoU) + class*(-p_fg).
    opt_costs = (cost_span, cost_giou, cost_class)."""
    cs, cg, cc = opt_costs
    prob = pred_logits.flatten(0, 1).softmax(-1)
    cost_class = -prob[:, [0] * len(tgt_spans)]
    out_spans = pred_spans.flatten(0, 1)
    cost_span = torch.cdist(out_spans, tgt_spans, p=1)
    cost_giou = -generalized_temporal_iou(span_cxw_to_xx(out_spans), span_cxw_to_xx(tgt_spans))
    return cs * cost_span + cg * cost_giou + cc * cost_class


# ------------------------------------------------------------------ whole path driver
def prepare_query_inputs(opt, q):
    """StartEndDataset._get_query_feat_by_qid (cone/ego4d_mad_dataloader.py:258-282):
    tokens truncated to max_q_l and L2-normalised (+eps); cls normalised (+eps)."""
    tok = np.asarray(q["token_features"], dtype=np.float32)[:opt.max_q_l]
    tok = l2_normalize_np(tok).astype(np.float32)
    cls = np.asarray(q.get("cls_features", q.get("eot_features")), dtype=np.float32)
    if cls.ndim == 2:
        cls = cls[0]
    cls = l2_normalize_np(cls).astype(np.float32)
    return torch.from_numpy(tok), torch.from_numpy(cls)


def prefilter(sd, opt, ann, video_feats, query_feats):
    """Stage A over a split: returns query_id -> full window rank list and the window
    scores (cone/inference.py:241-301)."""
    sd = as_torch_sd(sd)
    ctx_cache, ranks, scores = {}, OrderedDict(), OrderedDict()
    for row in ann:
        cid = row["clip_id"]
        if cid not in ctx_cache:
            v = torch.from_numpy(l2_normalize_np(np.asarray(video_feats[cid], dtype=np.float32)).astype(np.float32))
            ctx_cache[cid] = adapter_norm(sd, v, opt.adapter_module)
        _, cls = prepare_query_inputs(opt, query_feats[row["query_id"]])
        ws = window_scores(frame_scores(ctx_cache[cid], cls), opt.max_v_l)
        scores[row["query_id"]] = ws
        ranks[row["query_id"]] = rank_windows(ws)
    return ranks, scores


def build_batch(opt, ann_rows, video_feats, query_feats, ranks, motion_feats=None):
    """A5: eval branch of StartEndDataset.__getitem__ + start_end_collate
    (cone/ego4d_mad_dataloader.py:144-159, 229-234, 305-358).  Model-side video features
    are the RAW ones (H2).  ``motion_feats``: the RAW rows of the second visual source (:134-137, 150-158): the window
    model's input is sliced from it, the matching's from the appearance features -- both with the appearance length.
    Unlike the appearance reader (whose normalised copy is dropped: H2), the motion reader ``_get_video_motion_feat_by_vid``
    (:284-292) RETURNS its L2-normalised rows (``x / (|x| + 1e-5)``) whenever ``normalize_v`` is set (= not
    ``opt.no_norm_vfeat``, cone/inference.py:581), so that is what a two-source window model sees.  (With
    ``--no_norm_vfeat`` the reference's reader raises NameError; the raw rows are used here.)"""
    metas, vids, mots, txts, clss = [], [], [], [], []
    norm_v = not getattr(opt, "no_norm_vfeat", False)
    for row in ann_rows:
        tok, cls = prepare_query_inputs(opt, query_feats[row["query_id"]])
        v = torch.from_numpy(np.asarray(video_feats[row["clip_id"]], dtype=np.float32))
        mo = v
        if motion_feats is not None:
            m = np.asarray(motion_feats[row["clip_id"]], dtype=np.float32)
            mo = torch.from_numpy(l2_normalize_np(m).astype(np.float32) if norm_v else m)
        ctx_l = v.shape[0]
        for w in ranks[row["query_id"]][:opt.topk_window]:
            s, e = window_bounds(w, ctx_l, opt.max_v_l)
            vids.append(v[s:e])
            mots.append(mo[s:e])
            txts.append(tok)
            clss.append(cls)
            m = dict(row)
            m["duration"] = e - s
            m["video_start"] = s
            metas.append(m)
    src_vid, vid_mask = pad_sequences_1d(vids)
    src_mot, mot_mask = pad_sequences_1d(mots)
    src_txt, txt_mask = pad_sequences_1d(txts)
    return metas, dict(src_txt=src_txt, src_txt_mask=txt_mask, src_vid_motion=src_mot,
                       src_vid_motion_mask=mot_mask), dict(
        src_cls_txt=torch.stack(clss), src_vid_appear=src_vid.clone(), src_vid_appear_mask=vid_mask.clone())


def compute_mr_results(sd, opt, ann, video_feats, query_feats, ranks, capture=None, motion_feats=None):
    """cone/inference.py:30-100 with the DataLoader batching of eval_bsz queries."""
    sd = as_torch_sd(sd)
    mr_res = []
    for b0 in range(0, len(ann), opt.eval_bsz):
        rows = ann[b0:b0 + opt.eval_bsz]
        metas, mi, ci = build_batch(opt, rows, video_feats, query_feats, ranks, motion_feats)
        out = cone_forward(sd, opt, **mi)
        match = clip_matching(sd, opt, proposal=out["pred_spans"], **ci)
        if capture is not None:
            capture.append(dict(model_inputs=mi, clip_inputs=ci, outputs=out, matching=match, metas=metas))
        comp = compose_rows(opt, out["pred_logits"], out["pred_spans"], match,
                            [m["duration"] for m in metas], [m["video_start"] for m in metas])
        for m, r in zip(metas, comp):
            mr_res.append(dict(query_id=m["query_id"], query=m["query"], video_id=m["video_id"],
                               clip_id=m["clip_id"], pred_relevant_windows=round4_rows(r)))
        if opt.debug:
            break
    return mr_res


def eval_epoch(sd, opt, ann, video_feats, query_feats, motion_feats=None):
    """Stages A->C on an in-memory split; returns the three submission lists
    (fusion, proposal, matching) and the rank lists.  ``video_feats`` = the appearance source (pre-filter, matching),
    ``motion_feats`` = the window model's source when it is another one (cone/ego4d_mad_dataloader.py:63-71)."""
    with torch.no_grad():
        ranks, _ = prefilter(sd, opt, ann, video_feats, query_feats)
        mr = compute_mr_results(sd, opt, ann, video_feats, query_feats, ranks, motion_feats=motion_feats)
    return postprocess(mr, opt), ranks, mr


# ------------------------------------------------------------------ metrics
def ego4d_ground_truth_table(ground_truth):
    """standalone_eval/evaluate_ego4d_nlq.py:67-77 -- {(clip_uid, annotation_uid): annotation}."""
    table = {}
    for video in ground_truth["videos"]:
        for clip in video["clips"]:
            for ann in clip["annotations"]:
                table[(clip["clip_uid"], ann["annotation_uid"])] = ann
    return table


def iou_f64(pred, gt):
    """standalone_eval/evaluate_ego4d_nlq.py:41-60 for one target: numpy doubles, union clamped at 0,
    plain division (0/0 -> nan, x/0 -> inf)."""
    pred = np.asarray(pred, dtype=np.float64)
    ps, pe = pred[:, 0], pred[:, 1]
    inter = np.maximum(0.0, np.minimum(pe, gt[1]) - np.maximum(ps, gt[0]))
    union = np.maximum(0.0, np.maximum(pe, gt[1]) - np.minimum(ps, gt[0]))
    with np.errstate(all="ignore"):
        return 1.0 * inter / union


def evaluate_nlq_performance_ego4d(predictions, ground_truth, thresholds, topK):
    """standalone_eval/evaluate_ego4d_nlq.py:63-115 -> (results (n_thr, n_topK) float64, mIoU float64).
    mIoU averages the IoU of the first prediction of every query."""
    table = ego4d_ground_truth_table(ground_truth)
    hits = np.zeros((len(thresholds), len(topK), len(predictions)), dtype=bool)
    top1 = []
    for qi, p in enumerate(predictions):
        key = (p["clip_uid"], p["annotation_uid"])
        assert key in table, "Instance not present!"
        q = table[key]["language_queries"][p["query_idx"]]
        ov = iou_f64(p["predicted_times"], (q["clip_start_sec"], q["clip_end_sec"]))
        top1.append(ov[0])
        for t, thr in enumerate(thresholds):
            for r, k in enumerate(topK):
                hits[t, r, qi] = bool((ov > thr)[:k].any())
    return hits.mean(axis=-1), np.mean(np.asarray(top1).reshape(-1, 1))


def iou_f32(candidates, gt):
    """standalone_eval/evaluate_mad.py:33-38 -- torch fp32: inter clamped at 0, union NOT clamped."""
    start, end = candidates[:, 0].float(), candidates[:, 1].float()
    s, e = gt[0].float(), gt[1].float()
    inter = torch.minimum(end, e) - torch.maximum(start, s)
    union = torch.maximum(end, e) - torch.minimum(start, s)
    return inter.clamp(min=0) / union


def evaluate_nlq_performance_mad(submission, ground_truth, thresholds, topK):
    """standalone_eval/evaluate_mad.py:61-107 -> (n_topK, n_thr) float32 tensor."""
    truth = {d["query_id"]: d["timestamps"] for d in ground_truth}
    assert set(truth) == {e["query_id"] for e in submission}
    thr = torch.as_tensor(thresholds, dtype=torch.float32)
    ks = [int(k) for k in topK]
    out = torch.zeros(len(ks), len(thr))
    for e in submission:
        iou = iou_f32(torch.tensor(e["predicted_times"][:max(ks)]), torch.tensor(truth[e["query_id"]]))
        over = iou[:, None] > thr[None, :]
        for i, k in enumerate(ks):
            out[i] += over[:k].any(dim=0)
    out /= len(submission)
    return out


def windows_selection(query_id2windowidx, ground_truth, topK, clip_length, max_v_l):
    """standalone_eval/evaluate_pre_filtered_window.py:30-72 -> (n_topK,) float32 tensor: is any of the first
    r ranked windows one of the windows floor(start/S) .. ceil(end/S) that contain the target?"""
    S = int(max_v_l / 2)
    ks = [int(k) for k in topK]
    out = torch.zeros(len(ks))
    truth = {}
    for m in ground_truth:
        start, end = m["timestamps"][0] / clip_length, m["timestamps"][1] / clip_length
        truth[m["query_id"]] = (math.floor(start / S), math.ceil(end / S) + 1)
    assert set(truth) == set(query_id2windowidx)
    for qid, ranks in query_id2windowidx.items():
        lo, hi = truth[qid]
        inside = [lo <= w < hi for w in ranks[:max(ks)]]
        for i, k in enumerate(ks):
            out[i] += float(any(inside[:k]))
    out /= len(query_id2windowidx)
    return out


# ------------------------------------------------------------------ criterion forward
def hungarian_indices(opt_costs, pred_logits, pred_spans, tgt_spans_list):
    """HungarianMatcher.forward (cone/matcher.py:37-106): cost matrix over the whole batch, then an exact assignment per
    window on its own targets' columns (scipy's linear_sum_assignment there; here brute force over the <= 8! x C(8, k)
    candidate assignments, which is exact too).  Returns [(slot indices ascending, target indices)] per window."""
    from itertools import permutations
    B, Nq = pred_spans.shape[:2]
    out = []
    for b in range(B):
        tg = tgt_spans_list[b]
        C = matcher_cost(opt_costs, pred_logits[b:b + 1], pred_spans[b:b + 1], tg).view(Nq, -1)
        T = C.shape[1]
        k = min(Nq, T)
        best, arg = None, None
        if Nq <= T:
            for perm in permutations(range(T), k):
                c = sum(float(C[n, perm[n]]) for n in range(k))
                if best is None or c < best:
                    best, arg = c, [(n, perm[n]) for n in range(k)]
        else:
            for perm in permutations(range(Nq), k):
                c = sum(float(C[perm[j], j]) for j in range(k))
                if best is None or c < best:
                    best, arg = c, sorted((perm[j], j) for j in range(k))
        out.append(([p[0] for p in arg], [p[1] for p in arg]))
    return out


def criterion_layer(hyper, pred_logits, pred_spans, tgt_spans_list, neg_logits=None, saliency=None, pos_idx=None,
                    neg_idx=None, neg_saliency=None):
    """One decoder layer's losses of SetCriterion (cone/model.py:266-363): spans (L1 + GIoU on matched pairs), labels
    (class-weighted CE over all slots, the negative window's appended), class_error, saliency hinge."""
    costs = (hyper["set_cost_span"], hyper["set_cost_giou"], hyper["set_cost_class"])
    idx = hungarian_indices(costs, pred_logits, pred_spans, tgt_spans_list)
    src = torch.cat([pred_spans[b, i] for b, (i, _) in enumerate(idx)])
    tgt = torch.cat([tgt_spans_list[b][j] for b, (_, j) in enumerate(idx)])
    losses = {"loss_span": F.l1_loss(src, tgt, reduction="none").mean(),
              "loss_giou": (1 - torch.diag(generalized_temporal_iou(span_cxw_to_xx(src), span_cxw_to_xx(tgt)))).mean()}
    logits = pred_logits if neg_logits is None else torch.cat((pred_logits, neg_logits), dim=1)
    target = torch.full(logits.shape[:2], 1, dtype=torch.int64)
    for b, (i, _) in enumerate(idx):
        target[b, i] = 0
    w = torch.tensor([1.0, hyper["eos_coef"]])
    losses["loss_label"] = F.cross_entropy(logits.transpose(1, 2), target, w, reduction="none").mean()
    matched = torch.cat([pred_logits[b, i] for b, (i, _) in enumerate(idx)])
    losses["class_error"] = 100 - 100.0 * (matched.argmax(-1) == 0).float().mean()
    if saliency is not None:
        ar = torch.arange(saliency.shape[0])
        P = pos_idx.shape[1]
        pos = torch.stack([saliency[ar, pos_idx[:, c]] for c in range(P)], 1)
        neg = torch.stack([saliency[ar, neg_idx[:, c]] for c in range(P)], 1)
        ls = torch.clamp(hyper["saliency_margin"] + neg - pos, min=0).sum() / (len(pos) * P) * 2
        if neg_saliency is not None:
            nm = neg_saliency.max(1).values[:, None]
            ls = ls + torch.clamp(hyper["saliency_margin"] + nm - pos, min=0).sum() / (len(pos) * P) * 2
        losses["loss_saliency"] = ls
    return losses, idx


def adapter_nce(sim, temperature):
    """loss_adapter (cone/model.py:249-264)."""
    lg = sim / temperature
    d = torch.arange(len(lg))
    return (F.cross_entropy(lg, d) + F.cross_entropy(lg.T, d)) / 2
