"""Thin Python wrappers over the stage A / stage C entry points of libcone_hip.so.

Function names follow the reference (``temporal_nms`` = utils/temporal_nms.py:25,
``l2_normalize`` = utils/basic_utils.py:97, ...).  Everything executes in HIP kernels.
"""
from __future__ import annotations

import math

import torch

from . import _lib


def _dev():
    if not torch.cuda.is_available():
        raise _lib.ConeHipError("cone_amd needs a GPU: there is no CPU execution path")
    return torch.device("cuda", torch.cuda.current_device())


def l2_normalize(x: torch.Tensor, eps: float = 1e-5, clamp: bool = False) -> torch.Tensor:
    """x / (||x|| + eps) along the last dim (utils/basic_utils.py:97-99); with clamp=True
    x / max(||x||, eps), i.e. torch.nn.functional.normalize (run_on_video/cone_localizator.py:129)."""
    lib = _lib.load()
    x2 = x.to(torch.float32).contiguous().view(-1, x.shape[-1])
    out = torch.empty_like(x2)
    _lib.check(lib.cone_l2_normalize_rows(_lib.ptr(x2), x2.shape[0], x2.shape[1], eps, 1 if clamp else 0,
                                          _lib.ptr(out), _lib.stream()))
    return out.view(x.shape)


def num_windows(ctx_l: int, max_v_l: int) -> int:
    return math.ceil(ctx_l / int(max_v_l / 2)) + 1


def prefilter_scores(vid_ctx: torch.Tensor, cls_txt: torch.Tensor, max_v_l: int, frame_scores: bool = True,
                     split_bf16: bool = False):
    """cone/inference.py:284-296 for all queries of one video.

    vid_ctx (ctx_l, dv) adapted+normalised clip features, cls_txt (nq, dv).
    Returns (frame_scores (nq, ctx_l), window_scores (nq, num_window)).  The window max is fused into the
    frame-score stream; ``frame_scores=False`` skips writing the (nq, ctx_l) matrix (returned as None) -- the
    reference only computes it to take the window max, and so does every caller in this package.  ``split_bf16`` (opt-in,
    with ``frame_scores=False``): 8 or more queries run on the bf16 matrix cores, each fp32 product as six partial products of
    three-piece bf16 operands (fp32 accuracy, HBM-bound at 64 queries)."""
    lib = _lib.load()
    ctx_l, dv = vid_ctx.shape
    nq = cls_txt.shape[0]
    W, S = max_v_l, int(max_v_l / 2)
    nw = num_windows(ctx_l, max_v_l)
    fs = torch.empty(nq, ctx_l, device=vid_ctx.device) if frame_scores else None
    ws = torch.empty(nq, nw, device=vid_ctx.device)
    nbytes = (lib.cone_prefilter_scores_split_workspace(ctx_l, nq, W, dv) if split_bf16
              else lib.cone_prefilter_scores_workspace(ctx_l, nq, W))
    scratch = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=vid_ctx.device)
    if split_bf16:      # opt-in: >= 8 queries on the bf16 matrix cores as three-piece operands (cone_prefilter_scores_split)
        if frame_scores:
            raise ValueError("split_bf16 computes the window scores only (frame_scores=False)")
        _lib.check(lib.cone_prefilter_scores_split(_lib.ptr(vid_ctx, torch.float32), ctx_l, dv,
                                                   _lib.ptr(cls_txt, torch.float32), nq, W, S, _lib.ptr(ws),
                                                   _lib.ptr(scratch), scratch.numel(), _lib.stream()))
        return None, ws
    _lib.check(lib.cone_prefilter_scores(_lib.ptr(vid_ctx, torch.float32), ctx_l, dv,
                                         _lib.ptr(cls_txt, torch.float32), nq, W, S, _lib.ptr(fs), _lib.ptr(ws),
                                         _lib.ptr(scratch), scratch.numel(), _lib.stream()))
    return fs, ws


def prefilter_batched(ctx_arena, cls_norm, plan, max_v_l: int, k: int):
    """cone/inference.py:276-301 for every query of a split in three launches.  `plan` is the static
    index metadata built by FeatureStore.prefilter_plan().  Returns (topk_idx (nq,k) int32 with -1
    padding, frame_scores flat, win_scores flat)."""
    lib = _lib.load()
    dev = ctx_arena.device
    nq = cls_norm.shape[0]
    fs = torch.empty(plan["fs_total"], device=dev)
    ws = torch.empty(plan["win_total"], device=dev)
    idx = torch.empty(nq, k, dtype=torch.int32, device=dev)
    _lib.check(lib.cone_prefilter_batched(
        _lib.ptr(ctx_arena, torch.float32), ctx_arena.shape[1], _lib.ptr(cls_norm, torch.float32),
        _lib.ptr(plan["g_row0"], torch.int64), _lib.ptr(plan["g_ctx_l"], torch.int32),
        _lib.ptr(plan["g_q"], torch.int32), plan["ng"], plan["max_ctx_l"], _lib.ptr(plan["q_fs_off"], torch.int64),
        _lib.ptr(plan["q_win_off"], torch.int64), _lib.ptr(plan["q_ctx_l"], torch.int32), nq, max_v_l,
        int(max_v_l / 2), _lib.ptr(fs), _lib.ptr(ws), k, _lib.ptr(idx), _lib.stream()))
    return idx, fs, ws


def topk_windows(win_scores: torch.Tensor, k: int):
    """First k of the stable descending sort of each row (cone/inference.py:297-299, H6)."""
    lib = _lib.load()
    nq, nw = win_scores.shape
    idx = torch.empty(nq, k, dtype=torch.int32, device=win_scores.device)
    val = torch.empty(nq, k, device=win_scores.device)
    nbytes = lib.cone_topk_windows_workspace(nq, nw, k)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=win_scores.device)
    _lib.check(lib.cone_topk_windows_ws(_lib.ptr(win_scores, torch.float32), nq, nw, k, _lib.ptr(idx), _lib.ptr(val),
                                        _lib.ptr(ws), ws.numel(), _lib.stream()))
    return idx, val


def window_table_rows(win_idx, q_ctx_l, q_vid_off, tok_off, tok_len, q_base: int, eval_bsz: int, max_v_l: int,
                      batch_pad=None, n_batches: int = 0, row_q=None, row_slot=None):
    """A5 in one launch instead of ~50 index operations.  win_idx (nq, K) int32; the per-query vectors int32.  The window
    list is (row_q[b], row_slot[b]) (int32 maps from host metadata: a query owns min(K, windows of its video) rows), or the
    dense list of nq * K rows when both are None.  batch_pad (n_batches) int32 = the split's padding table, or None: derived
    from these windows (whole reference batches only).  Returns a dict of (n_rows) int32 columns + ``batch_pad``."""
    lib = _lib.load()
    nq, K = win_idx.shape
    dev = win_idx.device
    n_rows = nq * K if row_q is None else int(row_q.shape[0])
    out = torch.empty(7, n_rows, dtype=torch.int32, device=dev)
    derive = batch_pad is None
    if derive:
        batch_pad = torch.empty(n_batches, dtype=torch.int32, device=dev)
    p = _lib.ptr
    _lib.check(lib.cone_window_table(p(win_idx, torch.int32), nq, K, p(row_q, torch.int32), p(row_slot, torch.int32), n_rows,
                                     p(q_ctx_l, torch.int32), p(q_vid_off, torch.int32),
                                     p(tok_off, torch.int32), p(tok_len, torch.int32), int(q_base), int(eval_bsz),
                                     int(max_v_l), p(batch_pad, torch.int32), int(derive), int(batch_pad.numel()),
                                     *(p(out[i]) for i in range(7)), _lib.stream()))
    names = ("vid_row0", "vid_len", "video_start", "pad_len", "txt_row0", "txt_len", "cls_row")
    res = {k: out[i] for i, k in enumerate(names)}
    res["batch_pad"] = batch_pad
    return res


def compose_rows(logits, spans, match, duration, video_start, clip_length: float, sort: bool = True):
    """cone/inference.py:47-82 -> (B, Nq, 4) fp32 rows [st, ed, prob, match]."""
    lib = _lib.load()
    B, Nq = match.shape
    rows = torch.empty(B, Nq, 4, device=match.device)
    _lib.check(lib.cone_compose_rows(_lib.ptr(logits, torch.float32), _lib.ptr(spans, torch.float32),
                                     _lib.ptr(match, torch.float32), _lib.ptr(duration, torch.int32),
                                     _lib.ptr(video_start, torch.int32), float(clip_length), 1 if sort else 0, B, Nq,
                                     _lib.ptr(rows), _lib.stream()))
    return rows


def fuse_nms(cand: torch.Tensor, n_valid: torch.Tensor, nms_thd: float, max_before: int, max_after: int,
             cand_off: torch.Tensor = None, n_max: int = None):
    """Rounding + fusion + dict collapse + 3x NMS for nq queries (cone/inference.py:83,103-127,205-217).

    cand (nq, n_max, 4) fp32 (or fp64 rows that are already rounded), n_valid (nq,) int32 -- or ``cand_off`` (nq,) int64
    given: cand is ONE (rows, 4) matrix and query q owns rows cand_off[q] .. + n_valid[q] of it (at most ``n_max``).
    Returns rows (3, nq, max_after, 5) fp64, n (3, nq) int32, idx (3, nq, max_after) int32
    in the order fused / proposal / matching."""
    lib = _lib.load()
    if cand_off is None:
        nq, n_max, _ = cand.shape
    else:
        nq = int(n_valid.shape[0])
        if n_max is None or cand.dim() != 2 or cand.shape[1] != 4:
            raise ValueError("fuse_nms with cand_off takes a (rows, 4) matrix and the bound n_max")
    dev = cand.device
    # (the kernel writes every element: kept rows, zero rows / -1 past them, the counts -- no fill launches)
    # rows | n | idx live in ONE buffer: a caller that ships the kept rows to the host copies rows + counts in one transfer
    # (``rows.kept_buf``: the byte range of both)
    R, N = 3 * nq * max_after * 5 * 8, 3 * nq * 4
    N8 = (N + 7) // 8 * 8
    buf = torch.empty(R + N8 + 3 * nq * max_after * 4, dtype=torch.uint8, device=dev)
    rows = buf[:R].view(torch.float64).view(3, nq, max_after, 5)
    n = buf[R:R + N].view(torch.int32).view(3, nq)
    idx = buf[R + N8:].view(torch.int32).view(3, nq, max_after)
    rows.kept_buf = buf[:R + N]
    fn = lib.cone_fuse_nms_f64 if cand.dtype == torch.float64 else lib.cone_fuse_nms
    _lib.check(fn(_lib.ptr(cand), _lib.ptr(cand_off, torch.int64), _lib.ptr(n_valid, torch.int32), nq, int(n_max), float(nms_thd), int(max_before),
                  int(max_after), _lib.ptr(rows), _lib.ptr(n), _lib.ptr(idx), _lib.stream()))
    return rows, n, idx


def temporal_nms(predictions, nms_thd, max_after_nms=100):
    """utils/temporal_nms.py:25-74: list of [st, ed, score] -> kept list, same order/values."""
    if len(predictions) == 1:
        return predictions
    if len(predictions) == 0:
        return []
    lib = _lib.load()
    dev = _dev()
    pred = torch.tensor([[float(p[0]), float(p[1]), float(p[2])] for p in predictions], dtype=torch.float64,
                        device=dev)
    n = pred.shape[0]
    k = min(max_after_nms, n)
    keep = torch.full((max(k, 1),), -1, dtype=torch.int32, device=dev)
    kn = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(lib.cone_temporal_nms(_lib.ptr(pred), n, float(nms_thd), int(k), _lib.ptr(keep), _lib.ptr(kn),
                                     _lib.stream()))
    cnt = int(kn.item())
    keep = keep[:cnt].tolist()
    return [[predictions[i][0], predictions[i][1], predictions[i][2]] for i in keep]


def matcher_cost(pred_logits, pred_spans, tgt_spans, cost_span=10.0, cost_giou=1.0, cost_class=4.0):
    """cone/matcher.py:61-95 for one target span per window; returns (cost (B,Nq), argmin (B,))."""
    lib = _lib.load()
    B, Nq, _ = pred_spans.shape
    cost = torch.empty(B, Nq, device=pred_spans.device)
    best = torch.empty(B, dtype=torch.int32, device=pred_spans.device)
    _lib.check(lib.cone_matcher_cost(_lib.ptr(pred_logits, torch.float32), _lib.ptr(pred_spans, torch.float32),
                                     _lib.ptr(tgt_spans, torch.float32), B, Nq, cost_span, cost_giou, cost_class,
                                     _lib.ptr(cost), _lib.ptr(best), _lib.stream()))
    return cost, best
