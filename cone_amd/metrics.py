"""Validation metrics computed on the device from the kept rows.

Mirrors ``standalone_eval/evaluate_ego4d_nlq.py`` (R@K at IoU thresholds + mIoU, float64),
``standalone_eval/evaluate_mad.py`` (R@K, float32) and ``standalone_eval/evaluate_pre_filtered_window.py``
(window pre-filter recall) of the reference, which loop over queries in python on the decoded JSON.
Here the rows never leave HBM: one kernel counts, per (threshold, K), the queries whose first K predictions
contain an IoU above the threshold (``cone_eval_recall``); only the count table (and the per-query top-1 IoU
for mIoU) comes back.  Divisions by the number of queries are done on the host in the reference's dtype, so
the tables are bit-identical to the reference's (tests/golden/metrics.json).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _host_arrays(thresholds, topK):
    thr = np.ascontiguousarray(np.asarray([float(t) for t in thresholds], dtype=np.float64))
    ks = np.ascontiguousarray(np.asarray([int(k) for k in topK], dtype=np.int32))
    return thr, ks


def recall_counts(rows: torch.Tensor, n: torch.Tensor, gt: torch.Tensor, thresholds, topK, mode: int):
    """rows (nq, A, 5) fp64, n (nq) int32, gt (nq, 2) fp64 on the device -> (hits (n_thr, n_topK) int64,
    top1_iou (nq) fp64), both on the device."""
    lib = _lib.load()
    nq, A = rows.shape[0], rows.shape[1]
    if int(n.min()) < 1:
        raise ValueError("a query has no prediction (the reference indexes predicted_times[0])")
    thr, ks = _host_arrays(thresholds, topK)
    hits = torch.empty(len(thr), len(ks), dtype=torch.int64, device=rows.device)
    top1 = torch.empty(nq, dtype=torch.float64, device=rows.device)
    _lib.check(lib.cone_eval_recall(_lib.ptr(rows, torch.float64), _lib.ptr(n, torch.int32), _lib.ptr(gt, torch.float64),
                                    nq, A, thr.ctypes.data_as(C.c_void_p), len(thr), ks.ctypes.data_as(C.c_void_p),
                                    len(ks), mode, _lib.ptr(hits), _lib.ptr(top1), _lib.stream()))
    return hits, top1


def evaluate_nlq_performance_ego4d(rows, n, gt, thresholds, topK):
    """standalone_eval/evaluate_ego4d_nlq.py:63-115 -> (results (n_thr, n_topK) float64 ndarray, mIoU)."""
    hits, top1 = recall_counts(rows, n, gt, thresholds, topK, 0)
    results = hits.cpu().numpy().astype(np.float64) / float(rows.shape[0])
    miou = np.mean(top1.cpu().numpy().reshape(-1, 1))          # the reference averages a list of (1,) arrays
    return results, miou


def evaluate_nlq_performance_mad(rows, n, gt, thresholds, topK):
    """standalone_eval/evaluate_mad.py:61-107 -> (n_topK, n_thr) float32 tensor (CPU)."""
    hits, _ = recall_counts(rows, n, gt, thresholds, topK, 1)
    out = hits.t().contiguous().cpu().to(torch.float32)         # counts < 2^24: exact, like the += of bools
    out /= rows.shape[0]
    return out


def windows_selection(win_idx: torch.Tensor, gt: torch.Tensor, topK, clip_length: float, max_v_l: int):
    """standalone_eval/evaluate_pre_filtered_window.py:30-72 on the ranked window table (nq, k) int32 (-1 padded;
    k must reach max(topK) or the number of windows) -> (n_topK,) float32 tensor (CPU)."""
    lib = _lib.load()
    nq, k = win_idx.shape
    _, ks = _host_arrays([], topK)
    hits = torch.empty(len(ks), dtype=torch.int64, device=win_idx.device)
    _lib.check(lib.cone_eval_window_recall(_lib.ptr(win_idx, torch.int32), nq, k, _lib.ptr(gt, torch.float64),
                                           float(clip_length), int(max_v_l / 2), ks.ctypes.data_as(C.c_void_p),
                                           len(ks), _lib.ptr(hits), _lib.stream()))
    out = hits.cpu().to(torch.float32)
    out /= nq
    return out


# ------------------------------------------------------------------------------ ground truth -> (nq, 2)
def ego4d_targets(ann, ground_truth):
    """Target spans of the annotation rows (query_id = "<annotation_uid>_<idx>", cone/inference.py:133-140) from
    the nested NLQ json: standalone_eval/evaluate_ego4d_nlq.py:67-77,86-92."""
    table = {}
    for video in ground_truth["videos"]:
        for clip in video["clips"]:
            for a in clip["annotations"]:
                table[(clip["clip_uid"], a["annotation_uid"])] = a
    out = np.empty((len(ann), 2), dtype=np.float64)
    for i, r in enumerate(ann):
        parts = r["query_id"].split("_")
        assert len(parts) == 2
        key = (r["clip_id"], parts[0])
        assert key in table, "Instance not present!"
        q = table[key]["language_queries"][int(parts[1])]
        out[i] = (q["clip_start_sec"], q["clip_end_sec"])
    return out


def jsonl_targets(ann, ground_truth=None):
    """``timestamps`` of every annotation row (MAD: standalone_eval/evaluate_mad.py:78-80)."""
    src = {d["query_id"]: d["timestamps"] for d in (ground_truth if ground_truth is not None else ann)}
    return np.asarray([src[r["query_id"]] for r in ann], dtype=np.float64)


# ------------------------------------------------------------------------------ tables
def _table(rows, title=None):
    """terminaltables.AsciiTable when it is installed (what the reference prints); the same cells in a plain
    box otherwise -- the package is not part of this image, so byte parity of the box drawing is unverified."""
    try:
        import terminaltables
        t = terminaltables.AsciiTable(rows, title)
        for i in range(len(rows[0])):
            t.justify_columns[i] = "center"
        return t.table
    except ImportError:
        cells = [[str(c).split("\n") for c in r] for r in rows]
        widths = [max(len(line) for r in cells for line in r[i]) for i in range(len(rows[0]))]
        sep = "+" + "+".join("-" * (w + 2) for w in widths) + "+"
        out = [sep if not title else "+" + title + sep[len(title) + 1:]]
        for r in cells:
            for k in range(max(len(c) for c in r)):
                out.append("|" + "|".join(" " + (c[k] if k < len(c) else "").center(w) + " " for c, w in zip(r, widths)) + "|")
            out.append(sep)
        return "\n".join(out)


def display_results_ego4d(results, mIoU, thresholds, topK, title=None):
    """standalone_eval/evaluate_ego4d_nlq.py:22-38 (percentages, two decimals)."""
    head = [f"Rank@{k}\nmIoU@{t}" for k in topK for t in thresholds] + ["mIoU"]
    r = np.asarray(results) * 100
    body = [f"{r[j][i]:.02f}" for i in range(len(topK)) for j in range(len(thresholds))] + [f"{mIoU * 100:.02f}"]
    return _table([head, body], title)


def display_results_mad(results, thresholds, topK, title=None):
    """standalone_eval/evaluate_mad.py:15-30."""
    head = [f"Rank@{int(k)}\nmIoU@{float(t):.1f}" for k in topK for t in thresholds]
    r = results * 100
    body = [f"{float(r[i][j]):.02f}" for i in range(len(topK)) for j in range(len(thresholds))]
    return _table([head, body], title)


def display_window_results(results, topK, title=None):
    """standalone_eval/evaluate_pre_filtered_window.py:12-27."""
    r = results * 100
    return _table([[f"Rank@{int(k)}" for k in topK], [f"{float(r[i]):.02f}" for i in range(len(topK))]], title)
