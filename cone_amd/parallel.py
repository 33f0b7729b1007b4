"""Multi-GPU execution of the inference path: one process per GPU, ``torch.distributed`` over RCCL.

The reference is single-device; this is the MI355X-native extension the hot path
allows: after the pre-filter every (query, window) pair is independent until the per-query fusion + NMS.

Two shard modes
  * ``"window"`` -- the flat (query, window) list is cut into contiguous slices, one per rank; each rank
    runs the window model on its slice and the per-window proposal rows (Nq x 4 fp32 = 80 B / window)
    are ``all_gather``-ed so that the owner of a query holds all its candidates; fusion + NMS run on
    the owner; the kept rows are gathered for rank 0 to write.  This is BASELINE config 4: one long
    video / few queries still fill 8 GPUs.
  * ``"query"``  -- whole queries per rank (NMS is rank-local); only the kept rows are gathered.  The
    cheapest exchange; preferred when there are at least a few hundred queries per GPU.

One huge video (BASELINE configs 3 / 5: MAD-scale ctx_l, features 12.7 GB) shards the PRE-FILTER instead:
``ctx_shard`` cuts the window list into contiguous ranges, each rank holds only the clip rows its windows
cover (a W-S clip halo at the seams), scores them, keeps a local stable top-k and the ranks exchange
k (score, window) pairs per query -- ``prefilter_ctx_sharded``; the merged list equals the single-GPU
stable descending rank list bit for bit.

Messages are tiny (<= 1.6 KB / query) and latency-bound: ONE all_gather per batch of queries, never per
query.  xGMI is point-to-point (7 links per GPU), which a single small all_gather does not stress.

The communication helpers take the process group and plain tensors, so they run unchanged on the
``gloo`` backend (CPU tensors) -- that is how tests/test_parallel_cpu.py exercises them with 2 ranks.
"""
from __future__ import annotations

from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal split of range(n): the first (n % world) ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate per-rank tensors that differ only in dim 0 (rank order).  One size all_gather
    (8 B / rank) + one padded payload all_gather."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad.contiguous(), group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def run_window_sharded(n_windows: int, compute_rows: Callable[[int, int], torch.Tensor], group=None):
    """Each rank computes rows for its contiguous slice of the window list; returns the rows of ALL
    windows on every rank (the RCCL gather of per-window proposals ahead of the global NMS)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_windows, rank, world)
    local = compute_rows(lo, hi)
    assert local.shape[0] == hi - lo
    return all_gather_rows(local, group)


def run_query_sharded(n_queries: int, compute_kept: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]],
                      group=None):
    """Each rank runs stages A-C for its contiguous query shard; returns (rows, n) of all queries
    in annotation order on every rank.  rows (3, nq, max_after, 5), n (3, nq): gathered along dim 1."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(n_queries, rank, world)
    rows, n = compute_kept(lo, hi)
    rows_all = all_gather_rows(rows.transpose(0, 1).contiguous(), group).transpose(0, 1).contiguous()
    n_all = all_gather_rows(n.transpose(0, 1).contiguous(), group).transpose(0, 1).contiguous()
    return rows_all, n_all


def assemble_candidates(rows_all: torch.Tensor, q_of: torch.Tensor, slot: torch.Tensor, nq: int, K: int):
    """Scatter the gathered per-window rows (Nw, Nq, 4) into per-query candidate lists
    (nq, K*Nq, 4) in (window rank, slot) order -- the order cone/inference.py:141-149 extends
    ``predicted_times`` in."""
    Nq = rows_all.shape[1]
    cand = torch.zeros(nq, K * Nq, 4, dtype=rows_all.dtype, device=rows_all.device)
    cand.view(nq, K, Nq, 4)[q_of, slot] = rows_all
    return cand


# ---------------------------------------------------------------------------------- ctx-sharded pre-filter
def ctx_shard(ctx_l: int, max_v_l: int, rank: int, world: int) -> Tuple[int, int, int, int]:
    """Window range [w_lo, w_hi) owned by `rank` and the clip rows [f_lo, f_hi) those windows read
    (window i covers [max((i-1)S,0), min((i-1)S+W, ctx_l)), cone/inference.py:286-292; S = int(W/2))."""
    W, S = max_v_l, int(max_v_l / 2)
    nw = -(-ctx_l // S) + 1
    w_lo, w_hi = shard_range(nw, rank, world)
    if w_hi == w_lo:
        return w_lo, w_hi, 0, 0
    f_lo = max((w_lo - 1) * S, 0)
    f_hi = min(max((w_hi - 2) * S + W, S), ctx_l)       # window 0 alone reads [0, S)
    return w_lo, w_hi, f_lo, f_hi


def local_window_topk(ctx_local: torch.Tensor, shard: Tuple[int, int, int, int], cls_norm: torch.Tensor,
                      max_v_l: int, k: int, window_scores_fn: Callable, topk_fn: Callable):
    """Stable top-k of the windows this rank owns.  ``ctx_local`` = adapted + normalised clip rows
    [f_lo, f_hi) of the video.  f_lo is a multiple of S, so local window j' is global window
    w_lo + j' - 1 (w_lo > 0) or j' (w_lo = 0); the leading half window and the trailing partial windows of the
    local numbering belong to the neighbours and are dropped.  Returns (val (nq,k) fp32 with -inf padding,
    idx (nq,k) int32 GLOBAL window ids with -1 padding)."""
    w_lo, w_hi, f_lo, f_hi = shard
    nq = cls_norm.shape[0]
    n_own = w_hi - w_lo
    val = torch.full((nq, k), float("-inf"), dtype=torch.float32, device=cls_norm.device)
    idx = torch.full((nq, k), -1, dtype=torch.int32, device=cls_norm.device)
    if n_own == 0:
        return val, idx
    assert ctx_local.shape[0] == f_hi - f_lo, (ctx_local.shape, shard)
    ws = window_scores_fn(ctx_local, cls_norm, max_v_l)                 # (nq, local num_window)
    first = 0 if w_lo == 0 else 1
    own = ws[:, first:first + n_own].contiguous()
    assert own.shape[1] == n_own, (own.shape, shard)
    kk = min(k, n_own)
    li, lv = topk_fn(own, kk)
    val[:, :kk] = lv
    idx[:, :kk] = li.to(torch.int32) + w_lo
    return val, idx


def merge_topk(vals: torch.Tensor, idxs: torch.Tensor, k: int, topk_fn: Callable):
    """Exact merge of per-rank stable top-k lists, concatenated in rank order along dim 1: ranks own
    ascending window ranges and each list is (score desc, window asc), so a STABLE descending sort of
    the concatenation breaks every tie towards the lower window id -- the order of
    ``torch.sort(window_scores, descending=True, stable=True)`` on the whole video (H6)."""
    sel, v = topk_fn(vals.contiguous(), k)
    gi = torch.gather(idxs, 1, sel.to(torch.int64).clamp_(min=0))
    gi = torch.where(torch.isinf(v) & (v < 0), torch.full_like(gi, -1), gi)
    return gi, v


def prefilter_ctx_sharded(ctx_local: torch.Tensor, ctx_l: int, cls_norm: torch.Tensor, max_v_l: int, k: int,
                          group=None, window_scores_fn: Callable = None, topk_fn: Callable = None):
    """Pre-filter (cone/inference.py:284-299) of ONE video whose clip rows are sharded over the ranks of
    `group` as ``ctx_shard`` prescribes.  Every rank returns the same (idx (nq,k) int32 global window
    ids, -1 padded; val (nq,k)).  One all_gather of k x (4 + 4) B per query -- no feature row ever moves."""
    if window_scores_fn is None or topk_fn is None:
        from . import ops
        window_scores_fn = window_scores_fn or (lambda v, c, w: ops.prefilter_scores(v, c, w)[1])
        topk_fn = topk_fn or ops.topk_windows
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shard = ctx_shard(ctx_l, max_v_l, rank, world)
    val, idx = local_window_topk(ctx_local, shard, cls_norm, max_v_l, k, window_scores_fn, topk_fn)
    if world > 1:
        vb = [torch.empty_like(val) for _ in range(world)]
        ib = [torch.empty_like(idx) for _ in range(world)]
        dist.all_gather(vb, val.contiguous(), group=group)
        dist.all_gather(ib, idx.contiguous(), group=group)
        val, idx = torch.cat(vb, dim=1), torch.cat(ib, dim=1)
    return merge_topk(val, idx, k, topk_fn)


# ---------------------------------------------------------------------------------- drivers
@torch.no_grad()
def predict_split_distributed(model, store, opt, mode: str = "window", group=None):
    """Stages A->C across the ranks of `group`; rank 0 returns the three submission lists, the other
    ranks return None.  Every rank holds the same FeatureStore (features replicated: an Ego4D split
    is < 1 GB, the MAD-scale stress video 12.7 GB of the 288 GB per GPU)."""
    from . import inference as inf
    from . import ops
    rank = dist.get_rank(group)
    nq = len(store.ann)
    Nq = model.num_queries
    if mode == "query":
        def kept(lo, hi):
            sub = inf.FeatureStore.subset(store, lo, hi)
            dp = inf.device_pipeline(model, sub, opt)
            return dp["rows"], dp["n"]
        rows, n = run_query_sharded(nq, kept, group)
    elif mode == "window":
        win_idx = inf.prefilter(model, store, opt)          # replicated: HBM-bound and cheap
        wt = inf.window_table(store, opt, win_idx)
        feats = inf.project_features(model, store)
        n_win = int(wt["vid_row0"].shape[0])

        def rows_of(lo, hi):
            sl = {k: v[lo:hi] for k, v in wt.items()}
            if hi == lo:
                return torch.zeros(0, Nq, 4, device=store.device)
            return inf.run_windows(model, store, opt, sl, feats)["rows"]
        rows_all = run_window_sharded(n_win, rows_of, group)
        K = win_idx.shape[1]
        cand = assemble_candidates(rows_all, wt["q_of"], wt["slot"], nq, K)
        n_valid = ((win_idx >= 0).sum(1) * Nq).to(torch.int32)

        def kept(lo, hi):
            r, n_, _ = ops.fuse_nms(cand[lo:hi].contiguous(), n_valid[lo:hi].contiguous(), opt.nms_thd,
                                    opt.max_before_nms, opt.max_after_nms)
            return r, n_
        rows, n = run_query_sharded(nq, kept, group)
    else:
        raise ValueError(f"unknown shard mode {mode!r}")
    if rank != 0:
        return None
    return inf.format_results(store.ann, opt, rows, n)
