"""Multi-GPU execution of the inference path: one process per GPU, ``torch.distributed`` over RCCL.

The reference is single-device; this is the MI355X-native extension the hot path
allows: after the pre-filter every (query, window) pair is independent until the per-query fusion + NMS.

Two shard modes of ``predict_split_distributed``
  * ``"window"`` -- BASELINE config 4.  The flat (query, window) list is cut into contiguous slices, one per
    rank; each rank projects only the clips / text tokens its slice reads, runs the window model on the slice
    and the per-window proposal rows (Nq x 4 fp32 = 80 B / window) are exchanged with ONE fixed-size
    ``all_gather_into_tensor`` (the slice sizes follow from the replicated window table: no size exchange, no
    host sync).  Fusion + NMS over all queries then run on every rank (one workgroup per query, ~0.2 ms for
    1 000 queries -- cheaper than a second collective), so every rank ends the step holding the kept rows of the
    whole split; each rank formats the submission rows of its own query shard and rank 0 those of all.
    One long video / few queries still fill 8 GPUs.
  * ``"query"``  -- whole queries per rank (fusion + NMS are rank-local); only the kept rows are exchanged, again
    in one fixed-size all_gather.  The pre-filter is replicated (it is HBM-bound and cheap) so that the
    reference-batch padding of hazard H3 is that of the whole split whatever the cut points are.

One huge video (BASELINE configs 3 / 5: MAD-scale ctx_l, features 12.7 GB) shards the PRE-FILTER instead:
``ctx_shard`` cuts the window list into contiguous ranges, each rank holds only the clip rows its windows
cover (a W-S clip halo at the seams), scores them, keeps a local stable top-k and the ranks exchange
k (score, window) pairs per query -- ``prefilter_ctx_sharded``; the merged list equals the single-GPU
stable descending rank list bit for bit.

Messages are tiny (<= 1.6 KB / query) and latency-bound: ONE all_gather per batch of queries, never per
query.  xGMI is point-to-point (7 links per GPU), which a single small all_gather does not stress.

The drivers take their per-rank compute through a small ``hooks`` object (default: the HIP pipeline of
``cone_amd.inference``), so the very same sharding / exchange / assembly code runs on the ``gloo`` backend with
CPU tensors and a CPU checker injected as hooks -- that is how tests/test_parallel_cpu.py covers world sizes 2 and 3.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous near-equal split of range(n): the first (n % world) ranks get one extra item."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


_KEEP_IDX = {}       # (n_total, world, device) -> int64 row indices of the real rows inside the padded gather buffer


def _keep_index(n_total: int, world: int, device) -> torch.Tensor:
    """Rows of the (world * cap) gather buffer that are real: rank r holds shard_range(n_total, r, world) rows at
    r * cap.  Pure host arithmetic, uploaded once per (n_total, world, device) and reused by every later step."""
    key = (n_total, world, str(device))
    idx = _KEEP_IDX.get(key)
    if idx is None:
        cap = -(-n_total // world)
        rows = []
        for r in range(world):
            lo, hi = shard_range(n_total, r, world)
            rows.append(np.arange(r * cap, r * cap + (hi - lo), dtype=np.int64))
        if len(_KEEP_IDX) > 64:
            _KEEP_IDX.clear()
        idx = _KEEP_IDX[key] = torch.from_numpy(np.concatenate(rows)).to(device)
    return idx


def all_gather_fixed(local: torch.Tensor, n_total: int, group=None, virtual=None) -> torch.Tensor:
    """Concatenate, in rank order, per-rank tensors whose dim-0 sizes are ``shard_range(n_total, r, world)``
    -- known on every rank, so there is no size exchange and no host sync: ONE ``all_gather_into_tensor`` on
    a buffer padded to the largest shard (ceil(n_total / world) rows); the padding rows of the short ranks are dropped
    by an ``index_select`` with a cached device index (no boolean mask, no ``nonzero``).

    ``virtual=(rank, world)``: no process group -- the other ranks' shards are filled with copies of the local rows
    (the one-GPU proxy of bench.py's ``shard_proxy_8``: same buffers, same follow-up work, no collective)."""
    if virtual is not None:
        rank, world = virtual
    else:
        world = dist.get_world_size(group)
        rank = dist.get_rank(group)
    lo, hi = shard_range(n_total, rank, world)
    assert local.shape[0] == hi - lo, (local.shape, lo, hi)
    if world == 1:
        return local
    cap = -(-n_total // world)
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != cap:
        send = torch.zeros((cap,) + tail, dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    if virtual is not None:
        recv = send.contiguous().repeat((world,) + (1,) * len(tail))
    else:
        recv = torch.empty((world * cap,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    if n_total == world * cap:
        return recv
    # ranks < n_total % world hold `cap` rows, the others cap - 1: drop each short rank's one padding row
    return recv.index_select(0, _keep_index(n_total, world, recv.device))


def all_gather_rows(local: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate per-rank tensors that differ in dim 0 by amounts NOT known to the other ranks (rank order):
    one size all_gather (8 B / rank, one host sync) + one padded payload all_gather.  The drivers below never
    need it -- their shard sizes are static -- it serves ad-hoc gathers (tools, tests)."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    sizes = sizes.tolist()
    nmax = max(sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    recv = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, pad, group=group)
    return torch.cat([recv[r * nmax:r * nmax + s] for r, s in enumerate(sizes)], dim=0)


def _rank_world(group, virtual):
    return virtual if virtual is not None else (dist.get_rank(group), dist.get_world_size(group))


def run_window_sharded(n_windows: int, compute_rows: Callable[[int, int], torch.Tensor], group=None, virtual=None):
    """Each rank computes rows for its contiguous slice of the window list; returns the rows of ALL
    windows on every rank (the RCCL gather of per-window proposals ahead of the global NMS)."""
    rank, world = _rank_world(group, virtual)
    lo, hi = shard_range(n_windows, rank, world)
    local = compute_rows(lo, hi)
    assert local.shape[0] == hi - lo
    return all_gather_fixed(local, n_windows, group, virtual)


def pack_kept(rows: torch.Tensor, n: torch.Tensor) -> torch.Tensor:
    """(3, nq, A, 5) fp64 kept rows + (3, nq) counts -> ONE (nq, 3, A*5 + 1) fp64 message (count in the last
    slot; exact in fp64), so that a query shard's results travel in a single collective."""
    t, nq, A, _ = rows.shape
    msg = torch.empty(nq, t, A * 5 + 1, dtype=torch.float64, device=rows.device)
    msg[:, :, :A * 5] = rows.permute(1, 0, 2, 3).reshape(nq, t, A * 5)
    msg[:, :, A * 5] = n.t().to(torch.float64)
    return msg


def unpack_kept(msg: torch.Tensor):
    nq, t, w = msg.shape
    A = (w - 1) // 5
    rows = msg[:, :, :A * 5].reshape(nq, t, A, 5).permute(1, 0, 2, 3).contiguous()
    n = msg[:, :, A * 5].t().to(torch.int32).contiguous()
    return rows, n


def run_query_sharded(n_queries: int, compute_kept: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]],
                      group=None, virtual=None):
    """Each rank runs its contiguous query shard; returns (rows, n) of all queries in annotation order on
    every rank.  rows (3, nq, max_after, 5), n (3, nq): ONE fixed-size all_gather."""
    rank, world = _rank_world(group, virtual)
    lo, hi = shard_range(n_queries, rank, world)
    rows, n = compute_kept(lo, hi)
    return unpack_kept(all_gather_fixed(pack_kept(rows, n), n_queries, group, virtual))


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
    ids, -1 padded; val (nq,k)).  One all_gather of k x (4 + 4) B per query (the window ids travel as exact
    fp32 bit patterns next to the scores) -- no feature row ever moves."""
    if window_scores_fn is None or topk_fn is None:
        from . import ops
        window_scores_fn = window_scores_fn or (lambda v, c, w: ops.prefilter_scores(v, c, w, frame_scores=False)[1])
        topk_fn = topk_fn or ops.topk_windows
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    shard = ctx_shard(ctx_l, max_v_l, rank, world)
    val, idx = local_window_topk(ctx_local, shard, cls_norm, max_v_l, k, window_scores_fn, topk_fn)
    if world > 1:
        nq = val.shape[0]
        send = torch.stack([val, idx.view(torch.float32)], dim=0).contiguous()        # (2, nq, k): one message
        recv = torch.empty((world * 2, nq, k), dtype=torch.float32, device=send.device)
        dist.all_gather_into_tensor(recv, send, group=group)
        recv = recv.view(world, 2, nq, k)
        val = recv[:, 0].permute(1, 0, 2).reshape(nq, world * k).contiguous()
        idx = recv[:, 1].contiguous().view(torch.int32).permute(1, 0, 2).reshape(nq, world * k).contiguous()
    return merge_topk(val, idx, k, topk_fn)


# ---------------------------------------------------------------------------------- drivers
class HipHooks:
    """The per-rank compute of the drivers on the HIP pipeline (cone_amd.inference)."""

    def __init__(self, model):
        self.model = model
        self.num_queries = model.num_queries

    def prefilter(self, store, opt):
        from . import inference as inf
        return inf.prefilter(self.model, store, opt)

    def ctx_rows(self, store, f_lo: int, f_hi: int):
        """Adapted + normalised clip rows [f_lo, f_hi) of the (single) video of ``store`` -- what the pre-filter scores
        (cone/inference.py:250-260) -- computed for that range only."""
        from . import ops
        return self.model.adapter_norm(ops.l2_normalize(store.vid_raw[f_lo:f_hi], 1e-5))

    def cls_norm(self, store):
        from . import ops
        return store.cls_raw if store.cls_normalized else ops.l2_normalize(store.cls_raw, 1e-5)

    window_scores_fn = None     # defaults of prefilter_ctx_sharded: the fused HIP window scores / stable top-k
    topk_fn = None

    def project_video(self, store, row_range=None):
        from . import inference as inf
        return inf.project_video(self.model, store, row_range)

    def window_rows(self, store, opt, wt, video):
        """(Nw, Nq, 4) rows of the windows of table ``wt`` (all of them queries of ``store``)."""
        from . import inference as inf
        return inf.run_windows(self.model, store, opt, wt, inf.project_features(self.model, store, video))["rows"]

    def fuse_nms(self, cand, n_valid, opt, cand_off=None, n_max=None):
        """cand (nq, n_max, 4), or -- cand_off given -- ONE (rows, 4) matrix of which query q owns rows cand_off[q] .. +
        n_valid[q] (the gathered per-window rows as they are: inference.candidate_lists)."""
        from . import ops
        rows, n, _ = ops.fuse_nms(cand, n_valid, opt.nms_thd, opt.max_before_nms, opt.max_after_nms, cand_off=cand_off,
                                  n_max=n_max)
        return rows, n


def _slice_table(wt, lo, hi, q_lo, tok_base):
    """Rows [lo, hi) of a window table of the whole split, re-based onto the view that starts at query q_lo
    (whose first text token is row ``tok_base`` of the split's token arena)."""
    sl = {k: v[lo:hi] for k, v in wt.items()}
    sl["q_of"] = sl["q_of"] - q_lo
    sl["cls_row"] = sl["cls_row"] - q_lo
    sl["txt_row0"] = sl["txt_row0"] - tok_base
    return sl


def prefilter_one_video_ctx_sharded(store, opt, hooks, group=None):
    """Stage A of a split that holds ONE long video (BASELINE configs 3 / 5), sharded along ctx_l: this rank adapts,
    normalises and scores only the clip rows its window range covers (``ctx_shard``: 1 / world of the video + a W - S
    halo), keeps a local stable top-k, and ONE all_gather of k (score, window) pairs per query yields the same
    (nq, topk) window table on every rank -- bit-identical to the single-GPU pre-filter, ties included."""
    if len(store.ctx_l) != 1:
        raise ValueError("the ctx-sharded pre-filter takes a split over ONE video; several videos shard by query / window")
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ctx_l = int(store.ctx_l[0])
    _, _, f_lo, f_hi = ctx_shard(ctx_l, opt.max_v_l, rank, world)
    ctx_local = hooks.ctx_rows(store, f_lo, f_hi)
    idx, _ = prefilter_ctx_sharded(ctx_local, ctx_l, hooks.cls_norm(store), opt.max_v_l, opt.topk_window, group,
                                   window_scores_fn=getattr(hooks, "window_scores_fn", None),
                                   topk_fn=getattr(hooks, "topk_fn", None))
    return idx.contiguous()


@torch.no_grad()
def _window_sharded(store, opt, hooks, group, virtual, format_shard):
    """Window-sharded stages A->C with the replicated-feature store (an Ego4D / MAD split of many videos), for ANY mix of
    video lengths.  NOTHING of stage A or of the window table is replicated: the shape of the window list is host metadata
    (``inference.Selection``: query q owns rows row_off[q] .. row_off[q + 1]), so a rank knows its queries [a, b] from the
    cut alone and runs the pre-filter, the window table and the reference-batch padding (hazard H3) only for the
    eval_bsz-ALIGNED hull of that range -- whole reference batches, hence the split's own padding -- then the window model on
    its slice.  ONE fixed-size all_gather of the proposal rows; the gathered (n_win, Nq, 4) buffer IS the per-query candidate
    layout (a query's windows are adjacent: offsets + counts, no scatter); fusion + NMS of all queries on every rank."""
    from . import inference as inf
    rank, world = _rank_world(group, virtual)
    nq, K, Nq, bsz = len(store.ann), opt.topk_window, hooks.num_queries, opt.eval_bsz
    sel = inf.selection(store, opt, K)
    n_win = sel.n_rows
    dev = store.vid_raw.device
    win_hull, hull = None, (0, 0)

    def rows_of(lo, hi):
        nonlocal win_hull, hull
        if hi == lo:
            return torch.zeros(0, Nq, 4, device=dev)
        a, b = sel.query_of_row(lo), sel.query_of_row(hi - 1)
        ha, hb = (a // bsz) * bsz, min(nq, -(-(b + 1) // bsz) * bsz)
        # the two views (and the static index tables they cache on the device) depend on the annotations and the cut
        # only: built once per (store, cut), not once per step
        hull_store, sub = store.view(ha, hb), store.view(a, b + 1)
        win_hull, hull = hooks.prefilter(hull_store, opt), (ha, hb)
        wt = inf.window_table(hull_store, opt, win_hull)            # whole reference batches: the split's padding
        video = hooks.project_video(store, _video_row_range(store, a, b + 1))
        r0 = int(sel.row_off[ha])                                   # first row of the hull's table in the split's list
        table = _slice_table(wt, lo - r0, hi - r0, a - ha, int(store.tok_off[a]) - int(store.tok_off[ha]))
        return hooks.window_rows(sub, opt, table, video)
    rows_all = run_window_sharded(n_win, rows_of, group, virtual)
    cand, cand_off, n_valid, n_max = inf.candidate_lists(rows_all, store, opt, K)
    rows, n = hooks.fuse_nms(cand, n_valid, opt, cand_off, n_max)   # every rank, all queries: cheaper than a second collective
    q_lo, q_hi = shard_range(nq, rank, world)
    info = dict(rows=rows, n=n, win_idx=win_hull, win_idx_range=hull, n_windows=n_win, shard=(q_lo, q_hi), world=world)
    return _pending(store, opt, rows, n, rank, format_shard, info)


def _pending(store, opt, rows, n, rank, format_shard, info):
    """The host half of a distributed step as a ``PendingSplit``: everything the GPU does -- the collective included -- is
    enqueued when this returns (RCCL collectives are stream-ordered: the host does not wait for them); the rows this rank
    formats (its own query shard, or everything on rank 0) are on their way to pinned host memory behind an event, and
    ``result()`` waits for that event only, then builds the submission lists."""
    from . import inference as inf
    q_lo, q_hi = info["shard"]
    if format_shard:
        ann, r, c = store.ann[q_lo:q_hi], rows[:, q_lo:q_hi], n[:, q_lo:q_hi]
    elif rank == 0:
        ann, r, c = store.ann, rows, n
    else:
        return inf.PendingSplit(info, lambda: (None, info))
    ev = None
    if r.is_cuda:
        r, c = inf._to_pinned(r), inf._to_pinned(c)
        ev = torch.cuda.Event()
        ev.record()

    def finish():
        skel = inf.result_skeletons(ann, opt)          # host work that needs no result: under the GPU's time
        if ev is not None:
            ev.synchronize()
        return inf.format_results(ann, opt, r, c, skel), info
    return inf.PendingSplit(info, finish)


def predict_split_distributed(model, store, opt, mode: str = "window", group=None, hooks=None,
                              format_shard: bool = False, prefilter: str = "replicated", virtual=None):
    """``predict_split_distributed_async(...).result()``: one step at a time."""
    return predict_split_distributed_async(model, store, opt, mode, group, hooks, format_shard, prefilter, virtual).result()


@torch.no_grad()
def predict_split_distributed_async(model, store, opt, mode: str = "window", group=None, hooks=None,
                                    format_shard: bool = False, prefilter: str = "replicated", virtual=None):
    """Stages A->C across the ranks of `group`, in two halves like ``inference.predict_split_async``: the device half --
    the rank's share of stages A / B, the collective(s), fusion + NMS -- is enqueued here and nothing waits for it; the host
    half (wait for the kept rows this rank formats, build its submission lists) runs in ``PendingSplit.result()``.  A caller
    that evaluates step after step keeps one in flight: the lists of step i are built while the GPU runs step i + 1 (every
    rank issues its collectives in the same order, so steps in flight cannot cross).

    Every rank holds the same FeatureStore (features replicated:
    an Ego4D split is < 1 GB, the MAD-scale stress video 12.7 GB of the 288 GB per GPU).

    ``prefilter="replicated"``: stage A runs on every rank (HBM-bound and cheap for a split of short videos);
    ``prefilter="ctx"``: ONE long video -- stage A is sharded along ctx_l (``prefilter_one_video_ctx_sharded``: one small
    all_gather), then the window model is sharded by window as below: two collectives per step in all (BASELINE
    configs[4]: 64 queries x one MAD-length video).

    ``virtual=(rank, world)``: replay what that rank of a `world`-rank run computes, on one GPU and without a process
    group -- the gathers are filled with copies of the local shard (bench.py's ``shard_proxy_8``; replicated stage A only).

    Window mode with the replicated pre-filter takes ``_window_sharded``: stage A and the window table run only for the
    rank's own (batch-aligned) query range -- ``info['win_idx']`` then holds the window table of queries
    ``info['win_idx_range']`` instead of the whole split's.  Every cut point comes from host metadata (the shape of the
    window list does not depend on the scores): no rank ever reads a device result back to find its share.

    Returns ``(lists, info)``: ``info['rows'] / info['n']`` = the kept rows of ALL queries, on every rank (tensors);
    ``lists`` = the three submission lists -- of all queries on rank 0 and ``None`` elsewhere, or, with
    ``format_shard=True``, of the rank's own query shard ``info['shard']`` on every rank (the host formatting
    shards with the queries; a caller that wants one file concatenates the shards in rank order).  (Both through
    ``.result()`` of the returned PendingSplit; ``.info`` is valid at once, stream-ordered.)"""
    from . import inference as inf
    hooks = hooks or HipHooks(model)
    rank, world = _rank_world(group, virtual)
    nq = len(store.ann)
    Nq = hooks.num_queries
    if mode == "window" and prefilter == "replicated":
        return _window_sharded(store, opt, hooks, group, virtual, format_shard)      # a PendingSplit
    if prefilter == "ctx":
        if virtual is not None:
            raise ValueError("virtual ranks replay the replicated pre-filter only")
        win_idx = prefilter_one_video_ctx_sharded(store, opt, hooks, group)
    elif prefilter == "replicated":
        win_idx = hooks.prefilter(store, opt)           # replicated: HBM-bound and cheap
    else:
        raise ValueError(f"unknown pre-filter mode {prefilter!r}")
    batch_pad = inf.reference_batch_pad(store, opt, win_idx)
    sel = inf.selection(store, opt, win_idx.shape[1])       # the list's shape: host metadata, whatever win_idx holds
    q_lo, q_hi = shard_range(nq, rank, world)
    if mode == "query":
        def kept(lo, hi):
            if hi == lo:
                A = opt.max_after_nms
                return (torch.zeros(3, 0, A, 5, dtype=torch.float64, device=win_idx.device),
                        torch.zeros(3, 0, dtype=torch.int32, device=win_idx.device))
            sub = store.view(lo, hi)
            wi = win_idx[lo:hi].contiguous()
            wt = inf.window_table(sub, opt, wi, batch_pad)
            vid_rows = _video_row_range(store, lo, hi)
            rows = hooks.window_rows(sub, opt, wt, hooks.project_video(store, vid_rows))
            cand, cand_off, n_valid, n_max = inf.candidate_lists(rows, sub, opt, wi.shape[1])
            return hooks.fuse_nms(cand, n_valid, opt, cand_off, n_max)
        rows, n = run_query_sharded(nq, kept, group, virtual)
        n_windows = sel.n_rows
    elif mode == "window":
        wt = inf.window_table(store, opt, win_idx, batch_pad)
        n_win = sel.n_rows

        def rows_of(lo, hi):
            if hi == lo:
                return torch.zeros(0, Nq, 4, device=win_idx.device)
            # the slice's queries [a, b]: only their text tokens and the clips of their videos are projected
            a, b = sel.query_of_row(lo), sel.query_of_row(hi - 1)       # host arithmetic: no device read
            sub = store.view(a, b + 1)
            # the band of the videos of the slice's queries (for ONE long video that is the whole video: its top-k windows
            # lie anywhere, and 1 / world of the windows of 64 queries already touch most clips -- 33 000 clips: 26 GFLOP of
            # projection against 163 GFLOP of window model per rank at world 8)
            video = hooks.project_video(store, _video_row_range(store, a, b + 1))
            return hooks.window_rows(sub, opt, _slice_table(wt, lo, hi, a, int(store.tok_off[a])), video)
        rows_all = run_window_sharded(n_win, rows_of, group, virtual)
        cand, cand_off, n_valid, n_max = inf.candidate_lists(rows_all, store, opt, win_idx.shape[1])
        rows, n = hooks.fuse_nms(cand, n_valid, opt, cand_off, n_max)   # every rank, all queries: cheaper than a second collective
        n_windows = n_win
    else:
        raise ValueError(f"unknown shard mode {mode!r}")
    info = dict(rows=rows, n=n, win_idx=win_idx, n_windows=n_windows, shard=(q_lo, q_hi), world=world)
    return _pending(store, opt, rows, n, rank, format_shard, info)


def _video_row_range(store, q_lo, q_hi):
    """Arena rows [r0, r1) of the videos the queries [q_lo, q_hi) refer to (queries of a video are adjacent in the
    reference's annotation files, so this is a narrow band; correctness does not depend on it)."""
    vids = np.asarray(store.q_vid[q_lo:q_hi])
    v0, v1 = int(vids.min()), int(vids.max())
    return int(store.vid_off[v0]), int(store.vid_off[v1 + 1])
