"""World-size-2 and -3 tests of the multi-GPU path on the gloo backend (CPU tensors): shard ranges, the
fixed-size all_gather of per-window proposal rows / kept rows, and the window- and query-sharded drivers of
cone_amd.parallel themselves -- real FeatureStore (on the CPU device), real window table, real sharding and
assembly; only the per-rank compute is injected (the CPU oracle), so no GPU is needed."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cone_amd import parallel as par
from cone_amd import synth
from cone_amd.config import make_opt
from oracle import cone_oracle as O


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000, 20001):
        for world in (1, 2, 3, 8):
            spans = [par.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_rows(opt, sd, ann, vf, qf, ranks):
    """Stage B of the oracle for every (query, window) pair, one reference batch at a time: (Nw, Nq, 4)."""
    rows = []
    sd = O.as_torch_sd(sd)
    with torch.no_grad():
        for b0 in range(0, len(ann), opt.eval_bsz):
            metas, mi, ci = O.build_batch(opt, ann[b0:b0 + opt.eval_bsz], vf, qf, ranks)
            out = O.cone_forward(sd, opt, **mi)
            match = O.clip_matching(sd, opt, proposal=out["pred_spans"], **ci)
            comp = O.compose_rows(opt, out["pred_logits"], out["pred_spans"], match,
                                  [m["duration"] for m in metas], [m["video_start"] for m in metas])
            rows.extend(comp)
    return torch.tensor(rows, dtype=torch.float32)


class CheckerHooks:
    """Per-rank compute of predict_split_distributed on the CPU: the oracle's window rows, looked up by (query,
    rank slot), and its fusion + NMS.  While serving a window it CHECKS the window-table row the driver built
    for it -- clip range, text range, and the reference-batch padding of hazard H3 -- against the oracle's own
    collate of the whole split, so a shard that re-derives any of them from its own queries fails here."""
    num_queries = 5

    def __init__(self, opt, sd, ann, vf, qf, store, ranks=None):
        self.opt, self.ann, self.store = opt, ann, store
        self.ranks = ranks if ranks is not None else O.prefilter(sd, opt, ann, vf, qf)[0]
        sdt = O.as_torch_sd(sd)
        self.rows, self.meta = {}, {}
        with torch.no_grad():
            for b0 in range(0, len(ann), opt.eval_bsz):
                metas, mi, ci = O.build_batch(opt, ann[b0:b0 + opt.eval_bsz], vf, qf, self.ranks)
                out = O.cone_forward(sdt, opt, **mi)
                match = O.clip_matching(sdt, opt, proposal=out["pred_spans"], **ci)
                comp = O.compose_rows(opt, out["pred_logits"], out["pred_spans"], match,
                                      [m["duration"] for m in metas], [m["video_start"] for m in metas])
                pad = int(mi["src_vid_motion"].shape[1])
                qpos = {r["query_id"]: b0 + i for i, r in enumerate(ann[b0:b0 + opt.eval_bsz])}
                seen = {}
                for m, r in zip(metas, comp):
                    q = qpos[m["query_id"]]
                    slot = seen.get(q, 0)
                    seen[q] = slot + 1
                    self.rows[(q, slot)] = torch.tensor(r, dtype=torch.float32)
                    self.meta[(q, slot)] = (m["duration"], m["video_start"], pad)
        self.served = 0

    def prefilter(self, store, opt):
        K = opt.topk_window
        wi = torch.full((len(store.ann), K), -1, dtype=torch.int32)
        for i, r in enumerate(store.ann):
            lst = self.ranks[r["query_id"]][:K]
            wi[i, :len(lst)] = torch.tensor(lst, dtype=torch.int32)
        return wi

    # ctx-sharded pre-filter of ONE long video (exactly representable inputs: see _long_video_case): raw rows / raw cls
    def ctx_rows(self, store, f_lo, f_hi):
        self.ctx_rows_served = f_hi - f_lo
        return store.vid_raw[f_lo:f_hi].clone()

    def cls_norm(self, store):
        return store.cls_raw

    window_scores_fn = staticmethod(lambda v, c, W: _cpu_window_scores(v, c, W))
    topk_fn = staticmethod(lambda x, k: _cpu_topk(x, k))

    def project_video(self, store, row_range=None):
        r0, r1 = row_range if row_range is not None else (0, int(store.vid_raw.shape[0]))
        return dict(vid_base=r0, rows=(r0, r1))

    def window_rows(self, sub, opt, wt, video):
        out = []
        r0, r1 = video["rows"]
        for w in range(int(wt["vid_row0"].shape[0])):
            lq = int(wt["q_of"][w])
            assert 0 <= lq < len(sub.ann)
            q = lq + sub.q_base
            dur, vstart, pad = self.meta[(q, int(wt["slot"][w]))]
            assert int(wt["vid_len"][w]) == dur and int(wt["video_start"][w]) == vstart
            assert int(wt["pad_len"][w]) == pad, ("H3 padding", q, int(wt["pad_len"][w]), pad)
            row0 = int(self.store.vid_off[self.store.q_vid[q]]) + vstart
            assert int(wt["vid_row0"][w]) == row0 and r0 <= row0 and row0 + dur <= r1      # inside the projected band
            assert int(wt["txt_len"][w]) == sub.tok_len[lq] and int(wt["txt_row0"][w]) == int(sub.tok_off[lq])
            assert int(wt["cls_row"][w]) == lq
            out.append(self.rows[(q, int(wt["slot"][w]))])
            self.served += 1
        return torch.stack(out) if out else torch.zeros(0, 5, 4)

    def fuse_nms(self, cand, n_valid, opt, cand_off=None, n_max=None):
        A = opt.max_after_nms
        nq = int(n_valid.shape[0])
        r = torch.zeros(3, nq, A, 5, dtype=torch.float64)
        n = torch.zeros(3, nq, dtype=torch.int32)
        for qi in range(nq):
            if cand_off is None:
                mine = cand[qi, :int(n_valid[qi])]
            else:           # ONE (rows, 4) matrix: query qi owns rows cand_off[qi] .. + n_valid[qi]
                assert cand.dim() == 2 and int(n_valid[qi]) <= n_max
                mine = cand[int(cand_off[qi]):int(cand_off[qi]) + int(n_valid[qi])]
            rd = O.score_fusion(O.round4_rows(mine.tolist()))
            for t, idx in enumerate((2, 0, 1)):
                keep = O.post_processing_mr_nms(opt, rd, idx)
                n[t, qi] = len(keep)
                if keep:
                    r[t, qi, :len(keep)] = torch.tensor(keep, dtype=torch.float64)
        return r, n


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. gathers: ragged (sizes exchanged) and fixed (sizes = shard_range, no exchange)
        local = torch.arange((rank + 1) * 6, dtype=torch.float32).view(rank + 1, 2, 3) + 100 * rank
        got = par.all_gather_rows(local)
        exp = torch.cat([torch.arange((r + 1) * 6, dtype=torch.float32).view(r + 1, 2, 3) + 100 * r
                         for r in range(world)])
        assert torch.equal(got, exp)
        assert par.all_gather_rows(torch.zeros(0, 5)).shape == (0, 5)
        for n_total in (0, 1, world - 1, world, world + 1, 7 * world + 2):
            lo, hi = par.shard_range(n_total, rank, world)
            mine = torch.arange(lo, hi, dtype=torch.float64).view(-1, 1).repeat(1, 3)
            got = par.all_gather_fixed(mine, n_total)
            assert torch.equal(got, torch.arange(n_total, dtype=torch.float64).view(-1, 1).repeat(1, 3)), n_total
        rows = torch.rand(3, 4, 5, 5, dtype=torch.float64)
        n = torch.randint(0, 6, (3, 4), dtype=torch.int32)
        r2, n2 = par.unpack_kept(par.pack_kept(rows, n))
        assert torch.equal(r2, rows) and torch.equal(n2, n)

        # 2. the drivers themselves (real window table, shard cuts INSIDE reference batches, short videos so
        # that the H3 padding differs between batches) == the single-process oracle, bit for bit
        from cone_amd import inference as inf
        opt = make_opt("ego4d", nms_thd=0.5, topk_window=3, eval_bsz=4, max_after_nms=5, eval_split_name="test")
        sd = synth.make_state_dict(opt, 0)
        for nq, nv, seed, ctx_range in ((10, 3, 9, (20, 120)), (5, 2, 4, (60, 200)), (2, 1, 1, (30, 50))):
            ann, vf, qf = synth.make_dataset(opt, nq, nv, seed=seed, ctx_range=ctx_range)
            store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
            hooks = CheckerHooks(opt, sd, ann, vf, qf, store)
            (fo, po, mo), _, _ = O.eval_epoch(sd, opt, ann, vf, qf)
            for mode in ("window", "query"):
                hooks.served = 0
                lists, info = par.predict_split_distributed(None, store, opt, mode=mode, hooks=hooks)
                assert info["world"] == world
                for t, ref in enumerate((fo, po, mo)):        # every rank holds every query's kept rows
                    for qi in range(nq):
                        got_rows = info["rows"][t, qi, :int(info["n"][t, qi])].tolist()
                        assert got_rows == ref[qi]["predicted_times"], (mode, t, qi)
                if rank == 0:
                    assert lists == (fo, po, mo), mode
                else:
                    assert lists is None
                # the window model ran on a shard only, not on the whole split
                assert hooks.served <= -(-info["n_windows"] // world) + opt.topk_window, (mode, hooks.served)
                part, info2 = par.predict_split_distributed(None, store, opt, mode=mode, hooks=hooks, format_shard=True)
                lo, hi = info2["shard"]
                assert part == tuple(x[lo:hi] for x in (fo, po, mo)), mode
        # a view cut inside a reference batch refuses to derive the padding from its own windows
        ann, vf, qf = synth.make_dataset(opt, 10, 3, seed=9, ctx_range=(20, 120))
        store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
        hooks = CheckerHooks(opt, sd, ann, vf, qf, store)
        wi = hooks.prefilter(store, opt)
        sub = inf.FeatureStore.subset(store, 3, 8)
        try:
            inf.window_table(sub, opt, wi[3:8])
            raise AssertionError("unaligned view accepted without the split's padding table")
        except ValueError:
            pass
        full = inf.window_table(store, opt, wi)
        part = inf.window_table(sub, opt, wi[3:8], inf.reference_batch_pad(store, opt, wi))
        sel = (full["q_of"] >= 3) & (full["q_of"] < 8)
        assert torch.equal(part["pad_len"], full["pad_len"][sel]) and torch.equal(part["vid_row0"], full["vid_row0"][sel])
        assert len(set(full["pad_len"].tolist())) > 1         # the case is sensitive to the batch a window sits in
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_window_and_query_sharded_drivers(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=400)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(timeout=5) == "ok"


def _wide_worker(rank, world, port, out):
    """World 4 / 8 on ONE ragged split (videos of 1 .. 5 windows, fewer than top-k for most; shard cuts inside reference
    batches and inside queries; ranks that own a handful of windows): the window-sharded driver's hull / cut arithmetic,
    through the ASYNC entry with two steps in flight."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)            # up to 8 ranks on the container's 8 cores
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cone_amd import inference as inf
        opt = make_opt("ego4d", nms_thd=0.5, topk_window=4, eval_bsz=3, max_after_nms=5, eval_split_name="test")
        sd = synth.make_state_dict(opt, 0)
        ann, vf, qf = synth.make_dataset(opt, 11, 4, seed=6, ctx_range=(10, 200))
        store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
        sel = inf.selection(store, opt)
        assert not sel.dense and sel.n_rows >= world            # ragged, and every rank owns at least one window
        hooks = CheckerHooks(opt, sd, ann, vf, qf, store)       # checks every window-table row it is asked for (H3 padding too)
        (fo, po, mo), _, _ = O.eval_epoch(sd, opt, ann, vf, qf)
        h1 = par.predict_split_distributed_async(None, store, opt, mode="window", hooks=hooks, format_shard=True)
        served1 = hooks.served
        h2 = par.predict_split_distributed_async(None, store, opt, mode="window", hooks=hooks)      # second step enqueued first
        for h, shard_only in ((h1, True), (h2, False)):
            lists, info = h.result()
            assert info["world"] == world and info["n_windows"] == sel.n_rows
            for t, ref in enumerate((fo, po, mo)):              # every rank holds every query's kept rows
                for qi in range(len(ann)):
                    assert info["rows"][t, qi, :int(info["n"][t, qi])].tolist() == ref[qi]["predicted_times"], (t, qi)
            lo, hi = info["shard"]
            if shard_only:
                assert lists == tuple(x[lo:hi] for x in (fo, po, mo))
            else:
                assert (lists == (fo, po, mo)) if rank == 0 else (lists is None)
            ha, hb = info["win_idx_range"]                      # stage A ran on the batch-aligned hull of the rank's queries
            assert ha % opt.eval_bsz == 0 and (hb % opt.eval_bsz == 0 or hb == len(ann)) and hb - ha <= len(ann)
        wlo, whi = par.shard_range(sel.n_rows, rank, world)
        assert served1 == whi - wlo                             # the window model ran on the rank's slice, nothing more
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_gloo_window_sharded_async_driver_wide_worlds(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wide_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=400)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(timeout=5) == "ok"


# ---------------------------------------------------------------------------- ctx-sharded pre-filter
def _cpu_window_scores(v, c, W):
    return torch.stack([O.window_scores(O.frame_scores(v, c[q]), W) for q in range(c.shape[0])])


def _cpu_topk(x, k):
    val, idx = torch.sort(x, dim=1, descending=True, stable=True)
    return idx[:, :k].to(torch.int32), val[:, :k]


def _prefilter_case(ctx_l, W, seed, nq=3, dv=16):
    g = torch.Generator().manual_seed(seed)
    vid = torch.randn(ctx_l, dv, generator=g)
    vid = (vid * 4).round() / 4          # coarse values -> many exactly tied frame / window scores (H6)
    cls = (torch.randn(nq, dv, generator=g) * 2).round() / 2
    return vid, cls


def test_ctx_shard_geometry():
    for W in (90, 125):
        S = int(W / 2)
        for ctx_l in (1, S - 1, S, S + 1, W, W + 1, 901, 5000):
            nw = O.num_windows(ctx_l, W)
            for world in (1, 2, 3, 8):
                sh = [par.ctx_shard(ctx_l, W, r, world) for r in range(world)]
                assert sh[0][0] == 0 and sh[-1][1] == nw
                for w_lo, w_hi, f_lo, f_hi in sh:
                    for i in range(w_lo, w_hi):          # every owned window lies inside the rank's clip rows
                        a, b = O.window_bounds(i, ctx_l, W)
                        assert f_lo <= a and b <= f_hi, (ctx_l, W, world, i)
                    if w_hi > w_lo:
                        assert f_lo % S == 0
                        assert f_hi - f_lo <= (w_hi - w_lo + 1) * S + W   # halo only, never the whole video


def _ctx_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        for W, ctx_l, k in ((90, 901, 20), (90, 1, 5), (90, 44, 5), (90, 46, 3), (90, 333, 5), (125, 700, 30),
                            (125, 2000, 7)):
            vid, cls = _prefilter_case(ctx_l, W, seed=ctx_l + W)
            w_lo, w_hi, f_lo, f_hi = par.ctx_shard(ctx_l, W, rank, world)
            idx, val = par.prefilter_ctx_sharded(vid[f_lo:f_hi].clone(), ctx_l, cls, W, k,
                                                 window_scores_fn=_cpu_window_scores, topk_fn=_cpu_topk)
            full = _cpu_window_scores(vid, cls, W)
            for q in range(cls.shape[0]):
                ref = O.rank_windows(full[q])[:k]
                got = idx[q].tolist()
                assert got[:len(ref)] == ref, (W, ctx_l, k, q, got, ref)
                assert all(g == -1 for g in got[len(ref):])
                assert torch.equal(val[q, :len(ref)], full[q][ref])
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ctx_sharded_prefilter_equals_full_rank_list(world):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ctx_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(timeout=5) == "ok"


# ---------------------------------------------------------------------------- ctx-sharded pre-filter -> window-sharded model
def _long_video_case(opt, nq, ctx_l, seed):
    """ONE long video + nq queries whose pre-filter inputs are exactly representable (clip / cls features on a coarse
    grid: every frame score is exact in fp32 whatever the summation order, so the sharded and the whole-video scores
    agree bit for bit and ties -- hazard H6 -- are plentiful)."""
    ann, vf, qf = synth.make_dataset(opt, nq, 1, seed=seed, ctx_range=(ctx_l, ctx_l + 1))
    g = np.random.default_rng(seed)
    for k in vf:
        vf[k] = (np.round(g.standard_normal(vf[k].shape) * 4) / 4).astype(np.float32)
    for k in qf:
        key = "cls_features" if "cls_features" in qf[k] else "eot_features"
        qf[k][key] = (np.round(g.standard_normal(np.asarray(qf[k][key]).shape) * 2) / 2).astype(np.float32)
    return ann, vf, qf


def _composed_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cone_amd import inference as inf
        opt = make_opt("ego4d", nms_thd=0.5, topk_window=4, eval_bsz=3, max_after_nms=5, eval_split_name="test")
        sd = synth.make_state_dict(opt, 0)
        for nq, ctx_l, seed in ((7, 700, 3), (2, 95, 5)):
            ann, vf, qf = _long_video_case(opt, nq, ctx_l, seed)
            store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"), cls_normalized=True)
            # the single-process answer: rank lists of the whole video with the same (raw) scoring, then the oracle
            full = _cpu_window_scores(store.vid_raw, store.cls_raw, opt.max_v_l)
            ranks = {r["query_id"]: O.rank_windows(full[i]) for i, r in enumerate(ann)}
            hooks = CheckerHooks(opt, sd, ann, vf, qf, store, ranks=ranks)
            with torch.no_grad():
                mr = O.compute_mr_results(sd, opt, ann, vf, qf, ranks)
            fo, po, mo = O.postprocess(mr, opt)
            hooks.served = 0
            lists, info = par.predict_split_distributed(None, store, opt, mode="window", hooks=hooks, prefilter="ctx")
            # stage A ran on this rank's clip range only (1 / world of the video + the halo) ...
            w_lo, w_hi, f_lo, f_hi = par.ctx_shard(ctx_l, opt.max_v_l, rank, world)
            assert hooks.ctx_rows_served == f_hi - f_lo < ctx_l or world == 1 or ctx_l < 2 * opt.max_v_l
            # ... and produced the whole video's rank lists on every rank, bit for bit
            for qi, r in enumerate(ann):
                assert [w for w in info["win_idx"][qi].tolist() if w >= 0] == ranks[r["query_id"]][:opt.topk_window]
            for t, ref in enumerate((fo, po, mo)):
                for qi in range(nq):
                    assert info["rows"][t, qi, :int(info["n"][t, qi])].tolist() == ref[qi]["predicted_times"], (t, qi)
            assert (lists == (fo, po, mo)) if rank == 0 else (lists is None)
            assert hooks.served <= -(-info["n_windows"] // world) + opt.topk_window
        # virtual ranks (no collective: bench.py's shard proxy) follow the same cuts as real ones
        ann, vf, qf = synth.make_dataset(opt, 9, 2, seed=2, ctx_range=(60, 200))
        store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
        hooks = CheckerHooks(opt, sd, ann, vf, qf, store)
        _, real = par.predict_split_distributed(None, store, opt, mode="window", hooks=hooks, format_shard=True)
        served_real = hooks.served
        hooks.served = 0
        _, virt = par.predict_split_distributed(None, store, opt, mode="window", hooks=hooks, format_shard=True,
                                                virtual=(rank, world))
        assert virt["shard"] == real["shard"] and virt["n_windows"] == real["n_windows"]
        assert hooks.served == served_real                  # the same slice of the window list went through the model
        lo, hi = par.shard_range(real["n_windows"], rank, world)
        assert hooks.served == hi - lo
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_ctx_sharded_prefilter_feeds_window_sharded_model(world):
    """BASELINE configs[4]'s composition (one long video: pre-filter sharded along ctx_l -> window-sharded model -> one
    gather of proposal rows -> fusion + NMS): equals the single-process oracle bit for bit on every rank."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_composed_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=400)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert out.get(timeout=5) == "ok"
