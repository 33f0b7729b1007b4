"""World-size-2 tests of the multi-GPU plumbing on the gloo backend (CPU tensors): shard ranges, the
padded all_gather of per-window proposal rows, window- and query-sharded drivers.  The per-rank compute
is injected (here: the CPU oracle), so no GPU is needed."""
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


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. ragged all_gather
        local = torch.arange((rank + 1) * 6, dtype=torch.float32).view(rank + 1, 2, 3) + 100 * rank
        got = par.all_gather_rows(local)
        exp = torch.cat([torch.arange((r + 1) * 6, dtype=torch.float32).view(r + 1, 2, 3) + 100 * r
                         for r in range(world)])
        assert torch.equal(got, exp)
        assert par.all_gather_rows(torch.zeros(0, 5)).shape == (0, 5)

        # 2. window-sharded stage B + owner-side stage C == single-process oracle
        opt = make_opt("ego4d", nms_thd=0.5, topk_window=3, eval_bsz=2, max_after_nms=5)
        sd = synth.make_state_dict(opt, 0)
        ann, vf, qf = synth.make_dataset(opt, 5, 2, seed=9, ctx_range=(60, 200))
        ranks, _ = O.prefilter(sd, opt, ann, vf, qf)
        all_rows = _oracle_rows(opt, sd, ann, vf, qf, ranks)          # every rank could compute all; it only uses its slice
        n_win = all_rows.shape[0]
        rows = par.run_window_sharded(n_win, lambda lo, hi: all_rows[lo:hi].clone())
        assert torch.equal(rows, all_rows)
        K = opt.topk_window
        q_of = torch.tensor([qi for qi, r in enumerate(ann) for _ in ranks[r["query_id"]][:K]])
        slot = torch.tensor([s for r in ann for s in range(len(ranks[r["query_id"]][:K]))])
        cand = par.assemble_candidates(rows, q_of, slot, len(ann), K)

        def kept(lo, hi):          # owner-side fusion + NMS with the oracle (python doubles)
            A = opt.max_after_nms
            r = torch.zeros(3, hi - lo, A, 5, dtype=torch.float64)
            n = torch.zeros(3, hi - lo, dtype=torch.int32)
            for qi in range(lo, hi):
                nv = len(ranks[ann[qi]["query_id"]][:K]) * 5
                rd = O.score_fusion(O.round4_rows(cand[qi, :nv].tolist()))
                for t, idx in enumerate((2, 0, 1)):
                    keep = O.post_processing_mr_nms(opt, rd, idx)
                    n[t, qi - lo] = len(keep)
                    if keep:
                        r[t, qi - lo, :len(keep)] = torch.tensor(keep, dtype=torch.float64)
            return r, n
        rows_k, n_k = par.run_query_sharded(len(ann), kept)
        if rank == 0:
            (fo, po, mo), _, _ = O.eval_epoch(sd, opt, ann, vf, qf)
            for t, ref in enumerate((fo, po, mo)):
                for qi in range(len(ann)):
                    got_rows = rows_k[t, qi, :int(n_k[t, qi])].tolist()
                    assert got_rows == ref[qi]["predicted_times"], (t, qi)
            out.put("ok")
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_window_and_query_sharding():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
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
