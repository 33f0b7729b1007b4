#!/usr/bin/env python3
"""Benchmark of the CONE coarse-to-fine inference hot path on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 3 --warmup 1         # starts its own 8 ranks (torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = stages A->C (pre-filter, top-k windows, Moment-DETR window model, proposal matching,
fusion, 3x NMS, JSON rows) over BASELINE.json configs[1]: Ego4D-NLQ val-scale synthetic split
(1 000 queries over 50 videos, ctx_l~U[850,950), window_len 90, d 256, top-k 20 => 20 000 windows,
NMS 0.5) with every feature already resident in HBM.

Prints ONE JSON line (rank 0).  `value` (strong scaling): ONE such split per step whatever N -- at N > 1 it is sharded by
window over the ranks (BASELINE configs[3]: cone_amd.parallel.predict_split_distributed_async, one RCCL all_gather of the
per-window proposal rows ahead of the global NMS).  Extra objects on the same line:
  * `roofline`      -- the dominant kernel (the fp32-MFMA GEMM tile that accumulates the most time) priced from
                       hipEvent timings taken around each of its launches inside the timed region (cone_prof_*),
                       FLOPs = 2*M*N*K with the M actually processed;
  * `weak_scaling` (N > 1) -- every rank its own split of that size, the kept rows of all shards all-gathered (the N > 1
                       HEADLINE is BASELINE configs[3]: ONE split sharded by window, strong scaling -- see `value` below);
  * `dropin_forward`, `localizer` (N = 1) -- the reference's own Python API on the same kernels: its 640-window padded
                       batch through model(**inputs) + forward_clip_matching, and CONELocalizator.predict_moment;
  * `config5_sharded`, `prefilter_mad_ctx_sharded` (N > 1) -- BASELINE configs[4] and [2] over the N ranks: pre-filter
                       sharded along ctx_l, window model sharded by window;
  * `shard_proxy_2 / _4 / _8` (N = 1) -- rank 0's share of the 2 / 4 / 8-rank window-sharded split replayed on one GPU (no
                       collective) and the strong-scaling efficiency it projects;
  * `ms_per_step_dead_work_elided` -- the headline step WITHOUT the saliency head and the intermediate decoder layer's heads
                       (the headline computes both, as the reference's CONE.forward does -- and never reads them);
  * `config2_ragged` (N = 1) -- the headline workload on a ragged split (ctx_l ~ U[200, 1500)): per-window cost against the
                       dense split's, on the same sync-free path;
  * `prefilter_mad`, `latency_config1`, `config5` (N = 1) -- BASELINE configs[2], [0] and [4] on one GPU: the
                       MAD-scale pre-filter against the HBM roofline (1 and 64 queries), the single-query latency,
                       64 queries x one MAD-length video end to end;
  * `cpu_baseline`  -- the CPU oracle (a torch-CPU port of the reference path) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC: RCCL / device-memory sharing across the ranks' processes needs it on this driver (also when the ranks are
# started by torch.distributed.run directly rather than by self_launch below); harmless for one process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from cone_amd import _lib, ops, synth  # noqa: E402
from cone_amd import inference as inf  # noqa: E402
from cone_amd.config import make_opt  # noqa: E402
from cone_amd.model import build_model  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
HBM_PEAK_GBS = 8000.0           # same guide: HBM3E peak BW (spec); 6.29 TB/s measured float4 copy
HBM_COPY_GBS = 6290.0           # ... that measured copy rate (MI355X_MICROARCH.md, chip level)
KERNEL_NAMES = {0: "gemm_f32_kernel<128,128,false>", 1: "gemm_f32_kernel<128,128,true>",
                2: "gemm_f32_kernel<64,256,false>", 3: "enc_attn16_kernel", 4: "frame_score_kernel",
                5: "gemm_rows_kernel<16>", 6: "gemm_rows16_kernel", 7: "dec_cross_mfma_kernel", 8: "ffn_fused_kernel<false, false, 8, false>",
                9: "ffn_fused_kernel<true, false, 8, false>", 10: "ffn_wide_kernel<false>", 11: "ffn_wide_kernel<true>",
                12: "ffn_fused_kernel<false, false, 4, false>", 13: "ffn_fused_kernel<true, false, 4, false>",
                14: "gemm_rows_small_kernel"}
# one kind per KERNEL (include/cone_hip.h, cone_prof_collect): a record is one launch of that kernel and its own rows
GEMM_KINDS = (0, 1, 2, 5, 6, 8, 9, 10, 11, 12, 13, 14)   # records (kind, M, N, K, ms): 2*M*N*K FLOPs; the layer-tail kinds
#                                      (N = ff, K = 256): the feed-forward block = two such GEMMs, + the 256 x 256 output
#                                      projection for the kinds that start at the attention output
FFN_KINDS, FFN_PROJ_KINDS = (8, 9, 10, 11, 12, 13), (9, 11, 13)
# where each priced kernel is defined: the committed counter tables carry the hashes of these sources (tools/pmc_summary.py)
KERNEL_SOURCES = {"ffn_fused_kernel": ("ffn.hip", "common.h"), "frame_score_kernel": ("prefilter.hip", "common.h"),
                  "frame_score_mq_kernel": ("prefilter.hip", "common.h"), "gemm_rows16_kernel": ("gemm.hip", "common.h"),
                  "frame_score_mq3_kernel": ("prefilter.hip", "common.h")}


def rec_flops(kind, a, b, c):
    f = 2.0 * a * b * c
    if kind in FFN_KINDS:
        f *= 2.0
    if kind in FFN_PROJ_KINDS:
        f += 2.0 * a * 256 * 256
    return f


def csrc_hashes():
    """sha1 of every kernel source (cone_amd/csrc/*.hip, common.h): what a counter table was collected on."""
    import hashlib
    d = os.path.join(ROOT, "cone_amd", "csrc")
    return {f: hashlib.sha1(open(os.path.join(d, f), "rb").read()).hexdigest()[:12]
            for f in sorted(os.listdir(d)) if f.endswith((".hip", ".h", ".c"))}


PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")
PMC_PREFILTER_FILES = ("r06_pmc_prefilter.json", "r05_pmc_prefilter.json", "r04_pmc_prefilter.json")


def collect_profile():
    lib = _lib.load()
    cap = 1 << 16
    buf = np.zeros((cap, 5), dtype=np.float64)
    n = lib.cone_prof_collect(buf.ctypes.data, cap)
    if n < 0:
        raise RuntimeError(lib.cone_last_error().decode())
    return buf[:n]


def reference_window_flops(lv, lq, dv, dt, d=256, ff=1024, nq=5, enc=2, dec=2):
    """Algorithmic FLOPs of the reference's window model + matching for windows of lv clips and lq text tokens
    (2*M*N*K per dense layer, 4*L^2*d per attention, padding not counted; 486.1 MFLOP at
    lv=90, lq=20, dv=256, dt=768).  The reference computes all of it per window; this build de-duplicates the
    input projections and the first in_proj across windows and folds the decoder's memory K/V projections, so it
    executes fewer FLOPs than this for the same outputs."""
    lv, lq = np.asarray(lv, dtype=np.float64), np.asarray(lq, dtype=np.float64)
    L = lv + lq
    proj = 2 * lv * (dv * d + d * d) + 2 * lq * (dt * d + d * d)
    enc_f = enc * (2 * L * d * 3 * d + 2 * L * d * d + 4 * L * L * d + 2 * L * d * ff * 2)
    dec_f = dec * (2 * L * d * 2 * d                                   # memory K/V projection
                   + 2 * nq * d * 3 * d + 2 * nq * d * d + 4 * nq * nq * d          # self-attention over the slots
                   + 2 * nq * d * d * 2 + 4 * nq * L * d                            # cross-attention q/out + scores
                   + 2 * nq * d * ff * 2                                            # FFN
                   + 2 * nq * (d * 2 + d * d * 2 + d * 2))                          # class + span heads (aux too)
    rest = 2 * lv * d + 2 * nq * (dv * d * 2) + 2 * nq * dv                         # saliency, adapter on proposals, cosine
    return proj + enc_f + dec_f + rest


def executed_mfma_flops(rec, lv, lq, opt, steps, d=256, nq=5, heads=8):
    """FLOPs the MFMA kernels of `steps` steps executed: every dense layer from its launch record (2 M N K with the M the
    launch processed), the encoder attention cores as 4 L^2 d per window and layer, the folded decoder cross-attention as
    (fold W_k into the nq x heads queries) + scores + P.memory + W_v per window and layer."""
    lv, lq = np.asarray(lv, dtype=np.float64), np.asarray(lq, dtype=np.float64)
    L = lv + lq
    dense = sum(rec_flops(int(k), a, b, c) for k, a, b, c, ms in rec if int(k) in GEMM_KINDS)
    hd = d // heads
    enc = steps * opt.enc_layers * float((4.0 * L * L * d).sum())
    per_win = 2.0 * nq * heads * hd * d + 2.0 * nq * heads * L * d * 2 + 2.0 * nq * heads * d * hd
    dec = steps * opt.dec_layers * float(per_win.sum())
    return {"dense_layers": dense, "encoder_attention": enc, "decoder_cross_attention": dec, "total": dense + enc + dec}


def pmc_traffic(kernel, files=PMC_FILES):
    """HBM bytes per launch of `kernel` from the committed PMC passes of this same command (profiles/
    rNN_pmc_*.json, written by tools/pmc_summary.py: 2 x FETCH_SIZE + WRITE_SIZE -- the guide's gfx950 read
    correction -- from separate rocprofv3 --pmc runs; counters cannot be read from inside the process).  Same
    averaging as `achieved`: over all launches of the kernel in the profiled command."""
    for name in files:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                tab = json.load(f)
        except (OSError, ValueError):
            continue
        t = tab.get("cone::" + kernel)
        if t is None:       # template arguments may have been added since: same kernel name and first argument
            stem = "cone::" + kernel.split(",")[0].rstrip(">")
            cands = [v for k, v in tab.items() if k.startswith(stem) and isinstance(v, dict)]
            if not cands:
                continue
            t = max(cands, key=lambda v: v.get("launches", 0))
        # the table is only this kernel's traffic if it was collected on this kernel's source: the collection stamps the
        # sha1 of every file under cone_amd/csrc (key "_csrc"); another ffn.hip / prefilter.hip than today's => null
        stamp, now = tab.get("_csrc") or {}, csrc_hashes()
        files = KERNEL_SOURCES.get(kernel.split("<")[0], ())
        stale = [f for f in files if stamp.get(f) != now.get(f)]
        if not stamp or stale:
            return {"traffic": None,
                    "traffic_source": f"profiles/{name} was collected on another revision of "
                                      f"{', '.join(stale) or 'cone_amd/csrc (no stamp)'}: not this kernel's traffic"}
        return {"traffic": round(t["hbm_bytes_per_launch"]), "traffic_unit": "B/launch",
                "traffic_read": round(t["read_bytes_per_launch"]), "traffic_write": round(t["write_bytes_per_launch"]),
                "traffic_source": f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; reads = 2 x "
                                  f"FETCH_SIZE, the guide's gfx950 correction); collected on csrc "
                                  + " ".join(f"{f}@{stamp[f]}" for f in files)}
    return {"traffic": None}


def roofline_from_profile(rec):
    per = {}
    for kind, a, b, c, ms in rec:
        k = int(kind)
        if k not in GEMM_KINDS:
            continue
        d = per.setdefault(k, dict(ms=0.0, flops=0.0, launches=0))
        d["ms"] += ms
        d["flops"] += rec_flops(k, a, b, c)
        d["launches"] += 1
    if not per:
        return None, {}
    dom = max(per, key=lambda k: per[k]["ms"])
    d = per[dom]        # every record of a kind is one launch of THAT kernel over its own rows (wide / 64-row / small forms
    #                     have their own kinds): no row-count filter
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    all_ms = sum(v["ms"] for v in per.values())
    all_fl = sum(v["flops"] for v in per.values())
    roof = {"bound": "mfma", "achieved": round(achieved, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), "traffic": None,
            "kernel": KERNEL_NAMES[dom], "launches": d["launches"],
            "avg_launch_ms": round(d["ms"] / d["launches"], 4),
            "flops_per_launch": round(d["flops"] / d["launches"]),
            "all_gemm_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2)}
    roof.update(pmc_traffic(KERNEL_NAMES[dom]))
    if dom in FFN_KINDS:
        # algorithmic HBM bytes of a fused layer tail over M rows: its (M, 256) fp32 input rows (with the projection: attention
        # rows + residual rows), the (M, 256) output rows; the weights (<= 1.3 MB) are served by the L2 / Infinity Cache
        rows_io = 3 if dom in FFN_PROJ_KINDS else 2
        big = [a for kind, a, b, c, ms in rec if int(kind) == dom]        # per launch of THAT kernel, like `traffic`
        alg = rows_io * 1024.0 * sum(big) / max(1, len(big))
        roof["algorithmic_bytes"] = round(alg)
        roof["algorithmic_bytes_launches"] = len(big)
        if roof.get("traffic"):
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / alg, 3)
    shapes = {}
    for kind, a, b, c, ms in rec:       # the dominant kernel by (N, K): which layers pull the average down
        if int(kind) == dom and a >= 65536:
            d2 = shapes.setdefault(f"N{int(b)}_K{int(c)}", [0.0, 0.0, 0])
            d2[0] += rec_flops(dom, a, b, c)
            d2[1] += ms
            d2[2] += 1
    roof["by_shape_tflops"] = {k: [round(v[0] / (v[1] * 1e-3) / 1e12, 1), v[2]] for k, v in sorted(shapes.items())}
    extra = {KERNEL_NAMES[k]: {"ms": round(v["ms"], 3), "launches": v["launches"],
                               "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} for k, v in per.items()}
    # the row GEMM by shape: its average is dominated by the count of small launches, its time by ONE shape (the later
    # encoder layers' q | k | v projection, N = 768, K = 256, ~2 M rows)
    rs = {}
    for kind, a, b, c, ms in rec:
        if int(kind) == 6:
            d3 = rs.setdefault(f"N{int(b)}_K{int(c)}" + ("_big" if a >= 65536 else ""), [0.0, 0.0, 0])
            d3[0] += rec_flops(6, a, b, c); d3[1] += ms; d3[2] += 1
    if rs and KERNEL_NAMES[6] in extra:
        extra[KERNEL_NAMES[6]]["by_shape"] = {k: {"ms": round(v[1], 3), "launches": v[2],
                                                  "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1),
                                                  "frac_of_fp32_mfma_peak": round(v[0] / (v[1] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 3)}
                                              for k, v in sorted(rs.items(), key=lambda kv: -kv[1][1])[:6]}
    for kind in (3, 4, 7):
        sel = rec[rec[:, 0] == kind]
        if len(sel):
            extra[KERNEL_NAMES[kind]] = {"ms": round(float(sel[:, 4].sum()), 3), "launches": int(len(sel))}
    return roof, extra


def cpu_baseline(opt, sd, n_queries, n_videos):
    """The CPU oracle (torch-CPU port of the reference path) on a bounded sample of the workload.
    torch's intra-op pool is sized by a short probe (the window model's GEMMs are small: more threads
    than ~16-32 only add synchronisation cost on a many-core host); `cores` reports what was used."""
    from oracle import cone_oracle as O
    ann, vf, qf = synth.make_dataset(opt, n_queries, n_videos, seed=0)
    ncpu = os.cpu_count() or 1
    probe_q = max(2, min(8, n_queries // 8))
    best_t, best_rate = 1, 0.0
    for t in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
        torch.set_num_threads(t)
        t0 = time.time()
        O.eval_epoch(sd, opt, ann[:probe_q], vf, qf)
        rate = probe_q / (time.time() - t0)
        if rate > best_rate:
            best_t, best_rate = t, rate
    torch.set_num_threads(best_t)
    t0 = time.time()
    (_, _, _), ranks, mr = O.eval_epoch(sd, opt, ann, vf, qf)
    dt = time.time() - t0
    nwin = len(mr)
    return {"value": round(nwin / dt, 1), "unit": "windows/s", "cores": best_t, "kind": "port",
            "queries_per_s": round(n_queries / dt, 2), "host_cpus": ncpu,
            "sample": f"{n_queries} queries / {n_videos} videos / {nwin} windows of the same synthetic "
                      f"Ego4D-NLQ config, oracle eval_epoch end to end in {dt:.1f} s with {best_t} torch threads"}


# ------------------------------------------------------------------------------------------ other BASELINE configs
def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def _timed_in_flight(start, steps, warmup):
    """The headline's stepping for an extra: `start()` enqueues a step and returns its PendingSplit; one step stays in flight
    (its host half runs after the next step is enqueued); every step is finished inside the bracket."""
    def run(n):
        prev, out = None, None
        for _ in range(n):
            h = start()
            if prev is not None:
                out = prev.result()
            prev = h
        return prev.result() if prev is not None else out
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = run(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def bench_prefilter_mad(ctx_l=6_200_000, dv=512, W=125, topk=30, steps=5):
    """BASELINE configs[2]: MAD-scale long video (ctx_l x 512 fp32 = 12.7 GB resident in HBM, ~100 k windows of 125
    clips), window scores (frame scores with the window max fused into the stream: no (nq, ctx_l) matrix is written)
    + stable top-30 for 1 query (streaming kernel) and for 64 queries at once (fp32-MFMA tiles with the queries as LDS
    operand slabs, frames read once).  Roofline: SURVEY 8d's algorithmic bytes 4*ctx_l*dv + Q*4*(dv + num_window)
    over the hipEvent time of the frame-score kernel(s) of one query batch (the dominant kernel, > 85 % of the
    path) and, as `path_frac`, over the wall time of the whole pre-filter call sequence; `traffic` = HBM bytes per
    launch from the committed rocprofv3 PMC passes over tools/prefilter_bench.py (profiles/r03_pmc_prefilter.json)."""
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(0)
    vid = torch.randn(ctx_l, dv, device=dev, generator=g)
    vid = ops.l2_normalize(vid, 0.0)
    nw = ops.num_windows(ctx_l, W)
    out = {"workload": f"BASELINE.json configs[2]: one video of ctx_l={ctx_l} clips x d={dv} fp32 "
                       f"({ctx_l * dv * 4 / 1e9:.1f} GB resident), window_len={W}, {nw} windows, stable top-{topk}"}
    for nq in (1, 64):
        txt = ops.l2_normalize(torch.randn(nq, dv, device=dev, generator=g), 0.0)

        def call():
            _, ws = ops.prefilter_scores(vid, txt, W, frame_scores=False)
            return ops.topk_windows(ws, topk)
        _timed(call, 1, 2)
        lib.cone_prof_enable(1)
        dt, _ = _timed(call, steps, 0)
        rec = collect_profile()
        lib.cone_prof_enable(0)
        k_ms = float(rec[np.isin(rec[:, 0], (0, 4))][:, 4].sum()) / steps        # frame-score launches of one call
        alg = 4.0 * ctx_l * dv + nq * 4.0 * (dv + nw)
        ach = alg / (k_ms * 1e-3) / 1e9
        kernel = f"frame_score_kernel<{dv // 256}, 1, 4, 1>" if nq < 8 else "frame_score_mq_kernel<4, false>"
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": int(alg),
                # "also report vs 6.29e12" -- the guide's measured float4-copy rate of this part
                "frac_of_measured_copy_peak": round(ach / HBM_COPY_GBS, 4)}
        roof.update(pmc_traffic(kernel, PMC_PREFILTER_FILES))
        if roof.get("traffic"):
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / alg, 3)
        out[f"q{nq}"] = {"queries": nq, "ms_per_call": round(dt * 1e3, 3), "windows_per_s": round(nw * nq / dt, 1),
                         "frame_score_kernel_ms": round(k_ms, 3), "kernel": kernel, "roofline": roof,
                         "path_frac": round(alg / dt / 1e9 / HBM_PEAK_GBS, 4)}
        if nq >= 8:
            # the same launch against the OTHER roof: 2 * Q * ctx_l * dv exact-fp32 FLOPs over 4 * ctx_l * dv bytes = Q / 2
            # FLOP per byte; the ridge of this part is 157.3 TFLOP/s / 8 TB/s = 19.7, so from 40 queries on the contraction,
            # not the stream, is what bounds the kernel (and `roofline.frac` above cannot approach 1)
            fl = 2.0 * nq * ctx_l * dv
            out[f"q{nq}"]["roofline_mfma"] = {
                "bound": "mfma", "achieved": round(fl / (k_ms * 1e-3) / 1e12, 2), "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(fl / (k_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                "flop_per_byte": round(fl / alg, 1), "ridge_flop_per_byte": round(FP32_MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 1)}
        if nq >= 8:
            # OPT-IN beside it (never the headline form): the same call on the bf16 matrix cores, each fp32 product as six
            # partial products of three-piece bf16 operands (cone_prefilter_scores_split): fp32 accuracy, HBM-bound
            def call3():
                _, ws = ops.prefilter_scores(vid, txt, W, frame_scores=False, split_bf16=True)
                return ops.topk_windows(ws, topk)
            try:
                _, ws_a = ops.prefilter_scores(vid, txt, W, frame_scores=False)
                _, ws_b = ops.prefilter_scores(vid, txt, W, frame_scores=False, split_bf16=True)
                diff = float((ws_a - ws_b).abs().max())
                same = int((ops.topk_windows(ws_a, topk)[0] == ops.topk_windows(ws_b, topk)[0]).all(dim=1).sum())
                del ws_a, ws_b
                _timed(call3, 1, 2)
                lib.cone_prof_enable(1)
                dt3, _ = _timed(call3, steps, 0)
                rec3 = collect_profile()
                lib.cone_prof_enable(0)
                k3 = float(rec3[np.isin(rec3[:, 0], (0, 4))][:, 4].sum()) / steps
                roof3 = {"bound": "hbm", "achieved": round(alg / (k3 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (k3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes": int(alg)}
                roof3.update(pmc_traffic("frame_score_mq3_kernel", PMC_PREFILTER_FILES))
                if roof3.get("traffic"):
                    roof3["traffic_over_algorithmic"] = round(roof3["traffic"] / alg, 3)
                out[f"q{nq}_split_bf16x3"] = {
                    "note": "opt-in cone_prefilter_scores_split (NOT the default form): three-piece bf16 operands, six partial "
                            "products per fp32 product, fp32 accumulation; the 192 KiB of query pieces stream through a 3-stage "
                            "LDS ring by LDS-DMA, the twelve waves of a workgroup in lock-step on the 128-channel chunk",
                    "queries": nq, "ms_per_call": round(dt3 * 1e3, 3), "frame_score_kernel_ms": round(k3, 3),
                    "kernel": "frame_score_mq3_kernel", "roofline": roof3,
                    "path_frac": round(alg / dt3 / 1e9 / HBM_PEAK_GBS, 4),
                    "max_abs_diff_of_window_scores_vs_fp32": diff, "queries_with_identical_top_k": same}
            except Exception as e:      # noqa: BLE001
                out[f"q{nq}_split_bf16x3"] = {"error": repr(e)[:300]}
        del txt
    del vid
    torch.cuda.empty_cache()
    return out


def bench_latency_config1(sd_seed=0, steps=20):
    """BASELINE configs[0]: one query over one video of ctx_l = 900 clips => 22 windows, top-20
    => B = 20 windows, stages A->C + the JSON rows, as a latency figure -- launches issued one by one (eager) and the
    same sequence replayed as one hipGraph (opt.hip_graph: the resident-video / repeated-query serving form)."""
    opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, sd_seed).items()})
    ann, vf, qf = synth.make_dataset(opt, 1, 1, seed=0, ctx_range=(900, 901), lq_range=(12, 13))
    store = inf.FeatureStore(opt, ann, vf, qf)
    dt, (lists, dp) = _timed(lambda: inf.predict_split(model, store, opt), steps, 5)
    out = {"workload": "BASELINE.json configs[0]: 1 query x 1 video (ctx_l 900, 22 windows), top-20 => 20 windows, "
                       "stages A-C + JSON rows", "ms_per_query": round(dt * 1e3, 3),
           "windows_per_s": round(dp["n_windows"] / dt, 1), "queries_per_s": round(1.0 / dt, 1)}
    try:
        opt.hip_graph = True
        gdt, (glists, _) = _timed(lambda: inf.predict_split(model, store, opt), steps, 5)
        out["hip_graph"] = {"ms_per_query": round(gdt * 1e3, 3), "queries_per_s": round(1.0 / gdt, 1),
                            "same_rows_as_eager": glists == lists,
                            "note": "the ~110-launch sequence captured once and replayed as one graph launch per query"}
    except Exception as e:      # noqa: BLE001
        out["hip_graph"] = {"error": repr(e)[:300]}
    finally:
        opt.hip_graph = False
    return out


def reference_batch_tensors(model, store, opt, n_batch_queries=None):
    """The FIRST reference batch of the split as cone/inference.py:45-50 sees it: ``eval_bsz`` consecutive queries x their
    top-k windows (cone/config.py:59, cone/ego4d_mad_dataloader.py:146), collated like StartEndDataset's collate
    (:305-344): clips zero-padded to the batch's longest window, the query's tokens REPLICATED into each of its windows and
    zero-padded to the batch's longest query, prefix masks.  Returns the ``model_inputs`` dict + the window table rows."""
    nqb = min(n_batch_queries or opt.eval_bsz, len(store.ann))
    sub = store.view(0, nqb) if nqb < len(store.ann) else store
    win_idx = inf.prefilter(model, sub, opt)
    wt = inf.window_table(sub, opt, win_idx)
    vrow0, vlen = wt["vid_row0"].long(), wt["vid_len"].long()
    trow0, tlen = wt["txt_row0"].long(), wt["txt_len"].long()
    B, Lv, Lq = int(vrow0.shape[0]), int(vlen.max()), int(tlen.max())
    dev = store.device
    ar_v, ar_q = torch.arange(Lv, device=dev)[None], torch.arange(Lq, device=dev)[None]
    vmask, tmask = (ar_v < vlen[:, None]), (ar_q < tlen[:, None])
    tok = store.tok_raw if store.tok_normalized else ops.l2_normalize(store.tok_raw, 1e-5)
    cls = store.cls_raw if store.cls_normalized else ops.l2_normalize(store.cls_raw, 1e-5)
    rows = (vrow0[:, None] + ar_v).clamp_(max=store.vid_raw.shape[0] - 1)
    src_vid = (store.vid_raw[rows] * vmask[..., None]).contiguous()
    # a store with two visual sources (dataloader :134-158): the model input is cut from the motion arena, the matching's
    # from the appearance arena; one source: the same tensor serves both, as in the reference's collate
    # (the reference's motion reader hands out L2-normalised rows, :284-292: store.motion_rows())
    src_mot = src_vid if getattr(store, "mot_raw", None) is None else (store.motion_rows()[rows] * vmask[..., None]).contiguous()
    src_txt = tok[(trow0[:, None] + ar_q).clamp_(max=tok.shape[0] - 1)] * tmask[..., None]
    return dict(src_txt=src_txt.contiguous(), src_txt_mask=tmask.float(), src_vid_motion=src_mot,
                src_vid_motion_mask=vmask.float(), src_vid_appear=src_vid,
                src_cls_txt=cls[wt["cls_row"].long()].contiguous()), wt, sub


def bench_dropin_forward(model, store, opt, arena_exec_tflops, steps=20, warmup=3):
    """INTEGRATION.md Option A, measured: the reference's own call sites (cone/inference.py:45-50) on the reference's own
    batch -- ``outputs = model(**model_inputs)`` then ``model.forward_clip_matching(...)`` on eval_bsz x top-k zero-padded
    windows -- through cone_amd.model.CONE.forward = cone_forward_windows (compaction, projection of the valid rows,
    first-layer row caches, then the arena path's launches).  The dominant kernel's roofline from the launch records of the
    timed region; executed FLOPs / time against the arena path's figure for the same quantity."""
    lib = _lib.load()
    inputs, wt, sub = reference_batch_tensors(model, store, opt)
    B, Lv = inputs["src_vid_motion"].shape[:2]
    Lq = inputs["src_txt"].shape[1]
    mi = {k: inputs[k] for k in ("src_txt", "src_txt_mask", "src_vid_motion", "src_vid_motion_mask")}

    def call():
        # fresh mask tensor OBJECTS every batch (views: no launch), as a data loader delivers them: the model keeps the lengths
        # of the masks it saw last (forward_clip_matching gets the very tensor forward just saw, cone/inference.py:45-50)
        vm, tm = inputs["src_vid_motion_mask"].view_as(inputs["src_vid_motion_mask"]), mi["src_txt_mask"].view_as(mi["src_txt_mask"])
        o = model(src_txt=mi["src_txt"], src_txt_mask=tm, src_vid_motion=mi["src_vid_motion"], src_vid_motion_mask=vm)
        m = model.forward_clip_matching(inputs["src_cls_txt"], inputs["src_vid_appear"], vm, proposal=o["pred_spans"])
        return o, m
    dt, (o, mt) = _timed(call, steps, warmup)       # the figure: no launch timer (its two event records per launch cost a
    lib.cone_prof_enable(1)                         # 2.5 ms batch ~10 %; a 53 ms step nothing)
    dt_prof, _ = _timed(call, steps, 0)             # the kernel records: a second pass with the timer on
    rec = collect_profile()
    lib.cone_prof_enable(0)
    roof, kern = roofline_from_profile(rec)
    if roof:        # the committed counter table is per launch of the 20 000-window STEP: not this batch's launches
        for k in [k for k in roof if k.startswith("traffic")]:
            roof.pop(k)
        roof["traffic"] = None
    vl, tl = wt["vid_len"].cpu().numpy(), wt["txt_len"].cpu().numpy()
    ex = executed_mfma_flops(rec, vl, tl, opt, steps)
    etf = ex["total"] / (dt * steps) / 1e12         # FLOPs of the recorded pass = FLOPs of the timed pass (same batch)
    big_gemm = [f"N{int(b)}_K{int(c)}" for k, a, b, c, ms in rec if int(k) in (5, 6, 14) and (int(b) == 1024 or int(c) == 1024)]
    # the same windows through the arena entry: the two entries are one path, bit for bit
    feats = inf.project_features(model, sub)
    ar = model.forward_packed(feats["vproj"], wt["vid_row0"], wt["vid_len"], feats["tproj"], wt["txt_row0"], wt["txt_len"],
                              opt.max_v_l, int(tl.max()), l0=feats.get("l0"), saliency=True, aux=True)
    same = bool(torch.equal(ar["pred_logits"], o["pred_logits"]) and torch.equal(ar["pred_spans"], o["pred_spans"]))
    # the same two calls on a LARGER batch (--eval_bsz is a CLI argument of the reference, cone/config.py:59): 256 queries x
    # top-k windows per call -- the entry's rate once a batch fills the persistent grids several times over
    big = None
    if len(store.ann) >= 256:
        try:
            inputs2, wt2, _ = reference_batch_tensors(model, store, opt, n_batch_queries=256)
            vm2, tm2 = inputs2["src_vid_motion_mask"], inputs2["src_txt_mask"]

            def call2():
                vmv, tmv = vm2.view_as(vm2), tm2.view_as(tm2)
                o2 = model(src_txt=inputs2["src_txt"], src_txt_mask=tmv, src_vid_motion=inputs2["src_vid_motion"],
                           src_vid_motion_mask=vmv)
                return model.forward_clip_matching(inputs2["src_cls_txt"], inputs2["src_vid_motion"], vmv, proposal=o2["pred_spans"])
            dt2, _ = _timed(call2, max(3, steps // 4), 2)
            B2 = int(inputs2["src_vid_motion"].shape[0])
            big = {"windows": B2, "ms_per_batch": round(dt2 * 1e3, 3), "windows_per_s": round(B2 / dt2, 1)}
            del inputs2
        except Exception as e:      # noqa: BLE001
            big = {"error": repr(e)[:200]}
    return {"workload": f"the reference's batch (cone/inference.py:45-50): eval_bsz {opt.eval_bsz} x top-{opt.topk_window} = {B} "
                        f"zero-padded windows (Lv_pad {Lv}, Lq_pad {Lq}, {int(vl.sum() + tl.sum())} valid of {B * (Lv + Lq)} "
                        "token rows), model(**model_inputs) + model.forward_clip_matching(...) through CONE.forward -> "
                        "cone_forward_windows",
            "ms_per_batch": round(dt * 1e3, 3), "windows_per_s": round(B / dt, 1),
            "ms_per_batch_with_launch_timer": round(dt_prof * 1e3, 3),
            "launches_per_batch": int(len(rec) / steps), "roofline": roof,
            "layer_tail_launches": {k: v["launches"] // steps for k, v in kern.items() if k.startswith("ffn_")},
            "row_gemm_launches_with_1024": len(big_gemm),       # 0: no unfused linear1 / linear2 GEMM anywhere
            "executed_tflops": round(etf, 1), "executed_frac": round(etf / FP32_MFMA_PEAK_TFLOPS, 4),
            "executed_gflop_per_batch": {k: round(v / steps / 1e9, 2) for k, v in ex.items()},
            "arena_path_executed_tflops": round(arena_exec_tflops, 1) if arena_exec_tflops else None,
            "executed_rate_vs_arena_path": round(etf / arena_exec_tflops, 3) if arena_exec_tflops else None,
            "same_bits_as_arena_entry": same, "at_eval_bsz_256": big,
            "note": "a 640-window batch is 2.1 rounds of the persistent layer-tail grid (256 CUs x 128 rows) and its small "
                    "kernels are launch-bound: the arena path runs 20 000 windows per launch sequence"}


def bench_localizer(sd_seed=0, steps=30, ctx_l=900, n_tok=12):
    """run_on_video/cone_localizator.py:121-221 -- CONELocalizator.predict_moment(video_feats, text_feats): one query over
    one resident 900-clip video (BASELINE configs[0]'s shape), host tensors in, python list out."""
    from cone_amd.localizator import CONELocalizator
    opt = make_opt("ego4d")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, sd_seed).items()}
    loc = CONELocalizator(state_dict=sd)
    g = torch.Generator().manual_seed(3)
    vid = torch.randn(ctx_l, 256, generator=g).cuda()
    tok, cls = torch.randn(n_tok, 768, generator=g).cuda(), torch.randn(256, generator=g).cuda()
    dt, res = _timed(lambda: loc.predict_moment(vid, (tok, cls)), steps, 5)
    out = {"workload": f"CONELocalizator.predict_moment: 1 query ({n_tok} tokens) x 1 video (ctx_l {ctx_l}) resident on the "
                       "device, 20 windows, fused-score NMS, python list of [st, ed, score]",
           "ms_per_query": round(dt * 1e3, 3), "queries_per_s": round(1.0 / dt, 1), "moments": len(res)}
    try:        # opt-in: CONELocalizator(hip_graph=True) -- a shape seen before is one graph launch (inputs copied in)
        gloc = CONELocalizator(state_dict=sd, hip_graph=True)
        gdt, gres = _timed(lambda: gloc.predict_moment(vid, (tok, cls)), steps, 5)
        out["hip_graph"] = {"ms_per_query": round(gdt * 1e3, 3), "same_moments_as_eager": gres == res}
    except Exception as e:              # noqa: BLE001
        out["hip_graph"] = {"error": repr(e)[:200]}
    return out



def bench_config5(ctx_l=33_000, queries=64, steps=10):
    """BASELINE configs[4] on one GPU: 64 concurrent queries over one MAD-length video (ctx_l ~ 33 k clips = 110 min
    at 5 fps, d 512, window_len 125, top-30 => 1 920 windows), stages A->C + JSON rows."""
    opt = make_opt("mad", nms_thd=0.5, eval_split_name="test", topk_window=30)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 1).items()})
    ann, vf, qf = synth.make_dataset(opt, queries, 1, seed=0, ctx_range=(ctx_l, ctx_l + 1))
    store = inf.FeatureStore(opt, ann, vf, qf)
    dt, (_, dp) = _timed(lambda: inf.predict_split(model, store, opt), steps, 3)
    wt = dp["windows"]
    fl = reference_window_flops(wt["vid_len"].cpu().numpy(), wt["txt_len"].cpu().numpy(), 512, 512)
    try:
        model.set_option("split_bf16", 1)       # the opt-in path on the same workload (not the config's figure)
        dts, _ = _timed(lambda: inf.predict_split(model, store, opt), steps, 3)
    except Exception:
        dts = None
    return {"workload": f"BASELINE.json configs[4] on 1 GPU: {queries} queries x one MAD-length video (ctx_l {ctx_l}, "
                        f"d 512, window_len 125), top-30 => {dp['n_windows']} windows, stages A-C + JSON rows",
            "ms_per_step": round(dt * 1e3, 3), "windows_per_s": round(dp["n_windows"] / dt, 1),
            "queries_per_s": round(queries / dt, 1),
            "reference_algorithmic_tflops": round(float(fl.sum()) / dt / 1e12, 1),
            "ms_per_step_split_bf16": None if dts is None else round(dts * 1e3, 3)}


def bench_config2_ragged(model, opt, queries, videos, us_per_window_dense, steps=5, warmup=2):
    """BASELINE configs[1] on a RAGGED split: the same 1 000 queries x 50 videos with ctx_l ~ U[200, 1500) -- videos of fewer
    than top-20 windows (ctx_l <= 810) next to long ones, as real Ego4D-NLQ clips are.  The shape of the window list is host
    metadata (cone_amd.inference.Selection), so this split runs the very path of the dense one: one window-table launch, the
    per-window rows as the candidate lists, two pipeline chunks, no host sync (`sync_free`: the whole step also replays as
    one hipGraph, which cannot contain one, with the same rows)."""
    ann, vf, qf = synth.make_dataset(opt, queries, videos, seed=11, ctx_range=(200, 1500))
    store = inf.FeatureStore(opt, ann, vf, qf)
    sel = inf.selection(store, opt)
    dt, (lists, dp) = _timed_in_flight(lambda: inf.predict_split_async(model, store, opt), steps, warmup)     # the headline's stepping
    nw = dp["n_windows"]
    out = {"workload": f"BASELINE.json configs[1], ragged: {queries} queries x {videos} videos, ctx_l ~ U[200, 1500), "
                       f"window_len=90, d=256, topk_window=20, NMS 0.5: {nw} windows "
                       f"({int((sel.n_q < sel.K).sum())} queries own fewer than {sel.K})",
           "ms_per_step": round(dt * 1e3, 3), "n_windows": nw, "windows_per_s": round(nw / dt, 1),
           "queries_per_s": round(queries / dt, 1), "us_per_window": round(dt * 1e6 / nw, 4),
           "us_per_window_dense_split": round(us_per_window_dense, 4),     # the headline (both with one step in flight)
           "per_window_cost_vs_dense": round(dt * 1e6 / nw / us_per_window_dense, 4),
           "query_chunks": [list(c) for c in dp.get("chunks", [(0, queries)])]}
    saved = (opt.hip_graph if hasattr(opt, "hip_graph") else False, opt.pipeline_tail)
    try:
        opt.hip_graph, opt.pipeline_tail = True, 0.0
        gdt, (glists, _) = _timed_in_flight(lambda: inf.predict_split_async(model, store, opt), steps, 2)
        out["sync_free"] = {"hip_graph_ms_per_step": round(gdt * 1e3, 3), "same_rows_as_eager": glists == lists}
    except Exception as e:      # noqa: BLE001
        out["sync_free"] = {"error": repr(e)[:300]}
    finally:
        opt.hip_graph, opt.pipeline_tail = saved
    return out


def bench_shard_proxy(model, store, opt, ms_1gpu, steps=5, warmup=2, world=8):
    """ONE GPU's share of BASELINE configs[3] at `world` ranks, measured on this GPU without a process group: what rank 0
    of predict_split_distributed_async(mode="window") executes -- stage A of its batch-aligned query hull, project + window
    model + matching on its 1 / world slice of the window list, fusion + NMS over ALL queries (on a gather buffer filled with
    copies of the local rows), the JSON rows of its own query shard -- stepped like the N > 1 headline (one step in flight).
    projected_efficiency = ms_per_step(1 GPU) / (world x proxy ms): the strong-scaling efficiency a `world`-GPU run would show
    if the one 1.6 MB all_gather were free (it is latency-bound: one hop to each peer over xGMI)."""
    from cone_amd import parallel as par
    fn = lambda: par.predict_split_distributed_async(model, store, opt, mode="window", format_shard=True, virtual=(0, world))
    dt, (_, info) = _timed_in_flight(fn, steps, warmup)
    lo, hi = par.shard_range(info["n_windows"], 0, world)
    return {"what": f"rank 0 of {world} of the window-sharded split (BASELINE configs[3]) replayed on one GPU, no collective, "
                    "one step in flight (the N > 1 headline's stepping)",
            "world": world, "windows_of_rank": hi - lo, "queries_formatted": info["shard"][1] - info["shard"][0],
            "proxy_ms": round(dt * 1e3, 3), "ms_per_step_1gpu": round(ms_1gpu, 3),
            "projected_speedup": round(ms_1gpu / (dt * 1e3), 2),
            "projected_efficiency": round(ms_1gpu / (world * dt * 1e3), 4)}


def _mad_rows(lo, hi, dv, dev, block=62_000):
    """Rows [lo, hi) of THE synthetic MAD-scale video (row block b is drawn from generator seed 1000 + b, so every rank
    sees the same video whatever range it holds -- the halo rows of neighbouring ranks agree), L2-normalised."""
    parts = []
    for b in range(lo // block, (hi - 1) // block + 1):
        g = torch.Generator(device=dev).manual_seed(1000 + b)
        x = torch.randn(block, dv, device=dev, generator=g)
        parts.append(x[max(lo - b * block, 0):min(hi - b * block, block)])
    return ops.l2_normalize(torch.cat(parts, 0), 0.0)


def bench_prefilter_mad_ctx_sharded(dist, rank, world, timed_region, ctx_l=6_200_000, dv=512, W=125, topk=30):
    """BASELINE configs[2] over the N ranks: the 12.7 GB video sharded along ctx_l (each rank holds and streams 1 / N of the
    clips + a W - S halo), local fused window scores + stable top-30, ONE all_gather of 30 (score, window) pairs per
    query, exact merge (cone_amd.parallel.prefilter_ctx_sharded).  Roofline: the whole video's algorithmic bytes over
    the max-over-ranks time against N x 8 TB/s."""
    from cone_amd import parallel as par
    dev = torch.device("cuda", torch.cuda.current_device())
    w_lo, w_hi, f_lo, f_hi = par.ctx_shard(ctx_l, W, rank, world)
    local = _mad_rows(f_lo, f_hi, dv, dev)
    nw = ops.num_windows(ctx_l, W)
    out = {"workload": f"BASELINE.json configs[2] sharded along ctx_l over {world} ranks: ctx_l={ctx_l} x d={dv} fp32 "
                       f"({ctx_l * dv * 4 / 1e9:.1f} GB in all, {local.shape[0] * dv * 4 / 1e9:.2f} GB on rank 0), "
                       f"window_len={W}, {nw} windows, stable top-{topk}; one all_gather of {topk} (score, window) pairs per query",
           "ranks_seen": dist.get_world_size(), "collectives_per_call": 1}
    for nq in (1, 64):
        g = torch.Generator(device=dev).manual_seed(7)
        txt = ops.l2_normalize(torch.randn(nq, dv, device=dev, generator=g), 0.0)
        dt, _, _ = timed_region(lambda: par.prefilter_ctx_sharded(local, ctx_l, txt, W, topk))
        steps = timed_region.steps
        alg = 4.0 * ctx_l * dv + nq * 4.0 * (dv + nw)
        out[f"q{nq}"] = {"queries": nq, "ms_per_call": round(dt / steps * 1e3, 3),
                         "windows_per_s": round(nw * nq * steps / dt, 1),
                         "roofline": {"bound": "hbm", "achieved": round(alg * steps / dt / 1e9, 1),
                                      "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                      "frac": round(alg * steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4),
                                      "algorithmic_bytes": int(alg), "traffic": None,
                                      "note": "whole-path time (kernel + gather + merge), all ranks; per-kernel counter "
                                              "traffic: prefilter_mad (N = 1)"}}
    del local
    torch.cuda.empty_cache()
    return out


def bench_config5_sharded(dist, world, timed_region, ctx_l=33_000, queries=64):
    """BASELINE configs[4] as stated: 64 concurrent queries over ONE MAD-length video on the N ranks -- pre-filter sharded
    along ctx_l (one all_gather of top-30 pairs), window model sharded by window (one all_gather of proposal rows),
    fusion + NMS on every rank, JSON rows of the rank's query shard (predict_split_distributed(prefilter="ctx"))."""
    from cone_amd import parallel as par
    opt = make_opt("mad", nms_thd=0.5, eval_split_name="test", topk_window=30)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 1).items()})
    ann, vf, qf = synth.make_dataset(opt, queries, 1, seed=0, ctx_range=(ctx_l, ctx_l + 1))
    store = inf.FeatureStore(opt, ann, vf, qf)         # the same video and queries on every rank (features replicated)
    fn = lambda: par.predict_split_distributed(model, store, opt, mode="window", prefilter="ctx", format_shard=True)
    dt, _, (_, info) = timed_region(fn)
    steps = timed_region.steps
    return {"workload": f"BASELINE.json configs[4]: {queries} queries x one MAD-length video (ctx_l {ctx_l}, d 512, "
                        f"window_len 125), top-30 => {info['n_windows']} windows, over {world} ranks: ctx-sharded pre-filter "
                        "-> window-sharded model -> one gather of proposal rows -> fusion + NMS + JSON rows",
            "scaling": "strong", "ms_per_step": round(dt / steps * 1e3, 3),
            "windows_per_s": round(info["n_windows"] * steps / dt, 1), "queries_per_s": round(queries * steps / dt, 1),
            "ranks_seen": dist.get_world_size(), "collectives_per_step": 2}


def self_launch(n_gpus):
    """`python3 bench.py --gpus N` with N > 1 and no RANK in the environment: this process touches no GPU -- it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD process (never an exec),
    relays the child's stdout (rank 0's one JSON line) and stderr, and exits with its return code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher's own c10d rendezvous picks a free port itself (no bind-then-close race with another process)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n_gpus), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def rccl_preflight(dist, world, rank, nq, max_after, backend, device="cuda"):
    """One all_gather_into_tensor of the message the step's exchange sends -- the fp64 kept rows (3, nq, max_after, 5) + the
    int32 counts (3, nq) -- at this world size, before anything is timed.  A failure (xGMI / IPC / RCCL set-up) becomes a
    one-line reason on stderr and a non-zero exit instead of a hang or a stack of C++ frames in the middle of the bench."""
    try:
        t0 = time.perf_counter()
        rows = torch.full((3, nq, max_after, 5), float(rank), dtype=torch.float64, device=device)
        n = torch.full((3, nq), rank, dtype=torch.int32, device=device)
        rows_all = torch.empty((world * 3, nq, max_after, 5), dtype=torch.float64, device=device)
        n_all = torch.empty((world * 3, nq), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(rows_all, rows)
        dist.all_gather_into_tensor(n_all, n)
        if device == "cuda":
            torch.cuda.synchronize()
        want = torch.arange(world, dtype=torch.float64, device=device).repeat_interleave(3)
        ok = bool((rows_all[:, 0, 0, 0] == want).all()) and bool((n_all[:, 0].to(torch.float64) == want).all())
        if not ok:
            raise RuntimeError("gathered shards are not in rank order / not the values the ranks sent")
        return {"ok": True, "backend": backend, "world": world, "bytes_per_rank": int(rows.numel() * 8 + n.numel() * 4),
                "first_collective_ms": round((time.perf_counter() - t0) * 1e3, 2)}
    except Exception as e:          # noqa: BLE001
        sys.stderr.write(f"bench.py: collective preflight failed on rank {rank} of {world} ({backend}): "
                         f"{type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}\n")
        sys.stderr.flush()
        os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--queries", type=int, default=1000)
    ap.add_argument("--videos", type=int, default=50)
    ap.add_argument("--pipeline_chunks", type=int, default=None,
                    help="query chunks of the host/GPU software pipeline (default: automatic, see --pipeline_tail)")
    ap.add_argument("--set_option", action="append", default=[], metavar="NAME=VALUE",
                    help="diagnostics: flip an A/B switch of the model handle (cone_model_set_option), e.g. pos_tables=0")
    ap.add_argument("--window_batch", type=int, default=32768)
    ap.add_argument("--pipeline_tail", type=float, default=None,
                    help="fraction of the queries in the tail chunk of the host/GPU pipeline (default: automatic = 1/16 from 32 "
                         "reference batches on, i.e. at this size; 0 = one chunk)")
    ap.add_argument("--steps_in_flight", type=int, default=2,
                    help="2 (default): the GPU half of step i + 1 is enqueued before the host half of step i (waiting for its kept "
                         "rows, building its submission lists) runs -- cone_amd.inference.predict_split_async; every step is "
                         "complete inside the timed bracket.  1: one step at a time (predict_split)")
    ap.add_argument("--extras_timeout", type=float, default=420.0,
                    help="N > 1: seconds the extras behind the headline may take before rank 0 prints the line as it stands")
    ap.add_argument("--elide_dead_work", action="store_true",
                    help="A/B: the headline step WITHOUT the saliency head and the intermediate decoder layer's heads "
                         "(CONE.forward computes them, cone/inference.py never reads them); the default headline computes "
                         "both and reports the elided step beside it as ms_per_step_dead_work_elided")
    ap.add_argument("--need_saliency", action="store_true", help="accepted for older command lines: now the default")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_extras", action="store_true",
                    help="only the headline timed region (profiling runs): skip the full-forward region, the split_bf16 "
                         "region and the configs[0]/[2]/[4] figures")
    ap.add_argument("--cpu_queries", type=int, default=400)
    ap.add_argument("--mad_ctx_l", type=int, default=6_200_000,
                    help="clips of the MAD-scale stress video of prefilter_mad_ctx_sharded (N > 1)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # plain `python3 bench.py --gpus N`: become the launcher (no GPU call has happened in this process)
        sys.exit(self_launch(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks\n")
        sys.exit(2)
    if os.environ.get("CONE_BENCH_LAUNCH_CHECK") == "1":
        # launcher + rendezvous + preflight only, on CPU tensors over gloo (tests/test_host_cpu.py: no GPU in the build
        # container): the ranks the plain entry started find each other and exchange the step's message shape
        import torch.distributed as dist
        dist.init_process_group("gloo")
        pf = rccl_preflight(dist, world, rank, min(args.queries, 1000), 5, "gloo", device="cpu")
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "ranks_seen": dist.get_world_size(),
                              "collective_preflight": pf}))
        dist.destroy_process_group()
        return
    # test-only overrides so that the N > 1 code path can be exercised on a one-GPU box (tests/test_gpu_parity.py):
    # CONE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0, CONE_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks
    # on one device).  Neither is set by the driver's launch.
    if os.environ.get("CONE_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("CONE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    # Host side of the step is launch glue + list building: a large intra-op pool only adds wake-up latency
    # (torch.distributed.run sets OMP_NUM_THREADS=1 for the same reason); cpu_baseline() sizes its own pool.
    torch.set_num_threads(1)
    dist = None
    use_dist = "RANK" in os.environ          # launched by torch.distributed.run (also with a single rank)
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        preflight = rccl_preflight(dist, world, rank, min(args.queries, 1000), 5, backend)

    # the headline step computes everything CONE.forward returns (cone/model.py:112-127): saliency_scores and aux_outputs too
    full = not args.elide_dead_work
    opt = make_opt("ego4d", nms_thd=0.5, eval_split_name="test", topk_window=20, eval_bsz=32,
                   window_batch=args.window_batch, pipeline_chunks=args.pipeline_chunks, pipeline_tail=args.pipeline_tail,
                   need_saliency=full, need_aux=full)
    if full and args.pipeline_tail is None and args.pipeline_chunks is None \
            and -(-args.queries // opt.eval_bsz) >= 32:
        # the product's automatic chunking keeps a split whole when the caller asked for the per-window outputs of the whole
        # split; the bench reads none of them (it measures that they are COMPUTED), so it keeps the pipeline's two chunks
        opt.pipeline_tail = 1.0 / 16.0
    sd = synth.make_state_dict(opt, 0)
    model, _ = build_model(opt)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    multi = use_dist and world > 1
    # N > 1: the headline is BASELINE configs[3] -- ONE such split (the same on every rank: features replicated), sharded by
    # window over the ranks.  Total work is fixed as N grows: strong scaling.  (N = 1: the same split on one GPU.)
    ann, vf, qf = synth.make_dataset(opt, args.queries, args.videos, seed=0)
    store = inf.FeatureStore(opt, ann, vf, qf)          # features resident in HBM from here on
    if multi:
        from cone_amd import parallel as par
    lib = _lib.load()
    for kv in args.set_option:
        name, _, val = kv.partition("=")
        model.set_option(name, int(val or 1))

    inflight, last = [], [None]

    def finish_oldest():
        h = inflight.pop(0)
        out, dp = h.result()                # waits for the kept rows in pinned memory, builds the submission lists
        last[0] = ([out], dp)

    def start_headline():
        # the product's own drivers: stages A->C + the submission rows, device half enqueued, host half handed back.
        # N = 1: cone_amd.inference.predict_split_async; N > 1: cone_amd.parallel.predict_split_distributed_async -- the split
        # sharded by window, ONE RCCL all_gather of the per-window proposal rows (enqueued on the stream like a kernel), fusion
        # + NMS of all queries on every rank, the JSON rows of its own query shard on every rank
        if multi:
            return par.predict_split_distributed_async(model, store, opt, mode="window", format_shard=True)
        return inf.predict_split_async(model, store, opt)

    start = [start_headline]

    def step():
        # --steps_in_flight 2 (default): the GPU half of step i + 1 is enqueued before the host half of step i runs (the
        # product's eval loop does the same across splits); 1: one step at a time
        inflight.append(start[0]())
        while len(inflight) >= max(1, args.steps_in_flight):
            finish_oldest()
        return last[0]

    def fence():
        while inflight:                     # every enqueued step is finished (lists built) inside the timed bracket
            finish_oldest()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(fn):
        """W untimed + exactly K timed steps between barrier + synchronize fences; MAX over ranks."""
        for _ in range(args.warmup):
            fn()
        fence()
        lib.cone_prof_enable(1)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = fn()
        fence()
        dt = time.perf_counter() - t0
        if fn is step:
            res = last[0]                   # (a step's own result is complete once it is no longer in flight: here, all of them)
        rec = collect_profile()
        lib.cone_prof_enable(0)
        timed_region.per_rank = [dt]
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            allt = torch.empty(world, dtype=torch.float64, device="cuda")
            dist.all_gather_into_tensor(allt, t)
            timed_region.per_rank = [float(x) for x in allt.tolist()]
            dt = max(timed_region.per_rank)            # MAX over ranks
        return dt, rec, res

    timed_region.steps = args.steps
    dt, rec, (out, dp) = timed_region(step)
    n_windows = dp["n_windows"]

    # ---- the headline object first: nothing below can take it down (every extra is guarded on its own)
    res = None
    if rank == 0:
        roof, kern = roofline_from_profile(rec)
        res = {
            "metric": "windows/sec + queries/sec, Ego4D-NLQ win_len=90 d=256 top-k=20",
            "value": round(n_windows * args.steps / dt, 1),          # ONE split per step, whatever N: whole-job windows / s
            "unit": "windows/s",
            "queries_per_s": round(args.queries * args.steps / dt, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "ranks_seen": dist.get_world_size() if use_dist else 1,
            "ms_per_step_rank_max": round(max(timed_region.per_rank) / args.steps * 1e3, 3),
            "ms_per_step_rank_min": round(min(timed_region.per_rank) / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[3]: ONE Ego4D-NLQ val-scale synthetic split sharded by window over "
                                    f"{world} GPUs (one RCCL all_gather of the per-window proposal rows before the global NMS), "
                                    if multi else "BASELINE.json configs[1]: Ego4D-NLQ val-scale synthetic, ")
                                   + f"{args.queries} queries x {args.videos} videos, window_len=90, d=256, "
                                   f"topk_window=20, NMS 0.5, {n_windows} windows per step",
                       "collectives_per_step": 1 if multi else 0,
                       "window_batch": args.window_batch, "weights": "random-init (seed 0), reference architecture",
                       "steps_in_flight": max(1, args.steps_in_flight),
                       "query_chunks": [list(c) for c in dp.get("chunks", [(0, args.queries)])],
                       "outputs": "per window pred_logits, pred_spans, matching scores -> rows [st, ed, proposal, "
                                  "matching]; per query fused / proposal / matching top-5 after NMS as JSON rows"
                                  + ("; saliency_scores and the intermediate decoder layer's heads (aux_outputs) computed too, "
                                     "as CONE.forward does (cone/model.py:112-127) -- cone/inference.py never reads them (:54-59); "
                                     "ms_per_step_dead_work_elided is the same step without them"
                                     if full else
                                     "; --elide_dead_work: saliency and aux (intermediate-layer) heads NOT computed"),
                       "ranks_seen": dist.get_world_size() if use_dist else 1},
            "roofline": roof, "kernels": kern,
        }
        if use_dist:
            res["collective_preflight"] = preflight
        wt = dp.get("windows")
        if wt is None and multi:    # the sharded driver tabulates a rank's own windows only: the split's lengths, untimed
            wt = inf.window_table(store, opt, inf.prefilter(model, store, opt))
        if wt is not None:      # SURVEY 8d's pipeline-level figure: the reference's algorithmic FLOPs / wall time
            fl = reference_window_flops(wt["vid_len"].cpu().numpy(), wt["txt_len"].cpu().numpy(),
                                        opt.v_appear_feat_dim, opt.t_feat_dim)
            tf = float(fl.sum()) * args.steps / dt / 1e12           # ONE split per step
            ex = executed_mfma_flops(rec, wt["vid_len"].cpu().numpy(), wt["txt_len"].cpu().numpy(), opt, args.steps)
            if multi:           # the launch records are rank 0's (1 / world of the split's rows); the attention terms are
                ex["dense_layers"] *= world                          # computed from the whole split's window lengths
                ex["total"] = ex["dense_layers"] + ex["encoder_attention"] + ex["decoder_cross_attention"]
            etf = ex["total"] / dt / 1e12
            res["window_model"] = {"reference_mflop_per_window": round(float(fl.mean()) / 1e6, 1),
                                   "reference_algorithmic_tflops": round(tf, 1),
                                   "reference_equivalent_frac": round(tf / (world * FP32_MFMA_PEAK_TFLOPS), 4),
                                   "executed_tflops": round(etf, 1),
                                   "executed_frac": round(etf / (world * FP32_MFMA_PEAK_TFLOPS), 4),
                                   "executed_gflop_per_step": {k: round(v / args.steps / 1e9, 1) for k, v in ex.items()},
                                   "note": "reference_*: the REFERENCE's algorithmic FLOPs (padding excluded) over the whole step "
                                           "time -- what the reference would have to sustain for this throughput, not what this "
                                           "build executes; executed_*: the FLOPs the MFMA kernels of the step actually ran "
                                           "(dense layers from the launch records, attention cores from the window lengths: "
                                           "de-duplicated projections, folded decoder K/V) over the same time and the fp32-MFMA peak"}

    def note(name, value):
        if res is not None:
            res[name] = value

    # N > 1: the extras below contain collectives.  Should one of them ever stall (a rank that failed inside an extra while
    # the others wait for it), the headline -- measured and complete above -- must still come out: a watchdog on every rank
    # prints rank 0's line as it stands after --extras_timeout seconds and ends the process.
    import threading
    extras_done, emit_lock, emitted = threading.Event(), threading.Lock(), [False]

    def emit(extra=None):
        with emit_lock:
            if emitted[0]:
                return
            emitted[0] = True
            if rank == 0:
                line = dict(res)
                if extra:
                    line.update(extra)
                print(json.dumps(line), flush=True)

    def watchdog():
        if not extras_done.wait(args.extras_timeout):
            emit({"extras_timed_out_after_s": args.extras_timeout})
            os._exit(4)         # the headline is out, but a stalled extra / dead rank is a FAILED run for the launcher
    if use_dist and world > 1:
        threading.Thread(target=watchdog, daemon=True).start()

    def guarded(name, fn):
        """Run one extra; a failure becomes {"error": ...} under its name instead of taking the line down."""
        try:
            v = fn()
        except Exception as e:              # noqa: BLE001 -- torch OOM / RuntimeError / KeyError alike
            v = {"error": repr(e)[:500]}
            try:
                torch.cuda.synchronize()
            except Exception:               # noqa: BLE001
                pass
        if v is not None:
            note(name, v)

    serial_ms = [dt / args.steps * 1e3]      # the dense step one at a time: what the one-at-a-time extras below compare with
    # ---- the same K steps one at a time (predict_split: the host half of a step runs before the next step is enqueued)
    if not args.no_extras and max(1, args.steps_in_flight) > 1:
        def serial():
            keep, args.steps_in_flight = args.steps_in_flight, 1
            try:
                sdt1, _, (_, sdp1) = timed_region(step)
            finally:
                args.steps_in_flight = keep
            serial_ms[0] = sdt1 / args.steps * 1e3
            note("ms_per_step_one_at_a_time", round(sdt1 / args.steps * 1e3, 2))
            note("value_one_at_a_time", round(sdp1["n_windows"] * args.steps / sdt1, 1))
        guarded("one_at_a_time_error", serial)

    # ---- the eval pipeline's own default (cone_amd.inference): the outputs cone/inference.py never reads are not computed
    if full and not args.no_extras:
        def elided():
            opt.need_saliency = opt.need_aux = False
            try:
                fdt, _, (_, fdp) = timed_region(step)
            finally:
                opt.need_saliency = opt.need_aux = True
            note("ms_per_step_dead_work_elided", round(fdt / args.steps * 1e3, 2))
            note("value_dead_work_elided", round(fdp["n_windows"] * args.steps / fdt, 1))
        guarded("dead_work_elided_error", elided)

    if world == 1 and not args.no_extras and not any(kv.startswith("split_bf16") for kv in args.set_option):
        # OPT-IN path, reported beside the headline (which stays exact fp32): every layer tail on the bf16 matrix cores,
        # each fp32 product as six partial products of three-piece bf16 operands with fp32 accumulation (ffn_split.hip)
        def split_path():
            try:
                def one_chunk_outputs():            # per-window outputs of the whole split: one untimed unchunked step
                    tail0, opt.pipeline_tail = opt.pipeline_tail, 0.0
                    try:
                        step()
                        while inflight:
                            finish_oldest()
                        return last[0][1].get("outputs") or {}
                    finally:
                        opt.pipeline_tail = tail0
                ref_out = {k: v.clone() for k, v in one_chunk_outputs().items() if k in ("pred_logits", "pred_spans")}
                model.set_option("split_bf16", 1)
                sdt, srec, (_, sdp) = timed_region(step)
                sroof, _ = roofline_from_profile(srec)
                souts = one_chunk_outputs()
                diffs = {k: float((souts[k] - v).abs().max()) for k, v in ref_out.items()} if souts else None
                return {"note": "opt-in model option split_bf16=1 (NOT the headline): fp32 products of the fused layer tails as "
                                "six bf16 MFMA partial products of three-piece operands (x = xh + xm + xl exactly), fp32 "
                                "accumulation; error against float64 equal to the fp32 MFMA chain's (tools/probe/"
                                "split_bf16_probe.hip, tests); same reference fixtures, same tolerances",
                        "ms_per_step": round(sdt / args.steps * 1e3, 2),
                        "value": round(sdp["n_windows"] * args.steps / sdt, 1), "unit": "windows/s",
                        "queries_per_s": round(args.queries * args.steps / sdt, 1),
                        "layer_tail_tflops_fp32_equivalent": sroof["achieved"] if sroof else None,
                        "max_abs_diff_vs_default": diffs}
            except _lib.ConeHipError as e:      # a model shape the split kernels do not cover
                return {"note": f"split_bf16 not available for this model: {e}"}
            finally:
                model.set_option("split_bf16", 0)
        guarded("split_bf16x3", split_path)

    if multi:
        # the EASY curve beside the headline: every rank its own split of that size (seed = rank), the kept rows of all shards
        # all-gathered over RCCL -- per-GPU work fixed as N grows (weak scaling)
        def weak():
            annr, vfr, qfr = synth.make_dataset(opt, args.queries, args.videos, seed=rank)
            store_r = inf.FeatureStore(opt, annr, vfr, qfr)

            def start_weak():
                h = inf.predict_split_async(model, store_r, opt)
                dpw = h.info
                rows, n = dpw["rows"], dpw["n"]
                rows_all = torch.empty((world * rows.shape[0],) + tuple(rows.shape[1:]), dtype=rows.dtype, device=rows.device)
                n_all = torch.empty((world * n.shape[0],) + tuple(n.shape[1:]), dtype=n.dtype, device=n.device)
                dist.all_gather_into_tensor(rows_all, rows.contiguous())
                dist.all_gather_into_tensor(n_all, n.contiguous())
                dpw["rows_all"], dpw["n_all"] = rows_all, n_all
                return h
            start[0] = start_weak
            try:
                wdt, _, (_, wdp) = timed_region(step)
            finally:
                start[0] = start_headline
            return {"config": "BASELINE.json configs[1] on every rank (its own split, seed = rank), kept rows of all shards "
                              "all-gathered: per-GPU work fixed as N grows",
                    "scaling": "weak", "value": round(world * wdp["n_windows"] * args.steps / wdt, 1), "unit": "windows/s",
                    "queries_per_s": round(world * args.queries * args.steps / wdt, 1),
                    "ms_per_step": round(wdt / args.steps * 1e3, 2), "ranks_seen": dist.get_world_size(),
                    "collectives_per_step": 2}
        guarded("weak_scaling", weak)
        if not args.no_extras:
            # the other multi-GPU configs of BASELINE.json, as stated (collectives inside: every rank runs them)
            guarded("config5_sharded", lambda: bench_config5_sharded(dist, world, timed_region))
            guarded("prefilter_mad_ctx_sharded",
                    lambda: bench_prefilter_mad_ctx_sharded(dist, rank, world, timed_region, ctx_l=args.mad_ctx_l))

    if world == 1 and not args.no_extras:
        def dense_graph():
            # the headline workload with the step's launch sequence captured once and replayed as ONE hipGraph launch per step
            # (opt.hip_graph: the serving form for a split that is evaluated again at the same shapes); same stepping, same rows
            saved = (getattr(opt, "hip_graph", False), opt.pipeline_tail)
            try:
                opt.hip_graph, opt.pipeline_tail = True, 0.0
                gdt, (glists, _) = _timed_in_flight(lambda: inf.predict_split_async(model, store, opt), args.steps, 2)
            finally:
                opt.hip_graph, opt.pipeline_tail = saved
            return {"ms_per_step": round(gdt * 1e3, 2), "value": round(n_windows / gdt, 1), "unit": "windows/s",
                    "same_rows_as_eager": glists == out[0],
                    "note": "opt-in opt.hip_graph (NOT the headline): one graph launch per step instead of ~110 kernel launches"}
        guarded("hip_graph", dense_graph)
        guarded("dropin_forward", lambda: bench_dropin_forward(
            model, store, opt, (res or {}).get("window_model", {}).get("executed_tflops")))
        for w_ in (2, 4, 8):
            guarded(f"shard_proxy_{w_}", lambda w_=w_: bench_shard_proxy(model, store, opt, dt / args.steps * 1e3, world=w_))
        guarded("config2_ragged", lambda: bench_config2_ragged(model, opt, args.queries, args.videos,
                                                               dt / args.steps * 1e6 / n_windows, steps=args.steps))
        del store, dp, out
        model._ws.buf = None
        torch.cuda.empty_cache()
        guarded("latency_config1", bench_latency_config1)

        def localizer():
            v = bench_localizer()
            base = (res or {}).get("latency_config1", {}).get("ms_per_query")
            if base:
                v["latency_config1_ms_per_query"] = base
                v["vs_latency_config1"] = round(v["ms_per_query"] / base, 3)
            return v
        guarded("localizer", localizer)
        guarded("config5", bench_config5)
        guarded("prefilter_mad", bench_prefilter_mad)
    if world == 1 and not args.no_cpu_baseline and args.cpu_queries > 0:
        guarded("cpu_baseline", lambda: cpu_baseline(opt, sd, args.cpu_queries,
                                                     max(1, args.cpu_queries * args.videos // args.queries)))
    if res is not None and isinstance(res.get("roofline"), dict):
        # SURVEY 8(d)'s second roofline rides inside the headline's `roofline` object (the driver keeps that object): the
        # pre-filter's HBM roofline on BASELINE configs[2] at 1 and 64 queries, and the whole step's executed fraction
        pm = res.get("prefilter_mad") or {}
        for key, name in (("q1", "prefilter"), ("q64", "prefilter_q64")):
            r = (pm.get(key) or {}).get("roofline")
            if r:
                res["roofline"][name] = dict(r, kernel=pm[key].get("kernel"), workload="BASELINE.json configs[2], "
                                             f"{pm[key].get('queries')} query(ies): 4 ctx_l dv + Q 4 (dv + num_window) bytes "
                                             "over the hipEvent time of the frame-score launch")
        wm = res.get("window_model") or {}
        if "executed_frac" in wm:
            res["roofline"]["step_executed_frac"] = wm["executed_frac"]      # all MFMA FLOPs of the step / step time / 157.3
    extras_done.set()
    emit()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
