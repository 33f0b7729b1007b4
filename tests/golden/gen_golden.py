#!/usr/bin/env python3
"""Generate the golden fixtures of tests/golden/ by running the REFERENCE itself.

Runs ONLY in the build container (needs /root/reference; the GPU box never sees it):

    cd /root/repo && PYTHONPATH=/root/reference:/root/repo PYTHONDONTWRITEBYTECODE=1 \
        python tests/golden/gen_golden.py

What it does: imports the unmodified reference modules (``cone.model``,
``cone.inference``, ``utils.temporal_nms`` ...), with the two absent third-party
display / IO packages stubbed (``lmdb`` -- only ``lmdb.open`` in dataset
constructors, which the in-memory subclasses below bypass; ``terminaltables`` --
metric tables, val split only), loads weights made by ``cone_amd.synth`` (seeded,
checksummed) into the reference ``CONE`` module, runs it on small seeded inputs
and stores inputs' seeds + the reference's OUTPUTS as ``.npz`` / ``.json``.

Pinned deviation (SURVEY.md hazard H6): ``torch.sort`` is forced to
``stable=True`` while the reference's pre-filter runs, so tie order in the window
rank list is the canonical stable one.  Everything else is the reference verbatim.
"""
import io
import json
import math
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)

# ---- stubs for absent third-party packages -------------------------------------------------
sys.modules.setdefault("lmdb", types.ModuleType("lmdb"))
_tt = types.ModuleType("terminaltables")


class _AsciiTable:
    def __init__(self, data, title=None):
        self.data, self.title, self.justify_columns = data, title, {}

    @property
    def table(self):
        return "\n".join(" | ".join(map(str, r)) for r in self.data) + "\n"


_tt.AsciiTable = _AsciiTable
sys.modules.setdefault("terminaltables", _tt)

import cone.inference as ref_inf  # noqa: E402
from cone.ego4d_mad_dataloader import PreFilteringDataset, StartEndDataset  # noqa: E402
from cone.matcher import HungarianMatcher  # noqa: E402
from cone.model import build_model  # noqa: E402
from utils.basic_utils import l2_normalize_np_array  # noqa: E402
from utils.temporal_nms import temporal_nms as ref_temporal_nms  # noqa: E402

from cone_amd.config import make_opt  # noqa: E402
from cone_amd import synth  # noqa: E402
import inputs as gi  # noqa: E402  (tests/golden/inputs.py)

TRAIN_ONLY = dict(set_cost_span=10, set_cost_giou=1, set_cost_class=4, span_loss_coef=10,
                  giou_loss_coef=1, label_loss_coef=4, lw_saliency=1, adapter_loss=True,
                  adapter_loss_coef=1, eos_coef=.1, temperature=.07, saliency_margin=.2)


def ref_opt(preset, **kw):
    opt = make_opt(preset, **kw)
    for k, v in TRAIN_ONLY.items():
        setattr(opt, k, v)
    opt.device = torch.device("cpu")
    opt.pin_memory = False
    opt.num_workers = 0
    return opt


def ref_model(opt, seed):
    model, _ = build_model(opt)
    sd = synth.make_state_dict(opt, seed)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model.eval()
    return model, synth.state_dict_checksum(sd)


class MemPre(PreFilteringDataset):
    """PreFilteringDataset with the LMDB readers replaced by dict lookups."""

    def __init__(self, ann, video_feats, query_feats):
        self.data_mode = "context"
        self.query_data = ann
        self.video_data = list(dict.fromkeys(r["clip_id"] for r in ann))
        self.video2idx = {v: i for i, v in enumerate(self.video_data)}
        self._v, self._q = video_feats, query_feats

    def _get_video_appearance_feat_by_vid(self, vid):
        return torch.from_numpy(l2_normalize_np_array(self._v[vid]))

    def _get_query_feat_by_qid(self, qid):
        return l2_normalize_np_array(self._q[qid]["cls_features"])


class MemTxn:
    """An LMDB read transaction over in-memory arrays: ``get(key)`` -> the np.savez blob the reference's readers parse."""

    def __init__(self, entries):
        self._e = entries

    def get(self, key):
        buf = io.BytesIO()
        np.savez(buf, **self._e[bytes(key).decode()])
        return buf.getvalue()


class MemSE(StartEndDataset):
    """StartEndDataset (eval branch) with the LMDB readers replaced by dict lookups."""

    def __init__(self, opt, ann, video_feats, query_feats, motion_feats=None):
        self.max_q_l, self.max_v_l = opt.max_q_l, opt.max_v_l
        self.use_video, self.clip_len = True, opt.clip_length
        self.topk_window = opt.topk_window
        self.slide_window_size = int(opt.max_v_l / 2)
        self.eval, self.same_visual_path = True, True
        self.load_labels = False
        self.data = ann
        self.query_id2windowidx = None
        self._q = query_feats
        self.videofeat = {k: torch.from_numpy(v) for k, v in video_feats.items()}  # RAW (H2)
        if motion_feats is not None:        # a second visual source (the reference's motion_feat_dir != appearance_feat_dir)
            # through the reference's OWN reader (_get_video_motion_feat_by_vid, dataloader :284-292): unlike the appearance
            # reader (H2) it hands back the L2-NORMALISED rows whenever normalize_v is set -- the motion LMDB is an in-memory
            # transaction of np.savez blobs here, nothing else is replaced
            self.same_visual_path = False
            self.normalize_v = not opt.no_norm_vfeat
            self.motion_visual_txn = MemTxn({k: {"features": v} for k, v in motion_feats.items()})
            self.motion_videofeat = {k: self._get_video_motion_feat_by_vid(k) for k in motion_feats}

    def _get_query_feat_by_qid(self, qid):
        q = self._q[qid]
        tok = l2_normalize_np_array(q["token_features"][:self.max_q_l])
        return torch.from_numpy(tok), l2_normalize_np_array(q["cls_features"])


class StableSort:
    """Force torch.sort(stable=True) while active (H6)."""

    def __enter__(self):
        self._orig = torch.sort

        def stable_sort(x, *a, **k):
            k["stable"] = True
            return self._orig(x, *a, **k)

        torch.sort = stable_sort

    def __exit__(self, *exc):
        torch.sort = self._orig


# ---- fixtures ------------------------------------------------------------------------------
def gen_stage_b(name, preset, seed, lens_v, lens_q, **opt_kw):
    """CONE.forward + forward_clip_matching on one ragged padded batch (opt_kw: reference options, e.g. use_txt_pos)."""
    opt = ref_opt(preset, **opt_kw)
    model, cks = ref_model(opt, seed)
    inp = gi.stage_b_inputs(opt, 1000 + seed, lens_v, lens_q)
    vid, txt, vmask, tmask, cls = (inp[k] for k in ("src_vid", "src_txt", "vid_mask", "txt_mask", "src_cls_txt"))
    cap = {}
    model.transformer.encoder.register_forward_hook(lambda m, i, o: cap.__setitem__("memory", o))
    model.transformer.decoder.register_forward_hook(lambda m, i, o: cap.__setitem__("hs", o))
    model.input_vid_proj.register_forward_hook(lambda m, i, o: cap.__setitem__("vproj", o))
    model.input_txt_proj.register_forward_hook(lambda m, i, o: cap.__setitem__("tproj", o))
    with torch.no_grad():
        t = lambda a: torch.from_numpy(a)
        out = model(t(txt), t(tmask), t(vid), t(vmask))
        match = model.forward_clip_matching(t(cls), t(vid), t(vmask), proposal=out["pred_spans"])
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        preset=preset, weight_seed=seed, weight_checksum=cks, input_seed=1000 + seed,
        **({"use_txt_pos": 1} if opt_kw.get("use_txt_pos") else {}),     # (key present only in the fixtures that set it)
        **({"pre_norm": 1} if opt_kw.get("pre_norm") else {}),
        lens_v=np.array(lens_v), lens_q=np.array(lens_q),
        input_checksum=gi.checksum(vid, txt, cls),
        pred_logits=out["pred_logits"].numpy(), pred_spans=out["pred_spans"].numpy(),
        saliency_scores=out["saliency_scores"].numpy(), matching=match.numpy(),
        aux_logits=out["aux_outputs"][0]["pred_logits"].numpy(),
        aux_spans=out["aux_outputs"][0]["pred_spans"].numpy(),
        memory=cap["memory"].transpose(0, 1).numpy(),             # (B, L, d)
        hs=cap["hs"].permute(0, 2, 1, 3).numpy(),                  # (layers, B, Nq, d)
        vproj=cap["vproj"].numpy()[:, ::4], tproj=cap["tproj"].numpy()[:, ::2],
    )
    print("wrote", name)


def gen_stage_a(name, preset, seed, ctx_ls):
    """adapter+renorm, frame scores, window max, stable rank list for a few videos."""
    opt = ref_opt(preset)
    model, cks = ref_model(opt, seed)
    inputs = gi.stage_a_inputs(opt, 2000 + seed, ctx_ls)
    rec = dict(preset=preset, weight_seed=seed, weight_checksum=cks, input_seed=2000 + seed,
               ctx_ls=np.array(ctx_ls), input_checksum=gi.checksum(*[a for p in inputs for a in p]))
    S, W = int(opt.max_v_l / 2), opt.max_v_l
    for vi, ctx_l in enumerate(ctx_ls):
        raw, cls = inputs[vi]
        assert np.array_equal(gi.l2n(raw), l2_normalize_np_array(raw).astype(np.float32))
        v = torch.from_numpy(l2_normalize_np_array(raw))[None]
        with torch.no_grad():
            a = model.adapter_layer(v) + v                            # cone/inference.py:254-258
            a = (a / a.norm(dim=2, keepdim=True))[0]
        rec[f"adapted_{vi}"] = a.numpy()[::7]                        # every 7th clip
        for qi in range(3):
            c = torch.from_numpy(l2_normalize_np_array(cls[qi]))
            fs = torch.einsum("db,b->d", a, c).detach().cpu()          # :284
            nw = math.ceil(ctx_l / S) + 1
            wl = [torch.max(fs[max((i - 1) * S, 0):min((i - 1) * S + W, ctx_l)]) for i in range(nw)]
            wt = torch.Tensor(wl)
            with StableSort():
                _, idx = torch.sort(wt, descending=True)
            rec[f"frame_{vi}_{qi}"], rec[f"win_{vi}_{qi}"], rec[f"rank_{vi}_{qi}"] = \
                fs.numpy(), wt.numpy(), idx.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name)


def gen_e2e(name, preset, seed, n_queries, n_videos, ctx_range, **optkw):
    """The unmodified reference eval_epoch on an in-memory split -> submission files.  ``v_motion_feat_dim`` in optkw:
    the window model reads a SECOND visual source of that width (synth.make_motion_feats, seed = the data seed)."""
    opt = ref_opt(preset, nms_thd=0.5, eval_split_name="test", save_all=True, **optkw)
    model, cks = ref_model(opt, seed)
    ann, vf, qf = synth.make_dataset(opt, n_queries, n_videos, seed=3000 + seed, ctx_range=ctx_range)
    mf = synth.make_motion_feats(opt, vf, seed=3000 + seed) if "v_motion_feat_dim" in optkw else None
    captured = {}
    orig = ref_inf.get_eval_res

    def spy(*a, **k):
        res = orig(*a, **k)
        captured["mr_res"] = res[0]
        return res

    ref_inf.get_eval_res = spy
    se = MemSE(opt, ann, vf, qf, mf)
    with tempfile.TemporaryDirectory() as td:
        opt.results_dir = td
        opt.eval_path = os.path.join(td, "ann.jsonl")
        with open(opt.eval_path, "w") as fh:
            fh.write("\n".join(json.dumps(r) for r in ann))
        ext = "jsonl" if preset == "mad" else "json"
        fn = f"inference_{preset}_test_golden_preds.{ext}"
        try:
            with torch.no_grad(), StableSort():
                ref_inf.eval_epoch(model, MemPre(ann, vf, qf), se, opt, fn)
        except SystemExit:
            pass
        except (UnboundLocalError, NameError):
            pass  # MAD test split falls through to metric code without ground truth files
        files = {}
        for f in sorted(x for x in os.listdir(td) if x.startswith("inference_") and "preds.json" in x):
            with open(os.path.join(td, f)) as fh:
                files[f] = fh.read()
    ref_inf.get_eval_res = orig
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(dict(preset=preset, weight_seed=seed, weight_checksum=cks, data_seed=3000 + seed,
                       n_queries=n_queries, n_videos=n_videos, ctx_range=list(ctx_range), opt=optkw,
                       ranks=se.query_id2windowidx, mr_res=captured["mr_res"], files=files), f)
    print("wrote", name, list(files))


def gen_stage_c(name, seed):
    """score_fusion + post_processing_mr_nms + temporal_nms on crafted candidate lists."""
    rng = np.random.default_rng(4000 + seed)
    cases = []
    for ci in range(40):
        n = int(rng.choice([1, 2, 5, 37, 100, 150, 230]))
        st = np.round(rng.uniform(0, 480, n), 4)
        ed = np.round(st + rng.uniform(0.5, 60, n), 4)
        prop = np.round(rng.uniform(0, 1, n), 4)
        match = np.round(rng.uniform(-0.2, 0.4, n), 4)
        if ci % 3 == 0 and n > 4:            # duplicates (H5) and score ties
            st[n // 2], ed[n // 2] = st[1], ed[1]
            st[n - 1], ed[n - 1] = st[0], ed[0]
            prop[2] = prop[3]
            match[1] = match[4]
        if ci % 7 == 0:
            prop[:] = 0.5                     # min == max branch of normalize_score
        if ci % 11 == 0 and n > 3:
            st[2], ed[2] = 0.0, 0.0           # zero-length / union == 0
            st[3], ed[3] = -0.0, 0.0
        rows = [[float(a), float(b), float(c), float(d)] for a, b, c, d in zip(st, ed, prop, match)]
        for thd, max_before, max_after, split in ((0.5, 200, 5, "test"), (-1, 200, 5, "test"),
                                                  (0.3, 50, 10, "val"), (0.7, 200, 100, "test")):
            o = SimpleNamespace(nms_thd=thd, max_before_nms=max_before, max_after_nms=max_after,
                                eval_split_name=split)
            rd = ref_inf.score_fusion(rows)
            outs = [ref_inf.post_processing_mr_nms(o, rd, idx) for idx in (2, 0, 1)]
            cases.append(dict(rows=rows, nms_thd=thd, max_before_nms=max_before,
                              max_after_nms=max_after, fused=outs[0], proposal=outs[1], matching=outs[2]))
    direct = []
    for ci in range(20):
        n = int(rng.choice([1, 2, 3, 10, 64, 200]))
        st = rng.uniform(0, 100, n)
        ed = st + rng.uniform(0, 30, n)
        sc = np.round(rng.uniform(0, 1, n), 2)    # coarse scores -> many ties
        pred = [[float(a), float(b), float(c)] for a, b, c in zip(st, ed, sc)]
        for thd, ma in ((0.5, 5), (0.0, 100), (0.9, 3)):
            direct.append(dict(pred=pred, nms_thd=thd, max_after_nms=ma,
                               out=ref_temporal_nms([list(p) for p in pred], thd, ma)))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(dict(fusion_nms=cases, temporal_nms=direct), f)
    print("wrote", name)


def gen_matcher(name, seed):
    """HungarianMatcher cost matrix + assignment (cone/matcher.py:37-106), 1 target/window."""
    rng = np.random.default_rng(5000 + seed)
    B, Nq = 7, 5
    logits = rng.standard_normal((B, Nq, 2)).astype(np.float32)
    spans = np.stack([rng.uniform(0.2, 0.8, (B, Nq)), rng.uniform(0.05, 0.4, (B, Nq))], -1).astype(np.float32)
    tgt = np.stack([rng.uniform(0.2, 0.8, B), rng.uniform(0.05, 0.4, B)], -1).astype(np.float32)
    m = HungarianMatcher(cost_class=4, cost_span=10, cost_giou=1)
    captured = {}
    import cone.matcher as mm
    orig = mm.linear_sum_assignment

    def spy(c):
        captured.setdefault("C", []).append(np.asarray(c))
        return orig(c)

    mm.linear_sum_assignment = spy
    idx = m({"pred_logits": torch.from_numpy(logits), "pred_spans": torch.from_numpy(spans)},
            {"span_labels": [{"spans": torch.from_numpy(tgt[b:b + 1])} for b in range(B)]})
    mm.linear_sum_assignment = orig
    np.savez_compressed(os.path.join(HERE, name + ".npz"), logits=logits, spans=spans, tgt=tgt,
                        C=np.stack(captured["C"]), idx_i=np.stack([i.numpy() for i, _ in idx]),
                        idx_j=np.stack([j.numpy() for _, j in idx]))
    print("wrote", name)


def gen_criterion(name, seed):
    """SetCriterion.forward (cone/model.py:213-425) + HungarianMatcher with 1-3 targets per window
    (cone/matcher.py:37-106) on seeded random model outputs: every loss of the top layer and of the auxiliary
    layer, with and without negative-window outputs; loss_adapter on a random similarity matrix."""
    rng = np.random.default_rng(7000 + seed)
    opt = ref_opt("ego4d")
    _, criterion = build_model(opt)
    criterion.eval()
    B, Nq, L, P = 9, 5, 90, 2
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    mk_spans = lambda n: np.stack([rng.uniform(0.15, 0.85, n), rng.uniform(0.04, 0.5, n)], -1).astype(np.float32)
    layers = []
    for _ in range(2):
        layers.append(dict(pred_logits=f32(rng.standard_normal((B, Nq, 2)) * 2), pred_spans=mk_spans((B, Nq)).reshape(B, Nq, 2)))
    sal = f32(rng.standard_normal((B, L)))
    n_tgt = [1, 2, 1, 3, 1, 1, 2, 1, 5]
    tgt = [mk_spans(n) for n in n_tgt]
    pos_idx = rng.integers(0, L, (B, P))
    neg_idx = rng.integers(0, L, (B, P))
    neg = dict(pred_logits=f32(rng.standard_normal((B, Nq, 2)) * 2), saliency_scores=f32(rng.standard_normal((B, L))))
    t = torch.from_numpy
    outputs = dict(pred_logits=t(layers[1]["pred_logits"]), pred_spans=t(layers[1]["pred_spans"]), saliency_scores=t(sal),
                   aux_outputs=[dict(pred_logits=t(layers[0]["pred_logits"]), pred_spans=t(layers[0]["pred_spans"]))])
    targets = dict(span_labels=[dict(spans=t(x)) for x in tgt], saliency_pos_labels=t(pos_idx), saliency_neg_labels=t(neg_idx))
    neg_outputs = dict(pred_logits=t(neg["pred_logits"]), saliency_scores=t(neg["saliency_scores"]))
    with torch.no_grad():
        idx = criterion.matcher({k: v for k, v in outputs.items() if k != "aux_outputs"}, targets)
        idx_aux = criterion.matcher(outputs["aux_outputs"][0], targets)
        with_neg = criterion(outputs, targets, neg_outputs)
        without = criterion(outputs, targets, None)
        no_targets = criterion(outputs, None)
        sim = f32(rng.standard_normal((6, 6)))
        adapter = criterion.loss_adapter(dict(logits_per_video=t(sim)))
    tofl = lambda d: {k: float(v) for k, v in d.items()}
    fx = dict(seed=seed, B=B, Nq=Nq, L=L, n_tgt=n_tgt,
              layers=[{k: v.tolist() for k, v in l.items()} for l in layers], saliency=sal.tolist(),
              tgt=[x.tolist() for x in tgt], pos_idx=pos_idx.tolist(), neg_idx=neg_idx.tolist(),
              neg={k: v.tolist() for k, v in neg.items()}, sim=sim.tolist(),
              hyper=dict(eos_coef=opt.eos_coef, temperature=opt.temperature, saliency_margin=opt.saliency_margin,
                         set_cost_span=opt.set_cost_span, set_cost_giou=opt.set_cost_giou, set_cost_class=opt.set_cost_class),
              weight_dict={k: float(v) for k, v in criterion.weight_dict.items()},
              idx=[[i.tolist(), j.tolist()] for i, j in idx], idx_aux=[[i.tolist(), j.tolist()] for i, j in idx_aux],
              losses_with_neg=tofl(with_neg), losses_without_neg=tofl(without), losses_no_targets=tofl(no_targets),
              loss_adapter=tofl(adapter))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(fx, f)
    print("wrote", name, {k: round(v, 4) for k, v in fx["losses_with_neg"].items()})


def gen_localizer(name, seed):
    """run_on_video CONELocalizator.predict_moment on two synthetic videos (ckpt loading bypassed)."""
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__(d or {}, **kw)
            self.__dict__ = self

    ed.EasyDict = EasyDict
    sys.modules.setdefault("easydict", ed)
    import run_on_video.cone_localizator as loc
    a = loc.args
    opt = ref_opt("ego4d", clip_length=a.clip_length, topk_window=a.topk_window)
    model, cks = ref_model(opt, seed)
    obj = object.__new__(loc.CONELocalizator)
    obj.device = "cpu"
    obj.localizator = model
    obj.slide_window_size = int(a.max_v_l / 2)
    obj.max_v_l = a.max_v_l
    rng = np.random.default_rng(6000 + seed)
    cases = []
    for ctx_l, lq in ((901, 11), (1033, 20)):
        vid = rng.standard_normal((ctx_l, 256), dtype=np.float32) * 3
        tok = rng.standard_normal((lq, 768), dtype=np.float32)
        cls = rng.standard_normal((256,), dtype=np.float32)
        with StableSort():
            out = obj.predict_moment(torch.from_numpy(vid), (torch.from_numpy(tok), torch.from_numpy(cls)))
        cases.append(dict(ctx_l=ctx_l, lq=lq, out=out))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(dict(weight_seed=seed, weight_checksum=cks, input_seed=6000 + seed,
                       clip_length=a.clip_length, topk_window=a.topk_window, cases=cases), f)
    print("wrote", name)


def metric_inputs(seed, nq=48):
    """Synthetic submissions + ground truth for the metric evaluators (also used by the tests to rebuild
    device inputs): per query 1..9 rows [st, ed, prop, match, fused] at 4 decimals, with exact-threshold,
    disjoint, contained and degenerate (zero-length on a zero-length target) cases mixed in."""
    rng = np.random.default_rng(seed)
    preds, gts = [], []
    for q in range(nq):
        g0 = round(float(rng.uniform(0, 400)), 4)
        g1 = round(g0 + float(rng.uniform(1, 60)), 4)
        n = int(rng.integers(1, 10))
        rows = []
        for i in range(n):
            kind = rng.integers(0, 6)
            if kind == 0:                      # near the target
                s0 = g0 + float(rng.uniform(-20, 20)); e0 = s0 + float(rng.uniform(0.5, 70))
            elif kind == 1:                    # far away
                s0 = g1 + float(rng.uniform(50, 90)); e0 = s0 + float(rng.uniform(1, 30))
            elif kind == 2:                    # IoU exactly 0.5 / 0.3 of an integer-length target (strict > must fail)
                s0 = g0; e0 = g0 + (g1 - g0) * (0.5 if i % 2 else 0.3)
            elif kind == 3:                    # contains the target
                s0 = g0 - float(rng.uniform(0, 5)); e0 = g1 + float(rng.uniform(0, 5))
            elif kind == 4:                    # identical
                s0, e0 = g0, g1
            else:                              # zero length
                s0 = e0 = g0 + float(rng.uniform(0, 10))
            rows.append([round(s0, 4), round(e0, 4), round(float(rng.uniform(0, 1)), 4),
                         round(float(rng.uniform(-1, 1)), 4), round(float(rng.uniform(0, 2)), 4)])
        preds.append(rows)
        gts.append([g0, g1])
    # degenerate pair: zero-length target hit by a zero-length prediction -> 0/0
    gts[3] = [7.0, 7.0]                 # (kept off rank 1 so that mIoU stays finite; numpy gives nan > thr == False)
    while len(preds[3]) < 3:
        preds[3].append([1.0, 2.0, .1, .1, .1])
    preds[3][1][:2] = [7.0, 7.0]
    gts[5] = [10.0, 20.0]
    preds[5] = [[10.0, 13.0, .5, .5, .5], [10.0, 15.0, .4, .4, .4], [12.0, 18.0001, .3, .3, .3]]   # 0.3, 0.5, >0.6
    return preds, gts


def gen_metrics(name, seed):
    """standalone_eval/evaluate_ego4d_nlq.py:63-115, evaluate_mad.py:61-107, evaluate_pre_filtered_window.py:30-72."""
    import standalone_eval.evaluate_ego4d_nlq as ego4d_eval
    import standalone_eval.evaluate_mad as mad_eval
    import standalone_eval.evaluate_pre_filtered_window as window_eval
    preds, gts = metric_inputs(seed)
    nq = len(preds)
    # --- ego4d: nested ground truth, flat predictions
    gt_json = {"videos": [{"clips": []}]}
    predictions = []
    for q in range(nq):
        clip, ann_uid, qidx = f"clip{q // 5}", f"ann{q // 3}", q % 3
        clips = gt_json["videos"][0]["clips"]
        c = next((c for c in clips if c["clip_uid"] == clip), None)
        if c is None:
            c = {"clip_uid": clip, "annotations": []}
            clips.append(c)
        a = next((a for a in c["annotations"] if a["annotation_uid"] == ann_uid), None)
        if a is None:
            a = {"annotation_uid": ann_uid, "language_queries": [None, None, None]}
            c["annotations"].append(a)
        a["language_queries"][qidx] = {"clip_start_sec": gts[q][0], "clip_end_sec": gts[q][1]}
        predictions.append({"query_idx": qidx, "annotation_uid": ann_uid, "predicted_times": preds[q], "clip_uid": clip})
    for c in gt_json["videos"][0]["clips"]:
        for a in c["annotations"]:
            a["language_queries"] = [x if x is not None else {"clip_start_sec": 0.0, "clip_end_sec": 1.0}
                                     for x in a["language_queries"]]
    thr_e, topk_e = [0.3, 0.5], [1, 5, 10, 50, 100]                 # cone/inference.py:422-423
    with np.errstate(all="ignore"):
        res_e, miou_e = ego4d_eval.evaluate_nlq_performance(predictions, gt_json, thr_e, topk_e)
    # --- mad: jsonl-style
    sub = [{"query_id": f"q{q}", "predicted_times": preds[q], "video_id": f"v{q // 7}"} for q in range(nq)]
    gt_mad = [{"query_id": f"q{q}", "timestamps": gts[q]} for q in range(nq)]
    thr_m, topk_m = torch.tensor([0.1, 0.3, 0.5]), torch.tensor([1, 5, 10, 50, 100])    # cone/inference.py:333-334
    res_m = mad_eval.evaluate_nlq_performance(sub, gt_mad, thr_m, topk_m)
    # --- window pre-filter recall
    rng = np.random.default_rng(seed + 1)
    opt = SimpleNamespace(clip_length=0.535, max_v_l=90)
    q2w = {}
    for q in range(nq):
        nw = int(rng.integers(3, 60))
        q2w[f"q{q}"] = [int(x) for x in rng.permutation(nw)]
    topk_w = torch.tensor([1, 5, 10, 30, 50])
    res_w = window_eval.windows_selection(q2w, gt_mad, topk_w, opt)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(dict(seed=seed, preds=preds, gts=gts,
                       ego4d=dict(predictions=predictions, ground_truth=gt_json, thresholds=thr_e, topK=topk_e,
                                  results=np.asarray(res_e).tolist(), mIoU=float(miou_e)),
                       mad=dict(thresholds=[0.1, 0.3, 0.5], topK=topk_m.tolist(),
                                results=[[float(x) for x in r] for r in res_m.tolist()]),
                       window=dict(ranklists=q2w, clip_length=opt.clip_length, max_v_l=opt.max_v_l,
                                   topK=topk_w.tolist(), results=[float(x) for x in res_w.tolist()])), f)
    print("wrote", name, "mIoU", float(miou_e))


def main():
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "metrics":
        gen_metrics("metrics", 0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "criterion":
        gen_criterion("criterion", 0)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "prenorm":
        gen_stage_b("stageB_ego4d_prenorm", "ego4d", 4, [90, 45, 90, 17, 1, 63], [12, 5, 20, 9, 7, 17], pre_norm=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "two_sources":
        gen_e2e("e2e_ego4d_two_sources", "ego4d", 5, 10, 3, (100, 330), v_motion_feat_dim=128, eval_bsz=4, topk_window=4)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "txtpos":
        gen_stage_b("stageB_ego4d_txtpos", "ego4d", 3, [90, 45, 90, 17, 1, 63], [12, 5, 20, 9, 7, 17], use_txt_pos=True)
        return
    gen_stage_b("stageB_ego4d", "ego4d", 0, [90, 45, 90, 17, 1, 63], [12, 5, 20, 9, 7, 17])
    gen_stage_b("stageB_mad", "mad", 1, [125, 62, 3, 125], [25, 6, 11, 18])
    # --use_txt_pos (cone/config.py:115): text tokens carry TrainablePositionalEncoding(src_txt) (cone/model.py:106)
    gen_stage_b("stageB_ego4d_txtpos", "ego4d", 3, [90, 45, 90, 17, 1, 63], [12, 5, 20, 9, 7, 17], use_txt_pos=True)
    # --pre_norm (cone/config.py): normalize_before in every transformer layer + the encoder's final norm (cone/transformer.py:19-36)
    gen_stage_b("stageB_ego4d_prenorm", "ego4d", 4, [90, 45, 90, 17, 1, 63], [12, 5, 20, 9, 7, 17], pre_norm=True)
    gen_stage_a("stageA_ego4d", "ego4d", 0, [901, 900, 44, 91])
    gen_stage_a("stageA_mad", "mad", 1, [1250, 187])
    gen_e2e("e2e_ego4d", "ego4d", 0, 12, 3, (300, 420))
    gen_e2e("e2e_ego4d_small_bsz", "ego4d", 2, 9, 2, (80, 200), eval_bsz=4, topk_window=3)
    gen_e2e("e2e_mad", "mad", 1, 6, 2, (500, 800), topk_window=5)
    # two visual sources (cone/ego4d_mad_dataloader.py:63-81): motion features (128-d) into the window model, appearance
    # features (256-d) into the pre-filter and the proposal matching
    gen_e2e("e2e_ego4d_two_sources", "ego4d", 5, 10, 3, (100, 330), v_motion_feat_dim=128, eval_bsz=4, topk_window=4)
    gen_stage_c("stageC", 0)
    gen_matcher("matcher", 0)
    gen_criterion("criterion", 0)
    gen_localizer("localizer", 0)
    gen_metrics("metrics", 0)


if __name__ == "__main__":
    main()
