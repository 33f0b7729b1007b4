"""Coarse-to-fine inference driver on MI355X: the counterpart of ``cone/inference.py``.

Same CLI, same option handling (``cone_amd.config``), same checkpoint / ``opt.json`` inputs, same
output files (names, JSON / JSONL layout, row format ``[st, ed, proposal, matching, fused]``) and the
same stdout line ``total model running time:`` -- with stages A-C executed by HIP kernels:

  stage A  (cone/inference.py:241-301)  adapter + renorm over the whole clip arena, frame scores,
           window max, stable top-k                                  -> ``prefilter``
  stage B  (cone/inference.py:30-100)   window model + proposal matching on packed windows, gathered
           by index from device-resident arenas                      -> ``run_windows``
  stage C  (cone/inference.py:103-217)  4-dp rounding, fusion, dict collapse, 3x NMS, one workgroup
           per query                                                 -> ``fuse_and_nms``

Data layout in HBM (one split): ``vid_raw`` (sum ctx_l, dv) clip features of all videos back to back
(``vid_off`` row offsets), ``tok`` (sum Lq, dt) text token features, ``cls`` (nq, dv).  Windows are
(row0, len) pairs into these arenas -- nothing is padded or copied per window.

Scored splits (Ego4D val, MAD val / test) end with the reference's metric tables, counted on the device
(``cone_amd.metrics``).  Out of scope here: training-time evaluation hooks.
"""
from __future__ import annotations

import contextlib
import gc
import io
import json
import logging
import os
import time
import warnings
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .config import parse_test_options
from .model import build_model

logger = logging.getLogger(__name__)


# ------------------------------------------------------------------------------------ data
class AnnList(list):
    """The annotation rows of a store: a list that can carry its own derived metadata (the parsed submission keys), so that
    nothing about it has to be cached in a module-level table keyed by ``id()``."""
    __slots__ = ("_ego4d_keys",)

    def __getitem__(self, i):
        r = list.__getitem__(self, i)
        return AnnList(r) if isinstance(i, slice) else r


class FeatureStore:
    """Device-resident features of one evaluation split (replaces PreFilteringDataset +
    StartEndDataset of cone/ego4d_mad_dataloader.py for the eval path)."""

    def __init__(self, opt, annotations, video_feats, query_feats, device=None, tok_normalized=False,
                 cls_normalized=False, motion_feats=None, mot_normalized=False):
        """``video_feats``: clip_id -> RAW (ctx_l, dv) features (hazard H2: the window model sees them un-normalised);
        ``motion_feats`` (optional): clip_id -> RAW (ctx_l, dv_motion) features of a SECOND visual source -- the reference's
        ``motion_feat_dir`` when it is not ``appearance_feat_dir`` (cone/ego4d_mad_dataloader.py:63-81, 94-95): the window
        model (Moment-DETR) reads the motion features, the pre-filter and the proposal matching the appearance features
        (``video_feats``); the same clips, so the same number of rows per video.  None = one source for both (every shipped script).
        Unlike the appearance reader, the reference's motion reader hands the window model L2-NORMALISED rows
        (``_get_video_motion_feat_by_vid``, :284-292: ``x / (|x| + 1e-5)`` whenever ``normalize_v``): the arena ``mot_raw`` keeps what
        it was given, ``motion_rows()`` is what the window model reads (normalised on the device per step, like the text tokens);
        ``mot_normalized`` = the rows already went through that reader (``from_datasets``);
        ``query_feats``: query_id -> {token_features, cls_features | eot_features}.  ``tok_normalized`` /
        ``cls_normalized``: the text side already went through the reference's ``l2_normalize_np_array`` on the
        host (features taken from the reference's dataset objects, ``from_datasets``) and must not be normalised
        a second time."""
        self.opt = opt
        self.tok_normalized, self.cls_normalized = bool(tok_normalized), bool(cls_normalized)
        self.mot_normalized = bool(mot_normalized)
        self.ann = AnnList(annotations)
        if opt.data_ratio != 1:
            self.ann = self.ann[:int(len(self.ann) * opt.data_ratio)]   # dataloader :116-121
        dev = device or torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        qids = [r["query_id"] for r in self.ann]
        if len(set(qids)) != len(qids):
            raise ValueError("duplicate query_id in the annotation file")
        self.clip_ids = list(OrderedDict.fromkeys(r["clip_id"] for r in self.ann))
        self.clip2idx = {c: i for i, c in enumerate(self.clip_ids)}
        vids = [np.asarray(video_feats[c], dtype=np.float32) for c in self.clip_ids]
        self.ctx_l = [int(v.shape[0]) for v in vids]
        self.vid_off = np.concatenate([[0], np.cumsum(self.ctx_l)]).astype(np.int64)
        self.vid_raw = torch.from_numpy(np.concatenate(vids, 0)).to(dev)
        self.mot_raw = None         # the motion arena (same rows as vid_raw), or None: vid_raw serves both
        if motion_feats is not None:
            mots = [np.asarray(motion_feats[c], dtype=np.float32) for c in self.clip_ids]
            bad = [c for c, m, n in zip(self.clip_ids, mots, self.ctx_l) if int(m.shape[0]) != n]
            if bad:     # the reference slices both with the appearance length (dataloader :139-151); a shorter motion
                raise ValueError(f"motion and appearance features disagree on the number of clips of {bad[:3]}")  # video
            self.mot_raw = torch.from_numpy(np.concatenate(mots, 0)).to(dev)         # would silently shift its windows
        toks, clss = [], []
        for r in self.ann:
            q = query_feats[r["query_id"]]
            toks.append(np.asarray(q["token_features"], dtype=np.float32)[:opt.max_q_l])   # :272-273
            c = np.asarray(q["cls_features"] if "cls_features" in q else q["eot_features"], dtype=np.float32)
            clss.append(c[0] if c.ndim == 2 else c)                                         # :466-471
        self.tok_len = [int(t.shape[0]) for t in toks]
        self.tok_off = np.concatenate([[0], np.cumsum(self.tok_len)]).astype(np.int64)
        self.tok_raw = torch.from_numpy(np.concatenate(toks, 0)).to(dev)
        self.cls_raw = torch.from_numpy(np.stack(clss, 0)).to(dev)
        self.q_vid = np.array([self.clip2idx[r["clip_id"]] for r in self.ann], dtype=np.int64)
        self.q_base = 0         # annotation index of this store's first query in the split it was cut from
        self.nq_split = len(self.ann)
        self.max_tok_len = max(self.tok_len) if self.tok_len else 0     # of the SPLIT (views inherit it): kernel forms are
                                                                        # chosen from it, never from a chunk's own queries
        self._plan = None
        self._index = None

    @classmethod
    def subset(cls, store, lo: int, hi: int):
        """View of queries [lo, hi) of `store` (annotation order) sharing its device arenas -- a chunk of the
        software pipeline or the per-rank shard of a query-sharded multi-GPU run.  The view remembers where it
        sits in the split (``q_base``): the reference pads every batch of ``eval_bsz`` consecutive queries of
        the SPLIT to its longest window (hazard H3), so batch ids must stay global."""
        sub = cls.__new__(cls)
        sub.opt, sub.device = store.opt, store.device
        sub.tok_normalized, sub.cls_normalized = store.tok_normalized, store.cls_normalized
        sub.mot_normalized = store.mot_normalized
        sub.q_base, sub.nq_split = store.q_base + lo, store.nq_split
        sub.max_tok_len = store.max_tok_len
        sub.ann = store.ann[lo:hi]
        sub.clip_ids, sub.clip2idx, sub.ctx_l, sub.vid_off = store.clip_ids, store.clip2idx, store.ctx_l, store.vid_off
        sub.vid_raw, sub.mot_raw = store.vid_raw, store.mot_raw
        t0, t1 = int(store.tok_off[lo]), int(store.tok_off[hi])
        sub.tok_raw = store.tok_raw[t0:t1]
        sub.tok_len = store.tok_len[lo:hi]
        sub.tok_off = store.tok_off[lo:hi + 1] - t0
        sub.cls_raw = store.cls_raw[lo:hi]
        sub.q_vid = store.q_vid[lo:hi]
        sub._plan = None
        sub._index = None
        if getattr(store, "_index", None) is not None:      # slice the parent's device-side metadata: no H2D copy
            ix = store._index                               # (a pageable copy would wait for the stream to drain)
            sub._index = dict(q_ctx_l=ix["q_ctx_l"][lo:hi], q_vid_off=ix["q_vid_off"][lo:hi],
                              tok_off=ix["tok_off"][lo:hi + 1] - t0, tok_len=ix["tok_len"][lo:hi])
            if "tok_idx" in ix:
                sub._index["tok_idx"] = ix["tok_idx"][t0:t1]
        return sub

    def view(self, lo: int, hi: int):
        """``subset(self, lo, hi)``, cached on this store: a pipeline chunk or a rank's shard is the same view every step,
        and so are the static index tables it caches on the device (annotation metadata only)."""
        views = self.__dict__.setdefault("_views", {})
        v = views.get((lo, hi))
        # a view aliases the arenas it was cut from: refilling them IN PLACE (copy_) keeps it valid, assigning NEW tensors to
        # vid_raw / tok_raw / cls_raw does not -- such a view is rebuilt
        arenas = (self.vid_raw.data_ptr(), self.tok_raw.data_ptr(), self.cls_raw.data_ptr(),
                  0 if self.mot_raw is None else self.mot_raw.data_ptr())
        if v is None or v._arenas != arenas:
            if len(views) > 32:
                views.clear()
            v = views[(lo, hi)] = FeatureStore.subset(self, lo, hi)
            v._arenas = arenas
        return v

    def motion_rows(self, r0=0, r1=None):
        """Arena rows [r0, r1) as the WINDOW MODEL reads them (cone/ego4d_mad_dataloader.py:134-137, 150): one source -- the raw
        appearance rows (hazard H2); two sources -- the motion rows, L2-normalised with the reference's eps (its motion
        reader returns the normalised array, :284-292) unless they already are or ``--no_norm_vfeat`` is given."""
        r1 = int(self.vid_raw.shape[0]) if r1 is None else r1
        if self.mot_raw is None:
            return self.vid_raw[r0:r1]
        rows = self.mot_raw[r0:r1]
        if self.mot_normalized or getattr(self.opt, "no_norm_vfeat", False):
            return rows
        return ops.l2_normalize(rows, 1e-5)

    def index_tensors(self):
        """Static per-query index metadata on the device (depends on the annotation file only)."""
        if getattr(self, "_index", None) is None:
            dev = self.device
            qv = torch.from_numpy(self.q_vid).to(dev)
            self._index = dict(q_ctx_l=torch.tensor(self.ctx_l, device=dev)[qv],
                               q_vid_off=torch.from_numpy(self.vid_off).to(dev)[qv],
                               tok_off=torch.from_numpy(np.ascontiguousarray(self.tok_off)).to(dev),
                               tok_len=torch.tensor(self.tok_len, device=dev))
        if "tok_idx" not in self._index:    # index of every token row inside its query (--use_txt_pos: cone/model.py:106)
            self._index["tok_idx"] = torch.from_numpy(np.concatenate(
                [np.arange(n, dtype=np.int32) for n in self.tok_len] or [np.zeros(0, np.int32)])).to(self.device)
        return self._index

    def prefilter_plan(self):
        """Static index metadata of the pre-filter (depends on the annotation file only): groups of one
        video x up to 4 of its queries, and per-query offsets into the flat score buffers.  ``band`` = the arena rows
        [r0, r1) of the videos this store's queries refer to (a view of a few queries of a big split -- a rank's share
        -- adapts and scores only those); group row offsets are relative to r0."""
        if self._plan is not None:
            return self._plan
        S = int(self.opt.max_v_l / 2)
        by_vid = OrderedDict()
        for qi, v in enumerate(self.q_vid.tolist()):
            by_vid.setdefault(v, []).append(qi)
        r0 = int(min(self.vid_off[v] for v in by_vid)) if by_vid else 0
        r1 = int(max(self.vid_off[v + 1] for v in by_vid)) if by_vid else 0
        g_row0, g_ctx_l, g_q = [], [], []
        for v, qis in by_vid.items():
            for c0 in range(0, len(qis), 4):
                grp = qis[c0:c0 + 4]
                g_row0.append(int(self.vid_off[v]) - r0); g_ctx_l.append(self.ctx_l[v])
                g_q.append(grp + [-1] * (4 - len(grp)))
        q_ctx = np.array([self.ctx_l[v] for v in self.q_vid.tolist()], dtype=np.int64)
        q_nw = (q_ctx + S - 1) // S + 1
        dev = self.device
        t = lambda a, dt: torch.tensor(np.asarray(a), dtype=dt, device=dev)
        self._plan = dict(
            g_row0=t(g_row0, torch.int64), g_ctx_l=t(g_ctx_l, torch.int32), g_q=t(g_q, torch.int32).contiguous(),
            ng=len(g_row0), max_ctx_l=int(max(self.ctx_l[v] for v in by_vid)) if by_vid else 0, band=(r0, r1),
            q_fs_off=t(np.concatenate([[0], np.cumsum(q_ctx)[:-1]]), torch.int64),
            q_win_off=t(np.concatenate([[0], np.cumsum(q_nw)[:-1]]), torch.int64),
            q_ctx_l=t(q_ctx, torch.int32), fs_total=int(q_ctx.sum()), win_total=int(q_nw.sum()))
        return self._plan

    @classmethod
    def from_lmdb(cls, opt, device=None):
        """Read the reference's LMDBs of np.savez blobs (keys ``features`` / ``token_features`` +
        ``cls_features``|``eot_features``, keyed by clip_id / query_id utf-8; cone/ego4d_mad_dataloader.py:73-103,
        258-302, 453-473) -- the default data source of the CLI, as in the reference."""
        try:
            import lmdb
        except ImportError as e:
            raise ImportError("reading the reference LMDB feature stores needs the `lmdb` package "
                              "(or convert once with `python -m cone_amd.pack_features` where it is installed)") from e
        with open(opt.eval_path) as f:
            ann = [json.loads(l.strip("\n")) for l in f.readlines()]      # utils/basic_utils.py:51-53
        if opt.data_ratio != 1:
            ann = ann[:int(len(ann) * opt.data_ratio)]                     # dataloader :116-121 (before any read)

        def read_all(path, keys, fields):
            env = lmdb.open(path, readonly=True, create=False, max_readers=4096 * 8, readahead=False)
            txn = env.begin(buffers=True)
            out = {}
            for k in keys:
                dump = txn.get(k.encode())
                if dump is None:
                    raise KeyError(f"{path}: no entry for {k!r}")
                with io.BytesIO(dump) as reader:
                    blob = np.load(reader, allow_pickle=True)
                    out[k] = {f: np.asarray(blob[f]) for f in fields if f in blob}
            return out

        vf = {k: v["features"] for k, v in read_all(opt.appearance_feat_dir,
                                                    OrderedDict.fromkeys(r["clip_id"] for r in ann),
                                                    ["features"]).items()}
        mf = None
        if opt.motion_feat_dir != opt.appearance_feat_dir:        # dataloader :77-81, 105-111, 294-302
            mf = {k: v["features"] for k, v in read_all(opt.motion_feat_dir, OrderedDict.fromkeys(r["clip_id"] for r in ann),
                                                        ["features"]).items()}
        qf = read_all(opt.t_feat_dir, [r["query_id"] for r in ann],
                      ["token_features", "cls_features", "eot_features"])
        st = cls(SimpleNamespace(**dict(vars(opt), data_ratio=1)), ann, vf, qf, device=device, motion_feats=mf)
        st.opt = opt
        return st

    @classmethod
    def from_datasets(cls, opt, eval_inter_window_dataset, eval_intra_window_dataset, device=None):
        """Build the store from the reference's two dataset objects, as ``eval_epoch(model, inter_ds, intra_ds, opt,
        ...)`` receives them (cone/inference.py:227-228; cone/train.py:164-168): annotation rows and the RAM-resident
        RAW clip features from the StartEndDataset (``.data``, ``.videofeat``: dataloader :93-103, hazard H2), text
        features through its own accessor ``_get_query_feat_by_qid`` (:258-282) -- which hands back tokens that are
        already truncated and (unless ``normalize_t`` is off) L2-normalised, and a normalised cls vector, so the
        store is told not to normalise them again.  The PreFilteringDataset reads the same LMDBs (:409-431) and only
        contributes a consistency check here."""
        intra, inter = eval_intra_window_dataset, eval_inter_window_dataset
        ann = list(intra.data)
        if inter is not None and hasattr(inter, "query_data"):
            if [r["query_id"] for r in inter.query_data] != [r["query_id"] for r in ann]:
                raise ValueError("the two datasets list different queries")
        to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        vf = {c: to_np(intra.videofeat[c]) for c in OrderedDict.fromkeys(r["clip_id"] for r in ann)}
        mf = None
        if not getattr(intra, "same_visual_path", True):          # dataloader :94-95, 134-137: a second RAM-resident table
            mf = {c: to_np(intra.motion_videofeat[c]) for c in vf}
        qf = {}
        for r in ann:
            tok, cq = intra._get_query_feat_by_qid(r["query_id"])
            qf[r["query_id"]] = dict(token_features=to_np(tok), cls_features=to_np(cq))
        st = cls(SimpleNamespace(**dict(vars(opt), data_ratio=1)), ann, vf, qf, device=device,
                 tok_normalized=bool(getattr(intra, "normalize_t", True)), cls_normalized=True, motion_feats=mf,
                 mot_normalized=bool(getattr(intra, "normalize_v", True)))      # motion_videofeat: the reader's normalised rows
        st.opt = opt
        return st

    # -- packed arena file: one mmap-able .bin per split instead of an LMDB of compressed npz
    PACK_MAGIC = b"CONEFS01"

    def save_packed(self, path):
        """Write the split as ONE file: magic, header length, JSON header (annotation rows, clip ids, row
        offsets, dims), then the three fp32 arenas at 4 KiB-aligned offsets -- exactly the device layout, so
        loading is an mmap + one H2D copy per arena (no decompression, no per-query slicing)."""
        arrs = dict(vid_raw=self.vid_raw.cpu().numpy(), tok_raw=self.tok_raw.cpu().numpy(),
                    cls_raw=self.cls_raw.cpu().numpy())
        if self.mot_raw is not None:
            arrs["mot_raw"] = self.mot_raw.cpu().numpy()     # (an optional fourth arena: files without it read as before)
        head = dict(version=1, ann=self.ann, clip_ids=self.clip_ids, ctx_l=self.ctx_l, tok_len=self.tok_len,
                    normalized=dict(tok=self.tok_normalized, cls=self.cls_normalized, mot=self.mot_normalized), arrays={})
        off = 0
        for k, a in arrs.items():
            head["arrays"][k] = dict(offset=off, shape=list(a.shape))
            off += -(-a.nbytes // 4096) * 4096
        blob = json.dumps(head).encode()
        data0 = -(-(16 + len(blob)) // 4096) * 4096
        with open(path, "wb") as f:
            f.write(self.PACK_MAGIC + len(blob).to_bytes(8, "little") + blob)
            for k, a in arrs.items():
                f.seek(data0 + head["arrays"][k]["offset"])
                f.write(np.ascontiguousarray(a, dtype=np.float32).tobytes())
            f.truncate(data0 + off)
        return path

    @classmethod
    def from_packed(cls, opt, path, device=None):
        """Load a split written by save_packed (np.memmap -> device).  The annotation rows travel in the
        header; opt.data_ratio is applied like the reference's loader (dataloader :116-121)."""
        with open(path, "rb") as f:
            if f.read(8) != cls.PACK_MAGIC:
                raise ValueError(f"{path}: not a packed CONE feature store")
            n = int.from_bytes(f.read(8), "little")
            head = json.loads(f.read(n).decode())
        data0 = -(-(16 + n) // 4096) * 4096
        mm = {k: np.memmap(path, dtype=np.float32, mode="r", offset=data0 + v["offset"], shape=tuple(v["shape"]))
              for k, v in head["arrays"].items()}
        st = cls.__new__(cls)
        st.opt = opt
        nz = head.get("normalized", {})     # (files written before the flags: raw arenas)
        st.tok_normalized, st.cls_normalized = bool(nz.get("tok", False)), bool(nz.get("cls", False))
        st.mot_normalized = bool(nz.get("mot", False))
        dev = device or torch.device("cuda", torch.cuda.current_device())
        st.device = dev
        st.ann = AnnList(head["ann"])
        nq = len(st.ann) if opt.data_ratio == 1 else int(len(st.ann) * opt.data_ratio)
        st.ann = st.ann[:nq]
        st.clip_ids = list(head["clip_ids"])
        st.clip2idx = {c: i for i, c in enumerate(st.clip_ids)}
        st.ctx_l = [int(x) for x in head["ctx_l"]]
        st.vid_off = np.concatenate([[0], np.cumsum(st.ctx_l)]).astype(np.int64)
        st.tok_len = [int(x) for x in head["tok_len"]][:nq]
        st.tok_off = np.concatenate([[0], np.cumsum(st.tok_len)]).astype(np.int64)
        def up(a):      # read-only mapping -> device; on the CPU device take a private copy
            if dev.type == "cpu":
                return torch.from_numpy(np.array(a))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", UserWarning)        # "non-writable array": it is only read
                return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        st.vid_raw = up(mm["vid_raw"])
        st.mot_raw = up(mm["mot_raw"]) if "mot_raw" in mm else None
        st.tok_raw = up(mm["tok_raw"][:int(st.tok_off[-1])])
        st.cls_raw = up(mm["cls_raw"][:nq])
        st.q_vid = np.array([st.clip2idx[r["clip_id"]] for r in st.ann], dtype=np.int64)
        st.q_base, st.nq_split = 0, len(st.ann)
        st.max_tok_len = max(st.tok_len) if st.tok_len else 0
        st._plan = None
        st._index = None
        return st


# ------------------------------------------------------------------------------------ stage A
@torch.no_grad()
def prefilter(model, store: FeatureStore, opt, k=None, keep=None):
    """cone/inference.py:241-301.  Returns win_idx (nq, topk) int32 on device (-1 = no such window);
    ``k`` overrides opt.topk_window (the window-recall table ranks deeper than the model consumes).  Only the clip rows of
    the videos ``store``'s queries refer to are normalised, adapted and scored (the whole arena for a whole split).
    ``keep`` (a dict): receives ``cls_norm``, the normalised cls vectors stage B's matching reads too (dataloader :473 / :280:
    the same vectors) -- one launch less per step."""
    plan = store.prefilter_plan()
    r0, r1 = plan["band"]
    vid_norm = ops.l2_normalize(store.vid_raw[r0:r1], 1e-5)   # PreFilteringDataset :459
    ctx = model.adapter_norm(vid_norm)                        # :254-258, all videos in one pass
    cls_norm = store.cls_raw if store.cls_normalized else ops.l2_normalize(store.cls_raw, 1e-5)     # :473
    if keep is not None:
        keep["cls_norm"] = cls_norm
    win_idx, _, _ = ops.prefilter_batched(ctx, cls_norm, plan, opt.max_v_l, k or opt.topk_window)
    return win_idx


class Selection:
    """Shape of a split's window list, from HOST metadata alone: the pre-filter ranks every window of the query's video
    (cone/inference.py:286-299) and the dataset takes the first ``topk_window`` of the list (cone/ego4d_mad_dataloader.py:146),
    so query q owns ``n_q[q] = min(K, ceil(ctx_l / S) + 1)`` rows whatever the scores are -- rows ``row_off[q] ..
    row_off[q + 1]`` of the list, in rank order.  Nothing downstream of the pre-filter therefore needs to LOOK at its result
    to size or order anything: no ``nonzero``, no host sync, the launch queue runs ahead of the GPU for any mix of video
    lengths, and the step captures as a hipGraph.  ``dense``: every query owns exactly K rows (row b = (b // K, b % K)):
    the kernels compute the map themselves and the device-side maps are never built."""

    def __init__(self, store, K: int, S: int):
        ctx = np.asarray(store.ctx_l, dtype=np.int64)[np.asarray(store.q_vid, dtype=np.int64)]
        self.K = int(K)
        self.n_q = np.minimum(self.K, -(-ctx // S) + 1).astype(np.int64)
        self.row_off = np.concatenate([[0], np.cumsum(self.n_q)]).astype(np.int64)
        self.n_rows = int(self.row_off[-1])
        self.nq = int(self.n_q.shape[0])
        self.dense = bool((self.n_q == self.K).all())
        self._dev = {}

    def query_of_row(self, r: int) -> int:
        """Query that owns row r of the list (host arithmetic; r < n_rows)."""
        return int(np.searchsorted(self.row_off, r, side="right") - 1)

    def maps(self, device):
        """(q_of, slot) int64 and (row_q, row_slot) int32 of every row, on ``device`` (uploaded once per selection;
        the int32 pair is None for a dense list on the GPU: the kernels compute b // K, b % K)."""
        key = ("maps", str(device))
        hit = self._dev.get(key)
        if hit is None:
            q = np.repeat(np.arange(self.nq, dtype=np.int64), self.n_q)
            slot = np.arange(self.n_rows, dtype=np.int64) - self.row_off[:-1][q]
            q_of, sl = torch.from_numpy(q).to(device), torch.from_numpy(slot).to(device)
            i32 = (None, None) if self.dense else (q_of.to(torch.int32), sl.to(torch.int32))
            hit = self._dev[key] = (q_of, sl) + i32
        return hit

    def candidates(self, Nq: int, device):
        """(cand_off int64 (nq), n_valid int32 (nq)): query q owns rows cand_off[q] .. + n_valid[q] of the flat (n_rows * Nq, 4)
        candidate matrix -- the per-window rows ARE that matrix (a query's windows are adjacent, in rank order)."""
        key = ("cand", int(Nq), str(device))
        hit = self._dev.get(key)
        if hit is None:
            hit = self._dev[key] = (torch.from_numpy(self.row_off[:-1] * Nq).to(device),
                                    torch.from_numpy((self.n_q * Nq).astype(np.int32)).to(device))
        return hit


def selection(store: FeatureStore, opt, K=None) -> Selection:
    """The Selection of ``store`` for top-K (default opt.topk_window), cached on the store (annotation metadata only)."""
    K = int(K or opt.topk_window)
    S = int(opt.max_v_l / 2)
    cache = store.__dict__.setdefault("_sel", {})
    sel = cache.get((K, S))
    if sel is None:
        if len(cache) > 8:
            cache.clear()
        sel = cache[(K, S)] = Selection(store, K, S)
    return sel


def _window_geometry(store: FeatureStore, opt, win_idx):
    """(q_of, slot, start, vlen) of every selected window, row-major over (query, rank slot): torch index arithmetic
    (CPU tensors of the gloo tests; the GPU path's cross-check ``opt.window_table_torch``).  Row order from the
    Selection -- host metadata -- never from win_idx."""
    nq, K = win_idx.shape
    W, S = opt.max_v_l, int(opt.max_v_l / 2)
    st = store.index_tensors()
    q_of, slot, _, _ = selection(store, opt, K).maps(win_idx.device)
    wi = win_idx[q_of, slot].to(torch.int64)
    ctx_l = st["q_ctx_l"][q_of]
    start = torch.clamp((wi - 1) * S, min=0)
    end = torch.minimum((wi - 1) * S + W, ctx_l)
    return q_of, slot, start, end - start


def _use_table_kernel(opt, win_idx):
    return win_idx.shape[0] > 0 and win_idx.is_cuda and not getattr(opt, "window_table_torch", False)


def reference_batch_pad(store: FeatureStore, opt, win_idx):
    """Zero-padded clip length of every reference batch of the SPLIT ``store`` is (or was cut from): the
    reference collates ``eval_bsz`` consecutive queries x their top-k windows into one tensor padded to the
    longest window of that batch (cone/inference.py:306-313, utils/tensor_utils.py:36-39), and the proposal
    mean of forward_clip_matching divides by a length clipped to that padding (cone/model.py:186-199, hazard
    H3).  Returns (ceil(nq_split / eval_bsz),) int64; entries of batches this store holds no query of are 0."""
    nb = (store.nq_split + opt.eval_bsz - 1) // opt.eval_bsz
    if _use_table_kernel(opt, win_idx):
        return _table_launch(store, opt, win_idx, None, nb)["batch_pad"].to(torch.int64)     # one launch (cone_window_table)
    q_of, _, _, vlen = _window_geometry(store, opt, win_idx)
    bid = (q_of + store.q_base) // opt.eval_bsz
    return torch.zeros(nb, dtype=torch.int64, device=win_idx.device).scatter_reduce_(0, bid, vlen, reduce="amax")


def _table_launch(store: FeatureStore, opt, win_idx, batch_pad, nb):
    """cone_window_table over the Selection's row list: the table columns + ``batch_pad``."""
    st = store.index_tensors()
    i32 = st.get("i32")
    if i32 is None:
        i32 = st["i32"] = tuple(st[k].to(torch.int32).contiguous() for k in ("q_ctx_l", "q_vid_off", "tok_off", "tok_len"))
    bp = None if batch_pad is None else batch_pad.to(torch.int32).contiguous()
    _, _, row_q, row_slot = selection(store, opt, win_idx.shape[1]).maps(win_idx.device)
    return ops.window_table_rows(win_idx.contiguous(), *i32, store.q_base, opt.eval_bsz, opt.max_v_l, bp, nb,
                                 row_q=row_q, row_slot=row_slot)


def window_table(store: FeatureStore, opt, win_idx, batch_pad=None):
    """Eval branch of StartEndDataset.__getitem__ + collate (dataloader :144-159, 229-234, 305-344)
    as index arithmetic on the device: one row per (query, selected window), in annotation order -- for ANY mix of video
    lengths (a video of fewer than K windows contributes all of them; ``win_idx`` holds a query's valid entries first, the
    -1 tail is never read), with the row order taken from host metadata (``Selection``): nothing synchronises.

    ``batch_pad`` = reference_batch_pad() of the whole split.  Without it the padding is derived from the
    windows of ``store`` alone, which is only the reference's when ``store`` holds whole reference batches --
    a view cut inside a batch must be given the split's table (checked)."""
    nq, K = win_idx.shape
    st = store.index_tensors()
    if batch_pad is None:
        hi = store.q_base + nq
        if store.q_base % opt.eval_bsz or (hi % opt.eval_bsz and hi != store.nq_split):
            raise ValueError(f"queries [{store.q_base}, {hi}) of a {store.nq_split}-query split cut a reference "
                             f"batch (eval_bsz {opt.eval_bsz}): pass batch_pad=reference_batch_pad(split, ...)")
    if _use_table_kernel(opt, win_idx):
        nb = (store.nq_split + opt.eval_bsz - 1) // opt.eval_bsz
        wt = _table_launch(store, opt, win_idx, batch_pad, nb)          # the whole table is one launch
        wt.pop("batch_pad")
        wt["q_of"], wt["slot"] = selection(store, opt, K).maps(win_idx.device)[:2]
        return wt
    q_of, slot, start, vlen = _window_geometry(store, opt, win_idx)
    voff = st["q_vid_off"][q_of]
    if batch_pad is None:
        batch_pad = reference_batch_pad(store, opt, win_idx)
    bid = (q_of + store.q_base) // opt.eval_bsz
    i32 = lambda t: t.to(torch.int32).contiguous()
    tok_off, tok_len = st["tok_off"], st["tok_len"]
    return dict(q_of=q_of, slot=slot, vid_row0=i32(voff + start), vid_len=i32(vlen), video_start=i32(start),
                pad_len=i32(batch_pad[bid]), txt_row0=i32(tok_off[q_of]), txt_len=i32(tok_len[q_of]), cls_row=i32(q_of))


# ------------------------------------------------------------------------------------ stage B
@torch.no_grad()
def project_video(model, store: FeatureStore, row_range=None, ws=None):
    """Clip-side row work shared by every window that contains a clip: the input projection of each clip once
    (raw features: hazard H2) and, with the layer-0 cache, its first-layer q|k|v rows + the static position
    tables.  ``row_range`` = (r0, r1) restricts it to arena rows [r0, r1) (a rank that only runs the windows of
    some videos); ``vid_base`` is what window rows must be rebased by."""
    r0, r1 = row_range if row_range is not None else (0, int(store.vid_raw.shape[0]))
    vproj = model.project(0, store.motion_rows(r0, r1), ws=ws)   # the MOTION source (normalised when it is a second one)
    out = dict(vproj=vproj, vid_base=r0)
    if getattr(store.opt, "layer0_cache", True):
        out["l0_vid"] = model.layer0_rows(vproj, ws=ws)      # (the position tables are the model handle's own)
    return out


@torch.no_grad()
def project_text(model, store: FeatureStore, ws=None):
    """Text-side row work shared by all windows of a query: normalised tokens, their input projection and --
    with the layer-0 cache -- their first-layer q|k|v rows."""
    tok = store.tok_raw
    if not (store.tok_normalized or getattr(store.opt, "no_norm_tfeat", False)):
        tok = ops.l2_normalize(tok, 1e-5)                                   # dataloader :277-278 (normalize_t)
    tproj = model.project(1, tok, ws=ws)
    out = dict(tproj=tproj)
    if getattr(store.opt, "layer0_cache", True):
        out["l0_txt"] = model.layer0_rows(tproj, ws=ws)
        if getattr(model, "txt_pos_tables", False):     # --use_txt_pos: the tokens' own position rows, once per token as well
            out["txt_pos"], out["txt_pos_qk"] = model.text_positions(tproj, store.index_tensors()["tok_idx"])
    return out


@torch.no_grad()
def project_features(model, store: FeatureStore, video=None, cls_norm=None, text=None):
    """Row-wise work shared by every window that contains a clip and by all windows of a query
    (SURVEY.md H12): input projections of each clip / text token once, normalised cls vectors.  ``video`` =
    project_video() of the arena ``store`` shares (computed once per split, reused by its views); ``cls_norm`` = the
    normalised cls vectors of ``store`` when the pre-filter of the same call already made them."""
    video = video or project_video(model, store)
    text = text or project_text(model, store)
    tproj = text["tproj"]
    if cls_norm is None:
        cls_norm = store.cls_raw if store.cls_normalized else ops.l2_normalize(store.cls_raw, 1e-5)     # :280
    feats = dict(vproj=video["vproj"], vid_base=video["vid_base"], tproj=tproj, cls_norm=cls_norm)
    if "l0_vid" in video:
        feats["l0"] = dict(qkv_vid=video["l0_vid"], qkv_txt=text["l0_txt"] if "l0_txt" in text else model.layer0_rows(tproj),
                           max_v_l=store.opt.max_v_l)
        if "txt_pos" in text:
            feats["l0"]["txt_pos"], feats["l0"]["txt_pos_qk"] = text["txt_pos"], text["txt_pos_qk"]
    return feats


@torch.no_grad()
def run_windows(model, store: FeatureStore, opt, wt, feats=None, chunk=None):
    """CONE.forward + forward_clip_matching + row composition for every window of ``wt``.
    Returns rows (Nw, Nq, 4) fp32 [st, ed, proposal, matching] (sorted per window unless
    --no_sort_results) and the raw model outputs."""
    feats = feats or project_features(model, store)
    nw = wt["vid_row0"].shape[0]
    chunk = chunk or int(getattr(opt, "window_batch", 32768))
    # the longest query of the SPLIT, not of this chunk / shard view: the library picks kernel forms from Lv_max + Lq_max
    # (e.g. the rows-once cross-attention up to 128 tokens), and a window's bits must not depend on the chunk it rides in
    Lq_max = int(getattr(store, "max_tok_len", 0)) or max(store.tok_len)
    base = int(feats.get("vid_base", 0))
    outs = {k: [] for k in ("pred_logits", "pred_spans", "matching", "rows")}
    for c0 in range(0, nw, chunk):
        sl = slice(c0, min(c0 + chunk, nw))
        g = lambda k: wt[k][sl].contiguous()
        prow0 = g("vid_row0") - base if base else g("vid_row0")
        out = model.forward_packed(feats["vproj"], prow0, g("vid_len"), feats["tproj"], g("txt_row0"),
                                   g("txt_len"), opt.max_v_l, Lq_max, l0=feats.get("l0"),
                                   saliency=bool(getattr(opt, "need_saliency", False)),
                                   aux=bool(getattr(opt, "need_aux", False)))
        match = model.clip_matching_gathered(feats["cls_norm"], g("cls_row"), store.vid_raw, g("vid_row0"),
                                             g("vid_len"), g("pad_len"), out["pred_spans"])
        rows = ops.compose_rows(out["pred_logits"], out["pred_spans"], match, g("vid_len"), g("video_start"),
                                opt.clip_length, not opt.no_sort_results)
        outs["pred_logits"].append(out["pred_logits"]); outs["pred_spans"].append(out["pred_spans"])
        outs["matching"].append(match); outs["rows"].append(rows)
        if "saliency_scores" in out:
            outs.setdefault("saliency_scores", []).append(out["saliency_scores"])
        if "aux_outputs" in out:        # (n_dec - 1, B, Nq, 2) each, stacked over the intermediate decoder layers
            outs.setdefault("aux_logits", []).append(torch.stack([a["pred_logits"] for a in out["aux_outputs"]], 1))
            outs.setdefault("aux_spans", []).append(torch.stack([a["pred_spans"] for a in out["aux_outputs"]], 1))
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in outs.items()}      # one chunk: no copy


def compute_mr_results(model, store: FeatureStore, opt, win_idx=None):
    """cone/inference.py:30-100: the window-level submission list (python floats rounded to 4 dp)."""
    if win_idx is None:
        win_idx = prefilter(model, store, opt)
    if getattr(opt, "debug", False) and len(store.ann) > opt.eval_bsz:     # :93-94: stop after the first batch
        store, win_idx = FeatureStore.subset(store, 0, opt.eval_bsz), win_idx[:opt.eval_bsz].contiguous()
    wt = window_table(store, opt, win_idx)
    res = run_windows(model, store, opt, wt)
    rows = res["rows"].cpu().tolist()
    q_of = wt["q_of"].cpu().tolist()
    mr_res = []
    for w, r in enumerate(rows):
        meta = store.ann[q_of[w]]
        mr_res.append(dict(query_id=meta["query_id"], query=meta["query"], video_id=meta["video_id"],
                           clip_id=meta["clip_id"],
                           pred_relevant_windows=[[float(f"{e:.4f}") for e in row] for row in r]))
    return mr_res, {}


# ------------------------------------------------------------------------------------ stage C
def _rows_to_lists(rows, n):
    """(3, nq, max_after, 5) fp64 + counts -> [fused, proposal, matching][query] -> list of rows."""
    A = rows.shape[2]
    if rows.shape[1] == 0:
        return [[], [], []]
    n = n.cpu()
    rows = rows.cpu().tolist()
    if int(n.min()) == A:                   # every query kept max_after rows (the common case): nothing to trim
        return rows
    n = n.tolist()
    return [[r[:k] for r, k in zip(rows[t], n[t])] for t in range(3)]


def fuse_and_nms(cand, n_valid, opt):
    rows, n, _ = ops.fuse_nms(cand, n_valid, opt.nms_thd, opt.max_before_nms, opt.max_after_nms)
    return _rows_to_lists(rows, n)


def _group_submission(submission, opt):
    qid2 = OrderedDict()
    for item in submission:
        qid = item["query_id"]
        if qid not in qid2:
            if opt.dset_name == "ego4d":
                parts = qid.split("_")
                assert len(parts) == 2                                          # cone/inference.py:135-136
                qid2[qid] = {"query_idx": int(parts[1]), "annotation_uid": parts[0], "predicted_times": [],
                             "clip_uid": item["clip_id"]}
            else:
                qid2[qid] = {"query_id": qid, "predicted_times": [], "video_id": item["video_id"]}
        qid2[qid]["predicted_times"].extend(item["pred_relevant_windows"])
    return list(qid2.values())


def _postprocess(submission, opt):
    """cone/inference.py:130-202 on a window-level submission list (python rows)."""
    results = _group_submission(submission, opt)
    dev = torch.device("cuda", torch.cuda.current_device())
    n_max = max(1, max(len(r["predicted_times"]) for r in results))
    cand = torch.zeros(len(results), n_max, 4, dtype=torch.float64)
    nv = torch.zeros(len(results), dtype=torch.int32)
    for i, r in enumerate(results):
        pt = r["predicted_times"]
        nv[i] = len(pt)
        if pt:
            cand[i, :len(pt)] = torch.tensor(pt, dtype=torch.float64)
    lists = fuse_and_nms(cand.to(dev), nv.to(dev), opt)
    outs = []
    for t in range(3):
        lst = []
        for i, r in enumerate(results):
            o = r.copy()
            o["predicted_times"] = lists[t][i]
            lst.append(o)
        outs.append(lst)
    return tuple(outs)


def postprocessing_format_ego4d(submission, opt):
    assert opt.dset_name == "ego4d"
    return _postprocess(submission, opt)


def postprocessing_format_mad(submission, opt):
    assert opt.dset_name == "mad"
    return _postprocess(submission, opt)


def _save_jsonl(data, path):
    with open(path, "w") as f:
        f.write("\n".join(json.dumps(e) for e in data))


def write_submissions(opt, fusion, proposal, matching, save_submission_filename):
    """cone/inference.py:320-331, 386-417."""
    path = os.path.join(opt.results_dir, save_submission_filename)
    paths = [path]
    if opt.dset_name == "mad":
        _save_jsonl(fusion, path)
        if opt.save_all or opt.eval_modality != "both":
            pp, mp = path.replace("preds", "proposal_preds"), path.replace("preds", "matching_preds")
            _save_jsonl(proposal, pp); _save_jsonl(matching, mp)
            paths += [pp, mp]
    else:
        def dump(res, p):
            with open(p, "w") as f:
                json.dump({"version": "1.0", "challenge": "ego4d_nlq_challenge", "results": res}, f)
        dump(fusion, path)
        if opt.save_all or opt.eval_modality != "both":
            pp, mp = path.replace("preds", "proposal_preds"), path.replace("preds", "matching_preds")
            dump(proposal, pp); dump(matching, mp)
            paths += [pp, mp]
    return paths


def candidate_lists(rows, store: FeatureStore, opt, K: int):
    """Per-window rows (Nw, Nq, 4) -> the per-query candidate lists in (window rank, slot) order -- the order
    cone/inference.py:141-149 extends ``predicted_times`` in.  A query's windows are adjacent rows of the list, in rank
    order, so the rows already ARE the lists: returns ``(cand (Nw * Nq, 4) view, cand_off (nq) int64, n_valid (nq) int32,
    n_max)`` for ``ops.fuse_nms(cand, n_valid, ..., cand_off=cand_off, n_max=n_max)`` -- no scatter, no padded copy, for any
    mix of video lengths (offsets and counts are host metadata, uploaded once per store)."""
    sel = selection(store, opt, K)
    Nq = int(rows.shape[1])
    if rows.shape[0] != sel.n_rows:
        raise ValueError(f"{rows.shape[0]} window rows for a selection of {sel.n_rows}")
    cand_off, n_valid = sel.candidates(Nq, rows.device)
    return rows.reshape(-1, 4), cand_off, n_valid, sel.K * Nq


@torch.no_grad()
def device_pipeline(model, store: FeatureStore, opt, win_idx=None, batch_pad=None, video=None):
    """Stages A->C on the device only: returns the kept rows per query as tensors
    (rows (3, nq, max_after, 5) fp64, n (3, nq) int32) plus the intermediate tables.  ``win_idx`` /
    ``batch_pad`` / ``video``: results of prefilter / reference_batch_pad / project_video on the split ``store``
    was cut from (a view shares them instead of recomputing them over the whole arena).  Nothing in here synchronises
    with the device or reads a result back, whatever the video lengths (``Selection``).

    (Measured and not kept, round 5: forking the three independent fronts of a small split's step -- pre-filter, clip
    projections, text projections -- onto side streams, i.e. parallel branches of its hipGraph: 0.589 against 0.591 ms per
    single-query replay -- the runtime replays the branches one after the other.)"""
    keep = {}
    if win_idx is None:
        win_idx = prefilter(model, store, opt, keep=keep)
    wt = window_table(store, opt, win_idx, batch_pad)
    res = run_windows(model, store, opt, wt, project_features(model, store, video, keep.get("cls_norm")))
    rows = res["rows"]
    cand, cand_off, n_valid, n_max = candidate_lists(rows, store, opt, win_idx.shape[1])
    out_rows, out_n, out_idx = ops.fuse_nms(cand, n_valid, opt.nms_thd, opt.max_before_nms, opt.max_after_nms,
                                            cand_off=cand_off, n_max=n_max)
    return dict(rows=out_rows, n=out_n, idx=out_idx, win_idx=win_idx, windows=wt, cand=cand, cand_off=cand_off,
                n_valid=n_valid, n_windows=int(rows.shape[0]), outputs=res)


@contextlib.contextmanager
def _gc_paused():
    """The submission lists are ~35 small containers per query; building them with the cyclic collector armed
    triggers a full collection (tens of ms over the whole heap) every few splits.  None of it can be garbage."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def format_results(ann, opt, rows, n, skeletons=None):
    """A16 (cone/inference.py:130-202): the three submission lists from the kept rows."""
    with _gc_paused():
        return _format_results(ann, opt, rows, n, skeletons)


def _ego4d_keys(ann):
    """(query_idx, annotation_uid, clip_uid) of every row (cone/inference.py:133-140).  They depend on the annotation rows
    only: a store's own list (``AnnList``) keeps them (checked against its length and end points, in case it was edited)."""
    stamp = (len(ann), ann[0]["query_id"], ann[-1]["query_id"]) if len(ann) else (0, None, None)
    hit = getattr(ann, "_ego4d_keys", None) if isinstance(ann, AnnList) else None
    if hit is not None and hit[0] == stamp:
        return hit[1]
    keys = []
    for meta in ann:
        parts = meta["query_id"].split("_")
        assert len(parts) == 2
        keys.append((int(parts[1]), parts[0], meta["clip_id"]))
    if isinstance(ann, AnnList):
        ann._ego4d_keys = (stamp, keys)
    return keys


def result_skeletons(ann, opt):
    """The three submission lists with every field but ``predicted_times`` -- they depend on the annotations only, so the
    host builds them while the GPU is still working on the split (predict_split)."""
    with _gc_paused():
        if opt.dset_name == "ego4d":
            keys = _ego4d_keys(ann)
            return tuple([{"query_idx": k[0], "annotation_uid": k[1], "predicted_times": None, "clip_uid": k[2]}
                          for k in keys] for _ in range(3))
        return tuple([{"query_id": m["query_id"], "predicted_times": None, "video_id": m["video_id"]} for m in ann]
                     for _ in range(3))


_PT_KEY = "predicted_times"
_PYLISTS_WARNED = False


def _format_results(ann, opt, rows, n, skeletons=None):
    """rows (3, nq, max_after, 5) fp64, n (3, nq) int32 (device or host) -> the three lists.  One D2H copy each, then ONE
    pass of the C helper per list (cone_amd/csrc/pylists.c: floats, rows, per-query lists and the dict slot in one loop)
    instead of tensor.tolist() + a Python loop of 3 x nq assignments."""
    from . import _lib
    out = skeletons if skeletons is not None else result_skeletons(ann, opt)
    nq, A = int(rows.shape[1]), int(rows.shape[2])
    if nq == 0:
        return out
    rows_h = np.ascontiguousarray(rows.detach().cpu().numpy(), dtype=np.float64)
    n_h = np.ascontiguousarray(n.detach().cpu().numpy(), dtype=np.int32)
    try:
        fill = _lib.pylists().cone_fill_predicted_times
    except (_lib.ConeHipError, OSError, AttributeError) as e:
        # host-side helper only (plain C against Python.h, no GPU code): without it the same lists come from tolist() +
        # a Python loop -- slower, same values; the HIP library itself has no fallback anywhere
        global _PYLISTS_WARNED
        if not _PYLISTS_WARNED:
            _PYLISTS_WARNED = True
            warnings.warn(f"_cone_pylists.so unavailable ({e}); building the submission lists in Python")
        lists = _rows_to_lists(torch.from_numpy(rows_h), torch.from_numpy(n_h))
        for t in range(3):
            for item, pt in zip(out[t], lists[t]):
                item[_PT_KEY] = pt
        return out
    for t in range(3):
        fill(out[t], rows_h[t].ctypes.data, n_h[t].ctypes.data, nq, A, _PT_KEY)
    return out


def query_chunks(nq: int, opt):
    """Query ranges for the software pipeline of predict_split, cut at multiples of eval_bsz (every reference batch
    stays inside one chunk).  ``opt.pipeline_chunks = n``: n near-equal chunks; ``opt.pipeline_tail = f``: one head
    chunk and a tail of fraction f of the queries (the host builds the head's submission rows while the GPU runs
    the tail; 0: one chunk).  Default (neither set): a tail of 1/16 for splits of at least 32 reference batches,
    one chunk otherwise -- and one chunk whenever the caller wants the per-window outputs of the whole split
    (saliency / aux heads) or the hipGraph replay.  Measured on MI355X (config 2, 1 000 queries = 32 batches,
    20 000 windows, round 3, one box, 10 steps each): one chunk 54.3 / 54.2 / 54.9 ms per step, tail 1/16 53.9 /
    53.95 / 54.0, tail 1/32 55.0, tail 3/32 54.4 -- the tail's replay of the launch sequence costs ~0.6 ms of GPU time
    (small-M kernel forms) and hides ~1.1 ms of list building.  (Round 2, before the small-M forms: no gain.)"""
    nb = -(-nq // opt.eval_bsz)
    want = getattr(opt, "pipeline_chunks", None)
    if want is None:
        tail = getattr(opt, "pipeline_tail", None)
        if tail is None:
            whole = (getattr(opt, "hip_graph", False) or getattr(opt, "need_saliency", False)
                     or getattr(opt, "need_aux", False))
            tail = 1.0 / 16.0 if nb >= 32 and not whole else 0.0
        tail = float(tail)
        if tail <= 0 or nb < 8:
            return [(0, nq)]
        tb = max(1, int(round(nb * tail)))
        cut = (nb - tb) * opt.eval_bsz
        return [(0, cut), (cut, nq)]
    want = max(1, min(int(want), nb))
    cuts = sorted({min(nq, (-(-nb * i // want)) * opt.eval_bsz) for i in range(want + 1)})
    return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]


def _graph_key(model, opt):
    return (id(model), model._handle.value if getattr(model, "_handle", None) is not None else None, opt.topk_window,
            opt.eval_bsz, opt.max_v_l, opt.nms_thd, opt.max_before_nms, opt.max_after_nms, bool(opt.no_sort_results),
            bool(getattr(opt, "need_saliency", False)), bool(getattr(opt, "need_aux", False)),
            int(getattr(opt, "window_batch", 32768)), float(opt.clip_length))


@torch.no_grad()
def _graph_replay(model, store: FeatureStore, opt):
    """Stages A->C of ``store`` as ONE hipGraph launch (``opt.hip_graph``): the latency path for a split that is
    evaluated again and again at the same shapes -- a resident video queried repeatedly (BASELINE configs[0]: 1 query, 20
    windows: ~110 launches of a few microseconds each, bound by the host's launch rate when issued one by one).  The
    device pipeline never synchronises and takes every size from host metadata, so it captures as it is: after one
    eager warm-up (allocations, position tables, LDS attributes) the same call runs under stream capture; the store's
    arenas are the graph's inputs (refill them in place -- ``vid_raw.copy_`` / ``mot_raw.copy_`` (a second visual source) / ``tok_raw.copy_`` / ``cls_raw.copy_`` --
    for new features of the same shapes), the returned tensors its outputs (valid until the next replay).
    Any mix of video lengths: the shape of the window list is host metadata (``Selection``), so there is no data-dependent
    size anywhere."""
    cache = store.__dict__.setdefault("_graphs", {})
    # the captured launches point INTO the arenas: refilling them in place keeps a capture valid, assigning new tensors to
    # vid_raw / mot_raw / tok_raw / cls_raw does not -- such a store is captured again (as FeatureStore.view rebuilds its views)
    arenas = (store.vid_raw.data_ptr(), store.tok_raw.data_ptr(), store.cls_raw.data_ptr(),
              0 if store.mot_raw is None else store.mot_raw.data_ptr())
    key = _graph_key(model, opt) + arenas
    hit = cache.get(key)
    if hit is None:
        if len(cache) > 8:
            cache.clear()
        device_pipeline(model, store, opt)              # warm-up: workspace, kernel attributes
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            dp = device_pipeline(model, store, opt)
        # everything the captured launches point at must outlive the graph: the model (weights, position tables) and the
        # workspace buffer of THIS capture (the model's grow-only workspace may be replaced by a larger one later)
        hit = cache[key] = (g, dp, model, model._ws.buf)
    hit[0].replay()
    return dict(hit[1])


class PendingSplit:
    """A split whose device work is enqueued (``predict_split_async``).  ``info`` -- the run-info dict with the device-side
    result tensors -- is valid at once (stream-ordered); ``result()`` waits for the kept rows in pinned host memory, builds
    the submission lists and returns what ``predict_split`` returns.  A caller that evaluates split after split keeps one
    in flight: the lists of split i are built while the GPU runs split i + 1.

    With ``opt.hip_graph`` the tensors of ``info`` are the graph's STATIC outputs: the next replay of the same graph overwrites
    them, so they are valid only until the next ``predict_split_async`` on that store is enqueued (read them stream-ordered
    before it, or ``clone()`` them).  What ``result()`` returns is always safe: the kept rows were copied to this handle's own
    pinned buffers ahead of the next replay."""

    def __init__(self, info, finish):
        self.info, self._finish, self._res = info, finish, None

    def result(self):
        if self._finish is not None:
            self._res, self._finish = self._finish(), None
        return self._res


def _to_pinned(t):
    return torch.empty(t.shape, dtype=t.dtype, pin_memory=True).copy_(t, non_blocking=True)


def _kept_to_pinned(rows, n):
    """[rows, n] in pinned host memory, enqueued: one transfer when they share a buffer (ops.fuse_nms lays them out back to
    back), else one each."""
    buf = getattr(rows, "kept_buf", None)
    if buf is None or buf.numel() != rows.numel() * 8 + n.numel() * 4:
        return [_to_pinned(rows), _to_pinned(n)]
    host = _to_pinned(buf)
    R = rows.numel() * 8
    return [host[:R].view(torch.float64).view(rows.shape), host[R:].view(torch.int32).view(n.shape)]


def predict_split_async(model, store: FeatureStore, opt) -> PendingSplit:
    """``predict_split`` in two halves: everything the GPU does is enqueued here (nothing waits for it), the host half --
    waiting for the kept rows and building the submission lists -- runs in ``PendingSplit.result()``."""
    t0 = time.time()
    if getattr(opt, "debug", False) and len(store.ann) > opt.eval_bsz:
        win_idx = prefilter(model, store, opt)
        store = FeatureStore.subset(store, 0, opt.eval_bsz)
        dp = device_pipeline(model, store, opt, win_idx=win_idx[:opt.eval_bsz].contiguous())
        dp["ann"] = store.ann
        pend = [(store, dp, None)]
        chunks = [(0, len(store.ann))]
        info = dp
    else:
        chunks = query_chunks(len(store.ann), opt)
        if len(chunks) == 1:
            if getattr(opt, "hip_graph", False):
                dp = _graph_replay(model, store, opt)       # the whole launch sequence as ONE hipGraph launch
            else:
                dp = device_pipeline(model, store, opt)     # enqueued; nothing in it waits for the GPU
            pend = [(store, dp, None)]
            info = dp
        else:
            # clip-side work once for the split, shared by the chunks (they are views of one arena)
            store.index_tensors()
            win_idx = prefilter(model, store, opt)
            video = project_video(model, store)
            pend = []
            for lo, hi in chunks:
                sub = store.view(lo, hi)    # cached: the view's static index tables are uploaded once, not once per step
                pend.append((sub, device_pipeline(model, sub, opt, win_idx=win_idx[lo:hi], video=video), None))
            info = None
    # kept rows to pinned host memory behind an event per chunk (a graph replay's outputs are only valid until the next
    # replay: the copy is stream-ordered ahead of it)
    pend = [(sub, dp, (_kept_to_pinned(dp["rows"], dp["n"]), torch.cuda.Event())) for sub, dp, _ in pend]
    for _, _, (_, ev) in pend:
        ev.record()
    if info is None:
        # the split-level tensors of the info dict are enqueued right behind the last chunk, while the GPU is still busy with
        # it (built after the last chunk's rows were waited for, their four small launches + allocations sat in the GPU-idle
        # gap at the end of every step)
        info = dict(rows=torch.cat([p[1]["rows"] for p in pend], dim=1), n=torch.cat([p[1]["n"] for p in pend], dim=1),
                    n_windows=sum(p[1]["n_windows"] for p in pend), chunks=chunks, win_idx=win_idx,
                    windows={k: torch.cat([p[1]["windows"][k] for p in pend]) for k in ("vid_len", "txt_len")})

    def finish():
        outs = None
        for sub, dp, (host, ev) in pend:
            skel = result_skeletons(sub.ann, opt)           # host work that needs no result: under the GPU's time
            ev.synchronize()
            part = format_results(sub.ann, opt, host[0], host[1], skel)
            if outs is None:
                outs = part if len(pend) == 1 else tuple(list(x) for x in part)
            else:
                for dst, src in zip(outs, part):
                    dst.extend(src)
        info["model_seconds"] = time.time() - t0
        return outs, info

    return PendingSplit(info, finish)


def predict_split(model, store: FeatureStore, opt):
    """Stages A->C; returns the three submission lists (fused, proposal, matching) and run info.

    The queries run as a short software pipeline: all chunks are enqueued on the stream back to back (nothing in
    device_pipeline synchronises), their kept rows are copied to pinned host memory behind an event each, and the host
    builds the submission rows of chunk i while the GPU is still working on chunk i+1.  Results are identical to one big
    batch (rows of a GEMM are independent).  (``predict_split_async`` hands the host half back to the caller.)

    ``--debug`` keeps the reference's meaning (cone/inference.py:93-94): the pre-filter runs over the whole split,
    the window model stops after the first batch of ``eval_bsz`` queries, and only those queries are written."""
    return predict_split_async(model, store, opt).result()


EGO4D_VAL_GT = "data/ego4d_ori_data/nlq_val.json"      # hard-coded by the reference, cone/inference.py:420


@torch.no_grad()
def evaluate_split(model, store: FeatureStore, opt, dp, ground_truth=None, epoch_i=None):
    """The metric tables of cone/inference.py:332-381 (MAD) / :419-474 (Ego4D) from the device-resident kept
    rows ``dp`` of device_pipeline -- no JSON round trip, no python loop over queries (cone_amd.metrics).
    Returns (results, mIoU, results_proposal, mIoU_proposal, results_matching, mIoU_matching, score strings);
    like the reference's display functions, the returned tables are in percent."""
    from . import metrics as M
    dev = store.device
    rows, n = dp["rows"], dp["n"]
    if opt.dset_name == "mad":
        gt = torch.from_numpy(M.jsonl_targets(store.ann)).to(dev)
        thr, topk, wk = [0.1, 0.3, 0.5], [1, 5, 10, 50, 100], [1, 5, 10, 30, 50, 100, 200]
    else:
        if ground_truth is None:
            with open(getattr(opt, "ego4d_gt_path", None) or EGO4D_VAL_GT) as f:
                ground_truth = json.load(f)
        gt = torch.from_numpy(M.ego4d_targets(store.ann, ground_truth)).to(dev)
        thr, topk, wk = [0.3, 0.5], [1, 5, 10, 50, 100], [1, 5, 10, 30, 50]
    max_nw = max(-(-c // int(opt.max_v_l / 2)) + 1 for c in store.ctx_l)
    deep = prefilter(model, store, opt, k=min(max(wk), max_nw))          # the reference ranks every window
    wres = M.windows_selection(deep, torch.from_numpy(M.jsonl_targets(store.ann)).to(dev), wk, opt.clip_length,
                               opt.max_v_l)
    strs = [M.display_window_results(wres, wk, title=f"Window Pre-filtering Epoch {epoch_i}")]
    out = []
    for t, name in enumerate(("Fusion", "Proposal", "Matching")):
        if opt.dset_name == "mad":
            res, miou = M.evaluate_nlq_performance_mad(rows[t], n[t], gt, thr, topk), None
            strs.append(M.display_results_mad(res, thr, topk, title=f"{name} Epoch {epoch_i}"))
        else:
            res, miou = M.evaluate_nlq_performance_ego4d(rows[t], n[t], gt, thr, topk)
            strs.append(M.display_results_ego4d(res, miou, thr, topk, title=f"{name} Epoch {epoch_i}"))
        out += [res * 100, miou]
    return (*out, strs)


def eval_epoch(model, *args, epoch_i=None, criterion=None, tb_writer=None, ground_truth=None):
    """cone/inference.py:227-499: writes the prediction files; on the splits the reference scores (Ego4D val,
    MAD val / test) also the metric tables + ``.txt`` file.  Returns the reference's tuple
    ``(results, mIoU, [window, fusion, proposal, matching score strings], latest_file_paths)`` -- ``(None, None,
    [], paths)`` when there is nothing to score (the reference ``exit(0)``s there, :476-477).

    Two call shapes:
      ``eval_epoch(model, store, opt, save_submission_filename, ...)``                      -- a FeatureStore;
      ``eval_epoch(model, eval_inter_window_dataset, eval_intra_window_dataset, opt, save_submission_filename,
      epoch_i, criterion, tb_writer)``  -- the reference's own signature (:227-228), so that its second caller,
      the training loop (cone/train.py:164-168), can switch imports: the two dataset objects are read once into a
      device-resident store (``FeatureStore.from_datasets``, cached on the intra-window dataset object).
    ``criterion`` / ``tb_writer`` are accepted and ignored: the reference only feeds them eval-loss meters, which
    need the training criterion (out of scope here)."""
    if isinstance(args[0], FeatureStore):
        store, opt, save_submission_filename = args[0], args[1], args[2]
        rest = args[3:]
    else:
        inter_ds, intra_ds, opt, save_submission_filename = args[0], args[1], args[2], args[3]
        rest = args[4:]
        store = getattr(intra_ds, "_cone_amd_store", None)
        if store is None:
            store = FeatureStore.from_datasets(opt, inter_ds, intra_ds)
            try:
                intra_ds._cone_amd_store = store
            except AttributeError:
                pass
        store.opt = opt
    if len(rest) > 0 and epoch_i is None:
        epoch_i = rest[0]
    logger.info("Generate submissions")
    (fusion, proposal, matching), info = predict_split(model, store, opt)
    print("total model running time: ", info["model_seconds"])
    paths = write_submissions(opt, fusion, proposal, matching, save_submission_filename)
    scored = opt.eval_split_name == "val" or (opt.dset_name == "mad" and opt.eval_split_name == "test")
    if not scored:
        print("end of inference on test split")
        return None, None, [], paths
    if "ann" in info:           # --debug: only the first batch of queries went through the model
        store = FeatureStore.subset(store, 0, len(info["ann"]))
    res, miou, res_p, miou_p, res_m, miou_m, strs = evaluate_split(model, store, opt, info, ground_truth, epoch_i)
    for s_ in strs:
        print(s_, flush=True)
    sub_path = os.path.join(opt.results_dir, save_submission_filename)
    txt = sub_path.replace(".jsonl" if opt.dset_name == "mad" else ".json", ".txt")
    with open(txt, mode="w", encoding="utf-8") as f:
        for s_ in (strs[1:] if opt.dset_name == "mad" else strs):       # MAD omits the window table, :376-379
            f.write(s_)
    latest = [txt]
    if opt.eval_modality == "both":
        out_res, out_miou = res, miou
        latest.append(sub_path)
    elif opt.eval_modality == "proposal":
        out_res, out_miou = res_p, miou_p
        latest.append(sub_path.replace("preds", "proposal_preds"))
    else:
        # the reference tests for "clip", which the CLI cannot produce (H10): "matching" raises there
        raise UnboundLocalError("eval_modality %r is not scored by the reference" % (opt.eval_modality,))
    return out_res, (out_miou if opt.dset_name == "ego4d" else None), strs, latest


def setup_model(opt):
    """cone/inference.py:502-537 (inference part): build, load ``ckpt["model"]``."""
    model, criterion = build_model(opt)
    if opt.resume is None:
        raise ValueError("--resume <ckpt> is required")
    ckpt = torch.load(opt.resume, map_location="cpu", weights_only=False)
    model.load_state_dict(ckpt["model"])
    if getattr(opt, "split_bf16", False):
        model.set_option("split_bf16", 1)
    logger.info(f"Loaded model saved at epoch {ckpt.get('epoch')} from checkpoint: {opt.resume}")
    return model, criterion, None, None


def start_inference(argv=None):
    """cone/inference.py:540-607."""
    logging.basicConfig(format="%(asctime)s.%(msecs)03d:%(levelname)s:%(name)s - %(message)s",
                        datefmt="%Y-%m-%d %H:%M:%S", level=logging.INFO)
    opt = parse_test_options(argv)
    assert opt.eval_path is not None
    packed = getattr(opt, "packed_features", None)
    store = FeatureStore.from_packed(opt, packed) if packed else FeatureStore.from_lmdb(opt)
    model, _, _, _ = setup_model(opt)
    ext = "jsonl" if opt.dset_name == "mad" else "json"
    fn = f"inference_{opt.dset_name}_{opt.eval_split_name}_{opt.eval_id}_preds.{ext}"
    logger.info("Starting inference...")
    return eval_epoch(model, store, opt, fn)


if __name__ == "__main__":
    start_inference()
