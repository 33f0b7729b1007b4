"""Single-video, single-query localizer: the counterpart of ``run_on_video/cone_localizator.py``
(``CONELocalizator.predict_moment``), on the same HIP kernels as the dataset path.

Behaviour kept from the reference (it differs from ``cone/inference.py`` on purpose):
``F.normalize`` on clips and tokens, adapted features NOT re-normalised, raw cls vector for the window
ranking, every window treated as padded to (max_v_l, max_q_l) (so the proposal mean divides by the padded
slice length), spans scaled by ``max_v_l`` instead of the window's length, rows not sorted per window,
NMS on the fused score only over the first 100 candidates with threshold 0.5, 5 kept.

Deviation: for a video with fewer than ``topk_window`` windows the reference feeds all-padding windows to
the transformer (NaN outputs); here only the existing windows are scored.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops
from .config import MODEL_DEFAULTS
from .model import build_model

# run_on_video/cone_localizator.py:12-37
LOCALIZER_OPT = dict(MODEL_DEFAULTS, v_motion_feat_dim=256, v_appear_feat_dim=256, t_feat_dim=768, max_q_l=20,
                     max_v_l=90, topk_window=20, clip_length=0.5333, dset_name="ego4d")


class CONELocalizator:
    def __init__(self, load_checkpoint_path=None, device="cuda", state_dict=None, hip_graph=False, **overrides):
        """``hip_graph`` (cone_amd extension, opt-in): a call at a (video length, query length) shape seen before replays the
        whole launch sequence as ONE hipGraph launch -- the inputs are copied into the capture's own buffers first, the kept
        moments are read back as usual; same kernels, same order, same bits as the eager call."""
        self.hip_graph = bool(hip_graph)
        self.args = SimpleNamespace(**dict(LOCALIZER_OPT, **overrides))
        self.localizator, _ = build_model(self.args)
        if state_dict is None:
            if load_checkpoint_path is None:
                raise ValueError("either load_checkpoint_path or state_dict is required")
            state_dict = torch.load(load_checkpoint_path, map_location="cpu", weights_only=False)["model"]
        self.localizator.load_state_dict(state_dict)
        self.device = self.localizator.device
        self.slide_window_size = int(self.args.max_v_l / 2)
        self.max_v_l = self.args.max_v_l

    @torch.no_grad()
    def compute_window_ranklist(self, video_feats, text_cls_feat):
        """run_on_video/cone_localizator.py:83-100 (stable tie order)."""
        _, ws = ops.prefilter_scores(video_feats.contiguous(), text_cls_feat.reshape(1, -1).contiguous(), self.max_v_l,
                                      frame_scores=False)
        idx, _ = ops.topk_windows(ws, ws.shape[1])
        return idx[0].tolist()

    def _const(self, ctx_l: int, n_tok: int):
        """Index metadata of one (video length, query length) shape on the device, built once: the latency path uploads
        nothing per call (a pageable H2D copy waits for the stream to drain)."""
        cache = self.__dict__.setdefault("_consts", {})
        c = cache.get((ctx_l, n_tok))
        if c is None:
            if len(cache) > 64:
                cache.clear()
            dev, a = self.device, self.args
            K = min(a.topk_window, ops.num_windows(ctx_l, a.max_v_l))
            t = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
            c = cache[(ctx_l, n_tok)] = dict(
                K=K, q_ctx_l=t([ctx_l]), q_vid_off=t([0]), tok_off=t([0, n_tok]), tok_len=t([n_tok]),
                batch_pad=t([a.max_v_l]), full_w=torch.full((K,), a.max_v_l, dtype=torch.int32, device=dev),
                tok_idx=torch.arange(n_tok, dtype=torch.int32, device=dev),
                n_valid=t([K * self.localizator.num_queries]))
        return c

    @torch.no_grad()
    def _enqueue(self, video_feats, text_token_feats, text_cls_feat, c):
        """The device half of ``predict_moment``: everything up to the kept rows, no host round trip."""
        a, m = self.args, self.localizator
        K, W = c["K"], a.max_v_l
        vid = ops.l2_normalize(video_feats, 1e-5, clamp=True)                                  # :129
        tok = ops.l2_normalize(text_token_feats, 1e-5, clamp=True)                             # :133
        cls = text_cls_feat.reshape(1, -1)
        adapted = m.adapter_norm(vid, renorm=False)                                           # :135-138
        _, ws = ops.prefilter_scores(adapted, cls, W, frame_scores=False)                     # :83-100 (stable tie order)
        widx, _ = ops.topk_windows(ws, K)
        # every window counts as padded to max_v_l (the reference pads each to (max_v_l, max_q_l), :150-170)
        wt = ops.window_table_rows(widx, c["q_ctx_l"], c["q_vid_off"], c["tok_off"], c["tok_len"], 0, 1, W,
                                   batch_pad=c["batch_pad"], n_batches=1)
        vproj, tproj = m.project(0, vid), m.project(1, tok)
        out = m.forward_packed(vproj, wt["vid_row0"], wt["vid_len"], tproj, wt["txt_row0"], wt["txt_len"], W, a.max_q_l,
                               l0=m.layer0_cache(vproj, tproj, W, tok_index=c["tok_idx"]), saliency=False)
        match = m.clip_matching_gathered(cls, wt["cls_row"], vid, wt["vid_row0"], wt["vid_len"], wt["pad_len"],
                                         out["pred_spans"])
        rows = ops.compose_rows(out["pred_logits"], out["pred_spans"], match, c["full_w"], wt["video_start"], a.clip_length,
                                sort=False)                                                   # :191: scaled by max_v_l, no sort
        kept, n, _ = ops.fuse_nms(rows.reshape(1, K * m.num_queries, 4), c["n_valid"], 0.5, 100, 5)      # :200-219
        return kept, n

    def _replay(self, vid, tok, cls, c):
        """``hip_graph``: the launch sequence of this (video length, query length) shape, captured once behind an eager warm-up
        and replayed on the capture's own input buffers (the caller's tensors are copied in, stream-ordered)."""
        g = c.get("graph")
        if g is None:
            bufs = (torch.empty_like(vid), torch.empty_like(tok), torch.empty_like(cls))
            for b, x in zip(bufs, (vid, tok, cls)):
                b.copy_(x)
            self._enqueue(*bufs, c)                     # warm-up: workspace, kernel attributes
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._enqueue(*bufs, c)
            # what the captured launches point at must outlive the graph: the capture's workspace (the model's grow-only
            # scratch may be replaced by a larger one later)
            g = c["graph"] = (graph, bufs, out, self.localizator._ws.buf)
        for b, x in zip(g[1], (vid, tok, cls)):
            b.copy_(x, non_blocking=True)
        g[0].replay()
        return g[2]

    @torch.no_grad()
    def predict_moment(self, video_feats, text_feats):
        """run_on_video/cone_localizator.py:121-221 on the eval driver's kernels, enqueued without a host round trip until
        the kept moments are read back: window ranks stay on the device (the window table kernel turns them into row
        ranges), first-layer q|k|v once per clip / token, position tables, fused layer tails."""
        a, dev = self.args, self.device
        text_token_feats, text_cls_feat = text_feats
        if text_token_feats.shape[0] > a.max_q_l:
            raise ValueError(f"query has {text_token_feats.shape[0]} tokens > max_q_l={a.max_q_l}")
        vid = video_feats.to(dev, torch.float32).contiguous()
        tok = text_token_feats.to(dev, torch.float32).contiguous()
        cls = text_cls_feat.to(dev, torch.float32).reshape(-1).contiguous()
        c = self._const(int(vid.shape[0]), int(tok.shape[0]))
        kept, n = self._replay(vid, tok, cls, c) if self.hip_graph else self._enqueue(vid, tok, cls, c)
        kept, n = kept[0, 0].cpu(), int(n[0, 0])        # the call's one read-back
        return [[r[0], r[1], r[4]] for r in kept[:n].tolist()]
