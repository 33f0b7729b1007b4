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
    def __init__(self, load_checkpoint_path=None, device="cuda", state_dict=None, **overrides):
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

    @torch.no_grad()
    def predict_moment(self, video_feats, text_feats):
        a, m, dev = self.args, self.localizator, self.device
        text_token_feats, text_cls_feat = text_feats
        if text_token_feats.shape[0] > a.max_q_l:
            raise ValueError(f"query has {text_token_feats.shape[0]} tokens > max_q_l={a.max_q_l}")
        vid = ops.l2_normalize(video_feats.to(dev, torch.float32), 1e-5, clamp=True)          # :129
        tok = ops.l2_normalize(text_token_feats.to(dev, torch.float32), 1e-5, clamp=True)     # :133
        cls = text_cls_feat.to(dev, torch.float32).reshape(1, -1).contiguous()
        adapted = m.adapter_norm(vid, renorm=False)                                           # :135-138
        ranks = self.compute_window_ranklist(adapted, cls)
        widx = torch.tensor(ranks[:a.topk_window], dtype=torch.int64, device=dev)
        K, S, W, ctx_l = widx.shape[0], self.slide_window_size, a.max_v_l, vid.shape[0]
        start = torch.clamp((widx - 1) * S, min=0)
        vlen = torch.minimum((widx - 1) * S + W, torch.tensor(ctx_l, device=dev)) - start
        i32 = lambda t: t.to(torch.int32).contiguous()
        zeros = torch.zeros(K, dtype=torch.int32, device=dev)
        full = lambda v: torch.full((K,), v, dtype=torch.int32, device=dev)
        vproj, tproj = m.project(0, vid), m.project(1, tok)
        # the eval driver's path: first-layer q|k|v once per clip / token, position tables, fused layer tails (the reference
        # replicates and re-projects per window, run_on_video/cone_localizator.py:150-182)
        out = m.forward_packed(vproj, i32(start), i32(vlen), tproj, zeros, full(tok.shape[0]), W, a.max_q_l,
                               l0=m.layer0_cache(vproj, tproj, W), saliency=False)
        match = m.clip_matching_gathered(cls, zeros, vid, i32(start), i32(vlen), full(W), out["pred_spans"])
        rows = ops.compose_rows(out["pred_logits"], out["pred_spans"], match, full(W), i32(start), a.clip_length,
                                sort=False)                                                   # :191, no sort
        cand = rows.reshape(1, K * m.num_queries, 4).contiguous()
        nv = torch.tensor([K * m.num_queries], dtype=torch.int32, device=dev)
        kept, n, _ = ops.fuse_nms(cand, nv, 0.5, 100, 5)                                       # :200-219
        return [[r[0], r[1], r[4]] for r in kept[0, 0, :int(n[0, 0])].cpu().tolist()]
