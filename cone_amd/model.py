"""Host-side mirror of the reference ``CONE`` module for inference on MI355X.

Same constructor path (``build_model(args)``), state-dict keys, call signatures, argument
meaning and output dict as ``cone/model.py:16-152`` -- but ``forward`` /
``forward_clip_matching`` enqueue the hand-written HIP kernels of libcone_hip.so on the current
torch stream instead of running torch.nn modules.  Tensors are torch CUDA(HIP) tensors; torch is
only the allocator / stream owner here.

Not mirrored (out of scope): the training-only branches (``is_groundtruth`` matching, dropout,
autograd).  ``build_model`` returns ``(model, criterion)`` with the forward-value criterion of cone_amd.criterion.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .synth import state_dict_spec

import os as _os

_CHECK_MASKS = _os.environ.get("CONE_AMD_CHECK_MASKS", "0") == "1"
_TXT_POS_PREFIX = "txt_position_embed."  # present in checkpoints, read only with --use_txt_pos


def _dim_t_table(d: int) -> torch.Tensor:
    """temperature ** (2 * (i // 2) / d), fp32, exactly as cone/position_encoding.py:66-67."""
    dim_t = torch.arange(d, dtype=torch.float32)
    return 10000 ** (2 * (dim_t // 2) / d)


class Workspace:
    """Grow-only device scratch handed to the C ABI (the library never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes * 1.05) + 256, dtype=torch.uint8, device=device)
        return self.buf


class CONE:
    """Drop-in for the inference surface of the reference ``CONE`` nn.Module."""

    def __init__(self, args):
        # --use_txt_pos (cone/config.py:115; off in every shipped configuration): text tokens carry
        # TrainablePositionalEncoding(src_txt) instead of a zero position term (cone/model.py:106); the library then runs the
        # general path (x + pos materialised per token; no layer-0 caches / position tables)
        self.use_txt_pos = bool(getattr(args, "use_txt_pos", False))
        if getattr(args, "span_loss_type", "l1") != "l1":
            raise NotImplementedError("only span_loss_type='l1' (cone/config.py:134)")
        # --pre_norm (cone/config.py:120): normalize_before in every layer + the encoder's final norm; general path as well
        self.pre_norm = bool(getattr(args, "pre_norm", False))
        # two visual sources (cone/ego4d_mad_dataloader.py:63-71): MOTION features feed the window model (input_vid_proj,
        # cone/model.py:67), APPEARANCE features the pre-filter and the proposal matching (adapter, cone/model.py:80, 186-208);
        # every shipped script points both at one LMDB, the dims may differ when they do not
        self.args = args
        self.num_queries = args.num_queries
        self.adapter_module = args.adapter_module
        self.aux_loss = getattr(args, "aux_loss", True)
        self.n_input_proj = args.n_input_proj
        self.hidden_dim = args.hidden_dim
        self.device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        self._handle = None
        self._sd = None
        self._ws = Workspace()
        self.training = False

    # ---- nn.Module look-alikes ---------------------------------------------------------------
    def eval(self):
        return self

    def to(self, device):
        return self

    def __call__(self, *a, **k):
        return self.forward(*a, **k)

    def state_dict(self):
        return OrderedDict((k, v.clone()) for k, v in self._sd.items())

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts the reference checkpoint's ``ckpt["model"]`` unchanged (cone/inference.py:525-527)."""
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.ConeHipError("cone_amd needs a GPU: there is no CPU execution path")
        dev = torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        spec = state_dict_spec(self.args)
        sd = OrderedDict()
        for k, shape in spec.items():
            if k not in state_dict:
                if k.startswith(_TXT_POS_PREFIX) and not self.use_txt_pos:
                    continue
                raise KeyError(f"missing key in state_dict: {k}")
            v = state_dict[k]
            v = torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v
            if tuple(v.shape) != tuple(shape):
                raise ValueError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(shape)}")
            sd[k] = v.detach().to(device=dev, dtype=torch.float32).contiguous()
        if strict:
            extra = [k for k in state_dict if k not in spec]
            if extra:
                raise KeyError(f"unexpected keys in state_dict: {extra[:5]}")
        self._sd = sd
        self._pos_tabs = None
        self._dim_t = _dim_t_table(self.hidden_dim).to(dev)
        a = self.args
        w = _lib.Weights()
        w.hidden_dim, w.nheads, w.dim_ff = a.hidden_dim, a.nheads, a.dim_feedforward
        w.enc_layers, w.dec_layers, w.num_queries = a.enc_layers, a.dec_layers, a.num_queries
        w.n_input_proj, w.t_dim, w.v_dim, w.v_motion_dim = a.n_input_proj, a.t_feat_dim, a.v_appear_feat_dim, a.v_motion_feat_dim
        w.has_adapter = 1 if a.adapter_module == "linear" else 0
        # ABI 8: the handle's position tables cover windows of up to max_v_l clips (build_model's own parameter, cone/model.py:
        # 468-486); an args object without it gets the 255-clip tables.  Longer windows than that still run (general path)
        w.table_max_v_l = min(int(getattr(a, "max_v_l", 0) or 0), 255)
        p = lambda k: sd[k].data_ptr()

        def lin(dst, prefix):
            dst.w, dst.b = p(prefix + ".weight"), p(prefix + ".bias")

        def ln(dst, prefix):
            dst.g, dst.b = p(prefix + ".weight"), p(prefix + ".bias")

        def mha(dst, prefix):
            dst.in_proj_w, dst.in_proj_b = p(prefix + ".in_proj_weight"), p(prefix + ".in_proj_bias")
            lin(dst.out_proj, prefix + ".out_proj")

        for i in range(a.n_input_proj):
            ln(w.vid_proj_ln[i], f"input_vid_proj.{i}.LayerNorm"); lin(w.vid_proj[i], f"input_vid_proj.{i}.net.1")
            ln(w.txt_proj_ln[i], f"input_txt_proj.{i}.LayerNorm"); lin(w.txt_proj[i], f"input_txt_proj.{i}.net.1")
        for i in range(a.enc_layers):
            pre = f"transformer.encoder.layers.{i}"
            mha(w.enc[i].self_attn, pre + ".self_attn")
            lin(w.enc[i].linear1, pre + ".linear1"); lin(w.enc[i].linear2, pre + ".linear2")
            ln(w.enc[i].norm1, pre + ".norm1"); ln(w.enc[i].norm2, pre + ".norm2")
        for i in range(a.dec_layers):
            pre = f"transformer.decoder.layers.{i}"
            mha(w.dec[i].self_attn, pre + ".self_attn"); mha(w.dec[i].cross_attn, pre + ".multihead_attn")
            lin(w.dec[i].linear1, pre + ".linear1"); lin(w.dec[i].linear2, pre + ".linear2")
            ln(w.dec[i].norm1, pre + ".norm1"); ln(w.dec[i].norm2, pre + ".norm2"); ln(w.dec[i].norm3, pre + ".norm3")
        ln(w.dec_norm, "transformer.decoder.norm")
        w.query_embed = p("query_embed.weight")
        lin(w.class_embed, "class_embed")
        for i in range(3):
            lin(w.span_embed[i], f"span_embed.layers.{i}")
        lin(w.saliency_proj, "saliency_proj")
        if w.has_adapter:
            lin(w.adapter[0], "adapter_layer.layers.0"); lin(w.adapter[1], "adapter_layer.layers.1")
        w.pos_dim_t = self._dim_t.data_ptr()
        if self.pre_norm:
            w.pre_norm = 1
            ln(w.enc_norm, "transformer.encoder.norm")
        if self.use_txt_pos:
            w.txt_pos_embed = p(_TXT_POS_PREFIX + "position_embeddings.weight")
            w.txt_pos_rows = int(sd[_TXT_POS_PREFIX + "position_embeddings.weight"].shape[0])
            ln(w.txt_pos_ln, _TXT_POS_PREFIX + "LayerNorm")
        if self._handle is not None:
            lib.cone_model_destroy(self._handle)
            self._handle = None
        h = C.c_void_p()
        torch.cuda.synchronize()
        _lib.check(lib.cone_model_create(C.byref(w), C.byref(h)))
        self._handle = h
        return self

    def __del__(self):
        try:
            if self._handle is not None:
                _lib.load().cone_model_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    # ---- helpers --------------------------------------------------------------------------------
    def _h(self):
        if self._handle is None:
            raise _lib.ConeHipError("load_state_dict() must be called before running the model")
        return self._handle

    @staticmethod
    def _f32(t):
        return t.to(dtype=torch.float32).contiguous()

    def _lengths(self, mask):
        """Prefix mask (utils/tensor_utils.py:50-52: 1 = valid) -> int32 valid lengths (cone_mask_lengths: one launch).  The
        reference's collate (pad_sequences_1d) only ever produces PREFIX masks and the packed kernels take lengths; a mask with
        holes is not representable.  ``CONE_AMD_CHECK_MASKS=1`` verifies it (one device round trip per call) and raises
        ValueError.  Nothing is cached per mask tensor (one ~3 us launch per call): an identity / version-counter key goes
        stale under writes that bypass the counter (``.data.copy_``, DLPack / numpy aliases) and does not exist for inference
        tensors (``torch.inference_mode()``)."""
        m = mask.to(torch.float32).contiguous()
        n = torch.empty(m.shape[0], dtype=torch.int32, device=m.device)
        _lib.check(_lib.load().cone_mask_lengths(_lib.ptr(m), m.shape[0], m.shape[1], _lib.ptr(n), _lib.stream()))
        if _CHECK_MASKS:
            ar = torch.arange(m.shape[1], device=m.device)[None]
            if not bool(((ar < n[:, None]) == (m != 0)).all()):
                raise ValueError("CONE.forward takes prefix masks (1 ... 1 0 ... 0), as the reference's collate produces")
        return n

    # ---- CONE.forward (cone/model.py:82-128) ----------------------------------------------------
    def forward(self, src_txt, src_txt_mask, src_vid_motion, src_vid_motion_mask, taps: bool = False):
        lib, h = _lib.load(), self._h()
        vid, txt = self._f32(src_vid_motion), self._f32(src_txt)
        B, Lv, _ = vid.shape
        Lq = txt.shape[1]
        self._check_dim(vid, self.args.v_motion_feat_dim, "src_vid_motion")
        self._check_dim(txt, self.args.t_feat_dim, "src_txt")
        vlen, qlen = self._lengths(src_vid_motion_mask), self._lengths(src_txt_mask)
        dev = vid.device
        nq, nd, d = self.num_queries, self.args.dec_layers, self.hidden_dim
        logits = torch.empty(B, nq, 2, device=dev)
        spans = torch.empty(B, nq, 2, device=dev)
        sal = torch.empty(B, Lv, device=dev)
        want_aux = self.aux_loss and nd > 1
        t = _lib.Taps()
        aux_l = aux_s = mem = hs = None
        if want_aux:
            aux_l = torch.empty(nd - 1, B, nq, 2, device=dev)
            aux_s = torch.empty(nd - 1, B, nq, 2, device=dev)
            t.aux_logits, t.aux_spans = aux_l.data_ptr(), aux_s.data_ptr()
        if taps:
            mem = torch.empty(B, Lv + Lq, d, device=dev)
            hs = torch.empty(nd, B, nq, d, device=dev)
            t.memory, t.hs = mem.data_ptr(), hs.data_ptr()
        nbytes = lib.cone_forward_workspace(h, B, Lv, Lq)
        ws = self._ws.get(nbytes, dev)
        _lib.check(lib.cone_forward_windows(h, _lib.ptr(vid), _lib.ptr(vlen), _lib.ptr(txt), _lib.ptr(qlen), B,
                                            Lv, Lq, _lib.ptr(logits), _lib.ptr(spans), _lib.ptr(sal), C.byref(t),
                                            _lib.ptr(ws), ws.numel(), _lib.stream()))
        out = {"pred_logits": logits, "pred_spans": spans, "saliency_scores": sal}
        if want_aux:
            out["aux_outputs"] = [{"pred_logits": aux_l[i], "pred_spans": aux_s[i]} for i in range(nd - 1)]
        if taps:
            out["memory"], out["hs"] = mem, hs
        return out

    # ---- CONE.forward_clip_matching (cone/model.py:130-152) -------------------------------------
    def forward_clip_matching(self, src_cls_txt, src_vid_appear, src_vid_appear_mask, proposal=None,
                              is_groundtruth=False):
        if is_groundtruth:
            raise NotImplementedError("ground-truth proposal matching is a training-only branch")
        lib, h = _lib.load(), self._h()
        cls, vid, spans = self._f32(src_cls_txt), self._f32(src_vid_appear), self._f32(proposal)
        self._check_dim(vid, self.args.v_appear_feat_dim, "src_vid_appear")
        self._check_dim(cls, self.args.v_appear_feat_dim, "src_cls_txt")
        B, Lv, _ = vid.shape
        vlen = self._lengths(src_vid_appear_mask)
        match = torch.empty(B, self.num_queries, device=vid.device)
        nbytes = lib.cone_clip_matching_workspace(h, B)
        ws = self._ws.get(nbytes, vid.device)
        _lib.check(lib.cone_clip_matching(h, _lib.ptr(cls), _lib.ptr(vid), _lib.ptr(vlen), Lv, _lib.ptr(spans), B,
                                          _lib.ptr(match), _lib.ptr(ws), ws.numel(), _lib.stream()))
        return match

    # ---- arena-level entry points used by the eval driver ---------------------------------------
    def adapter_norm(self, vid_rows, renorm: bool = True):
        """cone/inference.py:254-258 over any number of clip rows (n, dv); renorm=False keeps
        adapter(x)+x un-normalised (run_on_video/cone_localizator.py:135-138)."""
        lib, h = _lib.load(), self._h()
        x = self._f32(vid_rows)
        self._check_dim(x, self.args.v_appear_feat_dim, "appearance clip features")
        out = torch.empty_like(x)
        nbytes = lib.cone_adapter_norm_workspace(h, x.shape[0])
        ws = self._ws.get(nbytes, x.device)
        _lib.check(lib.cone_adapter_norm(h, _lib.ptr(x), x.shape[0], _lib.ptr(out), 1 if renorm else 0, _lib.ptr(ws),
                                         ws.numel(), _lib.stream()))
        return out

    def project(self, which: int, rows, ws=None):
        """input_vid_proj (which=0) / input_txt_proj (which=1) on (n, din) rows -> (n, d).  ``ws``: the scratch to use (default:
        the model's own)."""
        lib, h = _lib.load(), self._h()
        x = self._f32(rows)
        self._check_dim(x, self.args.t_feat_dim if which else self.args.v_motion_feat_dim,
                        "text tokens" if which else "motion clip features")
        out = torch.empty(x.shape[0], self.hidden_dim, device=x.device)
        nbytes = lib.cone_project_workspace(h, which, x.shape[0])
        ws = (ws or self._ws).get(nbytes, x.device)
        _lib.check(lib.cone_project_tokens(h, which, _lib.ptr(x), x.shape[0], _lib.ptr(out), _lib.ptr(ws),
                                           ws.numel(), _lib.stream()))
        return out

    @staticmethod
    def _check_dim(t, dim, what):
        """The library takes plain pointers: a feature tensor of another width would be read with the wrong row stride."""
        if int(t.shape[-1]) != int(dim):
            raise ValueError(f"{what}: feature dim {int(t.shape[-1])}, the model was built for {int(dim)}")

    def set_option(self, name: str, value: int):
        """A/B switch of this model's handle (cone_model_set_option): parity tests and diagnostics only."""
        _lib.check(_lib.load().cone_model_set_option(self._h(), name.encode(), int(value)))
        return self

    def layer0_rows(self, proj_rows, ws=None):
        """First encoder layer's in_proj hoisted out of the window loop: q|k|v rows once per projected clip /
        text token (cone_layer0_project)."""
        lib, h = _lib.load(), self._h()
        qkv = torch.empty(proj_rows.shape[0], 3 * self.hidden_dim, device=proj_rows.device)
        nbytes = lib.cone_layer0_project_workspace(h, proj_rows.shape[0])        # (--pre_norm: norm1 of the rows first)
        ws = (ws or self._ws).get(nbytes, proj_rows.device) if nbytes else None
        _lib.check(lib.cone_layer0_project(h, _lib.ptr(proj_rows), proj_rows.shape[0], _lib.ptr(qkv), _lib.ptr(ws),
                                           ws.numel() if ws is not None else 0, _lib.stream()))
        return qkv

    def pos_tables(self, max_v_l: int):
        """Static position tables of this checkpoint (built once per model and window length): the sine rows of
        every (window length, position) and their W_qk^T images per encoder layer (cone_pos_tables)."""
        lib, h = _lib.load(), self._h()
        if getattr(self, "_pos_tabs", None) is None or self._pos_tabs[0] != max_v_l:
            rows = lib.cone_pos_table_rows(max_v_l)
            dev = self.device
            pos_rows = torch.empty(rows, self.hidden_dim, device=dev)
            pos_qk = torch.empty(self.args.enc_layers, rows, 2 * self.hidden_dim, device=dev)
            _lib.check(lib.cone_pos_tables(h, max_v_l, _lib.ptr(pos_rows), _lib.ptr(pos_qk), _lib.stream()))
            self._pos_tabs = (max_v_l, dict(pos_rows=pos_rows, pos_qk=pos_qk))
        return self._pos_tabs[1]

    @property
    def txt_pos_tables(self) -> bool:
        """--use_txt_pos checkpoints take the table path when the caller hands over the tokens' own position rows
        (``text_positions``)."""
        return self.use_txt_pos

    def text_positions(self, tproj, tok_index):
        """--use_txt_pos (cone/model.py:106): the position term of a text token is LayerNorm(src_txt[t] + position_embeddings[t])
        -- a per-TOKEN row, shared by all windows of its query like the row caches.  ``tok_index`` (n,) int32 = index of each
        projected token row inside its query.  Returns ``txt_pos`` (n, d) and ``txt_pos_qk`` (enc_layers, n, 2d), its images
        under every encoder layer's [W_q | W_k] (cone_layer0_text_positions)."""
        lib, h = _lib.load(), self._h()
        n = int(tproj.shape[0])
        if int(tok_index.shape[0]) != n:
            raise ValueError(f"text_positions: {n} token rows, {int(tok_index.shape[0])} indices")
        txt_pos = torch.empty(n, self.hidden_dim, device=tproj.device)
        txt_pos_qk = torch.empty(self.args.enc_layers, n, 2 * self.hidden_dim, device=tproj.device)
        _lib.check(lib.cone_layer0_text_positions(h, _lib.ptr(tproj), _lib.ptr(tok_index, torch.int32), n, _lib.ptr(txt_pos),
                                                  _lib.ptr(txt_pos_qk), _lib.stream()))
        return txt_pos, txt_pos_qk

    def layer0_cache(self, vproj, tproj, max_v_l: int, tok_index=None):
        """cone_layer0 for forward_packed: the per-row q|k|v caches.  The static position tables are the handle's own
        (built at cone_model_create, ABI 6); ``pos_tables()`` builds caller-owned ones (parity tests).  ``tok_index``: see
        ``text_positions`` -- a --use_txt_pos model needs it to take the table path."""
        l0 = dict(qkv_vid=self.layer0_rows(vproj), qkv_txt=self.layer0_rows(tproj), max_v_l=max_v_l)
        if self.txt_pos_tables and tok_index is not None:
            l0["txt_pos"], l0["txt_pos_qk"] = self.text_positions(tproj, tok_index)
        return l0

    def forward_packed(self, vproj, vid_row0, vid_len, tproj, txt_row0, txt_len, Lv_max, Lq_max, l0=None,
                       saliency: bool = True, aux: bool = False):
        """CONE.forward on windows given by index into projected token arenas.  ``saliency=False`` skips the
        saliency head (cone/inference.py computes and never reads it, :54-59); ``aux=True`` also runs decoder.norm +
        the class / span heads of the intermediate decoder layers (``aux_outputs``, cone/model.py:123-127 -- equally
        unread at inference)."""
        lib, h = _lib.load(), self._h()
        l0s = l0p = None
        if l0 is not None:
            dp = lambda k: l0[k].data_ptr() if l0.get(k) is not None else None
            l0s = _lib.Layer0(dp("qkv_vid"), dp("qkv_txt"), dp("pos_qk"), dp("pos_rows"), l0["max_v_l"], dp("txt_pos"),
                              dp("txt_pos_qk"), int(l0["txt_pos"].shape[0]) if l0.get("txt_pos") is not None else 0)
            l0p = C.byref(l0s)
        B = vid_row0.shape[0]
        dev = vproj.device
        nq, nd = self.num_queries, self.args.dec_layers
        logits = torch.empty(B, nq, 2, device=dev)
        spans = torch.empty(B, nq, 2, device=dev)
        sal = torch.empty(B, Lv_max, device=dev) if saliency else None
        tp = None
        want_aux = aux and nd > 1
        if want_aux:
            t = _lib.Taps()
            aux_l = torch.empty(nd - 1, B, nq, 2, device=dev)
            aux_s = torch.empty(nd - 1, B, nq, 2, device=dev)
            t.aux_logits, t.aux_spans = aux_l.data_ptr(), aux_s.data_ptr()
            tp = C.byref(t)
        nbytes = lib.cone_forward_packed_workspace(h, B, Lv_max, Lq_max, l0p)
        ws = self._ws.get(nbytes, dev)
        i32 = torch.int32
        _lib.check(lib.cone_forward_packed(h, _lib.ptr(vproj), _lib.ptr(vid_row0, i32), _lib.ptr(vid_len, i32),
                                           _lib.ptr(tproj), _lib.ptr(txt_row0, i32), _lib.ptr(txt_len, i32), B,
                                           Lv_max, Lq_max, _lib.ptr(logits), _lib.ptr(spans), _lib.ptr(sal), tp,
                                           l0p, _lib.ptr(ws), ws.numel(), _lib.stream()))
        out = {"pred_logits": logits, "pred_spans": spans}
        if saliency:
            out["saliency_scores"] = sal
        if want_aux:
            out["aux_outputs"] = [{"pred_logits": aux_l[i], "pred_spans": aux_s[i]} for i in range(nd - 1)]
        return out

    def clip_matching_gathered(self, cls, cls_row, vid, vid_row0, vid_len, pad_len, spans):
        lib, h = _lib.load(), self._h()
        B = vid_row0.shape[0]
        self._check_dim(vid, self.args.v_appear_feat_dim, "appearance clip features")
        self._check_dim(cls, self.args.v_appear_feat_dim, "query cls vectors")
        match = torch.empty(B, self.num_queries, device=vid.device)
        nbytes = lib.cone_clip_matching_workspace(h, B)
        ws = self._ws.get(nbytes, vid.device)
        i32 = torch.int32
        _lib.check(lib.cone_clip_matching_gathered(h, _lib.ptr(cls), _lib.ptr(cls_row, i32), _lib.ptr(vid),
                                                   _lib.ptr(vid_row0, i32), _lib.ptr(vid_len, i32),
                                                   _lib.ptr(pad_len, i32), _lib.ptr(spans), B, _lib.ptr(match),
                                                   _lib.ptr(ws), ws.numel(), _lib.stream()))
        return match


def build_model(args):
    """``build_model(args) -> (model, criterion)`` of cone/model.py:468-521.  The criterion is the forward-value mirror
    of ``SetCriterion`` (cone_amd.criterion: losses as numbers for evaluation-side meters, no autograd)."""
    from .criterion import build_criterion
    return CONE(args), build_criterion(args)
