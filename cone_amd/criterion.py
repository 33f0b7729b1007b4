"""Host-side mirror of the reference's ``SetCriterion`` and ``HungarianMatcher`` (cone/model.py:213-425,
cone/matcher.py:9-113) for the evaluation side: the forward VALUES of the training losses, computed by HIP kernels
(``csrc/criterion.hip``) from the model outputs -- assignment included.  No autograd: training itself is out of scope
(SURVEY.md section 2); this is what ``compute_mr_results(..., criterion=...)``-style loss meters need.

Same constructor arguments, ``weight_dict``, ``forward(outputs, targets, neg_outputs=None)`` signature, target layout
(``targets["span_labels"][b]["spans"]`` (T_b, 2) in (center, width); ``saliency_pos_labels`` / ``saliency_neg_labels``
(B, P)) and result keys (``loss_span``, ``loss_giou``, ``loss_label``, ``class_error``, ``loss_saliency`` and the
``_{i}`` variants of the auxiliary decoder layers) as the reference.
"""
from __future__ import annotations

import torch

from . import _lib


def _f32(t):
    return t.detach().to(dtype=torch.float32).contiguous()


def _i32(t):
    return t.detach().to(dtype=torch.int32).contiguous()


class HungarianMatcher:
    """cone/matcher.py:9-106.  ``__call__(outputs, targets)`` -> [(slot indices, target indices)] per window (int64,
    slot indices ascending like scipy's row indices)."""

    def __init__(self, cost_class: float = 1, cost_span: float = 1, cost_giou: float = 1, span_loss_type: str = "l1",
                 max_v_l: int = 75):
        if span_loss_type != "l1":
            raise NotImplementedError("only span_loss_type='l1' (cone/config.py:134)")
        assert cost_class != 0 or cost_span != 0 or cost_giou != 0, "all costs cant be 0"
        self.cost_class, self.cost_span, self.cost_giou = cost_class, cost_span, cost_giou
        self.foreground_label = 0

    def __call__(self, outputs, targets):
        assign = _layer_losses(self, None, outputs, targets, None, want_saliency=False)[1]
        out = []
        for row in assign.cpu().tolist():
            pairs = [(n, j) for n, j in enumerate(row) if j >= 0]
            out.append((torch.tensor([p[0] for p in pairs], dtype=torch.int64),
                        torch.tensor([p[1] for p in pairs], dtype=torch.int64)))
        return out

    forward = __call__


def _layer_losses(matcher, crit, outputs, targets, neg_outputs, want_saliency):
    """One launch pair of cone_criterion_forward for one decoder layer; returns (losses (5,) fp32 on device, assign)."""
    lib = _lib.load()
    logits = _f32(outputs["pred_logits"])
    dev = logits.device
    B, Nq = logits.shape[:2]
    spans = tgt = off = None
    if targets is not None:
        spans = _f32(outputs["pred_spans"])
        lst = [t["spans"] for t in targets["span_labels"]]
        sizes = [int(x.shape[0]) for x in lst]
        if max(sizes) > 8:
            raise NotImplementedError("more than 8 target spans per window")
        tgt = _f32(torch.cat([x.to(dev) for x in lst]))
        off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0).tolist()), dtype=torch.int32, device=dev)
    neg_logits = _f32(neg_outputs["pred_logits"]) if neg_outputs is not None else None
    sal = pos = neg = nsal = None
    L = P = L2 = 0
    if want_saliency and targets is not None and "saliency_pos_labels" in targets:
        sal = _f32(outputs["saliency_scores"])
        L = sal.shape[1]
        lab = []
        for key in ("saliency_pos_labels", "saliency_neg_labels"):
            t = targets[key]
            # the reference indexes saliency_scores[batch, label] (cone/model.py:335-338): torch wraps [-L, -1] and raises
            # outside [-L, L).  Labels come from the dataloader on the HOST: checked and wrapped there (no device sync);
            # device-resident labels are checked with one asynchronous device assert
            if t.is_cuda:
                torch._assert_async(((t >= -L) & (t < L)).all(), f"saliency label index out of range for {L} clips")
            elif t.numel() and (int(t.min()) < -L or int(t.max()) >= L):
                raise IndexError(f"saliency label index out of range for {L} clips: [{int(t.min())}, {int(t.max())}]")
            lab.append(_i32(torch.where(t < 0, t + L, t).to(dev)))
        pos, neg = lab
        P = pos.shape[1]
        if neg_outputs is not None:
            nsal = _f32(neg_outputs["saliency_scores"])
            L2 = nsal.shape[1]
    assign = torch.empty(B, Nq, dtype=torch.int32, device=dev)
    part = torch.empty(B, 8, device=dev)
    losses = torch.empty(5, device=dev)
    eos = crit.eos_coef if crit is not None else 1.0
    margin = crit.saliency_margin if crit is not None else 0.0
    P_ = _lib.ptr
    _lib.check(lib.cone_criterion_forward(P_(logits), P_(spans), P_(tgt), P_(off), P_(neg_logits), P_(sal), L, P_(pos),
                                          P_(neg), P, P_(nsal), L2, B, Nq, float(matcher.cost_span),
                                          float(matcher.cost_giou), float(matcher.cost_class), float(eos), float(margin),
                                          P_(assign), P_(part), P_(losses), _lib.stream()))
    return losses, assign


class SetCriterion:
    """cone/model.py:213-425 (forward values)."""

    def __init__(self, matcher, weight_dict, eos_coef, losses, temperature, span_loss_type, max_v_l, saliency_margin=1):
        if span_loss_type != "l1":
            raise NotImplementedError("only span_loss_type='l1' (cone/config.py:134)")
        self.matcher, self.weight_dict, self.losses = matcher, weight_dict, list(losses)
        self.temperature, self.span_loss_type, self.max_v_l = temperature, span_loss_type, max_v_l
        self.saliency_margin, self.eos_coef = saliency_margin, eos_coef
        self.foreground_label, self.background_label = 0, 1
        self.training = False

    def eval(self):
        return self

    def to(self, device):
        return self

    def loss_adapter(self, pos_outputs):
        """cone/model.py:249-264."""
        sim = _f32(pos_outputs["logits_per_video"])
        out = torch.empty(1, device=sim.device)
        _lib.check(_lib.load().cone_adapter_nce(_lib.ptr(sim), sim.shape[0], float(self.temperature), _lib.ptr(out),
                                                _lib.stream()))
        return {"loss_adapter": out[0]}

    def _pick(self, vals, top: bool, has_sal: bool, suffix: str = ""):
        d = {}
        if "spans" in self.losses:
            d["loss_span" + suffix], d["loss_giou" + suffix] = vals[0], vals[1]
        if "labels" in self.losses:
            d["loss_label" + suffix], d["class_error" + suffix] = vals[2], vals[3]
        if "saliency" in self.losses and top:
            d["loss_saliency"] = vals[4] if has_sal else 0
        return d

    def forward(self, outputs, targets, neg_outputs=None):
        if targets is None:                                             # :385-388
            vals, _ = _layer_losses(self.matcher, self, outputs, None, None, want_saliency=False)
            return {"loss_label": vals[2]}
        has_sal = "saliency_pos_labels" in targets
        vals, _ = _layer_losses(self.matcher, self, outputs, targets, neg_outputs, want_saliency=True)
        losses = self._pick(vals, True, has_sal)
        for i, aux in enumerate(outputs.get("aux_outputs", [])):       # :411-423 (saliency only in the top layer)
            v, _ = _layer_losses(self.matcher, self, aux, targets, neg_outputs, want_saliency=False)
            losses.update(self._pick(v, False, False, f"_{i}"))
        return losses

    __call__ = forward


def build_matcher(args):
    """cone/matcher.py:109-113."""
    g = lambda k, d: getattr(args, k, d)
    return HungarianMatcher(cost_span=g("set_cost_span", 10), cost_giou=g("set_cost_giou", 1),
                            cost_class=g("set_cost_class", 4), span_loss_type=g("span_loss_type", "l1"),
                            max_v_l=g("max_v_l", 75))


def build_criterion(args):
    """The criterion half of build_model (cone/model.py:499-521; defaults of cone/config.py:137-162)."""
    g = lambda k, d: getattr(args, k, d)
    weight_dict = {"loss_span": g("span_loss_coef", 10), "loss_giou": g("giou_loss_coef", 1),
                   "loss_label": g("label_loss_coef", 4), "loss_saliency": g("lw_saliency", 1.0)}
    if g("adapter_loss", True):                                         # cone/config.py:135
        weight_dict["loss_adapter"] = g("adapter_loss_coef", 1)
    if g("aux_loss", True):
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items() if k != "loss_saliency"})
        weight_dict.update(aux)
    return SetCriterion(matcher=build_matcher(args), weight_dict=weight_dict, losses=["spans", "labels", "saliency"],
                        eos_coef=g("eos_coef", 0.1), temperature=g("temperature", 0.07),
                        span_loss_type=g("span_loss_type", "l1"), max_v_l=g("max_v_l", 75),
                        saliency_margin=g("saliency_margin", 0.2))
