"""Options for the CONE inference hot path.

Mirrors the option surface of the reference (``cone/config.py:21-164``): the same
names, defaults and the ``opt.json`` round trip of ``TestOptions.parse``
(``cone/config.py:175-236``).  Only the options the inference path reads are
interpreted; training-only ones are carried through untouched so that an
``opt.json`` written by the reference loads unchanged.
"""
from __future__ import annotations

import argparse
import json
import os
from types import SimpleNamespace

# Options that the command line may override at test time; everything else is
# taken from <ckpt_dir>/opt.json (cone/config.py:190-193).
CLI_WINS = ("eval_path", "eval_split_name", "results_root", "num_workers", "nms_thd",
            "debug", "save_all", "max_before_nms", "max_after_nms", "max_pred_l",
            "min_pred_l", "eval_bsz", "data_ratio", "topk_window", "resume",
            "resume_all", "no_sort_results", "packed_features", "split_bf16")

MODEL_DEFAULTS = dict(
    hidden_dim=256, nheads=8, dim_feedforward=1024, enc_layers=2, dec_layers=2,
    num_queries=5, n_input_proj=2, pre_norm=False, position_embedding="sine",
    use_txt_pos=False, span_loss_type="l1", aux_loss=True, adapter_module="linear",
    dropout=0.1, input_dropout=0.5,
)

# Shipped hyper-parameters (cone/scripts/train_ego4d.sh:12-32, train_mad.sh:12-36).
PRESETS = {
    "ego4d": dict(dset_name="ego4d", max_v_l=90, max_q_l=20, clip_length=0.535,
                  v_motion_feat_dim=256, v_appear_feat_dim=256, t_feat_dim=768,
                  topk_window=20, eval_bsz=32),
    "mad": dict(dset_name="mad", max_v_l=125, max_q_l=25, clip_length=0.2,
                v_motion_feat_dim=512, v_appear_feat_dim=512, t_feat_dim=512,
                topk_window=30, eval_bsz=16),
}

EVAL_DEFAULTS = dict(
    nms_thd=-1.0, max_before_nms=200, max_after_nms=5, no_sort_results=False,
    eval_split_name="val", eval_modality="both", save_all=False, debug=False,
    num_workers=4, data_ratio=1.0, results_dir=".", device=0,
    no_norm_vfeat=False, no_norm_tfeat=False,       # cone/config.py:80-81 -> normalize_v / normalize_t of the datasets (cone/inference.py:581-582)
)


def make_opt(preset: str = "ego4d", **overrides) -> SimpleNamespace:
    """Namespace with every field the hot path reads."""
    d = dict(MODEL_DEFAULTS)
    d.update(EVAL_DEFAULTS)
    d.update(PRESETS[preset])
    d.update(overrides)
    return SimpleNamespace(**d)


def build_parser() -> argparse.ArgumentParser:
    """The inference CLI of cone/inference.py (cone/config.py:21-164, 229-236)."""
    p = argparse.ArgumentParser(description="CONE coarse-to-fine inference on MI355X")
    p.add_argument("--dset_name", type=str, choices=["ego4d", "mad"])
    p.add_argument("--eval_split_name", type=str, default="val")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--data_ratio", type=float, default=1.0)
    p.add_argument("--results_root", type=str, default="cone_results")
    p.add_argument("--device", type=int, default=0, help="0 gpu; the HIP path has no cpu mode")
    p.add_argument("--num_workers", type=int, default=4)
    p.add_argument("--no_pin_memory", action="store_true")
    p.add_argument("--topk_window", type=int, default=30)
    p.add_argument("--eval_bsz", type=int, default=32)
    p.add_argument("--resume", type=str, default=None)
    p.add_argument("--resume_all", action="store_true")
    p.add_argument("--max_q_l", type=int, default=20)
    p.add_argument("--max_v_l", type=int, default=90)
    p.add_argument("--clip_length", type=float, default=1.0)
    p.add_argument("--eval_path", type=str, default=None)
    p.add_argument("--packed_features", type=str, default=None,
                   help="(cone_amd extension) packed feature arena written by `python -m cone_amd.pack_features`; "
                        "replaces the LMDB readers")
    p.add_argument("--split_bf16", action="store_true",
                   help="(cone_amd extension, opt-in) transformer layer tails on the bf16 matrix cores: every fp32 product as "
                        "six partial products of three-piece bf16 operands, fp32 accumulation -- fp32-MFMA accuracy "
                        "(measured against float64), ~1.3x the step rate; the default computes them on the fp32 MFMA")
    p.add_argument("--no_norm_vfeat", action="store_true")
    p.add_argument("--no_norm_tfeat", action="store_true")
    p.add_argument("--motion_feat_dir", type=str)
    p.add_argument("--appearance_feat_dir", type=str)
    p.add_argument("--t_feat_dir", type=str)
    p.add_argument("--v_motion_feat_dim", type=int)
    p.add_argument("--v_appear_feat_dim", type=int)
    p.add_argument("--t_feat_dim", type=int)
    p.add_argument("--ctx_mode", type=str, default="video")
    p.add_argument("--adapter_module", default="none", type=str, choices=["linear", "none"])
    p.add_argument("--position_embedding", default="sine", type=str, choices=("sine", "learned"))
    p.add_argument("--enc_layers", default=2, type=int)
    p.add_argument("--dec_layers", default=2, type=int)
    p.add_argument("--dim_feedforward", default=1024, type=int)
    p.add_argument("--hidden_dim", default=256, type=int)
    p.add_argument("--input_dropout", default=0.5, type=float)
    p.add_argument("--dropout", default=0.1, type=float)
    p.add_argument("--use_txt_pos", action="store_true")
    p.add_argument("--nheads", default=8, type=int)
    p.add_argument("--num_queries", default=5, type=int)
    p.add_argument("--pre_norm", action="store_true")
    p.add_argument("--eval_modality", type=str, default="both",
                   choices=["both", "proposal", "matching"])
    p.add_argument("--save_all", action="store_true")
    p.add_argument("--n_input_proj", type=int, default=2)
    p.add_argument("--no_aux_loss", dest="aux_loss", action="store_false")
    p.add_argument("--span_loss_type", default="l1", type=str, choices=["l1"])
    p.add_argument("--no_sort_results", action="store_true")
    p.add_argument("--max_before_nms", type=int, default=200)
    p.add_argument("--max_after_nms", type=int, default=5)
    p.add_argument("--nms_thd", type=float, default=-1)
    # TestOptions (cone/config.py:229-236)
    p.add_argument("--eval_id", type=str)
    p.add_argument("--eval_results_dir", type=str, default=None)
    p.add_argument("--model_dir", type=str)
    return p


def parse_test_options(argv=None) -> SimpleNamespace:
    """``TestOptions().parse()`` (cone/config.py:175-222): ``--debug`` redirects results_root and zeroes
    num_workers (:179-181); options saved beside the checkpoint win, except the CLI_WINS whitelist (:184-193);
    ``results_dir`` stays the SAVED one unless ``--eval_results_dir`` is given (:194-195) -- a checkpoint directory
    whose opt.json carries no results_dir falls back to the checkpoint's own directory."""
    opt = build_parser().parse_args(argv)
    if opt.debug:
        opt.results_root = os.path.sep.join(opt.results_root.split(os.path.sep)[:-1] + ["debug_results"])
        opt.num_workers = 0
    if opt.resume is None:
        raise ValueError("--resume <ckpt> is required at inference")
    opt.model_dir = os.path.dirname(opt.resume)
    with open(os.path.join(opt.model_dir, "opt.json")) as f:
        saved = json.load(f)
    for k, v in saved.items():
        if k not in CLI_WINS:
            setattr(opt, k, v)
    if opt.eval_results_dir is not None:
        opt.results_dir = opt.eval_results_dir
    elif not getattr(opt, "results_dir", None):
        opt.results_dir = opt.model_dir
    opt.pin_memory = not opt.no_pin_memory
    return opt
