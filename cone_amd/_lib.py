"""ctypes binding of libcone_hip.so (include/cone_hip.h).

This is the whole FFI surface: every product code path goes through these calls, and a
missing / unbuilt library is a hard error (there is no CPU fallback anywhere in cone_amd).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcone_hip.so")

MAX_LAYERS = 8
MAX_PROJ = 3

c_float_p = C.c_void_p  # device pointers travel as integers


class Linear(C.Structure):
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p)]


class LNorm(C.Structure):
    _fields_ = [("g", C.c_void_p), ("b", C.c_void_p)]


class Mha(C.Structure):
    _fields_ = [("in_proj_w", C.c_void_p), ("in_proj_b", C.c_void_p), ("out_proj", Linear)]


class EncLayer(C.Structure):
    _fields_ = [("self_attn", Mha), ("linear1", Linear), ("linear2", Linear), ("norm1", LNorm),
                ("norm2", LNorm)]


class DecLayer(C.Structure):
    _fields_ = [("self_attn", Mha), ("cross_attn", Mha), ("linear1", Linear), ("linear2", Linear),
                ("norm1", LNorm), ("norm2", LNorm), ("norm3", LNorm)]


class Weights(C.Structure):
    _fields_ = [
        ("hidden_dim", C.c_int32), ("nheads", C.c_int32), ("dim_ff", C.c_int32), ("enc_layers", C.c_int32),
        ("dec_layers", C.c_int32), ("num_queries", C.c_int32), ("n_input_proj", C.c_int32),
        ("t_dim", C.c_int32), ("v_dim", C.c_int32), ("has_adapter", C.c_int32), ("v_motion_dim", C.c_int32),
        ("vid_proj_ln", LNorm * MAX_PROJ), ("vid_proj", Linear * MAX_PROJ),
        ("txt_proj_ln", LNorm * MAX_PROJ), ("txt_proj", Linear * MAX_PROJ),
        ("enc", EncLayer * MAX_LAYERS), ("dec", DecLayer * MAX_LAYERS),
        ("dec_norm", LNorm), ("query_embed", C.c_void_p), ("class_embed", Linear),
        ("span_embed", Linear * 3), ("saliency_proj", Linear), ("adapter", Linear * 2),
        ("pos_dim_t", C.c_void_p),
        ("txt_pos_embed", C.c_void_p), ("txt_pos_rows", C.c_int32), ("txt_pos_ln", LNorm),      # ABI 5: --use_txt_pos (NULL: off)
        ("pre_norm", C.c_int32), ("enc_norm", LNorm),                                            # ABI 5: --pre_norm (0: post-norm)
        ("table_max_v_l", C.c_int32),                                                            # ABI 8: tables for <= this many clips (0: 255)
    ]


class Layer0(C.Structure):
    _fields_ = [("qkv_vid", C.c_void_p), ("qkv_txt", C.c_void_p), ("pos_qk", C.c_void_p), ("pos_rows", C.c_void_p),
                ("max_v_l", C.c_int32),
                ("txt_pos", C.c_void_p), ("txt_pos_qk", C.c_void_p), ("n_txt", C.c_int64)]   # ABI 7: --use_txt_pos rows


class Taps(C.Structure):
    _fields_ = [("memory", C.c_void_p), ("hs", C.c_void_p), ("aux_logits", C.c_void_p),
                ("aux_spans", C.c_void_p)]


_SIGNATURES = {
    "cone_last_error": (C.c_char_p, []),
    "cone_abi_version": (C.c_int, []),
    "cone_model_create": (C.c_int, [C.POINTER(Weights), C.POINTER(C.c_void_p)]),
    "cone_model_destroy": (None, [C.c_void_p]),
    "cone_adapter_norm_workspace": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "cone_adapter_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                    C.c_void_p]),
    "cone_l2_normalize_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p,
                                         C.c_void_p]),
    "cone_num_windows": (C.c_int64, [C.c_int64, C.c_int]),
    "cone_prefilter_scores_workspace": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "cone_prefilter_scores": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_prefilter_scores_split_workspace": (C.c_size_t, [C.c_int64, C.c_int, C.c_int, C.c_int]),
    "cone_prefilter_scores_split": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_prefilter_batched": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "cone_topk_windows": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_topk_windows_workspace": (C.c_size_t, [C.c_int, C.c_int64, C.c_int]),
    "cone_topk_windows_ws": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "cone_project_workspace": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int64]),
    "cone_project_tokens": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                      C.c_size_t, C.c_void_p]),
    "cone_mask_lengths": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cone_forward_workspace": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "cone_forward_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Taps),
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_forward_packed_workspace": (C.c_size_t, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Layer0)]),
    "cone_forward_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(Taps), C.POINTER(Layer0), C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_pos_table_rows": (C.c_int64, [C.c_int]),
    "cone_pos_tables": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_layer0_project_workspace": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "cone_layer0_project": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_layer0_text_positions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_clip_matching_workspace": (C.c_size_t, [C.c_void_p, C.c_int]),
    "cone_clip_matching_gathered": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_clip_matching": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                     C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cone_window_table": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3
                          + [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_void_p]),
    "cone_compose_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cone_fuse_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_fuse_nms_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_temporal_nms": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "cone_matcher_cost": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_criterion_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "cone_adapter_nce": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "cone_prof_enable": (C.c_int, [C.c_int]),
    "cone_prof_collect": (C.c_int64, [C.c_void_p, C.c_int64]),
    "cone_eval_recall": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cone_eval_window_recall": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_void_p]),
    "cone_model_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    # cone_test_gemm(A, A2, a2_mod, W, bias, R, ln_g, ln_b, C, C2, ADD, M, N, K, flags, stream)
    "cone_test_gemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_void_p]),
    "cone_test_ffn": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_void_p]),
    "cone_test_proj_ffn": (C.c_int, [C.c_void_p] * 13 + [C.c_int, C.c_int, C.c_void_p]),
    "cone_test_proj_ffn_spread_scratch_bytes": (C.c_size_t, [C.c_int]),
    "cone_test_proj_ffn_spread": (C.c_int, [C.c_void_p] * 13 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "cone_test_ffn_split_image_bytes": (C.c_size_t, [C.c_int]),
    "cone_test_ffn_split": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "cone_test_rows_split_image_bytes": (C.c_size_t, [C.c_int]),
    "cone_test_rows_split": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "cone_test_proj_split_image_bytes": (C.c_size_t, []),
    "cone_test_proj_ffn_split": (C.c_int, [C.c_void_p] * 13 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "cone_test_enc_attn": (C.c_int, [C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "cone_test_dec_cross": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p]),
    "cone_test_dec_cross_slab_floats": (C.c_size_t, []),
    "cone_test_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                      C.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


class ConeHipError(RuntimeError):
    pass


def load():
    """Load libcone_hip.so; raises if it has not been built (python -m cone_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ConeHipError(
            f"{LIB_PATH} is missing: build the HIP extension with `python -m cone_amd.build` "
            "(hipcc, gfx950). cone_amd has no CPU fallback.")
    # torch ships its own libamdhip64.so.7; importing torch first makes our DT_NEEDED resolve to that
    # copy, so the process has ONE HIP runtime and torch's streams / allocations are ours too.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.cone_abi_version() != 8:
        raise ConeHipError("libcone_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


_pylists = None


def pylists():
    """The host-side list builder (cone_amd/csrc/pylists.c, ctypes.PyDLL: the call keeps the GIL)."""
    global _pylists
    if _pylists is None:
        path = os.path.join(HERE, "_cone_pylists.so")
        if not os.path.exists(path):
            raise ConeHipError(f"{path} is missing: build with `python -m cone_amd.build`")
        lib = C.PyDLL(path)
        fn = lib.cone_fill_predicted_times
        fn.restype = C.py_object
        fn.argtypes = [C.py_object, C.c_void_p, C.c_void_p, C.c_ssize_t, C.c_ssize_t, C.py_object]
        _pylists = lib
    return _pylists


def check(rc: int):
    if rc != 0:
        raise ConeHipError(f"libcone_hip error {rc}: {load().cone_last_error().decode()}")


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA(HIP) tensor (None -> NULL)."""
    import torch
    if t is None:
        return None
    if not t.is_cuda:
        raise ConeHipError("expected a tensor on the GPU")
    if not t.is_contiguous():
        raise ConeHipError("expected a contiguous tensor")
    if dtype is not None and t.dtype != dtype:
        raise ConeHipError(f"expected dtype {dtype}, got {t.dtype}")
    return C.c_void_p(t.data_ptr())


def stream():
    """The current HIP stream of the current device as a void*.  (The raw-stream getter is what torch's own compiled
    code paths call; it skips the Stream object that torch.cuda.current_stream() builds on every call -- the front of a
    step is ~30 launches issued while the GPU waits for the host.)"""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
