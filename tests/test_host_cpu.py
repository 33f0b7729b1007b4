"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header declares;
host logic (options, synthetic data, window geometry) agrees with the oracle.  No GPU compute."""
import json
import os
import re

import numpy as np
import pytest
import torch

from cone_amd import synth
from cone_amd.config import make_opt, parse_test_options
from oracle import cone_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_header_symbols():
    from cone_amd import _lib, build
    lib_path = build.build()
    assert os.path.exists(lib_path)
    lib = _lib.load()
    with open(os.path.join(ROOT, "include", "cone_hip.h")) as f:
        hdr = f.read()
    declared = sorted(set(re.findall(r"\b(cone_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cone_hip.h but not exported"
    assert set(declared) == set(_lib.EXPORTS), set(declared) ^ set(_lib.EXPORTS)
    assert lib.cone_abi_version() == 1
    assert lib.cone_num_windows(901, 90) == 22 and lib.cone_num_windows(1250, 125) == 22


def test_product_fails_loudly_without_gpu():
    from cone_amd import _lib
    from cone_amd.model import build_model
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    opt = make_opt("ego4d")
    model, crit = build_model(opt)
    assert crit is None
    with pytest.raises(_lib.ConeHipError):
        model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict(opt, 0).items()})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cone_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_synth_is_deterministic_and_matches_reference_keys():
    opt = make_opt("ego4d")
    a, b = synth.make_state_dict(opt, 3), synth.make_state_dict(opt, 3)
    assert synth.state_dict_checksum(a) == synth.state_dict_checksum(b)
    assert sum(v.size for v in a.values()) == 4355589 + 20 * 256 - 20 * 256   # 4 355 589 params
    assert a["transformer.encoder.layers.0.self_attn.in_proj_weight"].shape == (768, 256)
    mad = synth.make_state_dict(make_opt("mad"), 1)
    assert mad["input_txt_proj.0.net.1.weight"].shape == (256, 512)
    assert mad["adapter_layer.layers.1.weight"].shape == (512, 256)


def test_test_options_opt_json_round_trip(tmp_path):
    saved = vars(make_opt("ego4d"))
    saved.update(nms_thd=-1, topk_window=30, eval_bsz=32, device=0)
    (tmp_path / "opt.json").write_text(json.dumps(saved))
    ckpt = tmp_path / "model_best.ckpt"
    ckpt.write_bytes(b"")
    opt = parse_test_options(["--resume", str(ckpt), "--eval_split_name", "test", "--eval_path", "x.jsonl",
                              "--eval_id", "e1", "--nms_thd", "0.5", "--topk_window", "20", "--max_v_l", "7"])
    assert opt.nms_thd == 0.5 and opt.topk_window == 20          # CLI wins (cone/config.py:190-193)
    assert opt.max_v_l == 90 and opt.clip_length == 0.535         # opt.json wins
    assert opt.results_dir == str(tmp_path) and opt.model_dir == str(tmp_path)


def test_window_table_matches_oracle_collate():
    """Index arithmetic of cone_amd.inference.window_table == the eval branch of
    StartEndDataset.__getitem__ + collate as restated by the oracle (on CPU tensors)."""
    from cone_amd import inference as inf
    opt = make_opt("ego4d", topk_window=5, eval_bsz=3)
    ann, vf, qf = synth.make_dataset(opt, 10, 3, seed=4, ctx_range=(40, 300))
    store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
    rng = np.random.default_rng(0)
    ranks = {}
    win_idx = torch.full((len(ann), opt.topk_window), -1, dtype=torch.int32)
    for qi, r in enumerate(ann):
        nw = O.num_windows(vf[r["clip_id"]].shape[0], opt.max_v_l)
        perm = rng.permutation(nw).tolist()
        ranks[r["query_id"]] = perm
        k = min(nw, opt.topk_window)
        win_idx[qi, :k] = torch.tensor(perm[:k], dtype=torch.int32)
    wt = inf.window_table(store, opt, win_idx)
    w = 0
    for b0 in range(0, len(ann), opt.eval_bsz):
        metas, mi, ci = O.build_batch(opt, ann[b0:b0 + opt.eval_bsz], vf, qf, ranks)
        Lv_pad = mi["src_vid_motion"].shape[1]
        for j, m in enumerate(metas):
            assert int(wt["vid_len"][w]) == m["duration"] and int(wt["video_start"][w]) == m["video_start"]
            assert int(wt["pad_len"][w]) == Lv_pad
            r0 = int(wt["vid_row0"][w])
            assert torch.equal(store.vid_raw[r0:r0 + m["duration"]], mi["src_vid_motion"][j, :m["duration"]])
            t0, tl = int(wt["txt_row0"][w]), int(wt["txt_len"][w])
            assert tl == int(mi["src_txt_mask"][j].sum())
            assert ann[int(wt["q_of"][w])]["query_id"] == m["query_id"]
            w += 1
    assert w == wt["vid_len"].shape[0]


def test_packed_feature_store_round_trip(tmp_path):
    """one mmap-able arena file per split; loading reproduces the store bit for bit
    (CPU device: building a store involves no kernel)."""
    import numpy as np
    import torch
    from cone_amd import synth
    from cone_amd.config import make_opt
    from cone_amd.inference import FeatureStore
    opt = make_opt("ego4d")
    ann, vf, qf = synth.make_dataset(opt, 13, 3, seed=4, ctx_range=(50, 120))
    cpu = torch.device("cpu")
    a = FeatureStore(opt, ann, vf, qf, device=cpu)
    path = a.save_packed(str(tmp_path / "split.conefs"))
    b = FeatureStore.from_packed(opt, path, device=cpu)
    assert b.ann == a.ann and b.clip_ids == a.clip_ids and b.ctx_l == a.ctx_l and b.tok_len == a.tok_len
    for k in ("vid_raw", "tok_raw", "cls_raw"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    assert np.array_equal(a.vid_off, b.vid_off) and np.array_equal(a.tok_off, b.tok_off)
    assert np.array_equal(a.q_vid, b.q_vid)
    # data_ratio keeps a prefix of the queries like the reference's loader
    half = FeatureStore.from_packed(make_opt("ego4d", data_ratio=0.5), path, device=cpu)
    assert len(half.ann) == 6 and half.tok_raw.shape[0] == sum(a.tok_len[:6]) and half.cls_raw.shape[0] == 6
    with open(path, "rb") as f:
        assert f.read(8) == b"CONEFS01"
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"not a store")
    import pytest
    with pytest.raises(ValueError):
        FeatureStore.from_packed(opt, str(bad), device=cpu)
