"""CPU-side checks: the C-ABI library builds, loads and exports every symbol the header declares;
host logic (options, synthetic data, window geometry) agrees with the oracle.  No GPU compute."""
import json
import os
import re
import sys

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
    assert lib.cone_abi_version() == 8
    assert lib.cone_num_windows(901, 90) == 22 and lib.cone_num_windows(1250, 125) == 22


def test_product_fails_loudly_without_gpu():
    from cone_amd import _lib
    from cone_amd.model import build_model
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    opt = make_opt("ego4d")
    model, crit = build_model(opt)
    assert crit is not None and crit.weight_dict["loss_span"] == 10 and "loss_giou_0" in crit.weight_dict
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
    assert opt.model_dir == str(tmp_path)
    assert opt.results_dir == saved["results_dir"]                # the saved one, unless --eval_results_dir (:194-195)


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


def _write_reference_stores(tmp_path, opt, ann, vf, qf, eot=False):
    """The reference's on-disk inputs: annotation jsonl + the two LMDBs of np.savez blobs (fake lmdb backend)."""
    import fake_lmdb
    vdir = fake_lmdb.write_env(str(tmp_path / "vid_lmdb"), {c: {"features": v} for c, v in vf.items()})
    cls_key = "eot_features" if eot else "cls_features"
    tdir = fake_lmdb.write_env(str(tmp_path / "txt_lmdb"),
                               {q: {"token_features": d["token_features"],
                                    cls_key: d["cls_features"][None] if eot else d["cls_features"]}
                                for q, d in qf.items()})
    eval_path = tmp_path / "split.jsonl"
    eval_path.write_text("\n".join(json.dumps(r) for r in ann))
    return vdir, tdir, str(eval_path)


def test_feature_store_from_reference_lmdbs(tmp_path, monkeypatch):
    """FeatureStore.from_lmdb reads the reference's feature stores (np.savez blobs keyed by clip_id / query_id,
    ``features`` / ``token_features`` + ``cls_features`` or a 2-D ``eot_features``): same arenas as building the
    store from the arrays; data_ratio keeps a prefix; pack_features converts them to the packed arena file."""
    import sys
    import fake_lmdb
    from cone_amd.inference import FeatureStore
    from cone_amd import pack_features
    monkeypatch.setitem(sys.modules, "lmdb", fake_lmdb)
    cpu = torch.device("cpu")
    for preset, eot in (("ego4d", False), ("mad", True)):
        base = make_opt(preset)
        ann, vf, qf = synth.make_dataset(base, 9, 3, seed=2, ctx_range=(30, 90), lq_range=(3, 30))
        d = tmp_path / preset
        d.mkdir()
        vdir, tdir, eval_path = _write_reference_stores(d, base, ann, vf, qf, eot=eot)
        opt = make_opt(preset, eval_path=eval_path, motion_feat_dir=vdir, appearance_feat_dir=vdir, t_feat_dir=tdir)
        a = FeatureStore(opt, ann, vf, qf, device=cpu)
        b = FeatureStore.from_lmdb(opt, device=cpu)
        assert b.ann == a.ann and b.clip_ids == a.clip_ids and b.tok_len == a.tok_len
        assert max(b.tok_len) == base.max_q_l                       # tokens truncated to max_q_l (dataloader :272-273)
        for k in ("vid_raw", "tok_raw", "cls_raw"):
            assert torch.equal(getattr(a, k), getattr(b, k)), (preset, k)
        half = FeatureStore.from_lmdb(make_opt(preset, eval_path=eval_path, motion_feat_dir=vdir,
                                               appearance_feat_dir=vdir, t_feat_dir=tdir, data_ratio=0.5), device=cpu)
        assert len(half.ann) == 4 and half.cls_raw.shape[0] == 4
        # python -m cone_amd.pack_features: LMDBs -> one packed arena file
        saved = {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, str, bool, type(None)))}
        (d / "opt.json").write_text(json.dumps(saved))
        (d / "model_best.ckpt").write_bytes(b"")
        out = str(d / "split.conefs")
        pack_features.main(["--resume", str(d / "model_best.ckpt"), "--eval_path", eval_path, "--eval_split_name",
                            "val", "--out", out])
        c = FeatureStore.from_packed(opt, out, device=cpu)
        for k in ("vid_raw", "tok_raw", "cls_raw"):
            assert torch.equal(getattr(a, k), getattr(c, k)), (preset, k)
    with pytest.raises(KeyError):
        FeatureStore.from_lmdb(make_opt("ego4d", eval_path=eval_path, motion_feat_dir=vdir, appearance_feat_dir=vdir,
                                        t_feat_dir=vdir), device=cpu)       # queries looked up in the video store


class RefLikeDatasets:
    """Objects with the attributes of the reference's PreFilteringDataset / StartEndDataset that eval_epoch's
    reference call shape reads (cone/ego4d_mad_dataloader.py:19-103, 258-282, 397-431), over in-memory features."""

    def __init__(self, opt, ann, vf, qf, normalize_t=True):
        self.data = self.query_data = ann
        self.videofeat = {c: torch.from_numpy(v) for c, v in vf.items()}      # RAW features (hazard H2)
        self.same_visual_path, self.normalize_t, self.load_labels = True, normalize_t, False
        self._opt, self._qf = opt, qf
        self.query_id2windowidx = None

    def _get_query_feat_by_qid(self, qid):
        tok, cls = O.prepare_query_inputs(self._opt, self._qf[qid])
        if not self.normalize_t:
            tok = torch.from_numpy(np.asarray(self._qf[qid]["token_features"], dtype=np.float32)[:self._opt.max_q_l])
        return tok, cls.numpy()


def test_feature_store_from_reference_dataset_objects():
    from cone_amd.inference import FeatureStore
    opt = make_opt("ego4d")
    ann, vf, qf = synth.make_dataset(opt, 7, 2, seed=3, ctx_range=(30, 90), lq_range=(3, 30))
    ds = RefLikeDatasets(opt, ann, vf, qf)
    st = FeatureStore.from_datasets(opt, ds, ds, device=torch.device("cpu"))
    plain = FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
    assert st.tok_normalized and st.cls_normalized and not plain.tok_normalized
    assert torch.equal(st.vid_raw, plain.vid_raw) and st.tok_len == plain.tok_len
    ref_tok = torch.cat([O.prepare_query_inputs(opt, qf[r["query_id"]])[0] for r in ann])
    assert torch.equal(st.tok_raw, ref_tok)                                    # already normalised: taken as is
    raw = FeatureStore.from_datasets(opt, ds, RefLikeDatasets(opt, ann, vf, qf, normalize_t=False),
                                     device=torch.device("cpu"))
    assert not raw.tok_normalized and torch.equal(raw.tok_raw, plain.tok_raw)


def test_feature_store_with_two_visual_sources(tmp_path, monkeypatch):
    """motion_feat_dir != appearance_feat_dir (cone/ego4d_mad_dataloader.py:63-81, 94-95, 134-151): the store carries a second
    arena with the same rows; every constructor fills it (arrays, the reference's LMDBs, its dataset objects, the packed
    file), views alias it, and a motion source of another length is refused (the reference would slice it with the
    appearance length and silently shift the windows)."""
    import sys
    import fake_lmdb
    from cone_amd.inference import FeatureStore
    monkeypatch.setitem(sys.modules, "lmdb", fake_lmdb)
    cpu = torch.device("cpu")
    base = make_opt("ego4d", v_motion_feat_dim=128)
    ann, vf, qf = synth.make_dataset(base, 9, 3, seed=6, ctx_range=(30, 90))
    mf = synth.make_motion_feats(base, vf, seed=6)
    a = FeatureStore(base, ann, vf, qf, device=cpu, motion_feats=mf)
    assert a.mot_raw.shape == (a.vid_raw.shape[0], 128) and a.vid_raw.shape[1] == 256
    assert torch.equal(a.mot_raw, torch.from_numpy(np.concatenate([mf[c] for c in a.clip_ids])))
    assert FeatureStore(base, ann, vf, qf, device=cpu).mot_raw is None
    v = a.view(2, 6)
    assert v.mot_raw.data_ptr() == a.mot_raw.data_ptr() and v.vid_raw.data_ptr() == a.vid_raw.data_ptr()
    # the reference's stores: two video LMDBs
    vdir, tdir, eval_path = _write_reference_stores(tmp_path, base, ann, vf, qf)
    mdir = fake_lmdb.write_env(str(tmp_path / "motion_lmdb"), {c: {"features": m} for c, m in mf.items()})
    opt = make_opt("ego4d", v_motion_feat_dim=128, eval_path=eval_path, motion_feat_dir=mdir, appearance_feat_dir=vdir,
                   t_feat_dir=tdir)
    b = FeatureStore.from_lmdb(opt, device=cpu)
    for k in ("vid_raw", "mot_raw", "tok_raw", "cls_raw"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k
    # the reference's dataset objects (eval_epoch(model, inter_ds, intra_ds, ...))
    # -- whose motion table holds what its reader returns: the L2-NORMALISED rows (_get_video_motion_feat_by_vid, :284-292,
    # unlike the appearance reader: hazard H2), so the store must not normalise them a second time
    ds = RefLikeDatasets(base, ann, vf, qf)
    ds.same_visual_path, ds.normalize_v = False, True
    ds.motion_videofeat = {c: torch.from_numpy(O.l2_normalize_np(m).astype(np.float32)) for c, m in mf.items()}
    c = FeatureStore.from_datasets(base, ds, ds, device=cpu)
    assert c.mot_normalized and not a.mot_normalized and not b.mot_normalized
    assert torch.equal(c.mot_raw, torch.cat([ds.motion_videofeat[k] for k in a.clip_ids])) and torch.equal(c.vid_raw, a.vid_raw)
    assert c.motion_rows(3, 40).data_ptr() == c.mot_raw[3:40].data_ptr()           # taken as they are
    nn = FeatureStore(make_opt("ego4d", v_motion_feat_dim=128, no_norm_vfeat=True), ann, vf, qf, device=cpu, motion_feats=mf)
    assert nn.motion_rows(3, 40).data_ptr() == nn.mot_raw[3:40].data_ptr()         # --no_norm_vfeat: the raw rows
    one_src = FeatureStore(base, ann, vf, qf, device=cpu)
    assert one_src.motion_rows(2, 9).data_ptr() == one_src.vid_raw[2:9].data_ptr()    # one source: the raw appearance rows
    # the packed arena file carries the fourth arena and the flags; files without it load as before
    d = FeatureStore.from_packed(base, a.save_packed(str(tmp_path / "two.conefs")), device=cpu)
    assert torch.equal(d.mot_raw, a.mot_raw) and torch.equal(d.vid_raw, a.vid_raw) and not d.mot_normalized
    e = FeatureStore.from_packed(base, c.save_packed(str(tmp_path / "two_n.conefs")), device=cpu)
    assert e.mot_normalized and e.tok_normalized and e.cls_normalized and torch.equal(e.mot_raw, c.mot_raw)
    one = FeatureStore(base, ann, vf, qf, device=cpu)
    assert FeatureStore.from_packed(base, one.save_packed(str(tmp_path / "one.conefs")), device=cpu).mot_raw is None
    short = dict(mf)
    short[a.clip_ids[1]] = short[a.clip_ids[1]][:-1]
    with pytest.raises(ValueError):
        FeatureStore(base, ann, vf, qf, device=cpu, motion_feats=short)


def test_debug_and_results_dir_options(tmp_path):
    saved = vars(make_opt("ego4d"))
    saved.update(results_dir="/some/training/dir", results_root="runs/cone_results")
    (tmp_path / "opt.json").write_text(json.dumps(saved))
    ckpt = tmp_path / "model_best.ckpt"
    ckpt.write_bytes(b"")
    base = ["--resume", str(ckpt), "--eval_split_name", "test", "--eval_path", "x.jsonl", "--eval_id", "e"]
    opt = parse_test_options(base)
    assert opt.results_dir == "/some/training/dir" and not opt.debug           # saved results_dir stays (:194-195)
    opt = parse_test_options(base + ["--eval_results_dir", str(tmp_path), "--debug", "--results_root", "a/b"])
    assert opt.results_dir == str(tmp_path) and opt.debug and opt.num_workers == 0
    assert opt.results_root == os.path.join("a", "debug_results")             # cone/config.py:179-181


def test_three_piece_bf16_split_is_exact_and_six_products_carry_fp32():
    """The arithmetic claim behind the opt-in split_bf16 path (cone_amd/csrc/ffn_split.hip), checked on the CPU with torch's
    bfloat16 (round to nearest even, like v_cvt_pk_bf16_f32): h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) reconstruct
    every normal fp32 value exactly, and the six kept piece products differ from the fp32 product by <= 2^-22 of it."""
    import torch
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(200000, generator=g) * 3, torch.randn(50000, generator=g) * 1e-3,
                   torch.randn(50000, generator=g) * 1e4, torch.tensor([1.0, -1.0, 0.0, 3.14159274, 1e-30, 65504.0, 1 + 2 ** -23])])
    w = torch.randn(x.numel(), generator=g) / 16

    def split(t):
        h = t.to(torch.bfloat16).float()
        m = (t - h).to(torch.bfloat16).float()
        lo = (t - h - m).to(torch.bfloat16).float()
        return h, m, lo
    xh, xm, xl = split(x)
    assert torch.equal((xh.double() + xm.double() + xl.double()).float(), x)          # exact: 8 + 8 + 8 significant bits
    wh, wm, wl = split(w)
    assert torch.equal((wh.double() + wm.double() + wl.double()).float(), w)
    six = (xl.double() * wh + xh.double() * wl + xm.double() * wm + xh.double() * wm + xm.double() * wh + xh.double() * wh)
    exact = x.double() * w.double()
    rel = ((six - exact).abs() / exact.abs().clamp_min(1e-300))[exact != 0]
    assert float(rel.max()) <= 2.0 ** -22, float(rel.max())
    # two pieces / three products would not do: ~2^-16
    three = xh.double() * wm + xm.double() * wh + xh.double() * wh
    assert float(((three - exact).abs() / exact.abs().clamp_min(1e-300))[exact != 0].max()) > 2.0 ** -18


def test_query_chunks_of_the_host_gpu_pipeline():
    """cone_amd.inference.query_chunks (host logic of predict_split's software pipeline): chunks are cut at multiples of
    eval_bsz (every reference batch -- hazard H3's padding unit -- stays inside one chunk); default = a 1/16 tail from 32
    reference batches on, one chunk for smaller splits, on request, under hipGraph replay and when the caller wants the
    per-window outputs of the whole split."""
    from cone_amd import inference as inf
    mk = lambda **kw: make_opt("ego4d", topk_window=20, eval_bsz=32, **kw)
    assert inf.query_chunks(1000, mk()) == [(0, 960), (960, 1000)]
    assert inf.query_chunks(20000, mk()) == [(0, 18752), (18752, 20000)]
    assert inf.query_chunks(500, mk()) == [(0, 500)]
    for kw in (dict(pipeline_tail=0.0), dict(hip_graph=True), dict(need_saliency=True), dict(need_aux=True)):
        assert inf.query_chunks(1000, mk(**kw)) == [(0, 1000)], kw
    assert inf.query_chunks(1000, mk(pipeline_tail=0.125)) == [(0, 896), (896, 1000)]
    assert inf.query_chunks(100, mk(pipeline_tail=0.125)) == [(0, 100)]
    five = inf.query_chunks(41, make_opt("ego4d", topk_window=6, eval_bsz=4, pipeline_chunks=5))
    assert len(five) == 5 and five[0][0] == 0 and five[-1][1] == 41
    assert all(a % 4 == 0 for a, _ in five) and all(b == five[i + 1][0] for i, (_, b) in enumerate(five[:-1]))


def test_bench_plain_entry_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` with no RANK in the environment (how the driver calls the one-GPU bench) must not die
    on a WORLD_SIZE assert: it starts torch.distributed.run as a child process BEFORE any GPU call, the ranks rendezvous
    on 127.0.0.1 and pass the collective preflight (here: gloo on CPU tensors, CONE_BENCH_LAUNCH_CHECK=1 stops there),
    rank 0's JSON line and the exit code are relayed.  A wrong --gpus / WORLD_SIZE pair exits non-zero with a reason."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CONE_BENCH_LAUNCH_CHECK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--queries", "64"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["ranks_seen"] == 2
    pf = res["collective_preflight"]
    assert pf["ok"] and pf["world"] == 2 and pf["bytes_per_rank"] == 3 * 64 * 5 * 5 * 8 + 3 * 64 * 4
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


def test_views_inherit_the_split_level_query_length_bound():
    """Kernel forms are chosen from Lv_max + Lq_max; the Lq bound handed to the library is the SPLIT's longest query
    (FeatureStore.max_tok_len), also for a pipeline chunk / rank shard whose own queries are all shorter: a window's bits must
    not depend on the chunk it rides in (the rows-once cross-attention switches forms at 128 tokens)."""
    from cone_amd import inference as inf
    opt = make_opt("ego4d", topk_window=3, eval_bsz=4)
    ann, vf, qf = synth.make_dataset(opt, 12, 2, seed=5, ctx_range=(30, 120), lq_range=(3, 9))
    qf[ann[-1]["query_id"]]["token_features"] = np.ones((17, opt.t_feat_dim), np.float32)
    store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
    assert store.max_tok_len == 17
    head = store.view(0, 8)
    assert max(head.tok_len) <= 8 and head.max_tok_len == 17 and head.view(2, 4).max_tok_len == 17
    assert inf.FeatureStore.subset(store, 4, 12).max_tok_len == 17


def test_store_views_are_cached_and_follow_replaced_arenas():
    """FeatureStore.view: the same (lo, hi) view object step after step (its static index tables are uploaded once), rebuilt
    when the parent's arenas are REPLACED by new tensors (an in-place refill keeps the view: it aliases the arena)."""
    from cone_amd import inference as inf
    opt = make_opt("ego4d", topk_window=3, eval_bsz=4)
    ann, vf, qf = synth.make_dataset(opt, 9, 2, seed=3, ctx_range=(30, 120))
    store = inf.FeatureStore(opt, ann, vf, qf, device=torch.device("cpu"))
    v1 = store.view(2, 7)
    assert store.view(2, 7) is v1 and v1.q_base == 2 and len(v1.ann) == 5
    assert v1.tok_raw.data_ptr() == store.tok_raw[int(store.tok_off[2]):].data_ptr()
    store.tok_raw.mul_(2.0)                                   # in place: the view sees it
    assert store.view(2, 7) is v1 and torch.equal(v1.tok_raw, store.tok_raw[int(store.tok_off[2]):int(store.tok_off[7])])
    store.tok_raw = store.tok_raw.clone()                     # replaced: the cached view would be stale
    v2 = store.view(2, 7)
    assert v2 is not v1 and v2.tok_raw.data_ptr() == store.tok_raw[int(store.tok_off[2]):].data_ptr()
    # the shape of the window list is host metadata: a query owns min(K, ceil(ctx_l / S) + 1) rows
    sel = inf.selection(store, opt)
    S = int(opt.max_v_l / 2)
    assert sel.n_q.tolist() == [min(3, -(-store.ctx_l[v] // S) + 1) for v in store.q_vid.tolist()]
    assert sel.row_off[-1] == sel.n_rows and [sel.query_of_row(r) for r in range(sel.n_rows)] == \
        [q for q, n in enumerate(sel.n_q.tolist()) for _ in range(n)]
