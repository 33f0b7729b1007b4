"""Convert the reference's feature stores (LMDBs of compressed ``np.savez`` blobs, keys ``features`` /
``token_features`` + ``cls_features``|``eot_features`` -- feature_extraction/misc/convert_h5_to_lmdb.py:38-40,
feature_extraction/ego4d_merge_textual_cls_token_feature.py:45-47) of one annotation file into ONE packed arena
file in the device layout:

    python -m cone_amd.pack_features --resume <ckpt> --eval_path <jsonl> --eval_split_name val --out val.conefs
    python -m cone_amd.inference     --resume <ckpt> --eval_path <jsonl> --eval_split_name val --eval_id x \\
                                     --packed_features val.conefs

Needs the `lmdb` package for reading; runs on the CPU (no GPU involved).
"""
from __future__ import annotations

import sys

import torch

from .config import parse_test_options


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--out" not in argv:
        raise SystemExit("usage: python -m cone_amd.pack_features <inference options> --out FILE")
    i = argv.index("--out")
    out = argv[i + 1]
    del argv[i:i + 2]
    opt = parse_test_options(argv)
    from .inference import FeatureStore
    store = FeatureStore.from_lmdb(opt, device=torch.device("cpu"))
    store.save_packed(out)
    print(f"wrote {out}: {len(store.ann)} queries, {len(store.clip_ids)} videos, "
          f"{store.vid_raw.shape[0]} clips x {store.vid_raw.shape[1]}")


if __name__ == "__main__":
    main()
