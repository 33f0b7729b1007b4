"""Test-only stand-in for the ``lmdb`` package (not in this image): the slice of its API the reference's
dataloaders use -- ``lmdb.open(path, readonly=True, create=False, max_readers=..., readahead=False)``,
``env.begin(buffers=True)`` and ``txn.get(key_bytes)`` returning a buffer (None for a missing key)
(cone/ego4d_mad_dataloader.py:73-85, 263-302).  A "database" is a directory holding ``data.bin`` (the blobs back
to back) and ``index.json`` (key -> [offset, size]); ``write_env`` creates one with values that are ``np.savez``
blobs exactly as the reference's converters write them (feature_extraction/misc/convert_h5_to_lmdb.py:38-40,
feature_extraction/ego4d_merge_textual_cls_token_feature.py:45-47).

Install with ``sys.modules["lmdb"] = fake_lmdb`` (pytest: ``monkeypatch.setitem(sys.modules, "lmdb", fake_lmdb)``)."""
import io
import json
import os

import numpy as np


class Error(Exception):
    pass


class _Txn:
    def __init__(self, env, buffers):
        self.env, self.buffers = env, buffers

    def get(self, key, default=None):
        ent = self.env.index.get(bytes(key).decode())
        if ent is None:
            return default
        view = memoryview(self.env.blob)[ent[0]:ent[0] + ent[1]]
        return view if self.buffers else bytes(view)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class Environment:
    def __init__(self, path, readonly=False, create=True, **kwargs):
        if not os.path.isdir(path):
            raise Error(f"{path}: No such file or directory")
        with io.open(os.path.join(path, "index.json")) as f:
            self.index = json.load(f)
        with io.open(os.path.join(path, "data.bin"), "rb") as f:
            self.blob = f.read()

    def begin(self, buffers=False, write=False, **kwargs):
        return _Txn(self, buffers)

    def close(self):
        pass


def open(path, **kwargs):  # noqa: A001  (the package's own name)
    return Environment(path, **kwargs)


def write_env(path, entries):
    """entries: key -> dict of arrays; each value is stored as one compressed ``np.savez`` blob."""
    os.makedirs(path, exist_ok=True)
    index, off = {}, 0
    with io.open(os.path.join(path, "data.bin"), "wb") as f:
        for key, arrays in entries.items():
            buf = io.BytesIO()
            np.savez_compressed(buf, **arrays)
            b = buf.getvalue()
            f.write(b)
            index[key] = [off, len(b)]
            off += len(b)
    with io.open(os.path.join(path, "index.json"), "w") as f:
        json.dump(index, f)
    return path
