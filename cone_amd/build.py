"""Build libcone_hip.so (hipcc, gfx950 only) in-tree: ``python -m cone_amd.build``.

Objects are cached under ``cone_amd/csrc/_obj`` keyed by source mtime; the shared library lands
next to this file so that it travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcone_hip.so")
PYLISTS = os.path.join(HERE, "_cone_pylists.so")      # host helper (CPython C API): kept rows -> submission lists
SOURCES = ["api.hip", "gemm.hip", "rowops.hip", "attention.hip", "window_ops.hip", "prefilter.hip",
           "postproc.hip", "prof.hip", "dec_cross.hip", "metrics.hip", "ffn.hip", "criterion.hip", "dec_cross_mfma.hip", "ffn_split.hip", "ffn_wide.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# Files whose scalar arithmetic must round operation by operation like the reference's torch / Python
# code (HIP's __fmul_rn & co. are plain operators and would otherwise be contracted into fma).
NO_CONTRACT = {"window_ops.hip", "postproc.hip", "metrics.hip", "criterion.hip"}


def _deps_mtime():
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "cone_hip.h"), __file__]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".flags"          # the exact flag list the object was compiled with: an A/B build with other
    path = os.path.join(CSRC, src)  # CONE_HIPCC_FLAGS (thresholds are -D macros) never reuses a stale object
    extra = ["-ffp-contract=off"] if src in NO_CONTRACT else []
    extra += os.environ.get("CONE_HIPCC_FLAGS", "").split()     # A/B builds on the GPU box (tools/ab_variants.sh)
    flags = " ".join([*FLAGS, *extra])
    try:
        with open(stamp) as f:
            same_flags = f.read() == flags
    except OSError:
        same_flags = False
    if same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime()):
        return obj
    cmd = ["hipcc", *FLAGS, *extra, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(flags)
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(_compile, SOURCES))
    if (not os.path.exists(LIB)) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    try:
        build_pylists()
    except (RuntimeError, OSError) as e:    # no gcc / no Python.h on the box: the HIP library is complete without it,
        try:                                # a helper built from an OLDER pylists.c must not be picked up in its place
            if os.path.exists(PYLISTS) and os.path.getmtime(PYLISTS) < os.path.getmtime(os.path.join(CSRC, "pylists.c")):
                os.remove(PYLISTS)
        except OSError:
            pass
        sys.stderr.write(f"cone_amd.build: _cone_pylists.so not built ({str(e).splitlines()[0]}); "      # the host falls
                         "submission lists will be built by the Python loop\n")                         # back (slower)
    return LIB


def build_pylists() -> str:
    """csrc/pylists.c -> _cone_pylists.so (gcc; plain C against the interpreter's own headers, no GPU code)."""
    import sysconfig
    src = os.path.join(CSRC, "pylists.c")
    if os.path.exists(PYLISTS) and os.path.getmtime(PYLISTS) >= max(os.path.getmtime(src), os.path.getmtime(__file__)):
        return PYLISTS
    r = subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-I", sysconfig.get_paths()["include"], src, "-o", PYLISTS],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"gcc failed for pylists.c:\n{r.stdout}\n{r.stderr}")
    return PYLISTS


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
