"""Builds librlr_gpu.so (HIP kernels + C ABI + host engine) for gfx950, in-tree.

    python rust-local-rag_amd/build.py [--force] [--keep-temps]

hipcc cross-compiles without a GPU, so this runs in the authoring container; the built
.so travels to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
SO = os.path.join(HERE, "librlr_gpu.so")
ROOT = os.path.dirname(HERE)

SOURCES = ["scan.hip", "select.hip", "tail.hip", "exact.hip", "gemm.hip", "index.hip", "engine.cpp", "multi.cpp", "lexical.hip", "q8.hip",
           "jsonio.cpp"]
HEADERS = ["common.h", "kernels.h", "exact_dot.h", "lds_select.h", "staged_dot.h", "select_dev.h", "sort_emit.h", "pool_prepare.h",
           "engine_host.h", "lexical_internal.h", os.path.join(ROOT, "include", "rlr_gpu.h"),
           os.path.join(ROOT, "include", "rlr_engine.h"), os.path.join(ROOT, "include", "rlr_lexical.h")]

# -ffp-contract=off: a*b+c in source is a rounded multiply then a rounded add (the
# reference's arithmetic); FMAs appear only where the source spells fmaf().
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-I", os.path.join(ROOT, "include")]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_header() -> float:
    t = 0.0
    for h in HEADERS:
        p = h if os.path.isabs(h) else os.path.join(CSRC, h)
        if os.path.exists(p):
            t = max(t, os.path.getmtime(p))
    return t


def _compile(src: str, force: bool, keep_temps: bool) -> str:
    sp = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), _newest_header())
    if stale:
        cmd = [_hipcc(), *FLAGS, "-c", sp, "-o", obj]
        if src.endswith(".cpp"):
            cmd.insert(1, "-x")
            cmd.insert(2, "hip")
        if keep_temps:
            cmd += ["-save-temps=obj"]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


def build(force: bool = False, keep_temps: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, keep_temps), srcs))
    if force or not os.path.exists(SO) or any(os.path.getmtime(o) > os.path.getmtime(SO) for o in objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv))
