"""Builds libmt3hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

`python -m mt3_amd.build` or `__graft_entry__.build()`.  The .so is git-ignored but
travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmt3hip.so")
OBJ = os.path.join(ROOT, "build", "obj")

HIP_SOURCES = ["frontend.hip", "gemm.hip", "gemm_mx8.hip", "attention.hip", "enc_attention_x6.hip", "decode_ops.hip",
               "engine.hip"]
CPP_SOURCES = ["errors.cpp", "symbolic.cpp", "mx8_host.cpp"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm: /opt/rocm/bin/hipcc)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "mt3_hip.h"))
    return hs


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    if not force and not _stale(obj, [src] + _headers()):
        return obj
    inc = ["-I", os.path.join(ROOT, "include"), "-I", CSRC]
    if src.endswith(".hip"):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *inc, "-c", src, "-o", obj]
    else:
        # host-only C++: doubles must be evaluated exactly as written (symbolic.cpp)
        cmd = [_hipcc(), "-x", "c++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", *inc, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in HIP_SOURCES + CPP_SOURCES]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    if verbose:
        print("built", LIB, os.path.getsize(LIB), "bytes")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)
