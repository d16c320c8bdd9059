"""Build libflairhip.so (hipcc, gfx950 only) in-tree, next to this file.

The shared library is the product: the Python layer refuses to run without it (no CPU fallback).
Objects go to csrc/build/; the .so is git-ignored but travels to the GPU box with the snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
OUT = os.path.join(HERE, "libflairhip.so")
OBJ = os.path.join(CSRC, "build")

HIP_SOURCES = ["ffa_runtime.hip", "conv_igemm.hip", "conv3x3_ring.hip", "conv3x3_thin.hip", "conv7x7_stem.hip", "conv_wgrad.hip", "norm_pool.hip", "resample_loss.hip", "temporal.hip", "transformer.hip", "gemm.hip", "optim.hip"]
CXX_SOURCES = ["tile_grid.cpp", "tiff_codec.cpp"]
HEADERS = ["ffa_common.h", "ffa_common_host.h", os.path.join("..", "..", "include", "flairhip.h")]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _compile(src: str, flags) -> str:
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    stamp = obj + ".stamp"
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS]
    key = _digest(deps) + " " + " ".join(flags)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj
    cmd = [_hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{proc.stderr}")
    with open(stamp, "w") as f:
        f.write(key)
    return obj


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hip_flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"]
    cxx_flags = ["-O2", "-std=c++17", "-fPIC", "-x", "c++"]
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    jobs = [(s, hip_flags) for s in HIP_SOURCES] + [(s, cxx_flags) for s in CXX_SOURCES]
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda j: _compile(*j), jobs))
    link_key = _digest(objs)
    stamp = OUT + ".stamp"
    if not (os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == link_key):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"link failed:\n{proc.stderr}")
        with open(stamp, "w") as f:
            f.write(link_key)
    if verbose:
        print(f"[flairhip] built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
