#!/usr/bin/env python
"""Build a second copy of libflairhip.so with extra -D flags on ONE source (kernel A/B runs in one gpurun call):

  python tools/build_variant.py <name> <source.hip> -DFOO=1 [-DBAR=2 ...]   ->  csrc/build/libflairhip_<name>.so

Select it with FLAIRHIP_LIB=<path> (flairhip/lib.py).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))


def main():
    name, src, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
    from flairhip import build as B
    B.build()
    out_dir = os.path.join(B.CSRC, "build")
    obj = os.path.join(out_dir, f"{os.path.splitext(src)[0]}_{name}.o")
    so = os.path.join(out_dir, f"libflairhip_{name}.so")
    flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc"] + defs
    subprocess.run([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
    objs = [os.path.join(out_dir, os.path.splitext(s)[0] + ".o") for s in B.HIP_SOURCES + B.CXX_SOURCES if s != src]
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", so] + objs + [obj], check=True)
    print(so)


if __name__ == "__main__":
    main()
