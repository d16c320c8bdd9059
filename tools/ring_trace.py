#!/usr/bin/env python
"""Where a conv3x3_ring_kernel block spends its cycles, and the clock the chip holds under it (MI355X only).

Builds a second copy of the library with -DFFA_RING_TRACE=1 (wave 0 of every block stamps s_memtime /
s_memrealtime at block start and end and sums the cycles spent in the prologue, at phase-end synchronisations and in
tile epilogues) and runs one layer shape of tools/bench_kernels.py:

  python tools/ring_trace.py conv128 [fwd|dgrad]
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "flair-for-aigle_amd")
sys.path.insert(0, PKG)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build_trace_lib() -> str:
    from flairhip import build as B
    B.build()
    out_dir = os.path.join(B.CSRC, "build")
    obj = os.path.join(out_dir, "conv3x3_ring_trace.o")
    so = os.path.join(out_dir, "libflairhip_ringtrace.so")
    flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DFFA_RING_TRACE=1"]
    subprocess.run([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, "conv3x3_ring.hip"), "-o", obj], check=True)
    objs = [os.path.join(out_dir, os.path.splitext(s)[0] + ".o") for s in B.HIP_SOURCES + B.CXX_SOURCES
            if s != "conv3x3_ring.hip"] + [obj]
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", so] + objs, check=True)
    return so


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
        print(build_trace_lib())
        return
    shape = sys.argv[1] if len(sys.argv) > 1 else "conv128"
    kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    so = os.path.join(PKG, "csrc", "build", "libflairhip_ringtrace.so")
    if not os.path.exists(so):
        so = build_trace_lib()
    from flairhip import lib as L
    L.LIB_PATH = so
    import torch
    from flairhip import ops
    import bench_kernels as BK
    lib = L.load()
    name, cin, cout, k, stride, pad, H = next(s for s in BK.CONV_SHAPES if s[0] == shape)
    dev, dt, B = torch.device("cuda:0"), torch.bfloat16, BK.B
    cip, cop = ops.pad_channels(cin), ops.pad_channels(cout)
    x = torch.randn(B, H, H, cip, device=dev).to(dt)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    dy = torch.randn(B, H, H, cop, device=dev).to(dt)
    if kind == "fwd":
        pw = ops.pack_conv_weight(w, dt, stride, cip)
        run = lambda: ops.conv2d(x, pw, pad, cop)
    else:
        pw = ops.pack_conv_weight(w, dt, stride, cop, transpose=True)
        run = lambda: ops.conv2d(dy, pw, k - 1 - pad, cip)
    # two seconds of back-to-back launches first: the clock under sustained load is what matters
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    import time
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.0:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
        n += 50
    us = BK.timeit(run, iters=20)
    read = lib.ffa_ring_trace_read
    read.argtypes = [C.c_void_p, C.c_int]
    buf = (C.c_longlong * (1024 * 8))()
    read(buf, 1024 * 8)
    rows = [buf[i * 8:(i + 1) * 8] for i in range(1024) if buf[i * 8 + 1]]
    life = [r[1] - r[0] for r in rows]
    real = [r[3] - r[2] for r in rows]
    clk = sorted(l / max(r_, 1) * 100e6 for l, r_ in zip(life, real))
    t_first, t_last = min(r[0] for r in rows), max(r[1] for r in rows)
    m = lambda v: sum(v) / len(v)
    print(f"{shape} {kind}: {us:.1f} us per launch, {len(rows)} blocks traced")
    print(f"  in-kernel clock (s_memtime / s_memrealtime): median {clk[len(clk) // 2] / 1e9:.3f} GHz "
          f"(min {clk[0] / 1e9:.3f}, max {clk[-1] / 1e9:.3f})")
    print(f"  first block start -> last block end: {t_last - t_first} cycles; mean block lifetime {m(life):.0f} "
          f"(min {min(life)}, max {max(life)})")
    print(f"  per block: prologue {m([r[6] for r in rows]):.0f}, phase-end waits {m([r[4] for r in rows]):.0f} over "
          f"{m([r[7] for r in rows]):.0f} phases ({m([r[4] / max(r[7], 1) for r in rows]):.0f} per phase), "
          f"epilogues {m([r[5] for r in rows]):.0f}")


if __name__ == "__main__":
    main()
