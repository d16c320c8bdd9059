#!/usr/bin/env python
"""Phase timeline of conv_igemm_kernel blocks (developer tool, MI355X only).

Builds a second copy of the library with -DFFA_CONV_TRACE=1 (wave 0 of the first 64 blocks logs s_memtime at
every phase boundary), runs one layer shape and prints where a block's cycles go:

  python tools/conv_trace.py conv512            # shapes: see tools/bench_kernels.py CONV_SHAPES

Phases per 32-byte k-step chunk: issue (global loads of later chunks), compute (MFMAs + LDS fragment reads),
bar1 (wait for the other waves), store (registers -> LDS), bar2.
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
    obj = os.path.join(out_dir, "conv_igemm_trace.o")
    so = os.path.join(out_dir, "libflairhip_trace.so")
    # FFA_CONV_TRACE_FIRST=<block id> traces 64 blocks of a later dispatch round (steady state) instead of the first
    flags = [f"--offload-arch={B.ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-DFFA_CONV_TRACE=1",
             "-DFFA_WGRAD_TRACE=1", f"-DFFA_CONV_TRACE_FIRST={int(os.environ.get('FFA_CONV_TRACE_FIRST', '0'))}"]
    obj2 = os.path.join(out_dir, "conv_wgrad_trace.o")
    subprocess.run([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, "conv_igemm.hip"), "-o", obj], check=True)
    subprocess.run([B._hipcc()] + flags + ["-c", os.path.join(B.CSRC, "conv_wgrad.hip"), "-o", obj2], check=True)
    objs = [os.path.join(out_dir, os.path.splitext(s)[0] + ".o") for s in B.HIP_SOURCES + B.CXX_SOURCES
            if s not in ("conv_igemm.hip", "conv_wgrad.hip")] + [obj, obj2]
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", so] + objs, check=True)
    return so


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--build-only":
        print(build_trace_lib())
        return
    shape = sys.argv[1] if len(sys.argv) > 1 else "conv128"
    kind = sys.argv[2] if len(sys.argv) > 2 else "fwd"
    so = os.path.join(PKG, "csrc", "build", "libflairhip_trace.so")
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
    Ho = (H + 2 * pad - k) // stride + 1
    x = torch.randn(B, H, H, cip, device=dev).to(dt)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    dy = torch.randn(B, Ho, Ho, cop, device=dev).to(dt)
    read, clear = lib.ffa_conv_trace_read, lib.ffa_conv_trace_clear
    names = {1: "prologue", 2: "issue", 3: "compute", 4: "bar1", 5: "store", 6: "bar2", 7: "epilogue"}
    if kind == "wgrad":
        dw = torch.empty(cout, cin, k, k, device=dev)
        run = lambda: ops.conv_wgrad(x, dy, cout, cin, k, k, stride, pad, out=dw)
        read, clear = lib.ffa_wgrad_trace_read, lib.ffa_wgrad_trace_clear
        names = {1: "prologue", 2: "bar1", 3: "store", 4: "bar2", 5: "issue", 6: "compute", 7: "epilogue"}
    elif kind == "fwd":
        pw = ops.pack_conv_weight(w, dt, stride, cip)
        run = lambda: ops.conv2d(x, pw, pad, cop)
    else:
        pw = ops.pack_conv_weight(w, dt, stride, cop, transpose=True)
        run = lambda: ops.conv2d(dy, pw, k - 1 - pad, cip, dil=stride, out_hw=(H, H))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    clear()
    run()
    torch.cuda.synchronize()
    buf = (C.c_longlong * (64 * 256))()
    read.argtypes = [C.c_void_p, C.c_int]
    read(buf, 64 * 256)
    totals, blocks, spans = {}, 0, []
    for b in range(64):
        ev = [(v >> 56, v & ((1 << 56) - 1)) for v in buf[b * 256:(b + 1) * 256] if v]
        if len(ev) < 3:
            continue
        blocks += 1
        spans.append(ev[-1][1] - ev[0][1])
        for (s0, t0), (s1, t1) in zip(ev, ev[1:]):
            totals[s1] = totals.get(s1, 0) + (t1 - t0)
    if not blocks:
        print("no trace events (was the library built with FFA_CONV_TRACE=1?)")
        return
    span = sum(spans) / blocks
    print(f"{shape} {kind}: {blocks} blocks traced, mean block lifetime {span:.0f} cycles "
          f"(min {min(spans)}, max {max(spans)})")
    for s in sorted(totals):
        print(f"  {names.get(s, s):9s} {totals[s] / blocks:9.0f} cycles  {100.0 * totals[s] / blocks / span:5.1f} %")
    ev = [(v >> 56, v & ((1 << 56) - 1)) for v in buf[0:256] if v]
    print("  block 0 timeline (phase:+cycles):", " ".join(f"{str(names.get(s, s))[:4]}:{t - ev[0][1]}" for s, t in ev[:40]))


if __name__ == "__main__":
    main()
