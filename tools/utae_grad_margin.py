#!/usr/bin/env python
"""Diagnostic for tests/test_utae_gpu.py::test_utae_training_step_matches_the_references_own_autograd: prints, per
parameter, the gradient-norm ratio and cosine against the reference's own autograd (tests/golden/utae_train.npz),
worst first, plus the element-wise differences of the worst one -- to tell a summation-order effect (ReLU masks of
activations within rounding distance of zero flip: whole elements of a small gradient move) from a kernel error.

  python tools/utae_grad_margin.py [fp32|bf16]        (FLAIRHIP_LIB selects a library variant)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import test_utae_gpu as T  # noqa: E402
from flairhip import nn as hnn  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"

if "--second-lib" in sys.argv:  # every BatchNorm call also runs on a second build of the library, outputs compared
    import ctypes
    from flairhip import lib as _l, ops as _ops2
    _main = _l.load()
    _second = ctypes.CDLL(sys.argv[sys.argv.index("--second-lib") + 1])
    for _name, (_res, _args) in _l.SIGNATURES.items():
        _fn = getattr(_second, _name)
        _fn.restype, _fn.argtypes = _res, _args

    def _both(name):
        orig = getattr(_ops2, name)

        def wrapped(*a, **kw):
            a2 = [t.clone() if (name == "bn_stats" and i in (3, 4) and t is not None) else t for i, t in enumerate(a)]
            _l._lib = _second
            try:
                ref = orig(*a2, **kw)
            finally:
                _l._lib = _main
            out = orig(*a, **kw)
            ro = ref if isinstance(ref, tuple) else (ref,)
            oo = out if isinstance(out, tuple) else (out,)
            worst = 0.0
            for r, o in zip(ro, oo):
                if r is None:
                    continue
                worst = max(worst, (r.double() - o.double()).abs().max().item() / max(1e-12, r.double().abs().max().item()))
            shp = tuple(a[0].shape)
            flag = "  <-- differs" if worst > 1e-5 else ""
            if name == "bn_stats":  # (scale, shift, mean, rstd) of both builds against the float64 statistics of the input
                xd = a[0].double().reshape(-1, a[0].shape[-1])
                tm, tv = xd.mean(0), xd.var(0, unbiased=False)
                tr = 1.0 / torch.sqrt(tv + a[6])
                em = ((out[2].double() - tm).abs() / tm.abs().clamp_min(1e-3)).max().item()
                er_main = ((out[3].double() - tr).abs() / tr).max().item()
                er_second = ((ref[3].double() - tr).abs() / tr).max().item()
                c = ((out[3].double() - tr).abs() / tr).argmax().item()
                print(f"   rstd rel err vs float64: main {er_main:.2e} second {er_second:.2e} (channel {c}: mean {tm[c].item():.4g} var {tv[c].item():.4g}); mean err {em:.1e}")
            print(f"{name:10s} {shp} extra={[type(v).__name__ if not isinstance(v, (bool, int, float)) else v for v in a[1:]][-3:]} worst rel diff {worst:.2e}{flag}")
            return out
        setattr(_ops2, name, wrapped)

    for _n in ("bn_stats", "bn_apply", "bn_bwd", "channel_sums", "maxpool3x3s2_fwd", "nchw_to_nhwc", "nhwc_to_nchw"):
        _both(_n)

if "--check-bn" in sys.argv:  # every BatchNorm call of the step against a float64 torch evaluation of the same call
    from flairhip import ops as _ops
    _bwd, _stats, _apply, _sums = _ops.bn_bwd, _ops.bn_stats, _ops.bn_apply, _ops.channel_sums

    def _rel(a, b):
        return (a.double().cpu() - b.double().cpu()).abs().max().item() / max(1e-12, b.double().abs().max().item())

    def bn_bwd(x, dy, y, gamma, beta, mean, rstd, relu, want_dres):
        out = _bwd(x, dy, y, gamma, beta, mean, rstd, relu, want_dres)
        C = x.shape[-1]
        xd, gd = x.double().reshape(-1, C), dy.double().reshape(-1, C)
        xh = (xd - mean.double()) * rstd.double()
        g = gd
        if relu:
            yy = y.double().reshape(-1, C) if y is not None else xh * gamma.double() + beta.double()
            g = gd * (yy > 0)
        dbeta, dgamma = g.sum(0), (g * xh).sum(0)
        n = xd.shape[0]
        dx = gamma.double() * rstd.double() * (g - dbeta / n - xh * dgamma / n)
        print(f"bn_bwd   {tuple(x.shape)} relu={relu} y={'y' if y is not None else '-'} x.ptr%16={x.data_ptr() % 16} dy.ptr%16={dy.data_ptr() % 16} "
              f"contig={x.is_contiguous()},{dy.is_contiguous()}  dx {_rel(out[0].reshape(-1, C), dx):.1e} dgamma {_rel(out[2], dgamma):.1e} dbeta {_rel(out[3], dbeta):.1e}")
        return out

    def channel_sums(x):
        out = _sums(x)
        C = x.shape[-1]
        print(f"chan_sums {tuple(x.shape)} contig={x.is_contiguous()} sum {_rel(out[0], x.double().reshape(-1, C).sum(0)):.1e}")
        return out

    _ops.bn_bwd, _ops.channel_sums = bn_bwd, channel_sums
cuda = torch.device("cuda:0")
net, _ = T._model(cuda, precision)
net.train()
net.mlp_dropout = net.attn_dropout = 0.0
d = np.load(os.path.join(T.GOLD, "utae_train.npz"))
x, pos, tgt = (torch.tensor(d[k]).to(cuda) for k in ("x", "pos", "target"))
logits_nhwc, maps, attn = net.forward_nhwc(x, pos)
loss = hnn.HipCrossEntropyLoss(num_classes=19).to(cuda)(hnn.logits_view(logits_nhwc, 19), tgt)
loss.backward()
torch.cuda.synchronize()
rows = []
for k, p in net.named_parameters():
    g = p.grad.detach().float().cpu().flatten()
    ref_norm = float(d["norm__" + k])
    ref = torch.tensor(d["grad__" + k])
    if ref_norm < 1e-7:
        continue
    sample = g if g.numel() <= 4096 else g[:: max(1, g.numel() // 2048)]
    rows.append((F.cosine_similarity(sample, ref, dim=0).item(), g.double().norm().item() / ref_norm, k, sample, ref))
if "--dump" in sys.argv:
    tag = sys.argv[sys.argv.index("--dump") + 1]
    torch.save({k: p.grad.detach().float().cpu() for k, p in net.named_parameters()},
               os.path.join(ROOT, "gpurun_out", f"utae_grads_{tag}.pt"))
if "--diff" in sys.argv:
    a, b = sys.argv[sys.argv.index("--diff") + 1: sys.argv.index("--diff") + 3]
    ga, gb = (torch.load(os.path.join(ROOT, "gpurun_out", f"utae_grads_{t}.pt")) for t in (a, b))
    for k in ga:
        dd = (ga[k] - gb[k]).abs()
        if dd.max() > 1e-5 * gb[k].abs().max():
            idx = dd.flatten().argmax().item()
            print(f"{k:50s} shape {tuple(ga[k].shape)} max|d| {dd.max():.3e} of {gb[k].abs().max():.3e} at flat {idx}; nonzero diffs {(dd > 1e-6 * gb[k].abs().max()).sum().item()}")
    sys.exit(0)
rows.sort(key=lambda r: r[0])
print(f"loss {loss.item():.7f} (reference {float(d['loss']):.7f})")
for cos, ratio, k, _, _ in rows[:6]:
    print(f"cos {cos:.5f} ratio {ratio:.4f}  {k}")
cos, ratio, k, s, r = rows[0]
print("worst:", k)
print(" got", np.array2string(s.numpy()[:24], precision=5))
print(" ref", np.array2string(r.numpy()[:24], precision=5))
