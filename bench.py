#!/usr/bin/env python
"""Headline benchmark: 512x512x5 tiles/sec, training step (fwd + bwd + AdamW) of the 19-class U-Net
(ResNet-34 encoder) on synthetic tiles, data-parallel over N MI355X (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of 32 tiles per GPU (weak scaling): NCHW f32 batch ->
NHWC bf16, encoder / decoder conv stack (HIP implicit-GEMM on MFMA), BN / pool / upsample kernels, fused
softmax-CE (+argmax), full backward, gradient mean over ranks (RCCL, overlapped), AdamW + OneCycleLR.
Inputs are generated on the device before the timed region (random, never zeros: DVFS).

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel symbol of the step, timed live with HIP events on the stream the
                  kernels run on: achieved = algorithmic FLOPs of its launches / their summed duration
  cpu_baseline -- the oracle (torch-CPU fp32 restatement of the same step, kind "port") timed on this
                  box's host cores on a bounded sample (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3
TASK = "AERIAL_LABEL-COSIA"
MOD = "AERIAL_RGBI"
SETTLE_MAX_STEPS = 200  # untimed steps the settle phase may take before the counted warm-up


def _hk_tag(w, dtype) -> str:
    """the halo-depth template parameter the library picks for a 3x3 stride-1 operand (conv_igemm.hip launch_cfg):
    part of the kernel symbol, so the bench groups launches the way rocprofv3 does"""
    if not (w.kh == 3 and w.kw == 3 and w.stride == 1):
        return ""
    nchunks = w.ci_pitch * (2 if dtype == torch.bfloat16 else 4) // 32
    return ",hk4" if nchunks % 4 == 0 else (",hk2" if nchunks % 2 == 0 else ",hk1")


class KernelTimer:
    """HIP-event timing of the MFMA kernels, keyed by kernel symbol (template instantiation)."""

    def __init__(self):
        self.records = []  # (symbol, flops, start, end)
        self.hbm_records = []  # (symbol, algorithmic bytes, start, end): single-kernel HBM-bound launches
        self.enabled = False

    def install(self):
        from flairhip import ops
        timer = self
        orig_conv, orig_wgrad = ops.conv2d, ops.conv_wgrad

        def conv2d(x, w, pad, out_channels, *a, **kw):
            if not timer.enabled:
                return orig_conv(x, w, pad, out_channels, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_conv(x, w, pad, out_channels, *a, **kw)
            e.record()
            dil = kw.get("dil", 1)
            B, Ho, Wo = y.shape[0], y.shape[1], y.shape[2]
            flops = 2.0 * B * (Ho * Wo / (dil * dil)) * w.rows_real * w.ch_real * w.kh * w.kw
            tile = "8x32" if Wo >= 32 else "16x16"
            dt = "bf16" if x.dtype == torch.bfloat16 else "f32"
            if w.bco & 0x2000:  # thin-layout operand: conv3x3_thin_kernel (<= 32 channels in, <= 32 rows)
                sym = f"conv3x3_thin_kernel<bf16,ci{w.ci_pitch},rows{w.rows}>"
            elif w.bco & 0x1000:  # ring-layout operand: conv3x3_ring16_kernel (bf16, the default) / conv3x3_ring_kernel (f32)
                sym = (f"conv3x3_ring16_kernel<bf16,co64,{tile}>" if dt == "bf16"
                       else f"conv3x3_ring_kernel<{dt},{tile}{_hk_tag(w, x.dtype)}>")
            elif ops.conv_is_persistent(x.dtype, B, Ho, Wo, w, dil):  # blocks walking several tiles: its own symbol
                sym = f"conv3x3_persist_kernel<{dt},bco{w.bco},{tile}{_hk_tag(w, x.dtype)}>"
            else:
                sym = f"conv_igemm_kernel<{dt},{w.kh}x{w.kw},s{w.stride},bco{w.bco},{tile}{_hk_tag(w, x.dtype)}>"
            timer.records.append((sym, flops, s, e))
            return y

        def conv_wgrad(x, dy, co_real, ci_real, kh, kw_, stride, pad, *a, **kw):
            if not timer.enabled:
                return orig_wgrad(x, dy, co_real, ci_real, kh, kw_, stride, pad, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_wgrad(x, dy, co_real, ci_real, kh, kw_, stride, pad, *a, **kw)
            e.record()
            B, Ho, Wo = dy.shape[0], dy.shape[1], dy.shape[2]
            flops = 2.0 * B * Ho * Wo * co_real * ci_real * kh * kw_
            dt = "bf16" if x.dtype == torch.bfloat16 else "f32"
            wco = 2 if (dy.shape[-1] > 32 or not (kh == 3 and stride == 1)) else 1
            wci = 2 if (x.shape[-1] > 32 and dt == "bf16") else 1
            sym = f"conv_wgrad_kernel<{dt},{kh}x{kw_},s{stride},w{wco}x{wci}>+reduce"
            if dt == "bf16" and kh == 3 and stride == 1 and dy.shape[-1] <= 32 and x.shape[-1] <= 32:
                sym = "conv3x3_thin_wgrad_kernel<bf16>+reduce"
            timer.records.append((sym, flops, s, e))
            return r

        orig_dgrad_up, orig_up = ops.conv2d_dgrad_upcat, ops.conv2d_upcat

        def conv2d_dgrad_upcat(dy, wt, c1, c2):
            # same kernel instantiation as the plain 3x3 dgrad (the fused 2x2 pooling lives in its epilogue)
            if not timer.enabled:
                return orig_dgrad_up(dy, wt, c1, c2)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_dgrad_up(dy, wt, c1, c2)
            e.record()
            if r is not None:
                B, H, W = dy.shape[0], dy.shape[1], dy.shape[2]
                dt = "bf16" if dy.dtype == torch.bfloat16 else "f32"
                sym = f"conv_igemm_kernel<{dt},3x3,s1,bco{wt.bco},{'8x32' if W >= 32 else '16x16'}{_hk_tag(wt, dy.dtype)}>"
                if wt.bco & 0x2000:
                    sym = f"conv3x3_thin_kernel<bf16,ci{wt.ci_pitch},rows{wt.rows},pool>"
                timer.records.append((sym, 2.0 * B * H * W * wt.rows_real * wt.ch_real * 9, s, e))
            return r

        def conv2d_upcat(lo, skip, w, out_channels, *a, **kw):
            # the two-source instantiation (template flag UP): its own symbol in rocprofv3 as well
            if not timer.enabled:
                return orig_up(lo, skip, w, out_channels, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_up(lo, skip, w, out_channels, *a, **kw)
            e.record()
            if y is not None:
                B, H, W = y.shape[0], y.shape[1], y.shape[2]
                dt = "bf16" if lo.dtype == torch.bfloat16 else "f32"
                sym = f"conv_igemm_kernel<{dt},3x3,s1,bco{w.bco},{'8x32' if W >= 32 else '16x16'}{_hk_tag(w, lo.dtype)},up>"
                if w.bco & 0x2000:
                    sym = f"conv3x3_thin_kernel<bf16,ci{w.ci_pitch},rows{w.rows},up>"
                timer.records.append((sym, 2.0 * B * H * W * w.rows_real * w.ch_real * 9, s, e))
            return y

        # "normalise on load" forms (the BatchNorm + ReLU of the producing layer evaluated by the consumer): the same
        # kernel families with the prologue template flag
        orig_pro, orig_wpro = ops.conv2d_pro, ops.conv_wgrad_pro

        def conv2d_pro(x, w, out_channels, *a, **kw):
            if not timer.enabled:
                return orig_pro(x, w, out_channels, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_pro(x, w, out_channels, *a, **kw)
            e.record()
            B, Ho, Wo = y.shape[0], y.shape[1], y.shape[2]
            if w.bco & 0x2000:
                sym = f"conv3x3_thin_kernel<bf16,ci{w.ci_pitch},rows{w.rows}{',up' if kw.get('up') else ''},pro>"
            else:
                sym = f"conv3x3_ring16_kernel<bf16,co64,{'8x32' if Wo >= 32 else '16x16'}>"  # (+pro: same family)
            timer.records.append((sym, 2.0 * B * Ho * Wo * w.rows_real * w.ch_real * 9, s, e))
            return y

        def conv_wgrad_pro(x, dy, co_real, ci_real, *a, **kw):
            if not timer.enabled:
                return orig_wpro(x, dy, co_real, ci_real, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_wpro(x, dy, co_real, ci_real, *a, **kw)
            e.record()
            B, Ho, Wo = dy.shape[0], dy.shape[1], dy.shape[2]
            thin = dy.shape[-1] <= 32 and x.shape[-1] <= 32
            sym = ("conv3x3_thin_wgrad_kernel<bf16>+reduce" if thin else "conv_wgrad_kernel<bf16,3x3,s1,w2x2>+reduce")
            timer.records.append((sym, 2.0 * B * Ho * Wo * co_real * ci_real * 9, s, e))
            return r

        ops.conv2d, ops.conv_wgrad = conv2d, conv_wgrad
        ops.conv2d_dgrad_upcat, ops.conv2d_upcat = conv2d_dgrad_upcat, conv2d_upcat
        ops.conv2d_pro, ops.conv_wgrad_pro = conv2d_pro, conv_wgrad_pro

        # ---- HBM-bound kernels, one kernel per bracket (algorithmic bytes = every tensor read or written once) ----
        orig_bn_apply, orig_sce = ops.bn_apply, ops.softmax_ce

        def bn_apply(x, scale, shift, residual=None, relu=False, out=None):
            if not timer.enabled:
                return orig_bn_apply(x, scale, shift, residual, relu, out)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            y = orig_bn_apply(x, scale, shift, residual, relu, out)
            e.record()
            nb = x.numel() * x.element_size()
            dt = "bf16" if x.dtype == torch.bfloat16 else "f32"
            timer.hbm_records.append((f"bn_apply_kernel<{dt}>", nb * (3 if residual is not None else 2), s, e))
            return y

        def bn_bwd_stages(run_stage, x, has_y, has_dres):
            nb = x.numel() * x.element_size()
            dt = "bf16" if x.dtype == torch.bfloat16 else "f32"
            if not timer.enabled:
                run_stage(3)
                return
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            run_stage(4)  # channel_reduce_kernel<BnBwdOp> alone
            ev[1].record()
            run_stage(8)  # bn_bwd_finalize_kernel (per-channel vectors: no tensor traffic, not an HBM entry)
            ev[2].record()
            run_stage(2)  # bn_bwd_apply_kernel alone
            ev[3].record()
            timer.hbm_records.append((f"channel_reduce_kernel<{dt},BnBwdOp>", nb * (3 if has_y else 2), ev[0], ev[1]))
            timer.hbm_records.append((f"bn_bwd_apply_kernel<{dt}>",
                                      nb * ((3 if has_y else 2) + 1 + (1 if has_dres else 0)), ev[2], ev[3]))

        def softmax_ce(logits, targets, *a, **kw):
            if not timer.enabled:
                return orig_sce(logits, targets, *a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_sce(logits, targets, *a, **kw)
            e.record()
            nb = logits.numel() * logits.element_size()
            dt = "bf16" if logits.dtype == torch.bfloat16 else "f32"
            # logits read + dlogits written (stored pitch), targets read + predictions written (1 B per pixel each)
            timer.hbm_records.append((f"softmax_ce_tiled_kernel<{dt}>(+weight sum)", 2 * nb + 2 * targets.numel(), s, e))
            return r

        ops.bn_apply, ops.softmax_ce = bn_apply, softmax_ce
        ops.BN_BWD_STAGE_HOOK = bn_bwd_stages

    def summary(self):
        agg = defaultdict(lambda: [0.0, 0.0, 0])
        for sym, flops, s, e in self.records:
            a = agg[sym]
            a[0] += flops
            a[1] += s.elapsed_time(e) * 1e-3
            a[2] += 1
        return {k: {"flops": v[0], "seconds": v[1], "launches": v[2]} for k, v in agg.items()}

    def hbm_summary(self):
        agg = defaultdict(lambda: [0.0, 0.0, 0])
        for sym, nbytes, s, e in self.hbm_records:
            a = agg[sym]
            a[0] += nbytes
            a[1] += s.elapsed_time(e) * 1e-3
            a[2] += 1
        return {k: {"bytes": v[0], "seconds": v[1], "launches": v[2]} for k, v in agg.items()}


def pmc_traffic_source():
    """where roofline.traffic comes from: the counters are never collected inside the timed run"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not found:
        return None
    return ("committed " + os.path.relpath(found[-1], ROOT) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
            "of tools/collect_profiles.sh, gfx950 corrections applied by tools/pmc_summary.py); not measured in this run")


def pmc_traffic(symbol: str):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE are
    collected in their own rocprofv3 runs, never together with timing): profiles/r01_pmc_traffic.json.
    Returns None when no counter run matches the symbol."""
    import glob
    import re
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not found:
        return None
    kernels = json.load(open(found[-1]))["kernels"]  # the newest round's counter passes
    mh = re.match(r"(bn_apply_kernel|bn_bwd_apply_kernel|softmax_ce_kernel|softmax_ce_tiled_kernel)<(bf16|f32)>", symbol)
    mc = re.match(r"channel_reduce_kernel<(bf16|f32),(BnBwdOp|StatOp)>", symbol)
    if mc:
        want = "void channel_reduce_kernel<{}, {}>".format("ffa_bf16" if mc.group(1) == "bf16" else "float", mc.group(2))
        hit = [v for name, v in kernels.items() if name.startswith(want)]
        return round(hit[0]["hbm_bytes_per_launch"]) if hit else None
    if mh:
        want = "void {}<{}>".format(mh.group(1), "ffa_bf16" if mh.group(2) == "bf16" else "float")
        hit = [v for name, v in kernels.items() if name.startswith(want)]
        return round(hit[0]["hbm_bytes_per_launch"]) if hit else None
    m16 = re.match(r"conv3x3_ring16_kernel<bf16,co64,(\d+)x(\d+)>", symbol)
    if m16:  # template arguments <WCO, WPX, NT, TH, TW, OCC, PRO>
        want = "void conv3x3_ring16_kernel<1, 4, 4, {}, {}, ".format(*m16.group(1, 2))
        hit = [v for name, v in kernels.items() if name.startswith(want)]
        return round(max(hit, key=lambda v: v["launches"])["hbm_bytes_per_launch"]) if hit else None
    mr = re.match(r"conv3x3_ring_kernel<(bf16|f32),(\d+)x(\d+)", symbol)
    if mr:
        want = "void conv3x3_ring_kernel<{}, ".format("ffa_bf16" if mr.group(1) == "bf16" else "float")
        hit = [v for name, v in kernels.items() if name.startswith(want) and f", {mr.group(2)}, {mr.group(3)}, " in name]
        return round(max(hit, key=lambda v: v["launches"])["hbm_bytes_per_launch"]) if hit else None
    mp = re.match(r"conv3x3_persist_kernel<(bf16|f32),bco(\d+),(\d+)x(\d+),hk(\d)>", symbol)
    if mp:  # template arguments <T, BCO, TH, TW, HK>
        want = "void conv3x3_persist_kernel<{}, {}, {}, {}, {}>".format(
            "ffa_bf16" if mp.group(1) == "bf16" else "float", *mp.group(2, 3, 4, 5))
        hit = [v for name, v in kernels.items() if name.startswith(want)]
        return round(hit[0]["hbm_bytes_per_launch"]) if hit else None
    m = re.match(r"(conv_igemm_kernel|conv_wgrad_kernel)<(bf16|f32),(\d)x\d,s(\d),(?:bco(\d+)|w(\d)x(\d)),?(\d+x\d+)?(?:,hk(\d))?", symbol)
    if not m:
        return None
    kind, dt, k, st = m.group(1), ("ffa_bf16" if m.group(2) == "bf16" else "float"), m.group(3), m.group(4)
    best = None
    for name, v in kernels.items():
        if not name.startswith(f"void {kind}<{dt}, {k}, {k}, {st},"):
            continue
        if kind == "conv_igemm_kernel":
            th, tw = (m.group(8) or "8x32").split("x")
            if f", {m.group(5)}, " not in name or f", {th}, {tw}," not in name:
                continue
            if ("true>" in name) != symbol.endswith(",up>"):  # two-source instantiation (template flag UP)
                continue
            if m.group(9) and f", {tw}, {m.group(9)}, " not in name:  # halo depth HK follows the tile in the symbol
                continue
        else:
            if f", {m.group(6)}, {m.group(7)}," not in name:
                continue
        if best is None or v["launches"] > best["launches"]:
            best = v
    return None if best is None else round(best["hbm_bytes_per_launch"])


def cpu_baseline(budget_s: float = 14.0, B: int = 2):
    """The oracle's training step on the host cores (fp32, NCHW, eager, AdamW): B tiles per step."""
    import torch.nn.functional as F
    from oracle.unet_resnet34 import UnetResNet34
    # the GPU box gives one GPU's share of the host: 16 cores (more threads than that oversubscribe the cgroup)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))
    torch.set_num_threads(threads)
    torch.manual_seed(2025)
    model = UnetResNet34(5, 19).train()
    opt = torch.optim.AdamW(model.parameters(), lr=5e-5, weight_decay=0.01, betas=(0.9, 0.999))
    x = torch.randn(B, 5, 512, 512)
    t = torch.randint(0, 19, (B, 512, 512))
    w = torch.tensor([1.0] * 15 + [0.0] * 4)

    def step():
        loss = F.cross_entropy(model(x), t, weight=w)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    step()  # warm-up
    warm = time.perf_counter() - t0
    n = max(1, min(4, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    dt = time.perf_counter() - t0
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    del model, opt
    return {"value": round(B * n / dt, 4), "unit": "tiles/s", "cores": threads, "kind": "port",
            "sample": f"{n} timed step(s) of batch {B} (512x512x5, 19 classes, fp32 NCHW eager, AdamW) after 1 warm-up; "
                      f"oracle/unet_resnet34.py on '{cpu_model}'"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="tiles per GPU per step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--tile", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the short BASELINE configs[3] / configs[4] measurements appended to the line (N = 1 only)")
    ap.add_argument("--breakdown", action="store_true", help="print the per-kernel-symbol table to stderr")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of as one hipGraph replay")
    ap.add_argument("--ddp-graph", action="store_true", help="(default since round 3, kept for old command lines)")
    ap.add_argument("--ddp-eager", action="store_true",
                    help="N > 1: eager step with the all-reduces issued from autograd hooks (overlapped with backward) "
                         "instead of the default graph(fwd+bwd) -> bucketed all-reduce -> graph(AdamW)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # FFA_BENCH_BACKEND=gloo lets the multi-rank flow be rehearsed on a one-GPU box (ranks then share cuda:0)
    backend = os.environ.get("FFA_BENCH_BACKEND", "nccl")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # read once, when the GPU runtime initialises (next line)
    local_dev = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    # FFA_BENCH_FORCE_DDP=1 (one process): run the N > 1 PROGRAM -- graph(fwd + bwd) -> bucketed RCCL all-reduce ->
    # graph(AdamW) -- in a one-rank RCCL group, so that a one-GPU box executes exactly what the multi-GPU launch runs
    ddp = world > 1 or os.environ.get("FFA_BENCH_FORCE_DDP", "0") == "1"
    if ddp:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))
        elif backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL; one process per GPU
        else:
            dist.init_process_group(backend)

    from flairhip.configs import unet_resnet34_config
    from flairhip.distributed import GradSync
    from flair_hub.tasks.module_setup import build_segmentation_module

    total_steps = args.steps + args.warmup
    cfg = unet_resnet34_config(in_channels=5, precision=args.precision, batch_size=args.batch,
                               total_steps=2 * total_steps + 16 + SETTLE_MAX_STEPS)  # + graph warm-up, settle, roofline pass
    torch.manual_seed(cfg["hyperparams"]["seed"])
    task = build_segmentation_module(cfg, {MOD: args.tile}, "train").to(dev)
    task.train()
    opt_cfg = task.configure_optimizers()
    optimizer, scheduler = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    sync = GradSync(task.model, always_sync=(ddp and world == 1))

    g = torch.Generator(device=dev).manual_seed(2025 + rank)
    B, S = args.batch, args.tile
    x = torch.randn(B, 5, S, S, generator=g, device=dev)
    t = torch.randint(0, 19, (B, S, S), generator=g, device=dev, dtype=torch.uint8)
    batch = {MOD: x, TASK: t}

    timer = KernelTimer()
    timer.install()

    def eager_step(i):
        loss = task.training_step(batch, i)
        optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if sync._handles or not ddp:
            sync.finish()  # all-reduces were issued from the autograd hooks, underneath backward
        else:  # hook-less GradSync of the graph mode: hand the finished gradients over
            ps = [p for p in task.model.parameters() if p.grad is not None]
            sync.reduce_grads(ps, [p.grad for p in ps])
        optimizer.step()
        scheduler.step()
        return loss

    # single process: the whole step (forward, loss, metrics, backward, AdamW) is one hipGraph replay;
    # multi process: THE SAME captured kernels as two graphs with the collectives between them -- graph(forward + loss +
    # backward, weight gradients written into the flat buckets) -> bucketed RCCL all-reduce -> graph(AdamW)
    # (flairhip.graph.GraphedTrainStep); --ddp-eager: eager step, all-reduces from autograd hooks, overlapped with backward
    use_graph = (not args.no_graph) and (not ddp or not args.ddp_eager)
    graphed = None
    if use_graph:
        from flairhip.graph import GraphedTrainStep
        try:
            if not ddp:
                graphed = GraphedTrainStep(task, optimizer, batch, warmup_steps=3, after_step=scheduler.step)
            else:
                sync.remove()
                # hook-less mode reduces after backward, nothing overlaps: ONE collective over all gradients (98 MB) instead
                # of three 32 MiB buckets -- fewer launches and stream hand-overs, and a ring works best on large messages
                sync = GradSync(task.model, hooks=False, broadcast_from_rank0=False, always_sync=(world == 1),
                                bucket_bytes=1 << 30)
                graphed = GraphedTrainStep(task, optimizer, batch, warmup_steps=3, after_step=scheduler.step,
                                           grad_reduce=sync.reduce_grads)
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            use_graph = False
            if ddp:
                sync = GradSync(task.model, broadcast_from_rank0=False, always_sync=(world == 1))
    if use_graph and os.environ.get("FFA_BENCH_COPY_INPUTS", "0") != "1":
        # the synthetic batch lives in the graph's own static input buffers (GraphedTrainStep copies a batch it is handed
        # only when it sits elsewhere): like the eager step, the replay reads inputs that are already where the kernels
        # expect them -- an input pipeline writes its batches there (a 176 MB device-to-device copy per step otherwise)
        for k, v in batch.items():
            if torch.is_tensor(v):
                graphed.static_batch[k].copy_(v)
        batch = graphed.static_batch
    step = (lambda i: graphed(batch)) if use_graph else eager_step

    # Settle phase, before the counted warm-up: a fresh box starts with the GPU in a low power state and (eager mode)
    # an allocator that still grows -- the first 10-25 steps were measured at 23-25 ms against 15.5 afterwards on
    # some boxes.  Run windows of ten steps until two consecutive windows agree within 1.5 % (at most
    # SETTLE_MAX_STEPS steps / 3 s); every rank runs the same number of windows.
    prev, calm, t_settle = None, 0, time.perf_counter()
    for w in range(SETTLE_MAX_STEPS // 10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        calm = calm + 1 if (prev is not None and abs(dt - prev) <= 0.015 * prev) else 0
        prev = dt
        stop = calm >= 2 or time.perf_counter() - t_settle > 3.0
        if world > 1:  # a common decision (the collectives inside the step need every rank)
            flag = torch.tensor([1.0 if stop else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            stop = bool(flag.item() > 0.5)
        if stop:
            break
    if rank == 0:
        print(f"[bench] settle: {10 * (w + 1)} steps, last window {prev / 10 * 1e3:.2f} ms/step", file=sys.stderr)
    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = False  # the timed region carries no instrumentation (events cannot see inside a replay anyway)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    # roofline pass: the same step run eagerly a few times with HIP events around every MFMA kernel launch (all ranks
    # take part: the eager step of a multi-process run contains the gradient all-reduce)
    final_loss_t = loss.detach().clone()
    # nothing may keep the replayed step's autograd graph alive into the eager pass: its AccumulateGrad nodes belong
    # to the capture stream (the stream-mismatch warning of round 1, and the capture hazard trainers.py documents)
    del loss
    if graphed is not None:
        graphed.loss = graphed.loss.detach()
    roof_steps = 3
    timer.enabled = True
    for i in range(roof_steps):
        eager_step(args.warmup + args.steps + i)
    torch.cuda.synchronize()
    timer.enabled = False
    # second roofline pass, the precise one: the 3x3 MFMA launches carry their OWN start / stop events
    # (ffa_ktime_begin / _end: hipExtLaunchKernelGGL), i.e. the kernel's duration as rocprofv3 reports it; the event pairs
    # of the pass above bracket the launch call and with it ~4-5 us of dispatch gap per launch
    ktime = {}
    try:
        import ctypes as C
        from flairhip import lib as _fl
        _lib = _fl.load()
        cap = 4096
        _fl.check(_lib.ffa_ktime_begin(cap), "ktime_begin")
        for i in range(roof_steps):
            eager_step(args.warmup + args.steps + roof_steps + i)
        ms_buf, tag_buf = (C.c_float * cap)(), (C.c_int * cap)()
        n_timed = _lib.ffa_ktime_end(ms_buf, tag_buf, cap)
        for j in range(max(0, min(n_timed, cap))):
            a = ktime.setdefault(int(tag_buf[j]), [0.0, 0])
            a[0] += float(ms_buf[j]) * 1e-3
            a[1] += 1
    except Exception as e:  # measurement aid only
        print(f"[bench] kernel timing session unavailable ({type(e).__name__}: {e})", file=sys.stderr)
    if rank == 0 and args.breakdown:
        print("[bench] timed launches by tag (seconds, launches):", ktime, file=sys.stderr)
    loss = final_loss_t
    # the same K steps run eagerly (launch by launch, uninstrumented): a multi-GPU value must be compared with the
    # one-GPU value of the SAME step mode, so both are in every line
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        eager_step(args.warmup + args.steps + 2 * roof_steps + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed_eager = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed, elapsed_eager], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_eager = float(tt[0].item()), float(tt[1].item())

    final_loss = float(loss.item())
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        tiles_per_s = world * B * args.steps / elapsed
        summ = timer.summary()
        dom = max(summ.items(), key=lambda kv: kv[1]["seconds"])
        peak = MFMA_BF16_PEAK_TFLOPS if args.precision == "bf16" else MFMA_F32_PEAK_TFLOPS
        ach = dom[1]["flops"] / dom[1]["seconds"] / 1e12
        mfma_total = sum(v["seconds"] for v in summ.values()) * (args.steps / roof_steps)
        bracket_ms = dom[1]["seconds"] / dom[1]["launches"] * 1e3
        timing = "hipEventRecord pair around the launch call (includes the dispatch gap)"
        tag = {"conv3x3_ring16_kernel<bf16,co64,8x32>": 1, "conv3x3_ring16_kernel<bf16,co64,16x16>": 2}.get(dom[0])
        if tag in ktime and ktime[tag][1] == dom[1]["launches"]:
            # the same launches, timed by events attached to the kernel itself: this is the duration rocprofv3 reports
            ach = dom[1]["flops"] / ktime[tag][0] / 1e12
            avg_ms = ktime[tag][0] / ktime[tag][1] * 1e3
            timing = "start / stop events attached to the kernel launch (hipExtLaunchKernelGGL): the kernel's own duration"
        else:
            avg_ms = bracket_ms
        roofline = {"bound": "mfma", "kernel": dom[0], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": pmc_traffic(dom[0]), "traffic_source": pmc_traffic_source(),
                    "launches_per_step": dom[1]["launches"] / roof_steps,
                    "avg_launch_ms": round(avg_ms, 4), "timing": timing,
                    "avg_launch_ms_event_bracket": round(bracket_ms, 4),
                    "mfma_kernels_share_of_step": round(mfma_total / elapsed, 4)}
        if 16 in ktime:
            roofline["wgrad64_avg_launch_ms"] = round(ktime[16][0] / ktime[16][1] * 1e3, 4)
        if args.breakdown:
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["seconds"]):
                print(f"  {k:62s} {v['seconds'] / roof_steps * 1e3:8.3f} ms/step {v['launches'] // roof_steps:4d} launches "
                      f"{v['flops'] / v['seconds'] / 1e12:8.1f} TFLOP/s", file=sys.stderr)
            print(f"  MFMA kernels total {mfma_total / args.steps * 1e3:.3f} ms of {ms:.3f} ms/step; loss {final_loss:.4f}",
                  file=sys.stderr)
        out = {
            "metric": "512x512x5 tiles/sec (train fwd+bwd+AdamW), U-Net ResNet-34, 19 classes",
            "value": round(tiles_per_s, 2), "unit": "tiles/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"synthetic {S}x{S}x5 tiles, 19-class U-Net (ResNet-34 encoder), batch {B} per GPU, "
                                   f"train step (BASELINE.json configs[{1 if world == 1 else 2}])",
                       "global_batch": B * world, "tile": S, "parallelism": f"dp{world}"},
            "final_loss": round(final_loss, 5), "hip_graph": bool(use_graph),
            "step_mode": ("hipgraph(whole step)" if not ddp else "hipgraph(fwd+bwd) -> bucketed all-reduce -> hipgraph(adamw)")
                         if use_graph else ("eager" if not ddp else "eager, all-reduce from autograd hooks"),
            "ms_per_step_eager": round(elapsed_eager / args.steps * 1e3, 3),
            "roofline": roofline,
        }
        hsum = timer.hbm_summary()
        if hsum:
            # the HBM-bound group of the step, TIME-WEIGHTED: every bracketed streaming kernel (BatchNorm apply /
            # backward reduce / backward apply, loss) -- algorithmic bytes of all their launches over their summed
            # duration; the per-symbol rates sit beside it (the best symbol alone flattered the group in round 2)
            tb = sum(v["bytes"] for v in hsum.values())
            ts = sum(v["seconds"] for v in hsum.values())
            gbs = tb / ts / 1e9
            hd = max(hsum.items(), key=lambda kv: kv[1]["seconds"])
            out["roofline_hbm"] = {
                "bound": "hbm", "kernel": "time-weighted: " + " + ".join(sorted(hsum)), "achieved": round(gbs, 1),
                "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "peak_measured_copy": 6290.0,
                "frac_of_measured_copy": round(gbs / 6290.0, 4),
                "traffic": pmc_traffic(hd[0]), "traffic_kernel": hd[0],
                "launches_per_step": sum(v["launches"] for v in hsum.values()) / roof_steps,
                "algorithmic_bytes_per_step": round(tb / roof_steps),
                "per_kernel_GBps": {k: round(v["bytes"] / v["seconds"] / 1e9, 1) for k, v in sorted(hsum.items())},
                "hbm_kernels_share_of_step": round(ts * (args.steps / roof_steps) / elapsed, 4)}
            if args.breakdown:
                for k, v in sorted(hsum.items(), key=lambda kv: -kv[1]["seconds"]):
                    print(f"  {k:62s} {v['seconds'] / roof_steps * 1e3:8.3f} ms/step {v['launches'] // roof_steps:4d} "
                          f"launches {v['bytes'] / v['seconds'] / 1e9:8.0f} GB/s", file=sys.stderr)
        if world == 1 and not args.no_extras:
            # BASELINE.json configs[3] (Swin-T + UPerNet, 512 x 512, batch 32) and configs[4]'s per-GPU work (aerial +
            # DEM + Sentinel-2 series, COSIA + LPIS heads, batch 16): a short measurement each, AFTER the timed region of
            # the headline metric, so that a driver-run record carries them (round 2 had them in builder-run files only)
            del task, optimizer, scheduler, graphed, batch, x, t, sync
            torch.cuda.empty_cache()
            import importlib.util

            def tool(name):
                spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return mod
            for key, name, argv in (("swin_t", "bench_swin_train", ["--graph", "--steps", "10", "--warmup", "3"]),
                                    ("fusion_sentinel", "bench_fusion", ["--sentinel", "24", "--steps", "10", "--warmup", "3"])):
                try:
                    m = tool(name)
                    out[key] = m.measure(m.parse(argv))
                except Exception as e:  # an extra object never costs the headline line
                    out[key] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B=2)
            out["cpu_baseline_b8"] = cpu_baseline(B=8)  # B = 2 (config 1) and B = 8
        print(json.dumps(out))
    if ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
