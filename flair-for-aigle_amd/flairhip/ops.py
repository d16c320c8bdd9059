"""Functional layer over the C ABI: torch tensors in, torch tensors out, no autograd.

PyTorch is used here for device memory and the current HIP stream only.  Activations are
contiguous NHWC tensors ``[B, H, W, C]`` (bf16 or f32) whose channel count is a multiple of 16;
every function launches on ``torch.cuda.current_stream()`` and returns immediately.
"""
from __future__ import annotations

import os

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import lib as _l

CH_ALIGN = 16


def pad_channels(c: int, align: int = CH_ALIGN) -> int:
    return (c + align - 1) // align * align


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _l.BF16
    if t.dtype == torch.float32:
        return _l.F32
    raise TypeError(f"flairhip: unsupported dtype {t.dtype}")


def _dtype_id(dtype: torch.dtype) -> int:
    return _l.BF16 if dtype == torch.bfloat16 else _l.F32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk_nhwc(t: torch.Tensor, name: str) -> None:
    if not (t.is_cuda and t.dim() == 4 and t.is_contiguous()):
        raise ValueError(f"flairhip: {name} must be a contiguous CUDA NHWC tensor, got {tuple(t.shape)} "
                         f"strides {t.stride()} on {t.device}")


_workspaces = {}
_retired_workspaces = []


def workspace(nbytes: int, device: torch.device, slot: str = "default") -> torch.Tensor:
    """Grow-only scratch buffer per (device, slot); the library itself never allocates.  A buffer that is outgrown
    stays alive (``_retired_workspaces``): captured hipGraphs (GraphedTrainStep, GraphedCall) have its raw pointer
    baked into their kernel arguments, and the caching allocator would otherwise hand the freed block to another
    tensor that later replays then scribble split-K partials and statistics over."""
    key = (device.index, slot)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None:
            _retired_workspaces.append(buf)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


# --------------------------------------------------------------------------------------------------
# convolution

@dataclass
class PackedWeight:
    data: torch.Tensor  # packed operand (opaque bytes viewed as the compute dtype)
    rows: int           # padded row count (output channels of the GEMM)
    rows_real: int
    ci_pitch: int       # channel pitch of the activation it multiplies
    bco: int
    kh: int
    kw: int
    stride: int         # stride of the conv the operand is used for (1 for every dgrad operand)
    ch_real: int = 0    # real (unpadded) reduction channels


def pack_conv_weight(w_oihw: torch.Tensor, dtype: torch.dtype, stride: int, ci_pitch: int,
                     transpose: bool = False, scale: Optional[torch.Tensor] = None,
                     allow_ring: bool = True, allow_thin: bool = True, allow_stem: bool = True) -> PackedWeight:
    """OIHW f32 master weight -> MFMA operand for ffa_conv2d (forward, or dgrad when transpose=True).
    allow_ring=False keeps the operand in the conv_igemm layout (needed by the two-source / split-epilogue /
    zero-insertion calls); otherwise ffa_conv_plan picks the LDS-DMA ring layout where it applies.  allow_thin:
    bf16 layers with at most 32 stored input channels and 32 rows may get the register-resident layout of
    conv3x3_thin_kernel (plain, statistics, skip-less two-source and skip-less pooled-split calls).  allow_stem: the bf16
    forward operand of a 7x7 stride-2 PAD-3 convolution over <= 8 real input channels gets conv7x7_stem_kernel's layout."""
    lib = _l.load()
    if w_oihw.dtype != torch.float32 or not w_oihw.is_contiguous():
        raise ValueError("pack_conv_weight: master weight must be contiguous f32 OIHW")
    O, I, kh, kw = w_oihw.shape
    rows_real = I if transpose else O
    use_stride = 1 if transpose else stride
    did = _dtype_id(dtype)
    # the dgrad operand of a stride-2 layer is read through the zero insertion (dil = 2): conv_igemm only
    ring_ok = allow_ring and not (transpose and stride != 1)
    thin_ok = allow_thin and not (transpose and stride != 1)
    # the stem kernel stages the first 8 channels of a pixel only: forward operand of a <= 8-channel input
    stem_ok = allow_stem and not transpose and I <= 8  # (the caller vouches for pad = 3)
    bco = lib.ffa_conv_plan(did, kh, kw, use_stride, rows_real, ci_pitch,
                            (1 if ring_ok else 0) | (2 if thin_ok else 0) | (4 if stem_ok else 0))
    if bco <= 0:
        raise _l.FlairHipError(f"no conv kernel for {kh}x{kw} stride {use_stride}")
    blk = bco & 0xFFF
    rows = (rows_real + blk - 1) // blk * blk
    rg = lib.ffa_conv_row_group(kh)
    nbytes = lib.ffa_pack_conv_weight_bytes(did, rows, ci_pitch, kh, kw)
    if bco & _l.BCO_THIN:
        nbytes = lib.ffa_thin_pack_bytes(rows, ci_pitch)
    if bco & _l.BCO_STEM:
        nbytes = lib.ffa_stem_pack_bytes()
    dst = torch.empty(nbytes // (2 if dtype == torch.bfloat16 else 4), dtype=dtype, device=w_oihw.device)
    _l.check(lib.ffa_pack_conv_weight(did, w_oihw.data_ptr(), _ptr(scale), dst.data_ptr(), O, I, kh, kw,
                                      1 if transpose else 0, rows, ci_pitch, bco, rg, _stream()), "pack_conv_weight")
    return PackedWeight(dst, rows, rows_real, ci_pitch, bco, kh, kw, use_stride, O if transpose else I)


class PackBatch:
    """Descriptor tables for ffa_pack_conv_weights_batched / ffa_ring_pack_batched: re-packs many conv operands in one
    launch per layout.

    entries: (master weight OIHW f32, PackedWeight it feeds, transpose flag[, (first column, columns)]).  The tables hold raw device
    pointers, so they must be rebuilt when a parameter or a packed buffer is re-allocated."""

    def __init__(self, entries, dtype: torch.dtype):
        lib = _l.load()
        self.dtype_id = _dtype_id(dtype)
        self.keep = []
        plain = [e for e in entries if not (e[1].bco & (_l.BCO_RING | _l.BCO_THIN | _l.BCO_STEM))]
        self.stem = [e for e in entries if e[1].bco & _l.BCO_STEM]  # one operand per encoder: packed by its own small launch
        ring = [e for e in entries if e[1].bco & _l.BCO_RING]
        thin = [e for e in entries if e[1].bco & _l.BCO_THIN]
        self.n, self.table = len(plain), None
        self.n_ring, self.table_ring = len(ring), None
        self.n_thin, self.table_thin = len(thin), None
        dev = entries[0][0].device
        if plain:
            nb = lib.ffa_pack_desc_bytes()
            host = C.create_string_buffer(nb * len(plain))
            base = C.addressof(host)
            for i, e in enumerate(plain):
                w, pw, transpose = e[:3]
                O, I, kh, kw = w.shape
                if len(e) > 3:  # column block W[:, off : off + c] of a fusion 1x1 weight
                    off, c = e[3]
                    _l.check(lib.ffa_pack_desc_fill_cols(base + i * nb, w.data_ptr(), None, pw.data.data_ptr(), O, I, off, c,
                                                         kh, kw, 1 if transpose else 0, pw.rows, pw.ci_pitch, pw.bco,
                                                         lib.ffa_conv_row_group(kh), self.dtype_id), "pack_desc_fill_cols")
                else:
                    _l.check(lib.ffa_pack_desc_fill(base + i * nb, w.data_ptr(), None, pw.data.data_ptr(), O, I, kh, kw,
                                                    1 if transpose else 0, pw.rows, pw.ci_pitch, pw.bco,
                                                    lib.ffa_conv_row_group(kh), self.dtype_id), "pack_desc_fill")
                self.keep.append((w, pw))
            self.table = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
        if ring:
            nb = lib.ffa_ring_pack_desc_bytes()
            host = C.create_string_buffer(nb * len(ring))
            base = C.addressof(host)
            for i, (w, pw, transpose) in enumerate(ring):
                O, I, kh, kw = w.shape
                _l.check(lib.ffa_ring_pack_desc_fill(base + i * nb, w.data_ptr(), None, pw.data.data_ptr(), O, I,
                                                     1 if transpose else 0, pw.rows, pw.ci_pitch, self.dtype_id),
                         "ring_pack_desc_fill")
                self.keep.append((w, pw))
            self.table_ring = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)
        if thin:
            nb = lib.ffa_thin_pack_desc_bytes()
            host = C.create_string_buffer(nb * len(thin))
            base = C.addressof(host)
            for i, (w, pw, transpose) in enumerate(thin):
                O, I, kh, kw = w.shape
                _l.check(lib.ffa_thin_pack_desc_fill(base + i * nb, w.data_ptr(), None, pw.data.data_ptr(), O, I,
                                                     1 if transpose else 0, pw.rows, pw.ci_pitch), "thin_pack_desc_fill")
                self.keep.append((w, pw))
            self.table_thin = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev)

    def run(self) -> None:
        lib = _l.load()
        if self.table is not None:
            _l.check(lib.ffa_pack_conv_weights_batched(self.dtype_id, self.table.data_ptr(), self.n, _stream()),
                     "pack_conv_weights_batched")
        if self.table_ring is not None:
            _l.check(lib.ffa_ring_pack_batched(self.dtype_id, self.table_ring.data_ptr(), self.n_ring, _stream()),
                     "ring_pack_batched")
        if self.table_thin is not None:
            _l.check(lib.ffa_thin_pack_batched(self.table_thin.data_ptr(), self.n_thin, _stream()), "thin_pack_batched")
        for w, pw, transpose in self.stem:
            _l.check(lib.ffa_stem_pack(w.data_ptr(), None, pw.data.data_ptr(), w.shape[0], w.shape[1], _stream()), "stem_pack")


def conv_out_size(h: int, k: int, stride: int, pad: int) -> int:
    return (h + 2 * pad - k) // stride + 1


def conv_is_persistent(x_dtype, B: int, Ho: int, Wo: int, w: "PackedWeight", dil: int = 1) -> bool:
    """True when conv2d runs this operand on conv3x3_persist_kernel (its own symbol in rocprofv3) rather than
    conv_igemm_kernel; used by bench.py to group launches the way a profile does."""
    return bool(_l.load().ffa_conv_is_persistent(_dtype_id(x_dtype), B, Ho, Wo,
                                                 w.ci_pitch, w.rows, w.bco, w.kh, w.kw, w.stride, dil))


def conv2d(x: torch.Tensor, w: PackedWeight, pad: int, out_channels: int, bias: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, relu: bool = False, dil: int = 1,
           out_hw: Optional[Tuple[int, int]] = None, out: Optional[torch.Tensor] = None,
           stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[B,Ho,Wo,out_channels] = relu?(conv(x, w) + bias + residual).  out_channels is the stored pitch.
    ``stats`` (f32, >= conv_stat_rows * 2 * out_channels elements) receives per-tile channel sums / sums of squares
    of the stored output for a following training-mode BatchNorm (see bn_finalize)."""
    lib = _l.load()
    _chk_nhwc(x, "conv input")
    B, Hi, Wi, Ci = x.shape
    if Ci != w.ci_pitch:
        raise ValueError(f"conv2d: input pitch {Ci} != packed pitch {w.ci_pitch}")
    if out_hw is None:
        hv, wv = (Hi * 2, Wi * 2) if dil == 2 else (Hi, Wi)
        out_hw = (conv_out_size(hv, w.kh, w.stride, pad), conv_out_size(wv, w.kw, w.stride, pad))
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty((B, Ho, Wo, out_channels), dtype=x.dtype, device=x.device)
    if residual is not None and residual.shape != out.shape:
        raise ValueError("conv2d: residual shape mismatch")
    if bias is not None and bias.numel() < out_channels:
        raise ValueError("conv2d: bias shorter than the output pitch")
    if stats is not None:
        if stats.dtype != torch.float32 or stats.numel() < conv_stat_rows(B, Ho, Wo, w) * 2 * out_channels:
            raise ValueError("conv2d: statistics buffer too small or not f32")
        _l.check(lib.ffa_conv2d_stats(_dt(x), x.data_ptr(), w.data.data_ptr(), _ptr(bias), _ptr(residual),
                                      out.data_ptr(), stats.data_ptr(), B, Hi, Wi, Ci, Ho, Wo, out_channels, w.rows,
                                      w.bco, w.kh, w.kw, w.stride, pad, dil, 1 if relu else 0, _stream()),
                 "conv2d_stats")
        return out
    _l.check(lib.ffa_conv2d(_dt(x), x.data_ptr(), w.data.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                            B, Hi, Wi, Ci, Ho, Wo, out_channels, w.rows, w.bco, w.kh, w.kw, w.stride, pad, dil,
                            1 if relu else 0, _stream()), "conv2d")
    return out


def conv3x3_ring(x: torch.Tensor, w: PackedWeight, out_channels: int, bias: Optional[torch.Tensor] = None,
                 residual: Optional[torch.Tensor] = None, relu: bool = False, stats: Optional[torch.Tensor] = None,
                 pro_scale: Optional[torch.Tensor] = None, pro_shift: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The ring kernel called directly (w must be a ring-layout operand): conv3x3 pad 1 of x, or -- with pro_scale /
    pro_shift -- of relu(x * pro_scale[c] + pro_shift[c]) evaluated while the input is staged (the BatchNorm + ReLU
    of the producing layer folded into this convolution, no normalised tensor in memory)."""
    lib = _l.load()
    _chk_nhwc(x, "conv input")
    if not (w.bco & _l.BCO_RING):
        raise ValueError("conv3x3_ring: operand is not in the ring layout")
    B, H, W, Ci = x.shape
    if Ci != w.ci_pitch:
        raise ValueError(f"conv3x3_ring: input pitch {Ci} != packed pitch {w.ci_pitch}")
    if out is None:
        out = torch.empty((B, H, W, out_channels), dtype=x.dtype, device=x.device)
    if stats is not None and (stats.dtype != torch.float32 or
                              stats.numel() < conv_stat_rows(B, H, W, w) * 2 * out_channels):
        raise ValueError("conv3x3_ring: statistics buffer too small or not f32")
    if (pro_scale is None) != (pro_shift is None):
        raise ValueError("conv3x3_ring: prologue needs scale and shift")
    if pro_scale is not None and (pro_scale.numel() < Ci or pro_shift.numel() < Ci or
                                  pro_scale.dtype != torch.float32 or pro_shift.dtype != torch.float32):
        raise ValueError("conv3x3_ring: prologue vectors must be f32 with one entry per stored input channel")
    _l.check(lib.ffa_ring_conv3x3(_dt(x), x.data_ptr(), w.data.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                  _ptr(stats), _ptr(pro_scale), _ptr(pro_shift), B, H, W, Ci, out_channels, w.rows,
                                  1 if relu else 0, _stream()), "ring_conv3x3")
    return out


def conv2d_upcat(lo: torch.Tensor, skip: Optional[torch.Tensor], w: PackedWeight, out_channels: int,
                 stats: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, relu: bool = False
                 ) -> Optional[torch.Tensor]:
    """3x3 pad-1 conv over cat(nearest_x2(lo), skip) without materialising it.  Returns None when the library has no
    two-source kernel for this channel split (the caller then concatenates explicitly)."""
    lib = _l.load()
    _chk_nhwc(lo, "upcat lo")
    B, Hl, Wl, C1 = lo.shape
    C2 = 0
    if skip is not None:
        _chk_nhwc(skip, "upcat skip")
        if skip.shape[0] != B or skip.shape[1] != 2 * Hl or skip.shape[2] != 2 * Wl or skip.dtype != lo.dtype:
            raise ValueError("conv2d_upcat: skip must be [B, 2*Hl, 2*Wl, C2] of the same dtype")
        C2 = skip.shape[3]
    if w.kh != 3 or w.kw != 3 or w.stride != 1 or w.ci_pitch != C1 + C2:
        raise ValueError("conv2d_upcat: needs a 3x3 stride-1 operand packed for C1 + C2 input channels")
    out = torch.empty((B, 2 * Hl, 2 * Wl, out_channels), dtype=lo.dtype, device=lo.device)
    if bias is not None and bias.numel() < out_channels:
        raise ValueError("conv2d_upcat: bias shorter than the output pitch")
    rc = lib.ffa_conv2d_upcat(_dt(lo), lo.data_ptr(), _ptr(skip), w.data.data_ptr(), _ptr(bias), out.data_ptr(),
                              _ptr(stats), B, Hl, Wl, C1, C2, out_channels, w.rows, w.bco, 1 if relu else 0, _stream())
    if rc == _l.ERR_UNSUPPORTED:
        return None
    _l.check(rc, "conv2d_upcat")
    return out


# BatchNorm-backward reductions from the dgrad epilogue (ffa_conv2d_bnbwd).  Measured on MI355X, same session: 16.31 vs
# 16.24 ms/step without -- the epilogue's per-lane 16-byte reads of x (one cache line per lane) cost what the saved
# reduction pass gained -- so it stays OFF by default; FFA_FUSED_BN_BWD=1 turns it on (covered by the kernel tests).
FUSED_BN_BWD = os.environ.get("FFA_FUSED_BN_BWD", "0") == "1"


def conv2d_bnbwd(x: torch.Tensor, w: PackedWeight, pad: int, out_channels: int, bnx: torch.Tensor,
                 bn_scale: torch.Tensor, bn_shift: torch.Tensor, residual: Optional[torch.Tensor] = None, dil: int = 1,
                 out_hw: Optional[Tuple[int, int]] = None):
    """A (dgrad) convolution whose output dy is the gradient of relu(bn(bnx)): -> (dy, partials, rows) with the
    BatchNorm-backward reductions sum(g), sum(g*bnx) taken in the conv epilogue (see bn_bwd_partials)."""
    lib = _l.load()
    _chk_nhwc(x, "conv input")
    B, Hi, Wi, Ci = x.shape
    if out_hw is None:
        hv, wv = (Hi * 2, Wi * 2) if dil == 2 else (Hi, Wi)
        out_hw = (conv_out_size(hv, w.kh, w.stride, pad), conv_out_size(wv, w.kw, w.stride, pad))
    Ho, Wo = out_hw
    if tuple(bnx.shape) != (B, Ho, Wo, out_channels) or bnx.dtype != x.dtype or not bnx.is_contiguous():
        raise ValueError("conv2d_bnbwd: bnx must have the shape, pitch and dtype of the output")
    out = torch.empty((B, Ho, Wo, out_channels), dtype=x.dtype, device=x.device)
    rows = conv_stat_rows(B, Ho, Wo)
    part = workspace(rows * 2 * out_channels * 4, x.device, "bnpart").view(torch.float32)
    _l.check(lib.ffa_conv2d_bnbwd(_dt(x), x.data_ptr(), w.data.data_ptr(), _ptr(residual), out.data_ptr(),
                                  part.data_ptr(), bnx.data_ptr(), bn_scale.data_ptr(), bn_shift.data_ptr(), B, Hi, Wi,
                                  Ci, Ho, Wo, out_channels, w.rows, w.bco, w.kh, w.kw, w.stride, pad, dil, _stream()),
             "conv2d_bnbwd")
    return out, part, rows


def bn_bwd_partials(x: torch.Tensor, dy: torch.Tensor, part: torch.Tensor, rows: int, gamma, beta, mean, rstd):
    """BatchNorm (+ReLU, mask from x) backward from the partial sums of conv2d_bnbwd -> (dx, dgamma, dbeta)"""
    lib = _l.load()
    C_ = x.shape[-1]
    dx = torch.empty_like(x)
    dgb = torch.empty((2, C_), dtype=torch.float32, device=x.device)
    ws = workspace(lib.ffa_bn_workspace_bytes(C_), x.device, "bn")
    _l.check(lib.ffa_bn_bwd_partials(_dt(x), x.data_ptr(), dy.data_ptr(), part.data_ptr(), rows, _ptr(gamma), _ptr(beta),
                                     mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dgb[0].data_ptr(),
                                     dgb[1].data_ptr(), x.numel() // C_, C_, ws.data_ptr(), ws.numel(), _stream()),
             "bn_bwd_partials")
    return dx, dgb[0], dgb[1]


def conv2d_dgrad_upcat(dy: torch.Tensor, wt: PackedWeight, c1: int, c2: int):
    """Input gradient of conv2d_upcat -> (dlo [B,H/2,W/2,c1], dskip [B,H,W,c2] or None), or None when the channel
    split is not supported (the caller then runs the plain dgrad + upsample2x_concat_bwd)."""
    if not FUSED_UPCAT_BWD:
        return None
    lib = _l.load()
    _chk_nhwc(dy, "dgrad_upcat dy")
    B, H, W, Cdy = dy.shape
    if Cdy != wt.ci_pitch or wt.kh != 3 or wt.stride != 1 or H % 2 or W % 2:
        raise ValueError("conv2d_dgrad_upcat: operand / gradient mismatch")
    dlo = torch.empty((B, H // 2, W // 2, c1), dtype=dy.dtype, device=dy.device)
    dskip = torch.empty((B, H, W, c2), dtype=dy.dtype, device=dy.device) if c2 else None
    rc = lib.ffa_conv2d_dgrad_upcat(_dt(dy), dy.data_ptr(), wt.data.data_ptr(), dlo.data_ptr(), _ptr(dskip), B, H, W, Cdy,
                                    c1, c2, wt.rows, wt.bco, _stream())
    if rc == _l.ERR_UNSUPPORTED:
        return None
    _l.check(rc, "conv2d_dgrad_upcat")
    return dlo, dskip


FUSED_BN_STATS = os.environ.get("FFA_FUSED_BN_STATS", "1") != "0"


def conv_stat_rows(B: int, Ho: int, Wo: int, w: Optional[PackedWeight] = None) -> int:
    """rows of per-tile statistics the convolution with operand ``w`` writes (w=None: the conv_igemm tiling)"""
    if w is None:
        return int(_l.load().ffa_conv_stat_rows(B, Ho, Wo, 64, 64))
    return int(_l.load().ffa_conv_stat_rows(B, Ho, Wo, w.rows, w.bco))


def _bn_finalize(part: torch.Tensor, rows: int, npix: int, channels: int, gamma, beta, running_mean, running_var,
                 momentum: float, eps: float):
    lib = _l.load()
    dev = part.device
    out = torch.empty((4, channels), dtype=torch.float32, device=dev)
    ws = workspace(lib.ffa_bn_workspace_bytes(channels), dev, "bn")
    _l.check(lib.ffa_bn_finalize(part.data_ptr(), rows, npix, channels, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                 _ptr(running_var), momentum, eps, out[0].data_ptr(), out[1].data_ptr(),
                                 out[2].data_ptr(), out[3].data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
             "bn_finalize")
    return out[0], out[1], out[2], out[3]


def conv2d_bn_stats(x: torch.Tensor, w: PackedWeight, pad: int, out_channels: int, gamma, beta, running_mean,
                    running_var, momentum: float, eps: float):
    """conv + the batch statistics of its output in one kernel: -> (y0, scale, shift, mean, rstd); the running
    buffers are updated in place.  Equivalent to conv2d followed by bn_stats, minus one pass over y0."""
    if not FUSED_BN_STATS:  # A/B switch (FFA_FUSED_BN_STATS=0): the two-kernel path
        y0 = conv2d(x, w, pad, out_channels)
        return (y0,) + tuple(bn_stats(y0, gamma, beta, running_mean, running_var, momentum, eps))
    B, Hi, Wi, _ = x.shape
    Ho, Wo = conv_out_size(Hi, w.kh, w.stride, pad), conv_out_size(Wi, w.kw, w.stride, pad)
    rows = conv_stat_rows(B, Ho, Wo, w)
    part = workspace(rows * 2 * out_channels * 4, x.device, "bnpart").view(torch.float32)
    y0 = conv2d(x, w, pad, out_channels, stats=part)
    return (y0,) + _bn_finalize(part, rows, B * Ho * Wo, out_channels, gamma, beta, running_mean, running_var,
                                momentum, eps)


# "Normalise on load" (round 3): the BatchNorm + ReLU between two convolutions is evaluated by the CONSUMERS (the next
# conv's forward and weight gradient) while they stage their input, so the normalised tensor is never written.
# Implemented for every consumer kernel family of the U-Net (ring16, thin forward, thin / general weight gradient),
# bit-identical to the materialised form (tests/test_thin_conv_gpu.py, tests/test_ring_conv_gpu.py) -- and measured
# SLOWER in the step (same box, rocprofv3 kernel time per step 13.99 ms without, 14.36 ms with): the 21 saved
# bn_apply passes are worth 0.34 ms, the prologue costs conv3x3_ring16_kernel +10.7 us per launch (52.5 vs 41.8: an LDS
# read-modify-write of every halo piece and an earlier vmcnt drain, in a kernel that has no idle issue slots),
# conv_wgrad_kernel +12 us (68.7 vs 56.6) and the thin kernels +19 ... +51 us.  Off by default;
# FFA_NORM_ON_LOAD=1 turns it on.
NORM_ON_LOAD = os.environ.get("FFA_NORM_ON_LOAD", "0") == "1"


def pro_supported(w: "PackedWeight", dtype: torch.dtype, up: bool = False) -> bool:
    """True when conv2d_pro / conv_wgrad_pro exist for this operand: bf16, ring16 layout (>= 64 channels) or thin layout"""
    if dtype != torch.bfloat16 or w.kh != 3 or w.stride != 1:
        return False
    if w.bco & _l.BCO_THIN:
        return True
    return bool(w.bco & _l.BCO_RING) and not up and w.rows_real >= 64 and w.ch_real >= 64 and w.ci_pitch % 64 == 0


def conv2d_pro(x: torch.Tensor, w: PackedWeight, out_channels: int, pro_scale: torch.Tensor, pro_shift: torch.Tensor,
               bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, relu: bool = False,
               stats: Optional[torch.Tensor] = None, up: bool = False) -> torch.Tensor:
    """3x3 pad-1 conv of relu(x * pro_scale + pro_shift) without the normalised tensor (ffa_conv2d_pro); up: x is the
    low-resolution map of the skip-less nearest-x2 form (thin layout)."""
    lib = _l.load()
    _chk_nhwc(x, "conv input")
    B, Hs, Ws, Ci = x.shape
    H, W = (2 * Hs, 2 * Ws) if up else (Hs, Ws)
    if Ci != w.ci_pitch or pro_scale.numel() < Ci or pro_shift.numel() < Ci:
        raise ValueError("conv2d_pro: operand / prologue vector mismatch")
    out = torch.empty((B, H, W, out_channels), dtype=x.dtype, device=x.device)
    if stats is not None and (stats.dtype != torch.float32 or stats.numel() < conv_stat_rows(B, H, W, w) * 2 * out_channels):
        raise ValueError("conv2d_pro: statistics buffer too small or not f32")
    _l.check(lib.ffa_conv2d_pro(_dt(x), x.data_ptr(), w.data.data_ptr(), _ptr(bias), _ptr(residual), out.data_ptr(),
                                _ptr(stats), pro_scale.data_ptr(), pro_shift.data_ptr(), B, H, W, Ci, out_channels, w.rows,
                                w.bco, 1 if relu else 0, 1 if up else 0, _stream()), "conv2d_pro")
    return out


def conv2d_pro_bn_stats(x: torch.Tensor, w: PackedWeight, out_channels: int, pro_scale, pro_shift, gamma, beta,
                        running_mean, running_var, momentum: float, eps: float, up: bool = False):
    """conv2d_pro + batch statistics of its output -> (y0, scale, shift, mean, rstd)"""
    B, Hs, Ws, _ = x.shape
    H, W = (2 * Hs, 2 * Ws) if up else (Hs, Ws)
    rows = conv_stat_rows(B, H, W, w)
    part = workspace(rows * 2 * out_channels * 4, x.device, "bnpart").view(torch.float32)
    y0 = conv2d_pro(x, w, out_channels, pro_scale, pro_shift, stats=part, up=up)
    return (y0,) + _bn_finalize(part, rows, B * H * W, out_channels, gamma, beta, running_mean, running_var, momentum, eps)


def conv_wgrad_pro(x: torch.Tensor, dy: torch.Tensor, co_real: int, ci_real: int, pro_scale: torch.Tensor,
                   pro_shift: torch.Tensor, up: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW of the 3x3 pad-1 conv whose input was relu(x * pro_scale + pro_shift) (ffa_conv_wgrad_pro)"""
    lib = _l.load()
    _chk_nhwc(x, "wgrad input")
    _chk_nhwc(dy, "wgrad dy")
    B, Hs, Ws, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    did = _dt(x)
    need = lib.ffa_conv_wgrad_workspace_bytes(did, 3, 3, 1, Co, Ci, B, Ho, Wo)
    if need < 0:
        raise _l.FlairHipError("no wgrad kernel")
    ws = workspace(need, x.device, "wgrad")
    if out is None:
        out = torch.empty((co_real, ci_real, 3, 3), dtype=torch.float32, device=x.device)
    _l.check(lib.ffa_conv_wgrad_pro(did, x.data_ptr(), dy.data_ptr(), out.data_ptr(), pro_scale.data_ptr(),
                                    pro_shift.data_ptr(), B, Ho, Wo, Ci, Ho, Wo, Co, co_real, ci_real, 1 if up else 0, 0,
                                    ws.data_ptr(), ws.numel(), _stream()), "conv_wgrad_pro")
    return out


def upcat_supported(c1: int, c2: int, dtype: torch.dtype) -> bool:
    """Channel splits the two-source kernels take (ffa_conv2d_upcat: C1 covers whole halo channel groups;
    ffa_conv_wgrad_upcat: C1 is a multiple of a block's input channels)."""
    if not FUSED_UPCAT:
        return False
    eb = 2 if dtype == torch.bfloat16 else 4
    ci = c1 + c2
    nchunks = ci * eb // 32
    hk = 4 if nchunks % 4 == 0 else (2 if nchunks % 2 == 0 else 1)
    wci = 2 if (ci > 32 and eb == 2) else 1
    return c1 > 0 and c1 % 16 == 0 and c2 % 16 == 0 and hk > 1 and (c1 * eb) % (hk * 32) == 0 and c1 % (32 * wci) == 0


FUSED_UPCAT_BWD = os.environ.get("FFA_FUSED_UPCAT_BWD", "1") != "0"  # A/B: plain dgrad + up2_concat_bwd
FUSED_UPCAT = os.environ.get("FFA_FUSED_UPCAT", "1") != "0"  # A/B switch: materialise the decoder concat instead


def conv2d_upcat_bn_stats(lo: torch.Tensor, skip: Optional[torch.Tensor], w: PackedWeight, out_channels: int, gamma,
                          beta, running_mean, running_var, momentum: float, eps: float):
    """conv2d_upcat + batch statistics of its output -> (y0, scale, shift, mean, rstd)"""
    B, Hl, Wl, _ = lo.shape
    Ho, Wo = 2 * Hl, 2 * Wl
    rows = conv_stat_rows(B, Ho, Wo, w)
    part = workspace(rows * 2 * out_channels * 4, lo.device, "bnpart").view(torch.float32)
    y0 = conv2d_upcat(lo, skip, w, out_channels, stats=part)
    if y0 is None:
        raise _l.FlairHipError("conv2d_upcat: unsupported channel split (check upcat_supported first)")
    return (y0,) + _bn_finalize(part, rows, B * Ho * Wo, out_channels, gamma, beta, running_mean, running_var,
                                momentum, eps)


def conv_wgrad(x: torch.Tensor, dy: torch.Tensor, co_real: int, ci_real: int, kh: int, kw: int, stride: int,
               pad: int, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dW (OIHW f32, [co_real, ci_real, kh, kw]) from the conv input x and the output gradient dy."""
    lib = _l.load()
    _chk_nhwc(x, "wgrad input")
    _chk_nhwc(dy, "wgrad dy")
    B, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    did = _dt(x)
    need = lib.ffa_conv_wgrad_workspace_bytes(did, kh, kw, stride, Co, Ci, B, Ho, Wo)
    if need < 0:
        raise _l.FlairHipError(f"no wgrad kernel for {kh}x{kw} stride {stride}")
    ws = workspace(need, x.device, "wgrad")
    if out is None:
        out = torch.empty((co_real, ci_real, kh, kw), dtype=torch.float32, device=x.device)
    _l.check(lib.ffa_conv_wgrad(did, x.data_ptr(), dy.data_ptr(), out.data_ptr(), B, Hi, Wi, Ci, Ho, Wo, Co, co_real,
                                ci_real, kh, kw, stride, pad, 1 if accumulate else 0, ws.data_ptr(), ws.numel(),
                                _stream()), "conv_wgrad")
    return out


def conv_wgrad_upcat(lo: torch.Tensor, skip: Optional[torch.Tensor], dy: torch.Tensor, co_real: int,
                     out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """dW [co_real, C1 + C2, 3, 3] of the conv over cat(nearest_x2(lo), skip); None when unsupported."""
    lib = _l.load()
    _chk_nhwc(lo, "wgrad lo")
    _chk_nhwc(dy, "wgrad dy")
    B, Hl, Wl, C1 = lo.shape
    C2 = 0 if skip is None else skip.shape[3]
    Co = dy.shape[3]
    did = _dt(lo)
    need = lib.ffa_conv_wgrad_workspace_bytes(did, 3, 3, 1, Co, C1 + C2, B, 2 * Hl, 2 * Wl)
    if need < 0:
        return None
    ws = workspace(need, lo.device, "wgrad")
    if out is None:
        out = torch.empty((co_real, C1 + C2, 3, 3), dtype=torch.float32, device=lo.device)
    rc = lib.ffa_conv_wgrad_upcat(did, lo.data_ptr(), _ptr(skip), dy.data_ptr(), out.data_ptr(), B, Hl, Wl, C1, C2, Co,
                                  co_real, 0, ws.data_ptr(), ws.numel(), _stream())
    if rc == _l.ERR_UNSUPPORTED:
        return None
    _l.check(rc, "conv_wgrad_upcat")
    return out


# --------------------------------------------------------------------------------------------------
# normalisation / pooling

def bn_stats(x: torch.Tensor, gamma, beta, running_mean, running_var, momentum: float, eps: float):
    """Batch statistics of x (NHWC) -> (scale, shift, mean, rstd); updates the running buffers in place."""
    lib = _l.load()
    _chk_nhwc(x, "bn input")
    C_ = x.shape[-1]
    npix = x.numel() // C_
    dev = x.device
    out = torch.empty((4, C_), dtype=torch.float32, device=dev)
    ws = workspace(lib.ffa_bn_workspace_bytes(C_), dev, "bn")
    _l.check(lib.ffa_bn_stats(_dt(x), x.data_ptr(), npix, C_, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                              _ptr(running_var), momentum, eps, out[0].data_ptr(), out[1].data_ptr(),
                              out[2].data_ptr(), out[3].data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "bn_stats")
    return out[0], out[1], out[2], out[3]


def bn_eval_params(gamma, beta, running_mean, running_var, eps: float):
    lib = _l.load()
    C_ = running_mean.numel()
    out = torch.empty((2, C_), dtype=torch.float32, device=running_mean.device)
    _l.check(lib.ffa_bn_eval_params(C_, _ptr(gamma), _ptr(beta), running_mean.data_ptr(), running_var.data_ptr(), eps,
                                    out[0].data_ptr(), out[1].data_ptr(), _stream()), "bn_eval_params")
    return out[0], out[1]


def bn_apply(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, residual: Optional[torch.Tensor] = None,
             relu: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _l.load()
    _chk_nhwc(x, "bn_apply input")
    C_ = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    _l.check(lib.ffa_bn_apply(_dt(x), x.data_ptr(), _ptr(residual), out.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                              x.numel() // C_, C_, 1 if relu else 0, _stream()), "bn_apply")
    return out


# one-kernel BatchNorm backward for tensors that fit the chip's registers: OFF by default (FFA_BN_BWD_COOP=1 enables).
# Measured: a grid-wide barrier across the eight XCDs costs ~20-25 us, two of them more than the two launches and
# the two passes they save (72-76 us per call whatever the tensor size, against 24-55 us for reduce + finalize +
# apply on the tensors that fit; 15.49 -> 16.81 ms per training step).  Kept as the measured negative result.
FUSED_BN_BWD_COOP = os.environ.get("FFA_BN_BWD_COOP", "0") == "1"
_COOP_SYNC = {}


def _coop_sync(device) -> torch.Tensor:
    """barrier counters of ffa_bn_bwd_fused: zeroed once per device, re-armed by every launch"""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _COOP_SYNC.get(key)
    if t is None:
        t = _COOP_SYNC[key] = torch.zeros(4, dtype=torch.int32, device=device)
    return t


def coop_barrier_failed(device=None) -> bool:
    """True when a grid barrier of ffa_bn_bwd_fused ever timed out on this device (synchronises)"""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    t = _COOP_SYNC.get(key)
    return bool(t is not None and int(t[3].item()) != 0)


# set by bench.py's roofline pass: called as hook(run_stage, x, has_y, has_dres) instead of the single ffa_bn_bwd call
BN_BWD_STAGE_HOOK = None


def bn_bwd(x: torch.Tensor, dy: torch.Tensor, y: Optional[torch.Tensor], gamma, beta, mean, rstd, relu: bool,
           want_dres: bool, out_dgamma: Optional[torch.Tensor] = None, out_dbeta: Optional[torch.Tensor] = None):
    """-> (dx, dres or None, dgamma, dbeta).  With relu and y=None the ReLU mask is recomputed from x
    (only valid when no residual was added before the ReLU).  out_dgamma / out_dbeta (f32, C elements, contiguous):
    where the affine gradients are written -- the slots of a data-parallel gradient bucket (flairhip.nn._vec_out)."""
    lib = _l.load()
    C_ = x.shape[-1]
    dev = x.device
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    if out_dgamma is not None and out_dbeta is not None:
        for t in (out_dgamma, out_dbeta):
            if t.dtype != torch.float32 or t.numel() != C_ or not t.is_contiguous() or t.device != dev:
                raise ValueError("bn_bwd: output vectors must be contiguous f32 of C elements on the input's device")
        dgb = (out_dgamma, out_dbeta)
    else:
        dgb = torch.empty((2, C_), dtype=torch.float32, device=dev)
    ws = workspace(lib.ffa_bn_workspace_bytes(C_), dev, "bn")
    mode = 0 if not relu else (1 if y is not None else 2)
    if FUSED_BN_BWD_COOP and x.dtype == torch.bfloat16:
        sync = _coop_sync(dev)
        rc = lib.ffa_bn_bwd_fused(_dt(x), x.data_ptr(), dy.data_ptr(), _ptr(y), _ptr(gamma), _ptr(beta),
                                  mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), _ptr(dres), dgb[0].data_ptr(),
                                  dgb[1].data_ptr(), x.numel() // C_, C_, mode, ws.data_ptr(), ws.numel(),
                                  sync.data_ptr(), _stream())
        if rc == 0:
            return dx, dres, dgb[0], dgb[1]
        if rc != _l.ERR_UNSUPPORTED:
            _l.check(rc, "bn_bwd_fused")
    args = (_dt(x), x.data_ptr(), dy.data_ptr(), _ptr(y), _ptr(gamma), _ptr(beta), mean.data_ptr(), rstd.data_ptr(),
            dx.data_ptr(), _ptr(dres), dgb[0].data_ptr(), dgb[1].data_ptr(), x.numel() // C_, C_, mode, ws.data_ptr(),
            ws.numel())
    if BN_BWD_STAGE_HOOK is not None:  # measurement harness: the two stages as separate calls (same kernels)
        BN_BWD_STAGE_HOOK(lambda stages: _l.check(lib.ffa_bn_bwd_stages(*args, stages, _stream()), "bn_bwd"),
                          x, y is not None, dres is not None)
    else:
        _l.check(lib.ffa_bn_bwd(*args, _stream()), "bn_bwd")
    return dx, dres, dgb[0], dgb[1]


def channel_sums(x: torch.Tensor):
    """Per-channel (sum, sum of squares) over all pixels of an NHWC tensor, f32."""
    lib = _l.load()
    _chk_nhwc(x, "channel_sums input")
    C_ = x.shape[-1]
    out = torch.empty((2, C_), dtype=torch.float32, device=x.device)
    ws = workspace(lib.ffa_bn_workspace_bytes(C_), x.device, "bn")
    _l.check(lib.ffa_channel_sums(_dt(x), x.data_ptr(), x.numel() // C_, C_, out[0].data_ptr(), out[1].data_ptr(),
                                  ws.data_ptr(), ws.numel(), _stream()), "channel_sums")
    return out[0], out[1]


def maxpool3x3s2_fwd(x: torch.Tensor):
    lib = _l.load()
    _chk_nhwc(x, "maxpool input")
    B, H, W, C_ = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((B, Ho, Wo, C_), dtype=x.dtype, device=x.device)
    idx = torch.empty((B, Ho, Wo, C_), dtype=torch.uint8, device=x.device)
    _l.check(lib.ffa_maxpool3x3s2_fwd(_dt(x), x.data_ptr(), y.data_ptr(), idx.data_ptr(), B, H, W, C_, _stream()),
             "maxpool_fwd")
    return y, idx


def maxpool3x3s2_bwd(dy: torch.Tensor, idx: torch.Tensor, in_hw: Tuple[int, int],
                     add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = route(dy, idx) (+ add: another gradient of the pooled tensor's input, summed in the same pass)"""
    lib = _l.load()
    B, _, _, C_ = dy.shape
    H, W = in_hw
    dx = torch.empty((B, H, W, C_), dtype=dy.dtype, device=dy.device)
    if add is not None and (tuple(add.shape) != tuple(dx.shape) or add.dtype != dx.dtype or not add.is_contiguous()):
        raise ValueError("maxpool3x3s2_bwd: add must be a contiguous tensor of the input's shape and dtype")
    _l.check(lib.ffa_maxpool3x3s2_bwd(_dt(dy), dy.data_ptr(), idx.data_ptr(), _ptr(add), dx.data_ptr(), B, H, W, C_,
                                      _stream()), "maxpool_bwd")
    return dx


# --------------------------------------------------------------------------------------------------
# layout / resampling

def nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, cp: Optional[int] = None) -> torch.Tensor:
    """f32 NCHW batch tensor -> NHWC compute tensor with zero-filled pad channels."""
    lib = _l.load()
    if x.dtype != torch.float32 or not x.is_contiguous():
        x = x.float().contiguous()
    B, C_, H, W = x.shape
    cp = cp or pad_channels(C_)
    out = torch.empty((B, H, W, cp), dtype=dtype, device=x.device)
    _l.check(lib.ffa_nchw_to_nhwc(_dtype_id(dtype), x.data_ptr(), out.data_ptr(), B, C_, H, W, cp, _stream()),
             "nchw_to_nhwc")
    return out


def u8_nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, mean: torch.Tensor, std: torch.Tensor,
                    cp: Optional[int] = None) -> torch.Tensor:
    """uint8 [B,C,H,W] -> NHWC compute tensor holding (x - mean[c]) / std[c]; mean / std f32 device vectors [C]"""
    lib = _l.load()
    if x.dtype != torch.uint8 or x.ndim != 4 or not x.is_contiguous():
        raise ValueError("u8_nchw_to_nhwc: contiguous uint8 [B,C,H,W] expected")
    B, C_, H, W = x.shape
    if mean.numel() < C_ or std.numel() < C_ or mean.dtype != torch.float32 or std.dtype != torch.float32:
        raise ValueError("u8_nchw_to_nhwc: mean / std must be f32 vectors with one entry per channel")
    cp = pad_channels(C_) if cp is None else cp
    out = torch.empty((B, H, W, cp), dtype=dtype, device=x.device)
    _l.check(lib.ffa_u8_nchw_to_nhwc(_dtype_id(dtype), x.data_ptr(), out.data_ptr(), B, C_, H, W, cp, mean.data_ptr(),
                                     std.data_ptr(), _stream()), "u8_nchw_to_nhwc")
    return out


RAW_SAMPLE_KINDS = {torch.uint8: 0, torch.uint16: 1, torch.int16: 2, torch.float32: 3}  # FFA_SRC_*


def raw_nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, mean: torch.Tensor, std: torch.Tensor,
                     cp: Optional[int] = None) -> torch.Tensor:
    """raw raster samples (uint8 / uint16 / int16 / float32) [B,C,H,W] -> NHWC compute tensor holding
    (x - mean[c]) / std[c]; mean / std f32 device vectors [C]"""
    lib = _l.load()
    if x.dtype not in RAW_SAMPLE_KINDS or x.ndim != 4 or not x.is_contiguous():
        raise ValueError(f"raw_nchw_to_nhwc: contiguous [B,C,H,W] of uint8 / uint16 / int16 / float32 expected, "
                         f"got {x.dtype} {tuple(x.shape)}")
    B, C_, H, W = x.shape
    if mean.numel() < C_ or std.numel() < C_ or mean.dtype != torch.float32 or std.dtype != torch.float32:
        raise ValueError("raw_nchw_to_nhwc: mean / std must be f32 vectors with one entry per channel")
    cp = pad_channels(C_) if cp is None else cp
    out = torch.empty((B, H, W, cp), dtype=dtype, device=x.device)
    _l.check(lib.ffa_raw_nchw_to_nhwc(_dtype_id(dtype), RAW_SAMPLE_KINDS[x.dtype], x.data_ptr(), out.data_ptr(), B, C_,
                                      H, W, cp, mean.data_ptr(), std.data_ptr(), _stream()), "raw_nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, channels: int) -> torch.Tensor:
    lib = _l.load()
    _chk_nhwc(x, "nhwc_to_nchw input")
    B, H, W, cp = x.shape
    out = torch.empty((B, channels, H, W), dtype=torch.float32, device=x.device)
    _l.check(lib.ffa_nhwc_to_nchw(_dt(x), x.data_ptr(), out.data_ptr(), B, channels, H, W, cp, _stream()),
             "nhwc_to_nchw")
    return out


def upsample2x_concat_fwd(lo: torch.Tensor, skip: Optional[torch.Tensor]) -> torch.Tensor:
    lib = _l.load()
    _chk_nhwc(lo, "upsample input")
    B, Hl, Wl, C1 = lo.shape
    C2 = 0 if skip is None else skip.shape[-1]
    if skip is not None and tuple(skip.shape[:3]) != (B, 2 * Hl, 2 * Wl):
        raise ValueError(f"upsample2x_concat: skip {tuple(skip.shape)} does not match 2x of {tuple(lo.shape)}")
    out = torch.empty((B, 2 * Hl, 2 * Wl, C1 + C2), dtype=lo.dtype, device=lo.device)
    _l.check(lib.ffa_upsample_nearest2x_concat_fwd(_dt(lo), lo.data_ptr(), _ptr(skip), out.data_ptr(), B, Hl, Wl, C1,
                                                   C2, _stream()), "upsample2x_concat_fwd")
    return out


def upsample2x_concat_bwd(dcat: torch.Tensor, c1: int, skip_as_view: bool = False):
    """-> (dlo, dskip).  With ``skip_as_view`` the skip gradient is the channel slice of ``dcat`` itself (no copy):
    fine for a consumer that reads strided input, e.g. the sum with the encoder-side gradient of the same map."""
    lib = _l.load()
    _chk_nhwc(dcat, "upsample grad")
    B, H, W, C_ = dcat.shape
    c2 = C_ - c1
    dlo = torch.empty((B, H // 2, W // 2, c1), dtype=dcat.dtype, device=dcat.device)
    dskip = torch.empty((B, H, W, c2), dtype=dcat.dtype, device=dcat.device) if (c2 and not skip_as_view) else None
    _l.check(lib.ffa_upsample_nearest2x_concat_bwd(_dt(dcat), dcat.data_ptr(), dlo.data_ptr(), _ptr(dskip), B, H // 2,
                                                   W // 2, c1, c2, _stream()), "upsample2x_concat_bwd")
    if c2 and skip_as_view:
        dskip = dcat[..., c1:]
    return dlo, dskip


def bilinear_fwd(x: torch.Tensor, out_hw: Tuple[int, int]) -> torch.Tensor:
    lib = _l.load()
    _chk_nhwc(x, "bilinear input")
    B, Hi, Wi, C_ = x.shape
    Ho, Wo = out_hw
    y = torch.empty((B, Ho, Wo, C_), dtype=x.dtype, device=x.device)
    _l.check(lib.ffa_bilinear_fwd(_dt(x), x.data_ptr(), y.data_ptr(), B, Hi, Wi, Ho, Wo, C_, _stream()),
             "bilinear_fwd")
    return y


def bilinear_bwd(dy: torch.Tensor, in_hw: Tuple[int, int]) -> torch.Tensor:
    lib = _l.load()
    _chk_nhwc(dy, "bilinear grad")
    B, Ho, Wo, C_ = dy.shape
    Hi, Wi = in_hw
    dx = torch.empty((B, Hi, Wi, C_), dtype=dy.dtype, device=dy.device)
    # gather-form backward: fixed summation order, no scratch image (the workspace arguments stay in the ABI)
    _l.check(lib.ffa_bilinear_bwd(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, Hi, Wi, Ho, Wo, C_, None, 0, _stream()),
             "bilinear_bwd")
    return dx


# --------------------------------------------------------------------------------------------------
# loss / prediction

def softmax_ce(logits: torch.Tensor, targets: torch.Tensor, class_weights: torch.Tensor, num_classes: int,
               grad_scale: Optional[torch.Tensor] = None, want_grad: bool = False, want_pred: bool = False,
               want_sums: bool = False):
    """Weighted-mean cross-entropy over NHWC logits [B,H,W,Cp] and uint8 targets [B,H,W].

    -> (loss[1] f32, wsum[1] f32, dlogits or None, pred uint8 or None); with want_sums (needs want_grad) a fifth value:
    the per-class sums [Cp] f32 of dlogits over all pixels (the bias gradient of the layer that produced the logits),
    or None where the library cannot take them in the same pass.
    """
    lib = _l.load()
    _chk_nhwc(logits, "logits")
    if targets.dtype != torch.uint8 or not targets.is_contiguous():
        raise ValueError("softmax_ce: targets must be contiguous uint8 class indices")
    cp = logits.shape[-1]
    npix = logits.numel() // cp
    if targets.numel() != npix:
        raise ValueError("softmax_ce: target / logit pixel count mismatch")
    dev = logits.device
    res = torch.empty(2, dtype=torch.float32, device=dev)
    dlogits = torch.empty_like(logits) if want_grad else None
    pred = torch.empty(targets.shape, dtype=torch.uint8, device=dev) if want_pred else None
    if want_grad and grad_scale is None:
        grad_scale = torch.ones(1, dtype=torch.float32, device=dev)
    ws = workspace(lib.ffa_softmax_ce_workspace_bytes(), dev, "ce")
    if want_sums:
        sums = torch.empty(cp, dtype=torch.float32, device=dev) if want_grad else None
        rc = _l.ERR_UNSUPPORTED
        if sums is not None:
            rc = lib.ffa_softmax_ce_sums(_dt(logits), logits.data_ptr(), targets.data_ptr(), class_weights.data_ptr(),
                                         _ptr(grad_scale), res[0:1].data_ptr(), res[1:2].data_ptr(), _ptr(dlogits),
                                         _ptr(pred), sums.data_ptr(), npix, num_classes, cp, ws.data_ptr(), ws.numel(),
                                         _stream())
        if rc == 0:
            return res[0:1], res[1:2], dlogits, pred, sums
        if rc != _l.ERR_UNSUPPORTED:
            _l.check(rc, "softmax_ce_sums")
    _l.check(lib.ffa_softmax_ce(_dt(logits), logits.data_ptr(), targets.data_ptr(), class_weights.data_ptr(),
                                _ptr(grad_scale), res[0:1].data_ptr(), res[1:2].data_ptr(), _ptr(dlogits), _ptr(pred),
                                npix, num_classes, cp, ws.data_ptr(), ws.numel(), _stream()), "softmax_ce")
    if want_sums:
        return res[0:1], res[1:2], dlogits, pred, None
    return res[0:1], res[1:2], dlogits, pred


def scale_inplace(x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """x *= scale[0] (device f32 scalar) in place; the kernel returns at once when the scalar is exactly 1."""
    lib = _l.load()
    if not x.is_contiguous() or x.numel() % 8:
        raise ValueError("scale_inplace: contiguous tensor with a multiple of 8 elements expected")
    _l.check(lib.ffa_scale_inplace(_dt(x), x.data_ptr(), x.numel(), scale.data_ptr(), _stream()), "scale_inplace")
    return x


def predict_u8(logits: torch.Tensor, num_classes: int, mode: str = "argmax", crop: Optional[Tuple[int, int, int, int]] = None
               ) -> torch.Tensor:
    """NHWC logits -> uint8 prediction of the cropped window (y0, x0, h, w).

    'argmax' -> [B, h, w];  'class_prob' -> [B, K, h, w] = rint(softmax * 255)
    (flair_zonal_detection/postprocess.py:9-30 semantics; unknown modes raise ValueError like the reference).
    """
    lib = _l.load()
    _chk_nhwc(logits, "logits")
    if mode not in ("argmax", "class_prob"):
        raise ValueError(f"Unknown output type: {mode}")
    B, H, W, cp = logits.shape
    y0, x0, h, w = crop if crop is not None else (0, 0, H, W)
    shape = (B, h, w) if mode == "argmax" else (B, num_classes, h, w)
    out = torch.empty(shape, dtype=torch.uint8, device=logits.device)
    _l.check(lib.ffa_predict_u8(_dt(logits), 0 if mode == "argmax" else 1, logits.data_ptr(), out.data_ptr(), B, H, W,
                                num_classes, cp, y0, x0, h, w, _stream()), "predict_u8")
    return out


def onehot_to_index(onehot: torch.Tensor) -> torch.Tensor:
    lib = _l.load()
    if onehot.dtype != torch.float32 or not onehot.is_contiguous():
        onehot = onehot.float().contiguous()
    B, K, H, W = onehot.shape
    out = torch.empty((B, H, W), dtype=torch.uint8, device=onehot.device)
    _l.check(lib.ffa_onehot_to_index(onehot.data_ptr(), out.data_ptr(), B, K, H, W, _stream()), "onehot_to_index")
    return out


def confusion_matrix_update(counts: torch.Tensor, pred: torch.Tensor, target: torch.Tensor) -> None:
    """counts[target, pred] += 1 for uint8 class maps (counts: int64 [K, K], updated in place, no host sync)."""
    lib = _l.load()
    if pred.dtype != torch.uint8 or target.dtype != torch.uint8 or not (pred.is_contiguous() and target.is_contiguous()):
        raise ValueError("confusion_matrix_update: pred / target must be contiguous uint8 tensors")
    if counts.dtype != torch.int64 or counts.dim() != 2 or counts.shape[0] != counts.shape[1]:
        raise ValueError("confusion_matrix_update: counts must be int64 [K, K]")
    _l.check(lib.ffa_confusion_matrix(pred.data_ptr(), target.data_ptr(), pred.numel(), counts.shape[0],
                                      counts.data_ptr(), _stream()), "confusion_matrix")


# --------------------------------------------------------------------------------------------------
# host-side tile bookkeeping (no GPU involved)

def slice_grid(min_x, min_y, max_x, max_y, ref_left, ref_bottom, patch_size: int, margin: int, resolution: float):
    lib = _l.load()
    args = (float(min_x), float(min_y), float(max_x), float(max_y), float(ref_left), float(ref_bottom),
            int(patch_size), int(margin), float(resolution))
    n = lib.ffa_slice_grid(*args, None, 0)
    if n < 0:
        _l.check(int(n), "slice_grid")
    buf = (_l.Tile * max(n, 1))()
    n2 = lib.ffa_slice_grid(*args, buf, n)
    assert n2 == n
    return [buf[i] for i in range(n)]


def write_window(left, top, img_bounds, out_res, pred_h: int, pred_w: int):
    lib = _l.load()
    w = _l.Window()
    il, ib, ir, it = img_bounds
    _l.check(lib.ffa_write_window(float(left), float(top), float(il), float(ib), float(ir), float(it), float(out_res),
                                  int(pred_h), int(pred_w), C.byref(w)), "write_window")
    return w


# --------------------------------------------------------------------------------------------------
# U-TAE Sentinel branch (flair_hub/models/multitemp_model.py): small kernels around conv2d

def reflect_pad1(x: torch.Tensor) -> torch.Tensor:
    """[N,H,W,C] -> [N,H+2,W+2,C], reflect padding by one pixel (nn.Conv2d(padding_mode='reflect'))"""
    _chk_nhwc(x, "reflect_pad1 input")
    N, H, W, C = x.shape
    out = torch.empty((N, H + 2, W + 2, C), dtype=x.dtype, device=x.device)
    _l.check(_l.load().ffa_reflect_pad1(_dt(x), x.data_ptr(), out.data_ptr(), N, H, W, C, _stream()), "reflect_pad1")
    return out


def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, relu: bool = False,
               residual: Optional[torch.Tensor] = None, eps: float = 1e-5) -> torch.Tensor:
    """nn.GroupNorm over each image of an NHWC tensor: y = [residual +] relu?(gn(x))"""
    _chk_nhwc(x, "group_norm input")
    N, H, W, C = x.shape
    y = torch.empty_like(x)
    _l.check(_l.load().ffa_group_norm(_dt(x), x.data_ptr(), _ptr(residual), y.data_ptr(), gamma.data_ptr(),
                                      beta.data_ptr(), N, 1, H * W * C, 0, H * W, C, C, groups, eps, 1 if relu else 0,
                                      _stream()), "group_norm")
    return y


def group_norm_seq(x: torch.Tensor, B: int, T: int, gamma: torch.Tensor, beta: torch.Tensor, groups: int,
                   eps: float = 1e-5) -> torch.Tensor:
    """nn.GroupNorm over the T dates of every pixel: x is [B*T, h, w, C] (image n = b*T + t); statistics per
    (b, pixel, group) over T x C/groups values (LTAE2d.in_norm / out_norm with T = 1)"""
    _chk_nhwc(x, "group_norm_seq input")
    N, h, w, C = x.shape
    if N != B * T:
        raise ValueError("group_norm_seq: leading size is not B * T")
    y = torch.empty_like(x)
    P = h * w
    _l.check(_l.load().ffa_group_norm(_dt(x), x.data_ptr(), None, y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      B * P, P, T * P * C, C, T, P * C, C, groups, eps, 0, _stream()), "group_norm")
    return y


def positional_encoding(pos: torch.Tensor, d: int, repeat: int, period: float = 1000.0) -> torch.Tensor:
    """pos f32 [n] -> f32 [n, d * repeat] (PositionalEncoder)"""
    pos = pos.reshape(-1).float().contiguous()
    out = torch.empty((pos.numel(), d * repeat), dtype=torch.float32, device=pos.device)
    _l.check(_l.load().ffa_positional_encoding(pos.data_ptr(), out.data_ptr(), pos.numel(), d, repeat, period,
                                               _stream()), "positional_encoding")
    return out


def add_rowvec_(x: torch.Tensor, vec: torch.Tensor) -> torch.Tensor:
    """x[n, :, :, c] += vec[n, c] in place"""
    _chk_nhwc(x, "add_rowvec input")
    N, H, W, C = x.shape
    if tuple(vec.shape) != (N, C) or vec.dtype != torch.float32 or not vec.is_contiguous():
        raise ValueError("add_rowvec: vec must be contiguous f32 [N, C]")
    _l.check(_l.load().ffa_add_rowvec(_dt(x), x.data_ptr(), vec.data_ptr(), N, H * W, C, _stream()), "add_rowvec")
    return x


def ltae_attention(k: torch.Tensor, v: torch.Tensor, Q: torch.Tensor, pad: torch.Tensor, B: int, T: int):
    """-> (out [B,h,w,n_head*d_v], attn f32 [n_head,B,T,h*w]); k / v are [B*T,h,w,n_head*d_k / n_head*d_v]"""
    _chk_nhwc(k, "ltae keys")
    _chk_nhwc(v, "ltae values")
    N, h, w, KC = k.shape
    n_head, d_k = Q.shape
    d_v = v.shape[-1] // n_head
    if N != B * T or KC != n_head * d_k or v.shape[:3] != k.shape[:3]:
        raise ValueError("ltae_attention: inconsistent shapes")
    out = torch.empty((B, h, w, n_head * d_v), dtype=v.dtype, device=v.device)
    attn = torch.empty((n_head, B, T, h * w), dtype=torch.float32, device=v.device)
    _l.check(_l.load().ffa_ltae_attention(_dt(v), k.data_ptr(), v.data_ptr(), Q.data_ptr(), pad.data_ptr(),
                                          out.data_ptr(), attn.data_ptr(), B, T, h * w, n_head, d_k, d_v, _stream()),
             "ltae_attention")
    return out, attn


def temporal_aggregate(x: torch.Tensor, attn: torch.Tensor, pad: torch.Tensor, B: int, T: int, use_pad: bool
                       ) -> torch.Tensor:
    """x [B*T,H,W,C], attn f32 [n_head,B,T,H*W] -> [B,H,W,C] (Temporal_Aggregator 'att_group')"""
    _chk_nhwc(x, "aggregate input")
    N, H, W, C = x.shape
    out = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    _l.check(_l.load().ffa_temporal_aggregate(_dt(x), x.data_ptr(), attn.data_ptr(), pad.data_ptr(), out.data_ptr(), B,
                                              T, H * W, C, attn.shape[0], 1 if use_pad else 0, _stream()),
             "temporal_aggregate")
    return out


def detect_pad_images(x: torch.Tensor, value: float = 0.0) -> torch.Tensor:
    """x f32 [N, ...] -> u8 [N]: 1 where every element of image n equals `value`"""
    x = x.contiguous()
    N = x.shape[0]
    pad = torch.empty(N, dtype=torch.uint8, device=x.device)
    _l.check(_l.load().ffa_detect_pad_images(x.data_ptr(), pad.data_ptr(), N, x.numel() // N, value, _stream()),
             "detect_pad_images")
    return pad


def mask_images_(x: torch.Tensor, pad: torch.Tensor, value: float = 0.0) -> torch.Tensor:
    N = x.shape[0]
    _l.check(_l.load().ffa_mask_images(_dt(x), x.data_ptr(), pad.data_ptr(), N, x.numel() // N, value, _stream()),
             "mask_images")
    return x


# --------------------------------------------------------------------------------------------------
# Swin-Transformer / UPerNet (csrc/transformer.hip, csrc/gemm.hip)

ACT_NONE, ACT_GELU, ACT_DGELU, ACT_RELU = 0, 1, 2, 3


def linear(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE,
           residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           aux: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
           rows_per_scale: int = 0) -> torch.Tensor:
    """nn.Linear on the last dimension of a token tensor [..., K]: act(x w^T + bias) * row_scale + residual.
    w: [N, K] in x's dtype (nn.Linear.weight's layout), bias f32 [N].  aux ([..., N] bf16): with ACT_GELU it RECEIVES the
    pre-activation, with ACT_DGELU it supplies it (out = (x w^T) * gelu'(aux)).  row_scale f32 [M / rows_per_scale]."""
    if x.dtype not in (torch.bfloat16, torch.float32) or w.dtype != x.dtype:
        raise ValueError("linear: bf16 (MFMA token GEMM) or f32 (parity mode, plain FMA) operands of one dtype")
    if not x.is_contiguous() or not w.is_contiguous():
        raise ValueError("linear: operands must be contiguous")
    K = x.shape[-1]
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"linear: weight {tuple(w.shape)} does not match K = {K}")
    M = x.numel() // K
    if out is None:
        out = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=x.device)
    for name, t in (("residual", residual), ("aux", aux)):
        if t is not None and (t.shape != out.shape or not t.is_contiguous() or t.dtype != x.dtype):
            raise ValueError(f"linear: {name} must be contiguous bf16 and shaped like the output")
    if row_scale is not None and (row_scale.dtype != torch.float32 or rows_per_scale <= 0 or
                                  row_scale.numel() * rows_per_scale < M):
        raise ValueError("linear: row_scale must be f32 with one entry per rows_per_scale rows")
    _l.check(_l.load().ffa_linear_ex(_dt(x), x.data_ptr(), K, w.data_ptr(), _ptr(bias), _ptr(residual), N,
                                     out.data_ptr(), N, M, K, N, act, _ptr(aux), N, _ptr(row_scale), rows_per_scale,
                                     _stream()), "linear")
    return out


def space_to_depth(x: torch.Tensor, ps: int) -> torch.Tensor:
    """[B,H,W,C] -> [B,H/ps,W/ps,ps*ps*C], channel order (dy, dx, c)"""
    _chk_nhwc(x, "space_to_depth input")
    B, H, W, C = x.shape
    if H % ps or W % ps:
        raise ValueError(f"space_to_depth: {H}x{W} is not a multiple of the patch size {ps}")
    out = torch.empty((B, H // ps, W // ps, ps * ps * C), dtype=x.dtype, device=x.device)
    _l.check(_l.load().ffa_space_to_depth(_dt(x), x.data_ptr(), out.data_ptr(), B, H // ps, W // ps, C, ps, _stream()),
             "space_to_depth")
    return out


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
               stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.LayerNorm over the last dimension; ``stats`` (f32 [rows, 2]) receives (mean, rstd) for layer_norm_bwd"""
    if not x.is_contiguous():
        raise ValueError("layer_norm: input must be contiguous")
    C = x.shape[-1]
    out = torch.empty_like(x)
    _l.check(_l.load().ffa_layer_norm(_dt(x), x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      _ptr(stats), x.numel() // C, C, eps, _stream()), "layer_norm")
    return out


def layer_norm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, stats: torch.Tensor,
                   dres: Optional[torch.Tensor] = None):
    """-> (dx [+ dres], dgamma f32, dbeta f32)"""
    C = x.shape[-1]
    rows = x.numel() // C
    lib = _l.load()
    dx = torch.empty_like(x)
    dg = torch.empty(C, dtype=torch.float32, device=x.device)
    db = torch.empty(C, dtype=torch.float32, device=x.device)
    nb = lib.ffa_layer_norm_bwd_workspace_bytes(rows, C)
    ws = workspace(nb, x.device, "ln_bwd")
    _l.check(lib.ffa_layer_norm_bwd(_dt(x), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), stats.data_ptr(),
                                    _ptr(dres), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), rows, C, ws.data_ptr(), ws.numel(),
                                    _stream()), "layer_norm_bwd")
    return dx, dg, db


def patch_merge_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5,
                     stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PatchMerging's gather + LayerNorm(4C): [B,H,W,C] -> [B,H/2,W/2,4C]"""
    _chk_nhwc(x, "patch_merge input")
    B, H, W, C = x.shape
    out = torch.empty((B, H // 2, W // 2, 4 * C), dtype=x.dtype, device=x.device)
    _l.check(_l.load().ffa_patch_merge_norm(_dt(x), x.data_ptr(), out.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                            _ptr(stats), B, H, W, C, eps, _stream()), "patch_merge_norm")
    return out


def patch_merge_norm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, stats: torch.Tensor):
    """x [B,H,W,C] (the forward's input), dy [B,H/2,W/2,4C] -> (dx [B,H,W,C], dgamma, dbeta)"""
    B, H, W, C = x.shape
    rows = B * (H // 2) * (W // 2)
    lib = _l.load()
    dx = torch.empty_like(x)
    dg = torch.empty(4 * C, dtype=torch.float32, device=x.device)
    db = torch.empty(4 * C, dtype=torch.float32, device=x.device)
    nb = lib.ffa_layer_norm_bwd_workspace_bytes(rows, 4 * C)
    ws = workspace(nb, x.device, "ln_bwd")
    _l.check(lib.ffa_patch_merge_norm_bwd(_dt(x), x.data_ptr(), dy.data_ptr(), gamma.data_ptr(), stats.data_ptr(),
                                          dx.data_ptr(), dg.data_ptr(), db.data_ptr(), B, H, W, C, ws.data_ptr(),
                                          ws.numel(), _stream()), "patch_merge_norm_bwd")
    return dx, dg, db


def window_attention(qkv: torch.Tensor, qkv_bias: torch.Tensor, table: torch.Tensor, heads: int, ws: int, shift: int,
                     scale: float) -> torch.Tensor:
    """qkv [B,H,W,3C] -> [B,H,W,C]; table f32 [(2ws-1)^2, heads]"""
    _chk_nhwc(qkv, "window_attention input")
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    if table.shape != ((2 * ws - 1) ** 2, heads) or table.dtype != torch.float32 or not table.is_contiguous():
        raise ValueError("window_attention: relative position bias table must be contiguous f32 [(2ws-1)^2, heads]")
    if qkv_bias.numel() != C3 or qkv_bias.dtype != torch.float32:
        raise ValueError("window_attention: qkv bias must be f32 [3C]")
    out = torch.empty((B, H, W, C), dtype=qkv.dtype, device=qkv.device)
    _l.check(_l.load().ffa_window_attention(_dt(qkv), qkv.data_ptr(), out.data_ptr(), qkv_bias.data_ptr(),
                                            table.data_ptr(), B, H, W, C, heads, ws, shift, scale, _stream()),
             "window_attention")
    return out


def gelu(x: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    _l.check(_l.load().ffa_gelu(_dt(x), x.data_ptr(), out.data_ptr(), x.numel(), _stream()), "gelu")
    return out


def adaptive_avg_pool(x: torch.Tensor, size: int) -> torch.Tensor:
    _chk_nhwc(x, "adaptive_avg_pool input")
    B, H, W, C = x.shape
    out = torch.empty((B, size, size, C), dtype=x.dtype, device=x.device)
    _l.check(_l.load().ffa_adaptive_avg_pool(_dt(x), x.data_ptr(), out.data_ptr(), B, H, W, C, size, _stream()),
             "adaptive_avg_pool")
    return out


def bilinear_slice(x: torch.Tensor, out_hw: Tuple[int, int], out: Optional[torch.Tensor] = None, offset: int = 0,
                   addend: Optional[torch.Tensor] = None, align_corners: bool = False) -> torch.Tensor:
    """F.interpolate(x, out_hw, 'bilinear', align_corners) (+ addend) into out[..., offset:offset + C]"""
    _chk_nhwc(x, "bilinear_slice input")
    B, Hi, Wi, C = x.shape
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty((B, Ho, Wo, C), dtype=x.dtype, device=x.device)
    if out.shape[:3] != (B, Ho, Wo) or not out.is_contiguous() or out.dtype != x.dtype:
        raise ValueError("bilinear_slice: destination does not match")
    if addend is not None and (addend.shape != (B, Ho, Wo, C) or not addend.is_contiguous()):
        raise ValueError("bilinear_slice: addend must be dense [B,Ho,Wo,C]")
    _l.check(_l.load().ffa_bilinear_slice(_dt(x), x.data_ptr(), _ptr(addend), out.data_ptr(), B, Hi, Wi, Ho, Wo, C,
                                          out.shape[-1], offset, 1 if align_corners else 0, _stream()),
             "bilinear_slice")
    return out


def window_attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, qkv_bias: torch.Tensor, table: torch.Tensor, heads: int,
                         ws: int, shift: int, scale: float):
    """-> (dqkv [B,H,W,3C], dtable f32 [(2ws-1)^2, heads], dbias_pad f32 [3C]: what reaches the qkv bias through the
    padding tokens of the window grid)"""
    _chk_nhwc(qkv, "window_attention_bwd qkv")
    _chk_nhwc(dout, "window_attention_bwd dout")
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    if dout.shape != (B, H, W, C) or dout.dtype != qkv.dtype:
        raise ValueError("window_attention_bwd: dout must be [B,H,W,C] in the dtype of qkv")
    lib = _l.load()
    dqkv = torch.empty_like(qkv)
    dtable = torch.empty_like(table)
    dbias = torch.empty(C3, dtype=torch.float32, device=qkv.device)
    wsp = workspace(lib.ffa_window_attention_bwd_workspace_bytes(B, H, W, C, heads, ws), qkv.device, "attn_bwd")
    _l.check(lib.ffa_window_attention_bwd(_dt(qkv), qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(),
                                          qkv_bias.data_ptr(), table.data_ptr(), dtable.data_ptr(), dbias.data_ptr(), B,
                                          H, W, C, heads, ws, shift, scale, wsp.data_ptr(), wsp.numel(), _stream()),
             "window_attention_bwd")
    return dqkv, dtable, dbias


def bilinear_slice_bwd(dy: torch.Tensor, in_hw: Tuple[int, int], channels: int, offset: int = 0,
                       align_corners: bool = False) -> torch.Tensor:
    """gradient of bilinear_slice w.r.t. its input: dy [B,Ho,Wo,pitch] (slice [offset, offset+channels)) -> [B,Hi,Wi,channels]"""
    _chk_nhwc(dy, "bilinear_slice_bwd dy")
    B, Ho, Wo, P = dy.shape
    Hi, Wi = in_hw
    dx = torch.empty((B, Hi, Wi, channels), dtype=dy.dtype, device=dy.device)
    _l.check(_l.load().ffa_bilinear_slice_bwd(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, Hi, Wi, Ho, Wo, channels, P,
                                              offset, 1 if align_corners else 0, _stream()), "bilinear_slice_bwd")
    return dx


def adaptive_avg_pool_bwd(dy: torch.Tensor, in_hw: Tuple[int, int]) -> torch.Tensor:
    _chk_nhwc(dy, "adaptive_avg_pool_bwd dy")
    B, S, _, C = dy.shape
    H, W = in_hw
    dx = torch.empty((B, H, W, C), dtype=dy.dtype, device=dy.device)
    _l.check(_l.load().ffa_adaptive_avg_pool_bwd(_dt(dy), dy.data_ptr(), dx.data_ptr(), B, H, W, C, S, _stream()),
             "adaptive_avg_pool_bwd")
    return dx


def scale_rows(x: torch.Tensor, row_scale: torch.Tensor, rows_per_scale: int) -> torch.Tensor:
    C = x.shape[-1]
    out = torch.empty_like(x)
    _l.check(_l.load().ffa_scale_rows(_dt(x), x.data_ptr(), out.data_ptr(), row_scale.data_ptr(), x.numel() // C, C,
                                      rows_per_scale, _stream()), "scale_rows")
    return out


def column_sums(x: torch.Tensor) -> torch.Tensor:
    """f32 [C]: sum over every row of a contiguous [..., C] tensor (nn.Linear's bias gradient)"""
    if not x.is_contiguous():
        raise ValueError("column_sums: input must be contiguous")
    C = x.shape[-1]
    rows = x.numel() // C
    lib = _l.load()
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = workspace(lib.ffa_column_sums_workspace_bytes(rows, C), x.device, "colsum")
    _l.check(lib.ffa_column_sums(_dt(x), x.data_ptr(), out.data_ptr(), rows, C, ws.data_ptr(), ws.numel(), _stream()),
             "column_sums")
    return out


def updown2x_slice(x: torch.Tensor, channels: int, x_offset: int = 0, out: Optional[torch.Tensor] = None,
                   offset: int = 0) -> torch.Tensor:
    """bilinear x2 then bilinear x1/2 (align_corners=False) of x[..., x_offset:x_offset+channels] in one pass, written to
    out[..., offset:offset+channels] (dense output when out is None); the operator is symmetric (its own backward)"""
    _chk_nhwc(x, "updown2x_slice input")
    B, H, W, P = x.shape
    if out is None:
        out = torch.empty((B, H, W, channels), dtype=x.dtype, device=x.device)
    if out.shape[:3] != (B, H, W) or out.dtype != x.dtype or not out.is_contiguous():
        raise ValueError("updown2x_slice: destination does not match")
    _l.check(_l.load().ffa_updown2x_slice(_dt(x), x.data_ptr(), out.data_ptr(), B, H, W, channels, P, x_offset,
                                          out.shape[-1], offset, _stream()), "updown2x_slice")
    return out


def linear_wgrad(x: torch.Tensor, dy: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False,
                 with_bias: bool = False):
    """dW f32 [N, K] = dy^T x over all rows of the contiguous bf16 tensors x [..., K] and dy [..., N]; with_bias also
    returns db f32 [N] = column sums of dy from the same pass: (dW, db)"""
    if x.dtype not in (torch.bfloat16, torch.float32) or dy.dtype != x.dtype or not x.is_contiguous() or \
            not dy.is_contiguous():
        raise ValueError("linear_wgrad: contiguous bf16 or f32 operands of one dtype")
    K, N = x.shape[-1], dy.shape[-1]
    M = x.numel() // K
    if dy.numel() // N != M:
        raise ValueError("linear_wgrad: x and dy must have the same number of rows")
    lib = _l.load()
    if out is None:
        out = torch.empty((N, K), dtype=torch.float32, device=x.device)
    db = torch.empty(N, dtype=torch.float32, device=x.device) if with_bias else None
    ws = workspace(lib.ffa_linear_wgrad_workspace_bytes(M, N, K), x.device, "lin_wgrad")
    _l.check(lib.ffa_linear_wgrad(_dt(x), x.data_ptr(), K, dy.data_ptr(), N, out.data_ptr(), _ptr(db), M, K, N,
                                  1 if accumulate else 0, ws.data_ptr(), ws.numel(), _stream()), "linear_wgrad")
    return (out, db) if with_bias else out


# --------------------------------------------------------------------------------------------------
# U-TAE training (backward kernels of csrc/temporal.hip)

def reflect_pad1_bwd(dpad: torch.Tensor) -> torch.Tensor:
    _chk_nhwc(dpad, "reflect_pad1_bwd input")
    N, Hp, Wp, C = dpad.shape
    dx = torch.empty((N, Hp - 2, Wp - 2, C), dtype=dpad.dtype, device=dpad.device)
    _l.check(_l.load().ffa_reflect_pad1_bwd(_dt(dpad), dpad.data_ptr(), dx.data_ptr(), N, Hp - 2, Wp - 2, C, _stream()),
             "reflect_pad1_bwd")
    return dx


def _group_norm_bwd(x, dy, gamma, beta, groups, relu, geom, samples, eps):
    lib = _l.load()
    C = x.shape[-1]
    dx = torch.empty_like(x)
    partial = torch.empty((samples, 2 * C), dtype=torch.float32, device=x.device)
    Q, stride_hi, stride_lo, inner, inner_stride = geom
    _l.check(lib.ffa_group_norm_bwd(_dt(x), x.data_ptr(), dy.data_ptr(), dx.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), partial.data_ptr(), samples, Q, stride_hi, stride_lo, inner,
                                    inner_stride, C, groups, eps, 1 if relu else 0, _stream()), "group_norm_bwd")
    sums = column_sums(partial)
    return dx, sums[:C].clone(), sums[C:].clone()


def group_norm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int,
                   relu: bool = False, eps: float = 1e-5):
    """backward of group_norm (per image): -> (dx, dgamma, dbeta); a residual's gradient is dy itself"""
    N, H, W, C = x.shape
    return _group_norm_bwd(x, dy.contiguous(), gamma, beta, groups, relu, (1, H * W * C, 0, H * W, C), N, eps)


def group_norm_seq_bwd(x: torch.Tensor, dy: torch.Tensor, B: int, T: int, gamma: torch.Tensor, beta: torch.Tensor,
                       groups: int, eps: float = 1e-5):
    """backward of group_norm_seq (per pixel over the T dates)"""
    N, h, w, C = x.shape
    P = h * w
    return _group_norm_bwd(x, dy.contiguous(), gamma, beta, groups, False, (P, T * P * C, C, T, P * C), B * P, eps)


def ltae_attention_train(k: torch.Tensor, v: torch.Tensor, Q: torch.Tensor, pad: torch.Tensor, B: int, T: int,
                         drop: Optional[torch.Tensor] = None):
    """-> (out [B,h,w,n_head*d_v], attn f32 [n_head,B,T,h*w] after the attention dropout, prob: the clean softmax)"""
    _chk_nhwc(k, "ltae keys")
    _chk_nhwc(v, "ltae values")
    N, h, w, KC = k.shape
    n_head, d_k = Q.shape
    d_v = v.shape[-1] // n_head
    out = torch.empty((B, h, w, n_head * d_v), dtype=v.dtype, device=v.device)
    attn = torch.empty((n_head, B, T, h * w), dtype=torch.float32, device=v.device)
    prob = torch.empty_like(attn)
    _l.check(_l.load().ffa_ltae_attention_train(_dt(v), k.data_ptr(), v.data_ptr(), Q.data_ptr(), pad.data_ptr(),
                                                _ptr(drop), out.data_ptr(), attn.data_ptr(), prob.data_ptr(), B, T,
                                                h * w, n_head, d_k, d_v, _stream()), "ltae_attention_train")
    return out, attn, prob


def ltae_attention_bwd(k, v, Q, pad, drop, prob, dout, dattn_ext, B: int, T: int):
    """-> (dk, dv, dQ f32 [n_head, d_k])"""
    lib = _l.load()
    N, h, w, KC = k.shape
    n_head, d_k = Q.shape
    d_v = v.shape[-1] // n_head
    dk, dv = torch.empty_like(k), torch.empty_like(v)
    nblk = lib.ffa_ltae_attention_bwd_blocks(B, h * w, n_head)
    part = torch.empty((nblk, n_head * d_k), dtype=torch.float32, device=k.device)
    _l.check(lib.ffa_ltae_attention_bwd(_dt(v), k.data_ptr(), v.data_ptr(), Q.data_ptr(), pad.data_ptr(), _ptr(drop),
                                        prob.data_ptr(), dout.data_ptr(), _ptr(dattn_ext), dk.data_ptr(), dv.data_ptr(),
                                        part.data_ptr(), B, T, h * w, n_head, d_k, d_v, _stream()), "ltae_attention_bwd")
    dq = column_sums(part).view(n_head, d_k)
    return dk, dv, dq


def temporal_aggregate_bwd(x: torch.Tensor, attn: torch.Tensor, pad: torch.Tensor, dout: torch.Tensor, B: int, T: int,
                           use_pad: bool):
    """-> (dx like x, dattn f32 like attn)"""
    N, H, W, C = x.shape
    dx = torch.empty_like(x)
    dattn = torch.empty_like(attn)
    _l.check(_l.load().ffa_temporal_aggregate_bwd(_dt(x), x.data_ptr(), attn.data_ptr(), pad.data_ptr(), dout.data_ptr(),
                                                  dx.data_ptr(), dattn.data_ptr(), B, T, H * W, C, attn.shape[0],
                                                  1 if use_pad else 0, _stream()), "temporal_aggregate_bwd")
    return dx, dattn


def mean_stack(xs, divisor: Optional[float] = None) -> torch.Tensor:
    """(xs[0] + ... + xs[-1]) / divisor (default: their number) -- torch.mean(torch.stack(xs), 0) in one pass"""
    import ctypes as C
    xs = list(xs)
    if not 1 <= len(xs) <= 4:
        raise ValueError("mean_stack: 1 to 4 operands")
    x0 = xs[0]
    for x in xs:
        if x.shape != x0.shape or x.dtype != x0.dtype or not x.is_contiguous():
            raise ValueError("mean_stack: operands must be contiguous tensors of one shape and dtype")
    y = torch.empty_like(x0)
    ptrs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
    _l.check(_l.load().ffa_mean_stack(_dt(x0), ptrs, len(xs), float(len(xs) if divisor is None else divisor), y.data_ptr(),
                                      x0.numel(), _stream()), "mean_stack")
    return y


def mul(x: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    if x.shape != m.shape or x.dtype != m.dtype or not x.is_contiguous() or not m.is_contiguous():
        raise ValueError("mul: operands must be contiguous tensors of one shape and dtype")
    y = torch.empty_like(x)
    _l.check(_l.load().ffa_mul(_dt(x), x.data_ptr(), m.data_ptr(), y.data_ptr(), x.numel(), _stream()), "mul")
    return y
