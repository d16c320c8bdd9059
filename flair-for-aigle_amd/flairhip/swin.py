"""Swin-Transformer encoder + UPerNet decoder on libflairhip kernels.

The reference's default architecture `swin_*-upernet` (configs/train/config_models.yaml:5; the fork's zonal
configuration configs/config_model_zonal_segmentation.yaml:26 runs `swin_base_patch4_window12_384-upernet` on
512 px RGB tiles) comes out of flair_hub/models/monotemp_model.py:64-92 as
``smp.create_model(arch="upernet", encoder_name="tu-swin_...", classes, in_channels, img_size)``, split into
``.encoder`` and ``.decoder`` + ``.segmentation_head`` (:94-97).  Module / parameter names follow
smp 0.4.0 (TimmUniversalEncoder.model = timm FeatureListNet: patch_embed, layers_0..3; UPerNetDecoder: psp,
fpn_stages, fpn_bottleneck; SegmentationHead) so state dicts interchange; `layers.N.` spellings are accepted on load.

Arithmetic (all NHWC, a token = a pixel of the stage's map):
  PatchEmbed        ffa_space_to_depth + token GEMM (+ bias) + ffa_layer_norm
  SwinBlock         ffa_layer_norm -> qkv GEMM -> ffa_window_attention (shift / pad / partition / bias / mask / softmax / PV /
                    reverse by index arithmetic) -> proj GEMM with the residual add in its epilogue -> ffa_layer_norm ->
                    fc1 GEMM with GELU in its epilogue -> fc2 GEMM with the residual add in its epilogue
  PatchMerging      ffa_patch_merge_norm (2x2 gather + LayerNorm(4C)) -> reduction GEMM
  PSP / FPN         ffa_adaptive_avg_pool, 1x1 / 3x3 convolutions with evaluation-mode BatchNorm folded (ffa_conv2d),
                    ffa_bilinear_slice writing straight into the channel slice of the concat buffers (+ lateral addend)
  head              1x1 convolution + bias, x4 bilinear with align_corners=True (nn.UpsamplingBilinear2d)
bf16 mode uses the hand-written token GEMM (csrc/gemm.hip); the f32 parity mode runs the same layers through the f32
1x1 convolution kernel and a separate GELU pass.

Training: every block half / merging / embedding / decoder resampling step is one autograd node whose
backward runs on the same library -- input gradients through the token GEMM with the TRANSPOSED weight (gelu' of the kept
pre-activation and the DropPath factor in its epilogue), weight gradients through ffa_linear_wgrad (transposed-operand
GEMM, deterministic split over the tokens), bias gradients through ffa_channel_sums, ffa_layer_norm_bwd (with the residual
gradient added in the same pass), ffa_window_attention_bwd, ffa_bilinear_slice_bwd, ffa_adaptive_avg_pool_bwd.  timm's
DropPath (stochastic depth, drop_path_rate = 0.1 by default, linearly increasing over the blocks) is the per-sample
row scale of the residual GEMMs.  precision "32" trains through the same autograd nodes and entry points on their f32
kernels (plain-FMA GEMMs and attention backward): a parity mode for gradient checks against the CPU oracle, not a speed path.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import nn as hnn
from . import ops

SWIN_VARIANTS = {
    "tiny": (96, (2, 2, 6, 2), (3, 6, 12, 24)),
    "small": (96, (2, 2, 18, 2), (3, 6, 12, 24)),
    "base": (128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "large": (192, (2, 2, 18, 2), (6, 12, 24, 48)),
}
_NAME = re.compile(r"(?:tu[-_])?swin_(tiny|small|base|large)_patch(\d+)_window(\d+)_(\d+)(?:\..*)?")


def is_swin_name(name: str) -> bool:
    return _NAME.fullmatch(name) is not None


def parse_swin_name(name: str):
    m = _NAME.fullmatch(name)
    if not m:
        raise KeyError(f"not a Swin (v1) encoder name: {name}")
    dim, depths, heads = SWIN_VARIANTS[m.group(1)]
    return dim, depths, heads, int(m.group(3)), int(m.group(2))


class _Affine(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Linear(nn.Module):
    """nn.Linear's parameters (weight [out, in], optional bias); timm's init: trunc_normal(std=.02), zero bias"""

    def __init__(self, cin: int, cout: int, bias: bool = True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.trunc_normal_(self.weight, std=0.02)


class _ConvParams(nn.Module):
    """nn.Conv2d(cin, cout, k, stride=k)'s parameters with nn.Conv2d's default init (PatchEmbed.proj)"""

    def __init__(self, cin: int, cout: int, k: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        bound = 1.0 / (cin * k * k) ** 0.5
        nn.init.uniform_(self.bias, -bound, bound)


def _mod(**children) -> nn.Module:
    m = nn.Module()
    for k, v in children.items():
        setattr(m, k, v)
    return m


class _Attention(nn.Module):
    def __init__(self, dim: int, heads: int, ws: int):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)
        self.qkv = _Linear(dim, 3 * dim)
        self.proj = _Linear(dim, dim)


class _Block(nn.Module):
    def __init__(self, dim: int, resolution: int, heads: int, ws: int, shift: int):
        super().__init__()
        self.dim, self.heads = dim, heads
        self.ws = min(ws, resolution)                 # a window never exceeds the map ...
        self.shift = 0 if resolution <= ws else shift  # ... and a map that fits one window is not shifted
        self.norm1 = _Affine(dim)
        self.attn = _Attention(dim, heads, self.ws)
        self.norm2 = _Affine(dim)
        self.mlp = _mod(fc1=_Linear(dim, 4 * dim), fc2=_Linear(4 * dim, dim))


class _Stage(nn.Module):
    def __init__(self, dim_in: int, dim: int, resolution: int, depth: int, heads: int, ws: int, downsample: bool):
        super().__init__()
        if downsample:
            self.downsample = _mod(norm=_Affine(4 * dim_in), reduction=_Linear(4 * dim_in, dim, bias=False))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[_Block(dim, resolution, heads, ws, 0 if i % 2 == 0 else ws // 2)
                                      for i in range(depth)])


class _Operands:
    """bf16 / packed-f32 copies of the parameters a forward needs, rebuilt when a parameter changes"""

    def __init__(self):
        self.cache: Dict[str, tuple] = {}

    def get(self, tag: str, params: Sequence[Optional[torch.Tensor]], dtype: torch.dtype, build, extra=()):
        ver = tuple((None if p is None else (p._version, p.data_ptr())) for p in params) + (hnn.state_epoch(), dtype) + \
            tuple(extra)
        hit = self.cache.get(tag)
        if hit is None or hit[0] != ver:
            hit = (ver, build())
            self.cache[tag] = hit
        return hit[1]


def _lin_operand(ops_cache: _Operands, tag: str, weight: torch.Tensor, bias: Optional[torch.Tensor], dtype):
    """(weight operand, f32 bias or None): bf16 [N,K] for the token GEMM, a packed 1x1 conv operand in f32 mode"""
    def build():
        w = weight.detach().float().contiguous()
        if dtype == torch.bfloat16:
            return w.to(torch.bfloat16), (None if bias is None else bias.detach().float().contiguous())
        pw = ops.pack_conv_weight(w[:, :, None, None].contiguous(), dtype, 1, w.shape[1], allow_ring=False)
        b = torch.zeros(max(pw.rows, w.shape[0]), dtype=torch.float32, device=w.device)
        if bias is not None:
            b[: w.shape[0]] = bias.detach().float()
        return pw, b
    return ops_cache.get(tag, (weight, bias), dtype, build)


def _apply_linear(x: torch.Tensor, operand, n_out: int, act: int = ops.ACT_NONE,
                  residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    w, b = operand
    if x.dtype == torch.bfloat16:
        return ops.linear(x, w, b, act=act, residual=residual, out=out)
    if act == ops.ACT_NONE:
        return ops.conv2d(x, w, 0, n_out, bias=b, residual=residual, out=out)
    y = ops.gelu(ops.conv2d(x, w, 0, n_out, bias=b))
    if residual is not None:
        raise ValueError("activation and residual are not combined in the f32 path")
    return y


# --------------------------------------------------------------------------------------------------
# training: autograd nodes (bf16 on the MFMA kernels; f32 parity mode on the plain-FMA kernels of the same entry points)

def _wt(cache: _Operands, tag: str, weight: torch.Tensor, dtype: torch.dtype = torch.bfloat16):
    """(W, W^T) in the compute dtype, rebuilt when the parameter changes: forward operand [N,K] and input-gradient
    operand [K,N]"""
    def build():
        w = weight.detach().to(dtype).contiguous()
        return w, w.t().contiguous()
    return cache.get(tag + ":wt", (weight,), dtype, build)


def _wgrad_bias(x: torch.Tensor, dy: torch.Tensor):
    """(dW [N, K], db [N]) f32 of an nn.Linear from one pass over x and dy"""
    return ops.linear_wgrad(x, dy, with_bias=True)


def _wgrad(x: torch.Tensor, dy: torch.Tensor, n_out: int, n_in: int) -> torch.Tensor:
    """dW [n_out, n_in] f32 = dy^T x over all tokens (ffa_linear_wgrad: transposed-operand GEMM, deterministic split)"""
    return ops.linear_wgrad(x, dy)


def _colsum(dy: torch.Tensor) -> torch.Tensor:
    return ops.column_sums(dy)


def _drop_path_scale(rate: float, batch: int, device) -> Optional[torch.Tensor]:
    """timm DropPath(drop_prob, scale_by_keep=True): per sample 0 or 1 / keep_prob"""
    if rate <= 0.0:
        return None
    keep = 1.0 - rate
    return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep).div_(keep)


class _AttnHalf(torch.autograd.Function):
    """x + DropPath(proj(window_attention(qkv(LayerNorm(x)))))  -- the first half of a SwinTransformerBlock"""

    @staticmethod
    def forward(ctx, x, g1, b1, wqkv, bqkv, table, wproj, bproj, enc, blk, tag, rs):
        C = blk.dim
        rows = x.numel() // C
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        h = ops.layer_norm(x, g1.detach(), b1.detach(), stats=stats)
        wq, _ = _wt(enc._ops, tag + "qkv", wqkv, x.dtype)
        qkv = ops.linear(h, wq, bqkv.detach())
        scale = float((C // blk.heads) ** -0.5)
        att = ops.window_attention(qkv, bqkv.detach(), table.detach().contiguous(), blk.heads, blk.ws, blk.shift, scale)
        wp, _ = _wt(enc._ops, tag + "proj", wproj, x.dtype)
        rps = x.shape[1] * x.shape[2]
        y = ops.linear(att, wp, bproj.detach(), residual=x, row_scale=rs, rows_per_scale=rps if rs is not None else 0)
        ctx.enc, ctx.blk, ctx.tag, ctx.scale, ctx.rps = enc, blk, tag, scale, rps
        ctx.save_for_backward(x, stats, h, qkv, att, g1, wqkv, bqkv, table, wproj, rs)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, h, qkv, att, g1, wqkv, bqkv, table, wproj, rs = ctx.saved_tensors
        enc, blk, tag = ctx.enc, ctx.blk, ctx.tag
        C = blk.dim
        dy = dy.contiguous()
        dys = dy if rs is None else ops.scale_rows(dy, rs, ctx.rps)
        _, wpt = _wt(enc._ops, tag + "proj", wproj, x.dtype)
        datt = ops.linear(dys, wpt)
        dwp, dbp = _wgrad_bias(att, dys)
        dqkv, dtable, dbpad = ops.window_attention_bwd(qkv, datt, bqkv.detach(), table.detach().contiguous(), blk.heads,
                                                       blk.ws, blk.shift, ctx.scale)
        _, wqt = _wt(enc._ops, tag + "qkv", wqkv, x.dtype)
        dh = ops.linear(dqkv, wqt)
        dwq, dbq = _wgrad_bias(h, dqkv)
        dbq = dbq + dbpad
        dx, dg1, db1 = ops.layer_norm_bwd(x, dh, g1.detach(), stats, dres=dy)
        return dx, dg1, db1, dwq, dbq, dtable, dwp, dbp, None, None, None, None


class _MlpHalf(torch.autograd.Function):
    """x + DropPath(fc2(gelu(fc1(LayerNorm(x)))))  -- the second half of a SwinTransformerBlock"""

    @staticmethod
    def forward(ctx, x, g2, b2, w1, bb1, w2, bb2, enc, blk, tag, rs):
        C = blk.dim
        rows = x.numel() // C
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        h = ops.layer_norm(x, g2.detach(), b2.detach(), stats=stats)
        w1b, _ = _wt(enc._ops, tag + "fc1", w1, x.dtype)
        u = torch.empty(x.shape[:-1] + (4 * C,), dtype=x.dtype, device=x.device)
        a = ops.linear(h, w1b, bb1.detach(), act=ops.ACT_GELU, aux=u)
        w2b, _ = _wt(enc._ops, tag + "fc2", w2, x.dtype)
        rps = x.shape[1] * x.shape[2]
        y = ops.linear(a, w2b, bb2.detach(), residual=x, row_scale=rs, rows_per_scale=rps if rs is not None else 0)
        ctx.enc, ctx.blk, ctx.tag, ctx.rps = enc, blk, tag, rps
        ctx.save_for_backward(x, stats, h, u, a, g2, w1, w2, rs)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, h, u, a, g2, w1, w2, rs = ctx.saved_tensors
        enc, blk, tag = ctx.enc, ctx.blk, ctx.tag
        C = blk.dim
        dy = dy.contiguous()
        dys = dy if rs is None else ops.scale_rows(dy, rs, ctx.rps)
        _, w2t = _wt(enc._ops, tag + "fc2", w2, x.dtype)
        du = ops.linear(dys, w2t, act=ops.ACT_DGELU, aux=u)  # (dys W2) * gelu'(u)
        dw2, db2 = _wgrad_bias(a, dys)
        _, w1t = _wt(enc._ops, tag + "fc1", w1, x.dtype)
        dh = ops.linear(du, w1t)
        dw1, db1 = _wgrad_bias(h, du)
        dx, dg, db = ops.layer_norm_bwd(x, dh, g2.detach(), stats, dres=dy)
        return dx, dg, db, dw1, db1, dw2, db2, None, None, None, None


class _PatchMerge(torch.autograd.Function):
    """reduction(LayerNorm(2x2 gather))"""

    @staticmethod
    def forward(ctx, x, g, b, wred, enc, tag):
        B, H, W, C = x.shape
        stats = torch.empty((B * (H // 2) * (W // 2), 2), dtype=torch.float32, device=x.device)
        m = ops.patch_merge_norm(x, g.detach(), b.detach(), stats=stats)
        wr, _ = _wt(enc._ops, tag, wred, x.dtype)
        y = ops.linear(m, wr)
        ctx.enc, ctx.tag = enc, tag
        ctx.save_for_backward(x, stats, m, g, wred)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, m, g, wred = ctx.saved_tensors
        dy = dy.contiguous()
        _, wrt = _wt(ctx.enc._ops, ctx.tag, wred, x.dtype)
        dm = ops.linear(dy, wrt)
        dw = _wgrad(m, dy, wred.shape[0], wred.shape[1])
        dx, dg, db = ops.patch_merge_norm_bwd(x, dm, g.detach(), stats)
        return dx, dg, db, dw, None, None


class _PatchEmbed(torch.autograd.Function):
    """LayerNorm(Conv2d(k = s = patch)(x)) on the NHWC input tensor (which needs no gradient)"""

    @staticmethod
    def forward(ctx, x, wproj, bproj, g, b, enc):
        ps, cp = enc.patch, x.shape[-1]
        dim, cin = wproj.shape[0], wproj.shape[1]
        w2 = torch.zeros(dim, ps, ps, cp, dtype=torch.float32, device=x.device)
        w2[..., :cin] = wproj.detach().permute(0, 2, 3, 1)
        s2d = ops.space_to_depth(x, ps)
        t = ops.linear(s2d, w2.reshape(dim, -1).to(x.dtype), bproj.detach())
        stats = torch.empty((t.numel() // dim, 2), dtype=torch.float32, device=x.device)
        y = ops.layer_norm(t, g.detach(), b.detach(), stats=stats)
        ctx.geom = (ps, cp, dim, cin)
        ctx.save_for_backward(s2d, t, stats, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        s2d, t, stats, g = ctx.saved_tensors
        ps, cp, dim, cin = ctx.geom
        dt, dg, db = ops.layer_norm_bwd(t, dy.contiguous(), g.detach(), stats)
        dw2, dbias = _wgrad_bias(s2d, dt)
        dw = dw2.view(dim, ps, ps, cp)[..., :cin].permute(0, 3, 1, 2).contiguous()
        return None, dw, dbias, dg, db, None


class _AdaptivePool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.hw = (x.shape[1], x.shape[2])
        return ops.adaptive_avg_pool(x, size)

    @staticmethod
    def backward(ctx, dy):
        return ops.adaptive_avg_pool_bwd(dy.contiguous(), ctx.hw), None


class _ResizeCat(torch.autograd.Function):
    """cat([bilinear(t, out_hw) for t in tensors], channel) assembled in place (align_corners=False).  updown_last: one
    more slice = the LAST tensor upsampled x2 and resized back (UPerNet's placeholder FPN stage), as a 3-tap filter"""

    @staticmethod
    def forward(ctx, out_hw, updown_last, *tensors):
        B = tensors[0].shape[0]
        widths = [t.shape[-1] for t in tensors]
        total = sum(widths) + (widths[-1] if updown_last else 0)
        out = torch.empty((B, out_hw[0], out_hw[1], total), dtype=tensors[0].dtype, device=tensors[0].device)
        off = 0
        for t, c in zip(tensors, widths):
            ops.bilinear_slice(t, out_hw, out=out, offset=off)
            off += c
        if updown_last:
            ops.updown2x_slice(tensors[-1], widths[-1], out=out, offset=off)
        ctx.shapes = [(t.shape[1], t.shape[2], t.shape[3]) for t in tensors]
        ctx.updown_last = updown_last
        return out

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        grads, off = [], 0
        for i, (h, w, c) in enumerate(ctx.shapes):
            grads.append(ops.bilinear_slice_bwd(dy, (h, w), c, offset=off) if ctx.needs_input_grad[i + 2] else None)
            off += c
        if ctx.updown_last and grads[-1] is not None:
            h, w, c = ctx.shapes[-1]
            grads[-1] = grads[-1] + ops.updown2x_slice(dy, c, x_offset=off)  # the filter is its own transpose
        return (None, None, *grads)


class _ResizeAdd(torch.autograd.Function):
    """bilinear(x, out_hw, align_corners) (+ addend)"""

    @staticmethod
    def forward(ctx, x, addend, out_hw, align):
        ctx.hw, ctx.align, ctx.has_add = (x.shape[1], x.shape[2]), align, addend is not None
        return ops.bilinear_slice(x, out_hw, addend=addend, align_corners=align)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = ops.bilinear_slice_bwd(dy, ctx.hw, dy.shape[-1], align_corners=ctx.align) if ctx.needs_input_grad[0] else None
        return dx, (dy if ctx.has_add else None), None, None


class HipSwinEncoder(nn.Module):
    """smp's TimmUniversalEncoder over a timm Swin (features_only): forward(x_nhwc) ->
    [x, 0-channel placeholder at stride 2, f4, f8, f16, f32] (NHWC); ``out_channels`` = [in, 0, C, 2C, 4C, 8C]
    (the reference strips the two leading entries itself: flair_model.py:302-306, 506-518)."""

    def __init__(self, name: str, in_channels: int = 3, img_size: int = 512, drop_path_rate: float = 0.1):
        super().__init__()
        dim, depths, heads, ws, patch = parse_swin_name(name)
        # timm: SwinTransformer(drop_path_rate=0.1), the rate of block i = linspace(0, rate, sum(depths))[i]
        total = sum(depths)
        self.drop_path_rates = [drop_path_rate * i / max(total - 1, 1) for i in range(total)]
        if (dim // heads[0]) != 32:
            raise NotImplementedError("window attention kernels are built for a head dimension of 32")
        if ws > 12:
            raise NotImplementedError(f"window size {ws} > 12")
        self.name, self.in_channels, self.img_size, self.patch = name, in_channels, img_size, patch
        # channel pitch the model glue gives the NHWC input tensor: 8 instead of the conv stack's 16 halves the
        # space-to-depth tensor and the K of the patch-embedding GEMM (3 ... 5 real channels)
        self.input_pitch = 8 if in_channels <= 8 else ops.pad_channels(in_channels)
        self.dims = [dim * 2 ** i for i in range(4)]
        self.out_channels = [in_channels, 0] + self.dims
        model = nn.Module()
        model.patch_embed = _mod(proj=_ConvParams(in_channels, dim, patch), norm=_Affine(dim))
        res = img_size // patch
        for i in range(4):
            if i > 0:
                res = (res + 1) // 2
            setattr(model, f"layers_{i}", _Stage(self.dims[i - 1] if i else dim, self.dims[i], res, depths[i], heads[i],
                                                 ws, i > 0))
        self.model = model
        self._ops = _Operands()
        self._register_load_state_dict_pre_hook(self._rename_timm_keys)

    @staticmethod
    def _rename_timm_keys(state_dict, prefix, *args):
        """accept timm's un-flattened `layers.N.` spelling next to FeatureListNet's `layers_N.`"""
        for k in list(state_dict.keys()):
            if k.startswith(prefix) and ".layers." in k[len(prefix) - 1:]:
                nk = k[: len(prefix)] + re.sub(r"(^|\.)layers\.(\d+)\.", r"\1layers_\2.", k[len(prefix):])
                if nk != k and nk not in state_dict:
                    state_dict[nk] = state_dict.pop(k)

    # ---- layers --------------------------------------------------------------------------------------

    def _patch_embed(self, x: torch.Tensor) -> torch.Tensor:
        pe = self.model.patch_embed
        cp, ps = x.shape[-1], self.patch

        def build():  # Conv2d(k = s = patch) as a GEMM over (dy, dx, c) with the input's channel pitch
            w = pe.proj.weight.detach().float()
            dim, cin = w.shape[0], w.shape[1]
            w2 = torch.zeros(dim, ps, ps, cp, dtype=torch.float32, device=w.device)
            w2[..., :cin] = w.permute(0, 2, 3, 1)
            return w2.reshape(dim, ps * ps * cp)
        w2 = self._ops.get(f"pe_w{cp}", (pe.proj.weight,), torch.float32, build)
        operand = _lin_operand(self._ops, f"pe{cp}", w2, pe.proj.bias, x.dtype)
        t = _apply_linear(ops.space_to_depth(x, ps), operand, self.dims[0])
        return ops.layer_norm(t, pe.norm.weight.detach(), pe.norm.bias.detach())

    def _block(self, x: torch.Tensor, blk: _Block, tag: str) -> torch.Tensor:
        C, a, dt = blk.dim, blk.attn, x.dtype
        h = ops.layer_norm(x, blk.norm1.weight.detach(), blk.norm1.bias.detach())
        qkv_op = _lin_operand(self._ops, tag + "qkv", a.qkv.weight, a.qkv.bias, dt)
        qkv = _apply_linear(h, qkv_op, 3 * C)
        att = ops.window_attention(qkv, a.qkv.bias.detach(), a.relative_position_bias_table.detach().contiguous(),
                                   blk.heads, blk.ws, blk.shift, float((C // blk.heads) ** -0.5))
        x = _apply_linear(att, _lin_operand(self._ops, tag + "proj", a.proj.weight, a.proj.bias, dt), C, residual=x,
                          out=x)
        h = ops.layer_norm(x, blk.norm2.weight.detach(), blk.norm2.bias.detach())
        h = _apply_linear(h, _lin_operand(self._ops, tag + "fc1", blk.mlp.fc1.weight, blk.mlp.fc1.bias, dt), 4 * C,
                          act=ops.ACT_GELU)
        return _apply_linear(h, _lin_operand(self._ops, tag + "fc2", blk.mlp.fc2.weight, blk.mlp.fc2.bias, dt), C,
                             residual=x, out=x)

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        if not x.is_cuda:
            raise RuntimeError("HipSwinEncoder runs on the MI355X only (no CPU path in the product)")
        B, H, W, _ = x.shape
        if H % (self.patch * 8) or W % (self.patch * 8):
            raise ValueError(f"input {H}x{W} must be a multiple of {self.patch * 8}")
        train = self.training and torch.is_grad_enabled()
        feats = [x, x.new_empty((B, H // 2, W // 2, 0))]
        m = self.model
        if train:
            pe = m.patch_embed
            t = _PatchEmbed.apply(x, pe.proj.weight, pe.proj.bias, pe.norm.weight, pe.norm.bias, self)
        else:
            t = self._patch_embed(x)
        k = 0
        for i in range(4):
            stage = getattr(m, f"layers_{i}")
            if i > 0:
                ds = stage.downsample
                if train:
                    t = _PatchMerge.apply(t, ds.norm.weight, ds.norm.bias, ds.reduction.weight, self, f"s{i}red")
                else:
                    mm = ops.patch_merge_norm(t, ds.norm.weight.detach(), ds.norm.bias.detach())
                    t = _apply_linear(mm, _lin_operand(self._ops, f"s{i}red", ds.reduction.weight, None, t.dtype),
                                      self.dims[i])
            for j, blk in enumerate(stage.blocks):
                if min(t.shape[1], t.shape[2]) < blk.ws:
                    raise ValueError(f"stage {i} map {t.shape[1]}x{t.shape[2]} is smaller than the window {blk.ws} the "
                                     f"encoder was built for (img_size={self.img_size})")
                tag = f"s{i}b{j}"
                if train:
                    a, rate = blk.attn, self.drop_path_rates[k]
                    t = _AttnHalf.apply(t, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias,
                                        a.relative_position_bias_table, a.proj.weight, a.proj.bias, self, blk, tag,
                                        _drop_path_scale(rate, B, t.device))
                    t = _MlpHalf.apply(t, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias,
                                       blk.mlp.fc2.weight, blk.mlp.fc2.bias, self, blk, tag,
                                       _drop_path_scale(rate, B, t.device))
                else:
                    t = self._block(t, blk, tag)
                k += 1
            feats.append(t)
        return feats


# --------------------------------------------------------------------------------------------------
# UPerNet decoder + head (smp 0.4.0 decoders/upernet/decoder.py, base/heads.py)

def _conv_bn(cin: int, cout: int, k: int, padding: int = 0) -> nn.Sequential:
    """smp Conv2dReLU(use_batchnorm=True): children 0 = conv (no bias), 1 = BatchNorm2d, 2 = ReLU (stateless)"""
    return nn.Sequential(hnn.HipConv2d(cin, cout, k, 1, padding), hnn.HipBatchNorm2d(cout))


class _Slot(nn.Module):
    """stateless placeholder keeping nn.Sequential indices aligned with smp (AdaptiveAvgPool2d at index 0)"""


class HipUPerNetDecoder(nn.Module):
    def __init__(self, encoder_channels: Sequence[int], pyramid_channels: int = 256, segmentation_channels: int = 64,
                 sizes: Sequence[int] = (1, 2, 3, 6)):
        super().__init__()
        ch = list(encoder_channels)[::-1]  # [8C, 4C, 2C, C, 0, in]
        if any(c % 16 for c in ch[:-2]) or ch[0] % (16 * len(sizes)):
            raise NotImplementedError(f"encoder widths {ch} must be multiples of 16")
        self.sizes, self.pyramid = tuple(sizes), pyramid_channels
        psp = nn.Module()
        psp.blocks = nn.ModuleList(nn.Sequential(_Slot(), _conv_bn(ch[0], ch[0] // len(sizes), 1)) for _ in sizes)
        psp.out_conv = _conv_bn(ch[0] * 2, pyramid_channels, 1)
        self.psp = psp
        # one FPNBlock per remaining encoder channel entry (the last, for the input image, is never called: smp's
        # forward drops features[0]); a 0-channel skip has no lateral conv
        self.fpn_stages = nn.ModuleList(
            _mod(skip_conv=(_conv_bn(c, pyramid_channels, 1) if c != 0 else nn.Identity())) for c in ch[1:])
        self.fpn_bottleneck = _conv_bn((len(ch) - 1) * pyramid_channels, segmentation_channels, 3, padding=1)
        self._fold = _Operands()
        for m in self.modules():  # smp initialize_decoder
            if isinstance(m, hnn.HipConv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")

    def _conv1x1(self, x: torch.Tensor, seq) -> torch.Tensor:
        """evaluation-mode relu(BatchNorm(conv1x1(x))): in bf16 the token GEMM with the folded operand (scale * W as bf16
        [N, K], shift as its bias, ReLU in the epilogue) -- the PSP branches are GEMMs of 8 ... 288 tokens, which the
        pixel-tiled convolution kernel serves badly; f32 keeps the convolution kernel"""
        conv, bn = seq[0], seq[1]
        if x.dtype != torch.bfloat16 or x.shape[-1] != conv.in_channels or conv.out_channels % 8:
            return hnn.conv_bn_act(x, conv, bn, relu=True)

        def build():
            scale, shift = ops.bn_eval_params(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
            n = conv.out_channels
            w = conv.weight.detach().float().view(n, -1) * scale[:n, None]
            return w.to(torch.bfloat16).contiguous(), shift[:n].contiguous()
        # the training kernels update the running statistics through raw pointers: their epoch counter is part of the key
        w, b = self._fold.get(f"c{id(conv)}", (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var),
                              torch.bfloat16, build, extra=(getattr(bn, "_stats_epoch", 0),))
        return ops.linear(x, w, b, act=ops.ACT_RELU)

    @staticmethod
    def _is_updown(prev: torch.Tensor, size, target) -> bool:
        """the placeholder FPN stage doubles `prev` and the decoder resizes it straight back to `prev`'s own size"""
        h, w = prev.shape[1], prev.shape[2]
        return tuple(size) == (2 * h, 2 * w) and tuple(target) == (h, w)

    def _forward_train(self, *features: torch.Tensor) -> torch.Tensor:
        """the same graph through autograd nodes (training-mode BatchNorm inside conv_bn_act)"""
        H, W = features[0].shape[1], features[0].shape[2]
        target = (H // 4, W // 4)
        feats = list(features[1:])[::-1]
        x = feats[0]
        h, w = x.shape[1], x.shape[2]
        pooled = [hnn.conv_bn_act(_AdaptivePool.apply(x, s), blk[1][0], blk[1][1], relu=True)
                  for s, blk in zip(self.sizes, self.psp.blocks)]
        cat = _ResizeCat.apply((h, w), False, x, *pooled)
        fpn = [hnn.conv_bn_act(cat, self.psp.out_conv[0], self.psp.out_conv[1], relu=True)]
        updown = False
        for f, stage in zip(feats[1:], self.fpn_stages):
            size = (f.shape[1], f.shape[2])
            if f.shape[-1] == 0 and self._is_updown(fpn[-1], size, target):
                updown = True  # placeholder stage: x2 up, nothing added, resized back below -- never materialised
                continue
            lat = hnn.conv_bn_act(f, stage.skip_conv[0], stage.skip_conv[1], relu=True) if f.shape[-1] != 0 else None
            fpn.append(_ResizeAdd.apply(fpn[-1], lat, size, False))
        wide = _ResizeCat.apply(target, updown, *fpn)
        return hnn.conv_bn_act(wide, self.fpn_bottleneck[0], self.fpn_bottleneck[1], relu=True)

    def forward(self, *features: torch.Tensor) -> torch.Tensor:
        if self.training and torch.is_grad_enabled():
            return self._forward_train(*features)
        H, W = features[0].shape[1], features[0].shape[2]
        target = (H // 4, W // 4)
        feats = list(features[1:])[::-1]
        x = feats[0]
        B, h, w, C = x.shape
        # PSP: [x | up(conv(pool_s(x))) for s] assembled in place, then the 1x1 out_conv
        q = C // len(self.sizes)
        cat = torch.empty((B, h, w, 2 * C), dtype=x.dtype, device=x.device)
        ops.bilinear_slice(x, (h, w), out=cat, offset=0)  # same-size resize = copy into the slice
        for i, (s, blk) in enumerate(zip(self.sizes, self.psp.blocks)):
            p = self._conv1x1(ops.adaptive_avg_pool(x, s), blk[1])
            ops.bilinear_slice(p, (h, w), out=cat, offset=C + i * q)
        top = self._conv1x1(cat, self.psp.out_conv)
        fpn = [top]
        updown = False
        for f, stage in zip(feats[1:], self.fpn_stages):
            size = (f.shape[1], f.shape[2])
            if f.shape[-1] != 0:
                lat = self._conv1x1(f, stage.skip_conv)
                fpn.append(ops.bilinear_slice(fpn[-1], size, addend=lat))
            elif self._is_updown(fpn[-1], size, target):
                updown = True  # x2 up + resize back = one 3-tap filter pass below; the stride-2 map is never written
            else:
                fpn.append(ops.bilinear_slice(fpn[-1], size))
        P = self.pyramid
        wide = torch.empty((B, target[0], target[1], (len(fpn) + int(updown)) * P), dtype=x.dtype, device=x.device)
        for i, f in enumerate(fpn):
            ops.bilinear_slice(f, target, out=wide, offset=i * P)
        if updown:
            ops.updown2x_slice(fpn[-1], P, out=wide, offset=len(fpn) * P)
        return hnn.conv_bn_act(wide, self.fpn_bottleneck[0], self.fpn_bottleneck[1], relu=True)


class HipUPerNetHead(nn.Sequential):
    """smp SegmentationHead(kernel_size=1, upsampling=4): Conv2d + bias, nn.UpsamplingBilinear2d (align_corners=True)"""

    def __init__(self, cin: int, classes: int, upsampling: int = 4):
        super().__init__(hnn.HipConv2d(cin, classes, 1, 1, 0, bias=True), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self[0].weight)
        nn.init.constant_(self[0].bias, 0)
        self.classes, self.upsampling = classes, upsampling

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = hnn.conv_bias(x, self[0])
        if self.upsampling > 1:
            size = (y.shape[1] * self.upsampling, y.shape[2] * self.upsampling)
            if torch.is_grad_enabled() and y.requires_grad:
                return _ResizeAdd.apply(y, None, size, True)
            y = ops.bilinear_slice(y, size, align_corners=True)
        return y


class SwinUPerNet(nn.Module):
    """Counterpart of smp.create_model('upernet', 'tu-swin_...', classes=..., in_channels=..., img_size=...)."""

    def __init__(self, encoder_name: str, in_channels: int = 3, classes: int = 1, img_size: int = 512,
                 drop_path_rate: float = 0.1):
        super().__init__()
        self.encoder = HipSwinEncoder(encoder_name, in_channels, img_size, drop_path_rate)
        self.decoder = HipUPerNetDecoder(self.encoder.out_channels)
        self.segmentation_head = HipUPerNetHead(64, classes)

    def forward(self, x_nhwc: torch.Tensor) -> torch.Tensor:
        return self.segmentation_head(self.decoder(*self.encoder(x_nhwc)))
