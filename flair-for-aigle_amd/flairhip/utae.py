"""U-TAE, the Sentinel time-series encoder of FLAIR-HUB, on libflairhip kernels (evaluation-mode forward).

Counterpart of the reference's flair_hub/models/multitemp_model.py (UTAE :13-166, LTAE2d :169-284,
MultiHeadAttention :316-373, ConvLayer / ConvBlock / DownConvBlock / UpConvBlock :452-599, Temporal_Aggregator
:603-662) with the SAME parameter names, so a reference state dict loads unchanged.  Arithmetic: NHWC tensors with
image index n = b * T + t; reflect-padded 3x3 convolutions = ffa_reflect_pad1 + ffa_conv2d(pad 0) with the bias in
the conv epilogue; GroupNorm / attention / aggregation are the kernels of csrc/temporal.hip; evaluation-mode
BatchNorm is folded into the packed conv operand (scale) and its bias (shift), as everywhere else in the product.

Scope: the stride-1 configuration FLAIR hard-codes (flair_zonal_detection/model_utils.py:55-71:
str_conv k=3, s=1, p=1, agg_mode 'att_group', encoder_norm 'group', padding_mode 'reflect').  Training runs through
autograd nodes whose backward stays on the library (ffa_group_norm_bwd, ffa_reflect_pad1_bwd, ffa_ltae_attention_train /
_bwd with the reference's attention dropout 0.1, ffa_temporal_aggregate_bwd, the conv dgrad / wgrad kernels, training-mode
BatchNorm, nn.Dropout(0.2) after the L-TAE MLP as a mask multiply); there is no torch fallback.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn

from . import nn as hnn
from . import ops


class _Affine(nn.Module):
    """weight / bias holder (GroupNorm affine parameters)"""

    def __init__(self, c: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _Weights(nn.Module):
    """weight / bias holder with an arbitrary weight shape (Conv2d, ConvTranspose2d, Conv1d, Linear)"""

    def __init__(self, shape, n_bias: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*shape))
        self.bias = nn.Parameter(torch.zeros(n_bias))
        fan_in = self.weight[0].numel()
        nn.init.normal_(self.weight, std=(2.0 / max(fan_in, 1)) ** 0.5)


class _Slot(nn.Module):
    """parameter-free placeholder that keeps nn.Sequential indices aligned with the reference (its ReLU entries)"""


def _seq(*mods) -> nn.Sequential:
    return nn.Sequential(*mods)


def _conv_layer(nkernels, norm: str, k: int = 3) -> nn.Module:
    """ConvLayer (:452-497): .conv = Sequential(conv, norm, relu, conv, norm, relu, ...)"""
    layers = []
    for i in range(len(nkernels) - 1):
        layers.append(_Weights((nkernels[i + 1], nkernels[i], k, k), nkernels[i + 1]))
        layers.append(_Affine(nkernels[i + 1]) if norm == "group" else hnn.HipBatchNorm2d(nkernels[i + 1]))
        layers.append(_Slot())
    m = nn.Module()
    m.conv = _seq(*layers)
    return m


def _block(**children) -> nn.Module:
    m = nn.Module()
    for k, v in children.items():
        setattr(m, k, v)
    return m


# --------------------------------------------------------------------------------------------------
# training: autograd nodes

def _wgrad_conv(x, dy, weight_shape, pad):
    co, ci, kh, kw = weight_shape
    return ops.conv_wgrad(x, dy, co, ci, kh, kw, 1, pad)


def _bias_grad(d0, n):
    return ops.column_sums(d0)[:n].clone()


class _ConvGN(torch.autograd.Function):
    """[residual +] relu(GroupNorm4(conv3x3_reflect(x) + bias)): ConvLayer with norm='group' (:452-497)"""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, residual, net, tag, holder):
        xp = ops.reflect_pad1(x)
        pw, bias = net._packed(tag, holder, x.shape[-1])
        y0 = ops.conv2d(xp, pw, 0, ops.pad_channels(w.shape[0]), bias=bias)
        g, be = gamma.detach().float(), beta.detach().float()
        y = ops.group_norm(y0, g, be, 4, relu=True, residual=residual)
        ctx.net, ctx.tag, ctx.holder, ctx.has_res = net, tag, holder, residual is not None
        ctx.save_for_backward(xp, y0, g, be, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, y0, g, be, w = ctx.saved_tensors
        dy = dy.contiguous()
        d0, dgam, dbet = ops.group_norm_bwd(y0, dy, g, be, 4, relu=True)
        pwt, _ = ctx.net._packed(ctx.tag + ":T", ctx.holder, d0.shape[-1], transpose=True, with_bias=False)
        dxp = ops.conv2d(d0, pwt, 2, xp.shape[-1], out_hw=(xp.shape[1], xp.shape[2]))
        dx = ops.reflect_pad1_bwd(dxp) if ctx.needs_input_grad[0] else None
        dw = _wgrad_conv(xp, d0, w.shape, 0)
        db = _bias_grad(d0, w.shape[0])
        return dx, dw, db, dgam, dbet, (dy if ctx.has_res else None), None, None, None


class _ConvBN(torch.autograd.Function):
    """relu(BatchNorm_train(conv(x) [+ conv_b(x2)] + bias)).  reflect: 3x3 with reflect padding (ConvLayer norm='batch');
    otherwise a 1x1 convolution (skip_conv, the L-TAE MLP) or, with transpose, ConvTranspose2d(3, 1, 1) evaluated as a
    convolution with the transposed operand.  The second source is the other half of a channel concat (UpConvBlock's
    conv1 over cat([up, skip])): w's input-channel columns [0, c1) belong to x, the rest to x2."""

    @staticmethod
    def forward(ctx, x, x2, w, b, gamma, beta, net, tag, holder, bn, reflect, transpose, c1):
        def prep(t):
            return ops.reflect_pad1(t) if reflect else t
        n_out = w.shape[1] if transpose else w.shape[0]
        pitch = net._pitch(n_out)
        pad = 1 if transpose else 0
        xa = prep(x)
        if x2 is None:
            pw, bias = net._packed(tag, holder, x.shape[-1], transpose=transpose)
            y0 = ops.conv2d(xa, pw, pad, pitch, bias=bias)
            xb = None
        else:
            pa, bias = net._packed(tag + "a", holder, x.shape[-1], cols=slice(0, c1))
            pb, _ = net._packed(tag + "b", holder, x2.shape[-1], cols=slice(c1, None), with_bias=False)
            xb = prep(x2)
            t = ops.conv2d(xa, pa, pad, pitch, bias=bias)
            y0 = ops.conv2d(xb, pb, pad, pitch, residual=t)
        ctx.c1 = c1
        gam, bet = gamma.detach(), beta.detach()
        if pitch != n_out:
            # a class-score layer stored at the logits pitch: the statistics kernels work on the stored channels, so the
            # affine parameters and running buffers are padded for the call (pad channels: gamma 0 -> output 0)
            gam, bet = torch.zeros(pitch, device=x.device), torch.zeros(pitch, device=x.device)
            gam[:n_out], bet[:n_out] = gamma.detach(), beta.detach()
            rm, rv = torch.zeros(pitch, device=x.device), torch.ones(pitch, device=x.device)
            rm[:n_out], rv[:n_out] = bn.running_mean, bn.running_var
            scale, shift, mean, rstd = ops.bn_stats(y0, gam, bet, rm, rv, bn.momentum, bn.eps)
            bn.running_mean.copy_(rm[:n_out])
            bn.running_var.copy_(rv[:n_out])
        else:
            scale, shift, mean, rstd = ops.bn_stats(y0, gam, bet, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        bn.note_batch()
        y = ops.bn_apply(y0, scale, shift, relu=True)
        ctx.net, ctx.tag, ctx.holder, ctx.flags = net, tag, holder, (reflect, transpose, x2 is not None)
        ctx.n_out = n_out
        ctx.save_for_backward(xa, xb, y0, gam, bet, mean, rstd, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, xb, y0, gamma, beta, mean, rstd, w = ctx.saved_tensors
        reflect, transpose, two = ctx.flags
        net, tag, holder = ctx.net, ctx.tag, ctx.holder
        d0, _, dgam, dbet = ops.bn_bwd(y0, dy.contiguous(), None, gamma, beta, mean, rstd, True, False)
        n_out = ctx.n_out
        dgam, dbet = dgam[:n_out].clone(), dbet[:n_out].clone()
        db = _bias_grad(d0, n_out)
        k = w.shape[-1] if w.dim() == 4 else 1

        def unprep(dxp):
            return ops.reflect_pad1_bwd(dxp) if reflect else dxp

        if transpose:
            # y = C^T x with C the convolution whose OIHW weight is w [in, out, 3, 3]: dx = C dy, dW = wgrad(input dy, grad x)
            pwd, _ = net._packed(tag + ":D", holder, d0.shape[-1], with_bias=False)
            dx = ops.conv2d(d0, pwd, 1, xa.shape[-1]) if ctx.needs_input_grad[0] else None
            dw = ops.conv_wgrad(d0, xa, w.shape[0], w.shape[1], 3, 3, 1, 1)
            return dx, None, dw, db, dgam, dbet, None, None, None, None, None, None, None
        w4 = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
        dpad = k - 1  # full correlation of the unpadded convolution (pad 0)
        if not two:
            pwt, _ = net._packed(tag + ":T", holder, d0.shape[-1], transpose=True, with_bias=False)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = unprep(ops.conv2d(d0, pwt, dpad, xa.shape[-1], out_hw=(xa.shape[1], xa.shape[2])))
            dw = _wgrad_conv(xa, d0, w4, 0).view(w.shape)
            return dx, None, dw, db, dgam, dbet, None, None, None, None, None, None, None
        c1 = ctx.c1
        pat, _ = net._packed(tag + "a:T", holder, d0.shape[-1], transpose=True, with_bias=False, cols=slice(0, c1))
        pbt, _ = net._packed(tag + "b:T", holder, d0.shape[-1], transpose=True, with_bias=False, cols=slice(c1, None))
        dx = unprep(ops.conv2d(d0, pat, dpad, xa.shape[-1], out_hw=(xa.shape[1], xa.shape[2])))
        dx2 = unprep(ops.conv2d(d0, pbt, dpad, xb.shape[-1], out_hw=(xb.shape[1], xb.shape[2])))
        dwa = _wgrad_conv(xa, d0, (w4[0], c1, k, k), 0)
        dwb = _wgrad_conv(xb, d0, (w4[0], w4[1] - c1, k, k), 0)
        dw = torch.cat([dwa, dwb], dim=1).view(w.shape)
        return dx, dx2, dw, db, dgam, dbet, None, None, None, None, None, None, None


class _Conv1x1(torch.autograd.Function):
    """x W^T + b for a Conv1d(k=1) / nn.Linear weight on NHWC tokens (LTAE2d.inconv, MultiHeadAttention.fc1_k)"""

    @staticmethod
    def forward(ctx, x, w, b, net, tag, holder):
        pw, bias = net._packed(tag, holder, x.shape[-1])
        n_out = w.shape[0]
        y = ops.conv2d(x, pw, 0, ops.pad_channels(n_out), bias=bias)
        ctx.net, ctx.tag, ctx.holder = net, tag, holder
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        pwt, _ = ctx.net._packed(ctx.tag + ":T", ctx.holder, dy.shape[-1], transpose=True, with_bias=False)
        dx = ops.conv2d(dy, pwt, 0, x.shape[-1]) if ctx.needs_input_grad[0] else None
        dw = ops.conv_wgrad(x, dy, w.shape[0], w.shape[1], 1, 1, 1, 0).view(w.shape)
        return dx, dw, _bias_grad(dy, w.shape[0]), None, None, None


class _GNSeq(torch.autograd.Function):
    """GroupNorm over the dates of each pixel (LTAE2d.in_norm; out_norm with T = 1)"""

    @staticmethod
    def forward(ctx, x, gamma, beta, B, T, groups):
        g, be = gamma.detach().float(), beta.detach().float()
        ctx.geom = (B, T, groups)
        ctx.save_for_backward(x, g, be)
        return ops.group_norm_seq(x, B, T, g, be, groups)

    @staticmethod
    def backward(ctx, dy):
        x, g, be = ctx.saved_tensors
        B, T, groups = ctx.geom
        dx, dg, db = ops.group_norm_seq_bwd(x, dy, B, T, g, be, groups)
        return dx, dg, db, None, None, None


class _MaskImages(torch.autograd.Function):
    """TemporallySharedBlock.smart_forward: padded dates come out as pad_value and pass no gradient"""

    @staticmethod
    def forward(ctx, x, pad, value):
        ctx.save_for_backward(pad)
        return ops.mask_images_(x.clone(), pad, value)

    @staticmethod
    def backward(ctx, dy):
        (pad,) = ctx.saved_tensors
        return ops.mask_images_(dy.clone(), pad, 0.0), None, None


class _AddRows(torch.autograd.Function):
    """z + positional encoding (f32 [n, C], no gradient)"""

    @staticmethod
    def forward(ctx, z, pe):
        return ops.add_rowvec_(z.clone(), pe)

    @staticmethod
    def backward(ctx, dy):
        return dy, None


class _LtaeAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, keys, values, Q, pad, B, T, drop):
        q = Q.detach().float().contiguous()
        out, attn, prob = ops.ltae_attention_train(keys, values, q, pad, B, T, drop)
        ctx.geom = (B, T)
        ctx.save_for_backward(keys, values, q, pad, drop, prob)
        return out, attn

    @staticmethod
    def backward(ctx, dout, dattn):
        keys, values, q, pad, drop, prob = ctx.saved_tensors
        B, T = ctx.geom
        dattn = None if dattn is None else dattn.contiguous()
        dk, dv, dq = ops.ltae_attention_bwd(keys, values, q, pad, drop, prob, dout.contiguous(), dattn, B, T)
        return dk, dv, dq, None, None, None, None


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, attn, pad, B, T, use_pad):
        ctx.geom = (B, T, use_pad)
        ctx.save_for_backward(x, attn, pad)
        return ops.temporal_aggregate(x, attn, pad, B, T, use_pad)

    @staticmethod
    def backward(ctx, dy):
        x, attn, pad = ctx.saved_tensors
        B, T, use_pad = ctx.geom
        dx, dattn = ops.temporal_aggregate_bwd(x, attn, pad, dy.contiguous(), B, T, use_pad)
        return dx, dattn, None, None, None, None


class _Mul(torch.autograd.Function):
    """x * mask (nn.Dropout with a pre-scaled keep mask)"""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return ops.mul(x, mask)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return ops.mul(dy.contiguous(), mask), None


class _Add(torch.autograd.Function):
    """a + b through ffa_bn_apply's residual input ("out + conv2(out)" of UpConvBlock :598)"""

    @staticmethod
    def forward(ctx, a, b):
        one = torch.ones(a.shape[-1], dtype=torch.float32, device=a.device)
        return ops.bn_apply(a, one, torch.zeros_like(one), residual=b, relu=False)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


class HipUTAE(nn.Module):
    def __init__(self, input_dim: int, encoder_widths=(64, 64, 64, 128), decoder_widths=(32, 32, 64, 128),
                 out_conv=(32, 20), str_conv_k: int = 4, str_conv_s: int = 2, str_conv_p: int = 1,
                 agg_mode: str = "att_group", encoder_norm: str = "group", n_head: int = 16, d_model: int = 256,
                 d_k: int = 4, encoder: bool = False, return_maps: bool = False, pad_value=0,
                 padding_mode: str = "reflect", precision: str = "bf16"):
        super().__init__()
        if (str_conv_k, str_conv_s, str_conv_p) != (3, 1, 1) or agg_mode != "att_group" or encoder_norm != "group" \
                or padding_mode != "reflect":
            raise NotImplementedError(
                "HipUTAE covers the configuration FLAIR hard-codes (flair_zonal_detection/model_utils.py:55-71): strided "
                "convolutions k=3 s=1 p=1, agg_mode 'att_group', encoder_norm 'group', padding_mode 'reflect'; got "
                f"k={str_conv_k} s={str_conv_s} p={str_conv_p} agg={agg_mode} norm={encoder_norm} pad={padding_mode}")
        enc, dec = list(encoder_widths), list(decoder_widths if decoder_widths is not None else encoder_widths)
        if len(enc) != len(dec) or enc[-1] != dec[-1]:
            raise ValueError("encoder / decoder widths must have equal length and share the deepest width")
        if d_model % n_head or enc[-1] % n_head or any(c % 16 for c in enc + dec + [d_model, n_head * d_k]):
            raise NotImplementedError("channel widths must be multiples of 16 (and of the head count)")
        self.n_stages, self.enc, self.dec = len(enc), enc, dec
        self.input_dim, self.n_head, self.d_model, self.d_k = input_dim, n_head, d_model, d_k
        self.return_maps, self.encoder, self.pad_value = return_maps or encoder, encoder, float(pad_value)
        self.out_classes = list(out_conv)[-1]
        self.dtype = torch.bfloat16 if precision == "bf16" else torch.float32
        self.in_conv = _block(conv=_conv_layer([input_dim, enc[0], enc[0]], "group"))
        self.down_blocks = nn.ModuleList(
            _block(down=_conv_layer([enc[i], enc[i]], "group"), conv1=_conv_layer([enc[i], enc[i + 1]], "group"),
                   conv2=_conv_layer([enc[i + 1], enc[i + 1]], "group")) for i in range(self.n_stages - 1))
        ups = []
        for i in range(self.n_stages - 1, 0, -1):
            d_in, d_out, d_skip = dec[i], dec[i - 1], enc[i - 1]
            ups.append(_block(
                skip_conv=_seq(_Weights((d_skip, d_skip, 1, 1), d_skip), hnn.HipBatchNorm2d(d_skip), _Slot()),
                up=_seq(_Weights((d_in, d_out, 3, 3), d_out), hnn.HipBatchNorm2d(d_out), _Slot()),
                conv1=_conv_layer([d_out + d_skip, d_out], "batch"), conv2=_conv_layer([d_out, d_out], "batch")))
        self.up_blocks = nn.ModuleList(ups)
        te = nn.Module()
        te.inconv = _Weights((d_model, enc[-1], 1), d_model)
        te.attention_heads = nn.Module()
        te.attention_heads.Q = nn.Parameter(torch.randn(n_head, d_k) * (2.0 / d_k) ** 0.5)
        te.attention_heads.fc1_k = _Weights((n_head * d_k, d_model), n_head * d_k)
        te.in_norm, te.out_norm = _Affine(enc[-1]), _Affine(enc[-1])
        te.mlp = _seq(_Weights((enc[-1], d_model), enc[-1]), hnn.HipBatchNorm2d(enc[-1]), _Slot())
        self.temporal_encoder = te
        self.out_conv = _block(conv=_conv_layer([dec[0]] + list(out_conv), "batch"))
        self._cache = {}
        # training-time randomness of the reference: LTAE2d(dropout=0.2) after its MLP (:242,278),
        # ScaledDotProductAttention(attn_dropout=0.1) on the temporal attention masks (:387,399)
        self.mlp_dropout, self.attn_dropout = 0.2, 0.1

    # ---- operand preparation (cached per parameter version) ----------------------------------------

    def _packed(self, tag: str, holder: _Weights, ci_pitch: int, bn: Optional[hnn.HipBatchNorm2d] = None,
                transpose: bool = False, cols: Optional[slice] = None, with_bias: bool = True):
        """(packed operand, bias vector): conv weight as 4-D OIHW (Conv1d / Linear weights are 1x1 convolutions),
        evaluation-mode BatchNorm folded in: W' = scale * W, bias' = shift + scale * bias"""
        ver = (holder.weight._version, holder.weight.data_ptr(), holder.bias._version, hnn.state_epoch(),
               None if bn is None else (bn.weight._version, bn.bias._version, bn.running_mean._version,
                                        bn.running_var._version, getattr(bn, "_stats_epoch", 0)), self.dtype)
        hit = self._cache.get(tag)
        if hit is not None and hit[0] == ver:
            return hit[1], hit[2]
        w = holder.weight.detach().float()
        if w.dim() == 3:
            w = w.unsqueeze(-1)
        elif w.dim() == 2:
            w = w[:, :, None, None]
        if cols is not None:
            w = w[:, cols]
        w = w.contiguous()
        n_out = w.shape[1] if transpose else w.shape[0]
        scale = shift = None
        if bn is not None:
            scale, shift = ops.bn_eval_params(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                              bn.eps)
        pw = ops.pack_conv_weight(w, self.dtype, 1, ci_pitch, transpose=transpose, scale=scale, allow_ring=False,
                                  allow_thin=False)  # reflect-padded input, pad-0 convolution: conv_igemm only
        out_pitch = self._pitch(n_out)
        bias = torch.zeros(max(out_pitch, pw.rows), dtype=torch.float32, device=w.device)
        if with_bias:
            b = holder.bias.detach().float()
            bias[:n_out] = b if scale is None else shift[:n_out] + scale[:n_out] * b
        elif scale is not None:
            raise ValueError("a folded BatchNorm needs its shift in the bias vector")
        self._cache[tag] = (ver, pw, bias)
        return pw, bias

    def _pitch(self, n: int) -> int:
        """stored channel pitch of an n-channel tensor: class-score tensors of up to LOGIT_PITCH classes use the
        product's logits pitch (what the loss / argmax kernels expect), everything else the next multiple of 16"""
        return hnn.LOGIT_PITCH if (n == self.out_classes and n <= hnn.LOGIT_PITCH) else ops.pad_channels(n)

    def _conv_gn(self, x, seq, idx: int, tag: str, relu=True, residual=None):
        """reflect-pad conv3x3 + bias -> GroupNorm(4) -> ReLU (-> + residual): ConvLayer with norm='group'"""
        conv, gn = seq[idx], seq[idx + 1]
        pw, bias = self._packed(tag, conv, x.shape[-1])
        y = ops.conv2d(ops.reflect_pad1(x), pw, 0, ops.pad_channels(conv.weight.shape[0]), bias=bias)
        return ops.group_norm(y, gn.weight.detach().float(), gn.bias.detach().float(), 4, relu=relu, residual=residual)

    def _conv_bn(self, x, seq, idx: int, tag: str):
        """reflect-pad conv3x3 + bias -> BatchNorm (eval, folded) -> ReLU: ConvLayer with norm='batch'"""
        conv, bn = seq[idx], seq[idx + 1]
        pw, bias = self._packed(tag, conv, x.shape[-1], bn=bn)
        return ops.conv2d(ops.reflect_pad1(x), pw, 0, self._pitch(conv.weight.shape[0]), bias=bias, relu=True)

    # ---- forward -------------------------------------------------------------------------------------

    def forward_nhwc(self, x: torch.Tensor, batch_positions: torch.Tensor):
        """x f32 [B,T,C,H,W], batch_positions [B,T] -> (logits NHWC [B,H,W,pitch], maps NHWC (coarse to fine),
        attn f32 [n_head,B,T,H,W])"""
        if self.training and torch.is_grad_enabled():
            return self._forward_train(x, batch_positions)
        if self.training:
            raise NotImplementedError("HipUTAE in training mode runs under autograd only (BatchNorm batch statistics "
                                      "and dropout belong to the training step); call .eval() for inference")
        if not x.is_cuda:
            raise RuntimeError("HipUTAE runs on the MI355X only (no CPU path in the product)")
        B, T, C, H, W = x.shape
        if C != self.input_dim or H < 2 or W < 2:
            raise ValueError(f"expected [B,T,{self.input_dim},H>=2,W>=2], got {tuple(x.shape)}")
        N = B * T
        flat = x.reshape(N, C, H, W).float().contiguous()
        pad = ops.detect_pad_images(flat, self.pad_value)            # u8 [N], n = b * T + t
        # Temporal_Aggregator branches on pad_mask.any() (:609) only to skip a multiplication by (~pad_mask): applying the
        # mask always is the same arithmetic and needs no device -> host synchronisation (the step stays graph-capturable)
        any_pad = True
        cur = ops.nchw_to_nhwc(flat, self.dtype, ops.pad_channels(C))

        def shared(t):  # TemporallySharedBlock.smart_forward: padded dates come out as pad_value
            return ops.mask_images_(t, pad, self.pad_value)

        seq = self.in_conv.conv.conv
        cur = shared(self._conv_gn(self._conv_gn(cur, seq, 0, "in0"), seq, 3, "in1"))
        fmaps: List[torch.Tensor] = [cur]
        for i, blk in enumerate(self.down_blocks):
            t = self._conv_gn(fmaps[-1], blk.down.conv, 0, f"d{i}a")
            t = self._conv_gn(t, blk.conv1.conv, 0, f"d{i}b")
            t = self._conv_gn(t, blk.conv2.conv, 0, f"d{i}c", residual=t)  # out + conv2(out)
            fmaps.append(shared(t))

        # ---- L-TAE (:237-284) ----
        te = self.temporal_encoder
        z = ops.group_norm_seq(fmaps[-1], B, T, te.in_norm.weight.detach().float(), te.in_norm.bias.detach().float(),
                               self.n_head)
        pw, bias = self._packed("inconv", te.inconv, z.shape[-1])
        z = ops.conv2d(z, pw, 0, self.d_model, bias=bias)
        pe = ops.positional_encoding(batch_positions.to(x.device), self.d_model // self.n_head, self.n_head)
        ops.add_rowvec_(z, pe)
        pw, bias = self._packed("fc1_k", te.attention_heads.fc1_k, self.d_model)
        keys = ops.conv2d(z, pw, 0, self.n_head * self.d_k, bias=bias)
        o, attn = ops.ltae_attention(keys, z, te.attention_heads.Q.detach().float().contiguous(), pad, B, T)
        pw, bias = self._packed("mlp", te.mlp[0], self.d_model, bn=te.mlp[1])
        o = ops.conv2d(o, pw, 0, self.enc[-1], bias=bias, relu=True)
        out = ops.group_norm_seq(o, B, 1, te.out_norm.weight.detach().float(), te.out_norm.bias.detach().float(),
                                 self.n_head)

        # ---- decoder (:150-157, UpConvBlock :568-599) ----
        maps = [out]
        for i, blk in enumerate(self.up_blocks):
            skip_src = fmaps[-(i + 2)]
            if skip_src.shape[1:3] != fmaps[-1].shape[1:3]:
                raise NotImplementedError("attention masks are used at their own resolution (stride-1 U-TAE)")
            skip = ops.temporal_aggregate(skip_src, attn, pad, B, T, any_pad)
            pw, bias = self._packed(f"u{i}up", blk.up[0], out.shape[-1], bn=blk.up[1], transpose=True)
            d_out = blk.up[0].weight.shape[1]
            up = ops.conv2d(out, pw, 1, ops.pad_channels(d_out), bias=bias, relu=True)  # ConvTranspose2d(3,1,1)
            pw, bias = self._packed(f"u{i}skip", blk.skip_conv[0], skip.shape[-1], bn=blk.skip_conv[1])
            sk = ops.conv2d(skip, pw, 0, skip.shape[-1], bias=bias, relu=True)
            # conv1 over cat([up, sk]) without the concat: one convolution per source, chained through the residual
            c1, bn1 = blk.conv1.conv[0], blk.conv1.conv[1]
            pa, bias = self._packed(f"u{i}c1a", c1, up.shape[-1], bn=bn1, cols=slice(0, d_out))
            pb, _ = self._packed(f"u{i}c1b", c1, sk.shape[-1], bn=bn1, cols=slice(d_out, None))
            t = ops.conv2d(ops.reflect_pad1(up), pa, 0, ops.pad_channels(d_out), bias=bias)
            y = ops.conv2d(ops.reflect_pad1(sk), pb, 0, ops.pad_channels(d_out), residual=t, relu=True)
            y2 = self._conv_bn(y, blk.conv2.conv, 0, f"u{i}c2")
            one = torch.ones(y.shape[-1], dtype=torch.float32, device=y.device)
            out = ops.bn_apply(y2, one, torch.zeros_like(one), residual=y, relu=False)  # out + conv2(out)
            maps.append(out)
        seq = self.out_conv.conv.conv  # ConvBlock([dec[0]] + out_conv): conv -> BN -> ReLU per entry of out_conv
        logits = out
        for j in range(len(seq) // 3):
            logits = self._conv_bn(logits, seq, 3 * j, f"o{j}")
        return logits, maps, attn.view(self.n_head, B, T, H, W)

    def _forward_train(self, x: torch.Tensor, batch_positions: torch.Tensor):
        """training-mode forward_nhwc through autograd nodes (same return convention)"""
        if not x.is_cuda:
            raise RuntimeError("HipUTAE runs on the MI355X only (no CPU path in the product)")
        B, T, C, H, W = x.shape
        if C != self.input_dim or H < 2 or W < 2:
            raise ValueError(f"expected [B,T,{self.input_dim},H>=2,W>=2], got {tuple(x.shape)}")
        N = B * T
        flat = x.reshape(N, C, H, W).float().contiguous()
        pad = ops.detect_pad_images(flat, self.pad_value)
        any_pad = True  # see forward_nhwc: the mask is applied unconditionally (no host synchronisation)
        cur = ops.nchw_to_nhwc(flat, self.dtype, ops.pad_channels(C))

        def gn(t, seq, idx, tag, residual=None):
            conv, nrm = seq[idx], seq[idx + 1]
            return _ConvGN.apply(t, conv.weight, conv.bias, nrm.weight, nrm.bias, residual, self, tag, conv)

        def bnc(t, conv, bn, tag, reflect=True, transpose=False, t2=None, c1=0):
            return _ConvBN.apply(t, t2, conv.weight, conv.bias, bn.weight, bn.bias, self, tag, conv, bn, reflect,
                                 transpose, c1)

        def shared(t):
            return _MaskImages.apply(t, pad, self.pad_value)

        seq = self.in_conv.conv.conv
        cur = shared(gn(gn(cur, seq, 0, "in0"), seq, 3, "in1"))
        fmaps: List[torch.Tensor] = [cur]
        for i, blk in enumerate(self.down_blocks):
            t = gn(fmaps[-1], blk.down.conv, 0, f"d{i}a")
            t = gn(t, blk.conv1.conv, 0, f"d{i}b")
            t = gn(t, blk.conv2.conv, 0, f"d{i}c", residual=t)
            fmaps.append(shared(t))

        te = self.temporal_encoder
        z = _GNSeq.apply(fmaps[-1], te.in_norm.weight, te.in_norm.bias, B, T, self.n_head)
        z = _Conv1x1.apply(z, te.inconv.weight, te.inconv.bias, self, "inconv", te.inconv)
        pe = ops.positional_encoding(batch_positions.to(x.device), self.d_model // self.n_head, self.n_head)
        z = _AddRows.apply(z, pe)
        fk = te.attention_heads.fc1_k
        keys = _Conv1x1.apply(z, fk.weight, fk.bias, self, "fc1_k", fk)
        drop = None
        if self.attn_dropout > 0:
            keep = 1.0 - self.attn_dropout
            drop = torch.empty((self.n_head, B, T, H * W), dtype=torch.float32, device=x.device).bernoulli_(keep).div_(keep)
        o, attn = _LtaeAttention.apply(keys, z, te.attention_heads.Q, pad, B, T, drop)
        o = bnc(o, te.mlp[0], te.mlp[1], "mlp", reflect=False)
        if self.mlp_dropout > 0:
            keep = 1.0 - self.mlp_dropout
            o = _Mul.apply(o, torch.empty_like(o).bernoulli_(keep).div_(keep))
        out = _GNSeq.apply(o, te.out_norm.weight, te.out_norm.bias, B, 1, self.n_head)

        maps = [out]
        for i, blk in enumerate(self.up_blocks):
            skip_src = fmaps[-(i + 2)]
            if skip_src.shape[1:3] != fmaps[-1].shape[1:3]:
                raise NotImplementedError("attention masks are used at their own resolution (stride-1 U-TAE)")
            skip = _Aggregate.apply(skip_src, attn, pad, B, T, any_pad)
            d_out = blk.up[0].weight.shape[1]
            up = bnc(out, blk.up[0], blk.up[1], f"u{i}up", reflect=False, transpose=True)
            sk = bnc(skip, blk.skip_conv[0], blk.skip_conv[1], f"u{i}skip", reflect=False)
            y = bnc(up, blk.conv1.conv[0], blk.conv1.conv[1], f"u{i}c1", t2=sk, c1=d_out)
            y2 = bnc(y, blk.conv2.conv[0], blk.conv2.conv[1], f"u{i}c2")
            out = _Add.apply(y2, y)
            maps.append(out)
        seq = self.out_conv.conv.conv
        logits = out
        for j in range(len(seq) // 3):
            logits = bnc(logits, seq[3 * j], seq[3 * j + 1], f"o{j}")
        return logits, maps, attn.view(self.n_head, B, T, H, W)

    def forward(self, input: torch.Tensor, batch_positions: Optional[torch.Tensor] = None, return_att: bool = False):
        """the reference's return convention (:158-166), NCHW float32 tensors"""
        logits, maps, attn = self.forward_nhwc(input, batch_positions)
        to_nchw = lambda t, c: ops.nhwc_to_nchw(t, c)
        widths = [self.dec[-1]] + [self.dec[i - 1] for i in range(self.n_stages - 1, 0, -1)]
        maps_nchw = [to_nchw(m, c) for m, c in zip(maps, widths)]
        if self.encoder:
            return maps_nchw[-1], maps_nchw
        out = to_nchw(logits, self.out_classes)
        if return_att:
            return out, attn
        if self.return_maps:
            return out, maps_nchw
        return out
