"""nn.Module / autograd layer over flairhip.ops.

Modules keep torch-compatible parameters (conv weights OIHW f32, BatchNorm weight / bias /
running stats) under the names segmentation_models_pytorch 0.4.0 emits, so reference checkpoints
load unchanged (flair_hub/models/checkpoint.py:225-228 hard-codes some of those names).  The
forward / backward arithmetic is entirely libflairhip kernels on NHWC tensors; torch supplies
autograd bookkeeping, parameters and the optimizer.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops

LOGIT_PITCH = 32  # channel pitch of logits tensors (>= any class count of the reference's tasks)


def _as_nhwc_grad(g: torch.Tensor) -> torch.Tensor:
    return g if g.is_contiguous() else g.contiguous()


def _dw_out(conv) -> Optional[torch.Tensor]:
    """Where the weight gradient of ``conv`` should be written: its slot in the data-parallel gradient bucket
    (flairhip.distributed.GradSync) when there is one and no gradient has been accumulated yet -- autograd then adopts
    the returned tensor as ``weight.grad`` and the all-reduce reads it in place -- else None (a fresh tensor)."""
    w = conv.weight
    buf = getattr(w, "_ffa_grad_buf", None)
    if buf is None or w.grad is not None or buf.shape != w.shape:
        return None
    # one taker per step: a convolution applied twice in one forward would otherwise get the slot for both of its
    # weight gradients (the second kernel overwrites the first, autograd then sums two aliases of the last one).  The
    # state epoch moves with every optimizer step / graph replay; a second use inside it falls back to a fresh tensor,
    # which autograd accumulates onto the slot.
    if getattr(w, "_ffa_buf_taken", None) == _STATE_EPOCH:
        return None
    w._ffa_buf_taken = _STATE_EPOCH
    return buf.view(buf.shape)  # a fresh alias: autograd only adopts a gradient tensor nobody else holds


def _vec_out(p: torch.Tensor, n: int) -> Optional[torch.Tensor]:
    """_dw_out for a per-channel parameter (BatchNorm weight / bias): its slot in the data-parallel gradient bucket, or
    None.  Without it every such gradient is a separate tensor that GradSync copies into its slot -- 92 small
    device-to-device copies per step for the U-Net (0.24 ms of 2.5-us copy kernels, found in the one-rank RCCL rehearsal)."""
    buf = getattr(p, "_ffa_grad_buf", None)
    if buf is None or p.grad is not None or buf.numel() != n or buf.dtype != torch.float32 or buf.dim() != 1:
        return None
    if getattr(p, "_ffa_buf_taken", None) == _STATE_EPOCH:
        return None
    p._ffa_buf_taken = _STATE_EPOCH
    return buf.view(buf.shape)


def _bn_bwd(x, dy, y, gamma, beta, mean, rstd, relu, want_dres):
    """ops.bn_bwd with the affine gradients written straight into their bucket slots when there are any"""
    n = x.shape[-1]
    og, ob = _vec_out(gamma, n), _vec_out(beta, n)
    if og is None or ob is None:
        og = ob = None
    return ops.bn_bwd(x, dy, y, gamma, beta, mean, rstd, relu, want_dres, out_dgamma=og, out_dbeta=ob)


class HipConv2d(nn.Module):
    """Parameter holder mirroring nn.Conv2d (weight OIHW f32, optional bias)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 bias: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        self._cache = {}

    @property
    def in_pitch(self) -> int:
        return ops.pad_channels(self.in_channels)

    @property
    def out_pitch(self) -> int:
        return LOGIT_PITCH if self.bias is not None and self.out_channels <= LOGIT_PITCH else ops.pad_channels(
            self.out_channels)

    def packed(self, dtype: torch.dtype, transpose: bool = False, scale: Optional[torch.Tensor] = None,
               tag: str = "", ring: bool = True, thin: bool = True) -> ops.PackedWeight:
        """MFMA operand for the current weight values; re-packed only when the parameter changed.  ring=False keeps
        the conv_igemm layout (operands of the two-source / split-epilogue / bn-backward-epilogue kernels); thin=False
        keeps a <= 32-channel layer off conv3x3_thin_kernel (it has no BatchNorm-backward-partials epilogue)."""
        ring = ring and self.padding == 1  # the ring kernel is a pad-1 kernel
        thin = thin and self.padding == 1
        if not tag:
            tag = "plain" if not thin else ("" if ring else "igemm")
        key = (dtype, transpose, tag)
        ver = (self.weight._version, self.weight.data_ptr(), None if scale is None else scale._version, _STATE_EPOCH)
        hit = self._cache.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        w = self.weight.detach()
        if not w.is_contiguous():
            w = w.contiguous()
        pitch = self.out_pitch if transpose else self.in_pitch
        pw = ops.pack_conv_weight(w, dtype, self.stride, pitch, transpose=transpose, scale=scale, allow_ring=ring,
                                  allow_thin=thin, allow_stem=(self.padding == 3))
        self._cache[key] = (ver, pw)
        return pw

    def extra_repr(self) -> str:
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, bias={self.bias is not None}")


_STATE_EPOCH = 0


def state_epoch() -> int:
    return _STATE_EPOCH


def bump_state_epoch() -> None:
    """Parameters / buffers changed on the device without Python noticing (a hipGraph replay runs the optimizer and
    the BatchNorm running-statistics updates in place: no tensor ``_version`` moves).  Every cache keyed on versions
    -- packed MFMA operands, eval-mode BatchNorm folds -- carries this epoch as well and is rebuilt on next use."""
    global _STATE_EPOCH
    _STATE_EPOCH += 1


def _after_optimizer_step(optimizer, args, kwargs) -> None:
    bump_state_epoch()


# torch's fused (single-kernel) Adam / AdamW / SGD update parameters WITHOUT moving their ``_version`` (checked on
# torch 2.10: fused=True leaves p._version unchanged, the foreach / single-tensor paths bump it), so a cache keyed on
# versions alone would keep feeding the convolutions the weights of the first step.  Every optimizer step in the
# process therefore advances the state epoch.
from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook  # noqa: E402

_register_step_hook(_after_optimizer_step)


class PackPlan:
    """Keeps the MFMA operands of all conv layers of a module tree current with ONE kernel launch per optimizer
    step (93 separate pack launches cost 0.5 ms per step at batch 32).  Layers enter the plan once they have been
    packed individually (first forward / backward), so unused layers never do."""

    def __init__(self, root: nn.Module):
        self.convs = [m for m in root.modules() if isinstance(m, HipConv2d)]
        self.batch = None
        self.sig = None       # identity of the buffers the table points at
        self.versions = None  # parameter versions the packed operands correspond to

    def refresh(self, dtype: torch.dtype) -> None:
        entries, sig, slots = [], [], []
        for c in self.convs:
            for transpose in (False, True):
                for tag in ("", "igemm", "plain"):
                    hit = c._cache.get((dtype, transpose, tag))
                    if hit is None:
                        continue
                    entries.append((c.weight.detach(), hit[1], transpose))
                    sig.append((c.weight.data_ptr(), hit[1].data.data_ptr()))
                    slots.append((c, (dtype, transpose, tag)))
            # column blocks of fusion 1x1 weights (FusionHandler.conv_f: one operand per modality and direction)
            for key, hit in c._cache.items():
                if isinstance(key, tuple) and key and key[0] == "fusion_slice" and key[3] == dtype:
                    pw = hit[1]
                    if pw.bco & (ops._l.BCO_RING | ops._l.BCO_THIN):
                        continue
                    entries.append((c.weight.detach(), pw, key[5], (key[1], key[2])))
                    sig.append((c.weight.data_ptr(), pw.data.data_ptr()))
                    slots.append((c, key))
        if not entries:
            return
        versions = [c.weight._version for c, _ in slots]
        if self.batch is None or sig != self.sig:
            if not all(e[0].is_contiguous() for e in entries):
                return
            self.batch, self.sig, self.versions = ops.PackBatch(entries, dtype), sig, None
        # cache stamps: (version, data_ptr, scale version, epoch) for whole weights, (version, data_ptr, epoch) for blocks
        stale = [i for i, (c, key) in enumerate(slots)
                 if c._cache[key][0][0] != versions[i] or c._cache[key][0][-1] != _STATE_EPOCH]
        if not stale:
            return
        self.batch.run()
        for (c, key), v in zip(slots, versions):
            stamp = (v, c.weight.data_ptr(), _STATE_EPOCH) if key[0] == "fusion_slice" else (v, c.weight.data_ptr(), None, _STATE_EPOCH)
            c._cache[key] = (stamp, c._cache[key][1])


class HipBatchNorm2d(nn.Module):
    """Parameter / buffer holder mirroring nn.BatchNorm2d (eps 1e-5, momentum 0.1)."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self._pending_batches = 0
        self._register_state_dict_hook(HipBatchNorm2d._flush_hook)
        self.register_load_state_dict_post_hook(HipBatchNorm2d._loaded_hook)

    def note_batch(self) -> None:
        self._pending_batches += 1  # folded into the device counter lazily (no per-step launch)
        # the running statistics were just rewritten through raw pointers (no tensor version moved): eval-mode folds
        # of THIS layer computed earlier are stale even if no optimizer step follows (statistics recalibration passes)
        self._stats_epoch = getattr(self, "_stats_epoch", 0) + 1

    @staticmethod
    def _loaded_hook(module, incompatible_keys):
        # the loaded num_batches_tracked is the truth: batches counted before the load must not be added on top
        module._pending_batches = 0
        module._stats_epoch = getattr(module, "_stats_epoch", 0) + 1  # eval-mode folds of the old statistics are stale

    @staticmethod
    def _flush_hook(module, state_dict, prefix, local_metadata):
        if module._pending_batches:
            module.num_batches_tracked += module._pending_batches
            module._pending_batches = 0
            state_dict[prefix + "num_batches_tracked"] = module.num_batches_tracked

    def extra_repr(self) -> str:
        return f"{self.num_features}, eps={self.eps}, momentum={self.momentum}"


# --------------------------------------------------------------------------------------------------
# autograd functions (all tensors NHWC)

class _ConvBnAct(torch.autograd.Function):
    """y = relu?( batch_norm_train(conv(x)) (+ residual) )"""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, conv: HipConv2d, bn: HipBatchNorm2d, relu: bool):
        pw = conv.packed(x.dtype)
        y0, scale, shift, mean, rstd = ops.conv2d_bn_stats(x, pw, conv.padding, conv.out_pitch, gamma, beta,
                                                          bn.running_mean, bn.running_var, bn.momentum, bn.eps)
        bn.note_batch()
        y = ops.bn_apply(y0, scale, shift, residual=residual, relu=relu)
        ctx.conv, ctx.relu, ctx.has_res = conv, relu, residual is not None
        # the forward output is kept for the ReLU mask only when a residual was added; otherwise the mask is
        # recomputed from the pre-norm tensor (one fewer tensor read per backward pass, one fewer kept alive)
        ctx.save_for_backward(x, y0, y if (relu and residual is not None) else None, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y0, y, gamma, beta, mean, rstd = ctx.saved_tensors
        conv = ctx.conv
        dy = _as_nhwc_grad(dy)
        want_res = ctx.has_res and ctx.needs_input_grad[4]
        d0, dres, dgamma, dbeta = _bn_bwd(y0, dy, y, gamma, beta, mean, rstd, ctx.relu, want_res)
        dx = None
        if ctx.needs_input_grad[0]:
            pwt = conv.packed(x.dtype, transpose=True)
            k = conv.kernel_size
            dx = ops.conv2d(d0, pwt, k - 1 - conv.padding, x.shape[-1], dil=conv.stride,
                            out_hw=(x.shape[1], x.shape[2]))
        dw = None
        if ctx.needs_input_grad[1]:
            dw = ops.conv_wgrad(x, d0, conv.out_channels, conv.in_channels, conv.kernel_size, conv.kernel_size,
                                conv.stride, conv.padding, out=_dw_out(conv))
        return dx, dw, dgamma, dbeta, dres, None, None, None


class _UpConvBnAct(torch.autograd.Function):
    """y = relu(bn_train(conv3x3(cat(nearest_x2(lo), skip)))) -- the first half of a U-Net DecoderBlock -- with the
    upsampled + concatenated tensor never materialised: forward conv and wgrad read lo (at y/2, x/2) and skip
    directly.  Only the input gradient still passes through a full-resolution dcat (split by up2_concat_bwd)."""

    @staticmethod
    def forward(ctx, lo, skip, weight, gamma, beta, conv: HipConv2d, bn: HipBatchNorm2d):
        y0, scale, shift, mean, rstd = ops.conv2d_upcat_bn_stats(lo, skip, conv.packed(lo.dtype, ring=False), conv.out_pitch, gamma,
                                                                 beta, bn.running_mean, bn.running_var, bn.momentum,
                                                                 bn.eps)
        bn.note_batch()
        y = ops.bn_apply(y0, scale, shift, relu=True)
        ctx.conv = conv
        ctx.save_for_backward(lo, skip, y0, gamma, beta, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        lo, skip, y0, gamma, beta, mean, rstd = ctx.saved_tensors
        conv = ctx.conv
        d0, _, dgamma, dbeta = _bn_bwd(y0, _as_nhwc_grad(dy), None, gamma, beta, mean, rstd, True, False)
        dlo = dskip = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            pwt = conv.packed(d0.dtype, transpose=True, ring=False)
            c1, c2 = lo.shape[-1], (0 if skip is None else skip.shape[-1])
            pair = ops.conv2d_dgrad_upcat(d0, pwt, c1, c2)  # dlo pooled in the dgrad epilogue, dskip written directly
            if pair is None:
                dcat = ops.conv2d(d0, pwt, 1, conv.in_pitch)
                pair = ops.upsample2x_concat_bwd(dcat, c1, skip_as_view=True)
            dlo, dskip = pair
        dw = None
        if ctx.needs_input_grad[2]:
            dw = ops.conv_wgrad_upcat(lo, skip, d0, conv.out_channels, out=_dw_out(conv))
            if dw is None:
                raise RuntimeError("conv_wgrad_upcat refused a channel split that upcat_supported accepted")
        return dlo, dskip, dw, dgamma, dbeta, None, None


class _DecoderBlock(torch.autograd.Function):
    """A whole U-Net DecoderBlock in training mode as one autograd node:
        y = relu(bn_b(conv_b( relu(bn_a(conv_a( cat(nearest_x2(lo), skip) ))) )))
    conv_a reads lo / skip directly (two-source kernels), and bn_a's backward reductions come out of conv_b's dgrad
    epilogue -- which needs both layers in one node, the gradient tensor between them carries no side data."""

    @staticmethod
    def forward(ctx, lo, skip, wa, ga, ba, wb, gb, bb, blk):
        ca, na, cb, nb = blk.conv1[0], blk.conv1[1], blk.conv2[0], blk.conv2[1]
        xa, sca, sha, ma, ra = ops.conv2d_upcat_bn_stats(lo, skip, ca.packed(lo.dtype, ring=False), ca.out_pitch, ga, ba,
                                                         na.running_mean, na.running_var, na.momentum, na.eps)
        na.note_batch()
        pwb = cb.packed(lo.dtype)
        # normalise-on-load: conv_b (and its weight gradient) read relu(xa * sca + sha) straight from xa
        ctx.fold = ops.NORM_ON_LOAD and ops.pro_supported(pwb, lo.dtype) and not ops.FUSED_BN_BWD
        if ctx.fold:
            ya = None
            xb, scb, shb, mb, rb = ops.conv2d_pro_bn_stats(xa, pwb, cb.out_pitch, sca, sha, gb, bb, nb.running_mean,
                                                           nb.running_var, nb.momentum, nb.eps)
        else:
            ya = ops.bn_apply(xa, sca, sha, relu=True)
            xb, scb, shb, mb, rb = ops.conv2d_bn_stats(ya, pwb, cb.padding, cb.out_pitch, gb, bb,
                                                       nb.running_mean, nb.running_var, nb.momentum, nb.eps)
        nb.note_batch()
        y = ops.bn_apply(xb, scb, shb, relu=True)
        ctx.blk = blk
        ctx.save_for_backward(lo, skip, xa, ya, xb, ga, ba, ma, ra, sca, sha, gb, bb, mb, rb)
        return y

    @staticmethod
    def backward(ctx, dy):
        lo, skip, xa, ya, xb, ga, ba, ma, ra, sca, sha, gb, bb, mb, rb = ctx.saved_tensors
        blk = ctx.blk
        ca, cb = blk.conv1[0], blk.conv2[0]
        db_, _, dgb, dbb = _bn_bwd(xb, _as_nhwc_grad(dy), None, gb, bb, mb, rb, True, False)
        dwb = None
        if ctx.needs_input_grad[5]:
            if ctx.fold:
                dwb = ops.conv_wgrad_pro(xa, db_, cb.out_channels, cb.in_channels, sca, sha, out=_dw_out(cb))
            else:
                dwb = ops.conv_wgrad(ya, db_, cb.out_channels, cb.in_channels, 3, 3, 1, 1, out=_dw_out(cb))
        pbt = cb.packed(db_.dtype, transpose=True, ring=not ops.FUSED_BN_BWD, thin=not ops.FUSED_BN_BWD)
        if ops.FUSED_BN_BWD:
            dya, part, rows = ops.conv2d_bnbwd(db_, pbt, 1, xa.shape[-1], xa, sca, sha)
            da, dga, dba = ops.bn_bwd_partials(xa, dya, part, rows, ga, ba, ma, ra)
        else:
            dya = ops.conv2d(db_, pbt, 1, xa.shape[-1])
            da, _, dga, dba = _bn_bwd(xa, dya, None, ga, ba, ma, ra, True, False)
        dlo = dskip = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            pat = ca.packed(da.dtype, transpose=True, ring=False)
            c1, c2 = lo.shape[-1], (0 if skip is None else skip.shape[-1])
            pair = ops.conv2d_dgrad_upcat(da, pat, c1, c2)
            if pair is None:
                pair = ops.upsample2x_concat_bwd(ops.conv2d(da, pat, 1, ca.in_pitch), c1, skip_as_view=True)
            dlo, dskip = pair
        dwa = None
        if ctx.needs_input_grad[2]:
            dwa = ops.conv_wgrad_upcat(lo, skip, da, ca.out_channels, out=_dw_out(ca))
            if dwa is None:
                raise RuntimeError("conv_wgrad_upcat refused a channel split that upcat_supported accepted")
        return dlo, dskip, dwa, dga, dba, dwb, dgb, dbb, None


class _BasicBlock(torch.autograd.Function):
    """One ResNet BasicBlock in training mode as a single autograd node:
        y = relu( bn2(conv2( relu(bn1(conv1(x))) )) + identity ),   identity = x  or  bn_d(conv_d(x)).
    Same kernels as the layer-wise path; what the fusion buys is the backward of the fork at x: the gradient of
    the identity branch enters the last dgrad convolution through its epilogue's residual input instead of being
    summed by a separate elementwise pass over the block's input (16 such passes per step in ResNet-34)."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2, wd, gd, bd, blk, fork=False):
        c1, n1, c2, n2 = blk.conv1, blk.bn1, blk.conv2, blk.bn2
        down = blk.downsample is not None
        ctx.fork = fork
        x1, sc1, sh1, m1, r1 = ops.conv2d_bn_stats(x, c1.packed(x.dtype), c1.padding, c1.out_pitch, g1, b1,
                                                   n1.running_mean, n1.running_var, n1.momentum, n1.eps)
        n1.note_batch()
        pw2 = c2.packed(x.dtype)
        # normalise-on-load: conv2 (and its weight gradient) read relu(x1 * sc1 + sh1) straight from x1
        ctx.fold = ops.NORM_ON_LOAD and ops.pro_supported(pw2, x.dtype) and not ops.FUSED_BN_BWD and c2.padding == 1
        y1 = None if ctx.fold else ops.bn_apply(x1, sc1, sh1, relu=True)
        if down:
            cd, nd = blk.downsample[0], blk.downsample[1]
            xd, scd, shd, md, rd = ops.conv2d_bn_stats(x, cd.packed(x.dtype), cd.padding, cd.out_pitch, gd, bd,
                                                       nd.running_mean, nd.running_var, nd.momentum, nd.eps)
            nd.note_batch()
            idt = ops.bn_apply(xd, scd, shd, relu=False)
        else:
            xd = md = rd = None
            idt = x
        if ctx.fold:
            x2, sc2, sh2, m2, r2 = ops.conv2d_pro_bn_stats(x1, pw2, c2.out_pitch, sc1, sh1, g2, b2, n2.running_mean,
                                                           n2.running_var, n2.momentum, n2.eps)
        else:
            x2, sc2, sh2, m2, r2 = ops.conv2d_bn_stats(y1, pw2, c2.padding, c2.out_pitch, g2, b2,
                                                       n2.running_mean, n2.running_var, n2.momentum, n2.eps)
        n2.note_batch()
        y = ops.bn_apply(x2, sc2, sh2, residual=idt, relu=True)
        ctx.blk = blk
        ctx.save_for_backward(x, x1, y1, x2, y, xd, g1, b1, m1, r1, g2, b2, m2, r2, gd, bd, md, rd, sc1, sh1)
        # fork: the block's input also feeds the U-Net decoder; handing the alias out of THIS node lets its backward
        # take the decoder's gradient of x as the residual input of a dgrad conv instead of an elementwise sum
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dx_skip=None):
        x, x1, y1, x2, y, xd, g1, b1, m1, r1, g2, b2, m2, r2, gd, bd, md, rd, sc1, sh1 = ctx.saved_tensors
        blk = ctx.blk
        if dx_skip is not None:
            dx_skip = _as_nhwc_grad(dx_skip)
        c1, c2 = blk.conv1, blk.conv2
        dy = _as_nhwc_grad(dy)

        def dgrad(conv, d, like, residual=None):
            k = conv.kernel_size
            return ops.conv2d(d, conv.packed(d.dtype, transpose=True), k - 1 - conv.padding, like.shape[-1],
                              dil=conv.stride, out_hw=(like.shape[1], like.shape[2]), residual=residual)

        def wgrad(conv, inp, d):
            return ops.conv_wgrad(inp, d, conv.out_channels, conv.in_channels, conv.kernel_size, conv.kernel_size,
                                  conv.stride, conv.padding, out=_dw_out(conv))

        d2, dres, dg2, db2 = _bn_bwd(x2, dy, y, g2, b2, m2, r2, True, True)
        dw2 = None
        if ctx.needs_input_grad[4]:
            dw2 = (ops.conv_wgrad_pro(x1, d2, c2.out_channels, c2.in_channels, sc1, sh1, out=_dw_out(c2)) if ctx.fold
                   else wgrad(c2, y1, d2))
        if ops.FUSED_BN_BWD:
            # bn1's backward reductions come out of conv2's dgrad epilogue (one pass over x1 and dy1 fewer)
            dy1, part, rows = ops.conv2d_bnbwd(d2, c2.packed(d2.dtype, transpose=True, ring=False, thin=False), c2.kernel_size - 1 - c2.padding,
                                               x1.shape[-1], x1, sc1, sh1)
            d1, dg1, db1 = ops.bn_bwd_partials(x1, dy1, part, rows, g1, b1, m1, r1)
        else:
            dy1 = dgrad(c2, d2, x1)
            d1, _, dg1, db1 = _bn_bwd(x1, dy1, None, g1, b1, m1, r1, True, False)
        dw1 = wgrad(c1, x, d1) if ctx.needs_input_grad[1] else None
        dwd = dgd = dbd = None
        if blk.downsample is not None:
            cd = blk.downsample[0]
            dd, _, dgd, dbd = _bn_bwd(xd, dres, None, gd, bd, md, rd, False, False)
            dwd = wgrad(cd, x, dd) if ctx.needs_input_grad[7] else None
            dres = dgrad(cd, dd, x, residual=dx_skip) if ctx.needs_input_grad[0] else None  # second residual slot
        elif dx_skip is not None:
            dres = dres + dx_skip  # identity block used as a fork: no free epilogue slot (not the U-Net's case)
        dx = dgrad(c1, d1, x, residual=dres) if ctx.needs_input_grad[0] else None
        return dx, dw1, dg1, db1, dw2, dg2, db2, dwd, dgd, dbd, None, None


class _ConvBias(torch.autograd.Function):
    """y = conv(x) + bias  (segmentation head; output pitch LOGIT_PITCH, pad channels zero)"""

    @staticmethod
    def forward(ctx, x, weight, bias, conv: HipConv2d):
        pw = conv.packed(x.dtype)
        bpad = None
        if bias is not None:
            bpad = torch.zeros(conv.out_pitch, dtype=torch.float32, device=x.device)
            bpad[: conv.out_channels] = bias.detach()
        y = ops.conv2d(x, pw, conv.padding, conv.out_pitch, bias=bpad)
        ctx.conv, ctx.has_bias = conv, bias is not None
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        conv = ctx.conv
        dy = _as_nhwc_grad(dy)
        dx = None
        if ctx.needs_input_grad[0]:
            pwt = conv.packed(x.dtype, transpose=True)
            k = conv.kernel_size
            dx = ops.conv2d(dy, pwt, k - 1 - conv.padding, x.shape[-1], dil=conv.stride,
                            out_hw=(x.shape[1], x.shape[2]))
        dw = ops.conv_wgrad(x, dy, conv.out_channels, conv.in_channels, conv.kernel_size, conv.kernel_size,
                            conv.stride, conv.padding, out=_dw_out(conv)) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            s = _take_dlogit_sums(dy)  # the loss kernel summed its own gradient per class: no pass over dy
            if s is None:
                s, _ = ops.channel_sums(dy)
            db = s[: conv.out_channels].clone()
        return dx, dw, db, None


def _fusion_slice(conv: HipConv2d, w: torch.Tensor, off: int, c: int, dtype: torch.dtype, pitch: int,
                  transpose: bool) -> ops.PackedWeight:
    """MFMA operand of the column block W[:, off:off+c] of a fusion 1x1 weight, cached until the weight changes
    (an inference loop packs each block once instead of once per batch)"""
    key = ("fusion_slice", off, c, dtype, pitch, transpose)
    ver = (conv.weight._version, conv.weight.data_ptr(), _STATE_EPOCH)
    hit = conv._cache.get(key)
    if hit is None or hit[0] != ver:
        hit = (ver, ops.pack_conv_weight(w[:, off:off + c].contiguous(), dtype, 1, pitch, transpose=transpose))
        conv._cache[key] = hit
    return hit[1]


class _FusionConv1x1(torch.autograd.Function):
    """y = bias + sum_m W[:, slice_m] x_m -- the 1x1 convolution over the channel concat of several sources
    (FusionHandler.conv_f, flair_hub/models/flair_model.py:470-475,533-541) evaluated WITHOUT the concat: one
    1x1 conv per source, chained through the conv epilogue's residual input.  Saves a write + a read of the
    concatenated map per stage and needs no concat kernel in either direction."""

    @staticmethod
    def forward(ctx, weight, bias, conv: HipConv2d, splits, *xs):
        out_pitch = ops.pad_channels(conv.out_channels)
        bpad = None
        if bias is not None:
            bpad = torch.zeros(out_pitch, dtype=torch.float32, device=xs[0].device)
            bpad[: conv.out_channels] = bias.detach()
        w = weight.detach()
        y, off = None, 0
        for m, (x, c) in enumerate(zip(xs, splits)):
            pw = _fusion_slice(conv, w, off, c, x.dtype, x.shape[-1], False)
            y = ops.conv2d(x, pw, 0, out_pitch, bias=bpad if m == 0 else None, residual=y)
            off += c
        ctx.conv, ctx.splits, ctx.has_bias, ctx.out_pitch = conv, tuple(splits), bias is not None, out_pitch
        ctx.save_for_backward(weight, *xs)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, *xs = ctx.saved_tensors
        conv, dy = ctx.conv, _as_nhwc_grad(dy)
        w = weight.detach()
        dxs, dws, off = [], [], 0
        for m, (x, c) in enumerate(zip(xs, ctx.splits)):
            dx = None
            if ctx.needs_input_grad[4 + m]:
                pwt = _fusion_slice(conv, w, off, c, x.dtype, ctx.out_pitch, True)
                dx = ops.conv2d(dy, pwt, 0, x.shape[-1], out_hw=(x.shape[1], x.shape[2]))
            dxs.append(dx)
            if ctx.needs_input_grad[0]:
                dws.append(ops.conv_wgrad(x, dy, conv.out_channels, c, 1, 1, 1, 0))
            off += c
        dw = torch.cat(dws, dim=1) if ctx.needs_input_grad[0] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            s, _ = ops.channel_sums(dy)
            db = s[: conv.out_channels].clone()
        return (dw, db, None, None, *dxs)


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, idx = ops.maxpool3x3s2_fwd(x)
        ctx.save_for_backward(idx)
        ctx.in_hw = (x.shape[1], x.shape[2])
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        return ops.maxpool3x3s2_bwd(_as_nhwc_grad(dy), idx, ctx.in_hw)


class _MaxPoolFork(torch.autograd.Function):
    """(pooled, x) from x: the ResNet stem output feeds the max-pool AND the U-Net skip.  As one node the two
    gradients of x are summed inside the max-pool backward kernel instead of by a separate pass over x."""

    @staticmethod
    def forward(ctx, x):
        y, idx = ops.maxpool3x3s2_fwd(x)
        ctx.save_for_backward(idx)
        ctx.in_hw = (x.shape[1], x.shape[2])
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        (idx,) = ctx.saved_tensors
        add = None if dskip is None else _as_nhwc_grad(dskip)
        if dy is None:
            return add
        return ops.maxpool3x3s2_bwd(_as_nhwc_grad(dy), idx, ctx.in_hw, add=add)


class _UpConcat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lo, skip):
        ctx.c1 = lo.shape[-1]
        return ops.upsample2x_concat_fwd(lo, skip)

    @staticmethod
    def backward(ctx, dcat):
        # every skip feature of the U-Net also feeds the next encoder stage, so its gradient is summed with another
        # one right away: hand autograd the channel slice of dcat instead of a copy (consumers that need a
        # contiguous tensor make one themselves, _as_nhwc_grad)
        dlo, dskip = ops.upsample2x_concat_bwd(_as_nhwc_grad(dcat), ctx.c1, skip_as_view=True)
        return dlo, dskip


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_hw):
        ctx.in_hw = (x.shape[1], x.shape[2])
        return ops.bilinear_fwd(x, out_hw)

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear_bwd(_as_nhwc_grad(dy), ctx.in_hw), None


class _ToNHWC(torch.autograd.Function):
    """f32 NCHW batch tensor -> NHWC compute tensor (no gradient needed for network inputs, but the
    generic-logits path of the loss uses the backward)."""

    @staticmethod
    def forward(ctx, x, dtype, cp):
        ctx.c = x.shape[1]
        return ops.nchw_to_nhwc(x, dtype, cp)

    @staticmethod
    def backward(ctx, dy):
        return ops.nhwc_to_nchw(_as_nhwc_grad(dy), ctx.c), None, None


# Per-class sums of the loss gradient, handed from the loss node to the node of the layer that produced the logits (its
# bias gradient = the column sums of dlogits).  Keyed by the gradient buffer's address and geometry; an entry lives from
# the loss node's backward to its consumer's backward within one backward pass (the buffer itself is alive in between)
# and every loss forward clears what an unconsumed earlier pass may have left.
_DLOGIT_SUMS = {}


def _take_dlogit_sums(dy: torch.Tensor):
    hit = _DLOGIT_SUMS.pop(dy.data_ptr(), None)
    if hit is None or hit[0] != tuple(dy.shape) or hit[1] != dy.dtype or not dy.is_contiguous():
        return None
    # the buffer must be exactly what the loss node handed out: if autograd accumulated a second gradient into it in
    # place (one logits tensor feeding two losses) its version counter has moved and the sums describe one addend only
    if dy._version != hit[3]:
        return None
    return hit[2]


class _SoftmaxCE(torch.autograd.Function):
    """Weighted-mean CE over NHWC logits.  When the logits need a gradient the forward pass writes dlogits too
    (for an upstream gradient of 1, in the same pass over the logits); backward multiplies them by the actual
    grad_output read on the device -- a no-op kernel when it is exactly 1 (plain ``loss.backward()``) -- so the
    usual training step reads the logits once instead of twice and never synchronises with the host."""

    @staticmethod
    def forward(ctx, logits, targets, weights, num_classes, holder):
        need = ctx.needs_input_grad[0]
        _DLOGIT_SUMS.clear()  # nothing of an earlier backward pass may outlive the next loss evaluation
        loss, wsum, dlogits, pred, sums = ops.softmax_ce(logits, targets, weights, num_classes, want_grad=need,
                                                         want_pred=True, want_sums=True)
        if holder is not None:
            holder["pred"] = pred
            holder["wsum"] = wsum
        ctx.dlogits, ctx.sums = dlogits, sums
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dlogits, ctx.dlogits = ctx.dlogits, None
        if dlogits is None:
            raise RuntimeError("softmax-CE backward called twice (the fused gradient buffer was already consumed)")
        gs = g.detach().reshape(1).float().contiguous()
        sums, ctx.sums = ctx.sums, None
        out = ops.scale_inplace(dlogits, gs)
        if sums is not None:  # the sums belong to an upstream gradient of 1, like dlogits before the rescale
            _DLOGIT_SUMS[out.data_ptr()] = (tuple(out.shape), out.dtype, sums * gs, out._version)
        return out, None, None, None, None


def conv_bn_act(x, conv: HipConv2d, bn: HipBatchNorm2d, relu: bool = True, residual=None):
    """Training: conv -> batch-stat BN -> (+residual) -> ReLU.  Eval: BN folded into the conv
    (scale into the packed weights, shift as bias) with the residual add / ReLU in the conv epilogue."""
    if bn.training:
        return _ConvBnAct.apply(x, conv.weight, bn.weight, bn.bias, residual, conv, bn, relu)
    pw, shift = _eval_folded(conv, bn, x.dtype)
    return ops.conv2d(x, pw, conv.padding, conv.out_pitch, bias=shift, residual=residual, relu=relu)


def _eval_folded(conv: HipConv2d, bn: HipBatchNorm2d, dtype: torch.dtype, ring: bool = True):
    """(packed weights with the eval-mode BatchNorm scale folded in, shift vector), cached per parameter version"""
    ver = (conv.weight._version, conv.weight.data_ptr(), bn.weight._version, bn.bias._version,
           bn.running_mean._version, bn.running_var._version, dtype, _STATE_EPOCH, getattr(bn, "_stats_epoch", 0))
    ck = "eval_fold" if ring else "eval_fold_igemm"
    hit = conv._cache.get(ck)
    if hit is None or hit[0] != ver:
        scale, shift = ops.bn_eval_params(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                          bn.eps)
        w = conv.weight.detach()
        pw = ops.pack_conv_weight(w if w.is_contiguous() else w.contiguous(), dtype, conv.stride, conv.in_pitch,
                                  scale=scale, allow_ring=ring and conv.padding == 1, allow_thin=conv.padding == 1,
                                  allow_stem=conv.padding == 3)
        hit = (ver, pw, shift)
        conv._cache[ck] = hit
    return hit[1], hit[2]


def up_conv_bn_act(lo, skip, conv: HipConv2d, bn: HipBatchNorm2d):
    """relu(bn(conv3x3(cat(nearest_x2(lo), skip)))): two-source kernels when the channel split allows, otherwise the
    explicit upsample + concat followed by conv_bn_act."""
    c1, c2 = lo.shape[-1], (0 if skip is None else skip.shape[-1])
    if conv.kernel_size == 3 and conv.stride == 1 and conv.padding == 1 and conv.in_channels == c1 + c2 and \
            ops.upcat_supported(c1, c2, lo.dtype):
        if bn.training:
            return _UpConvBnAct.apply(lo, skip, conv.weight, bn.weight, bn.bias, conv, bn)
        pw, shift = _eval_folded(conv, bn, lo.dtype, ring=False)
        y = ops.conv2d_upcat(lo, skip, pw, conv.out_pitch, bias=shift, relu=True)
        if y is not None:
            return y
    return conv_bn_act(up_concat(lo, skip), conv, bn, relu=True)


def decoder_block(lo, skip, blk):
    """Training-mode forward of a DecoderBlock module (conv1 = [conv, bn], conv2 = [conv, bn]); None when the channel
    split needs the explicit concat (the caller then chains up_conv_bn_act / conv_bn_act)."""
    ca, cb = blk.conv1[0], blk.conv2[0]
    c1, c2 = lo.shape[-1], (0 if skip is None else skip.shape[-1])
    if not (ca.kernel_size == 3 and ca.stride == 1 and ca.padding == 1 and ca.in_channels == c1 + c2 and
            cb.kernel_size == 3 and cb.stride == 1 and cb.padding == 1 and ops.upcat_supported(c1, c2, lo.dtype)):
        return None
    return _DecoderBlock.apply(lo, skip, ca.weight, blk.conv1[1].weight, blk.conv1[1].bias, cb.weight,
                               blk.conv2[1].weight, blk.conv2[1].bias, blk)


def basic_block(x, blk, fork: bool = False):
    """Training-mode forward of a BasicBlock module (conv1, bn1, conv2, bn2, downsample) as one autograd node.
    fork=True -> (y, alias of x): hand the alias to the other consumer of x (the U-Net skip)."""
    if blk.downsample is not None:
        cd, nd = blk.downsample[0], blk.downsample[1]
        extra = (cd.weight, nd.weight, nd.bias)
    else:
        extra = (None, None, None)
    return _BasicBlock.apply(x, blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight,
                             blk.bn2.bias, *extra, blk, fork)


def conv_bias(x, conv: HipConv2d):
    return _ConvBias.apply(x, conv.weight, conv.bias, conv)


def fusion_conv1x1(xs, splits, conv: HipConv2d):
    """conv(cat(xs, channel)) for a 1x1 ``conv`` whose input channels are the sources' real channels in order."""
    if conv.kernel_size != 1 or conv.stride != 1 or conv.padding != 0:
        raise ValueError("fusion_conv1x1: expects a 1x1 stride-1 convolution")
    if sum(splits) != conv.in_channels:
        raise ValueError(f"fusion_conv1x1: sources carry {sum(splits)} channels, the conv expects {conv.in_channels}")
    return _FusionConv1x1.apply(conv.weight, conv.bias, conv, tuple(splits), *xs)


class _MeanStack(torch.autograd.Function):
    """torch.mean(torch.stack(xs), 0) of same-shaped NHWC maps (FusionHandler case 3)"""

    @staticmethod
    def forward(ctx, *xs):
        ctx.n = len(xs)
        return ops.mean_stack([x.contiguous() for x in xs])

    @staticmethod
    def backward(ctx, g):
        gi = ops.mean_stack([_as_nhwc_grad(g)], divisor=ctx.n)
        return tuple(gi for _ in range(ctx.n))  # the same tensor for every branch: nobody writes a gradient in place


def mean_stack(xs):
    xs = list(xs)
    return xs[0] if len(xs) == 1 else _MeanStack.apply(*xs)


def max_pool(x):
    return _MaxPool.apply(x)


def max_pool_fork(x):
    """-> (max_pool(x), x): use the returned alias of x for the second consumer (see _MaxPoolFork)"""
    return _MaxPoolFork.apply(x)


def up_concat(lo, skip):
    return _UpConcat.apply(lo, skip)


def bilinear(x, out_hw):
    if (x.shape[1], x.shape[2]) == tuple(out_hw):
        return x  # align_corners=False resize to the same size is the identity (flair_model.py:327 on U-Net@512)
    return _Bilinear.apply(x, tuple(out_hw))


def to_nhwc(x_nchw: torch.Tensor, dtype: torch.dtype, cp: Optional[int] = None) -> torch.Tensor:
    return _ToNHWC.apply(x_nchw, dtype, cp)


def logits_view(y_nhwc: torch.Tensor, num_classes: int) -> torch.Tensor:
    """[B,K,H,W]-shaped view of NHWC logits, as callers of the reference expect (NCHW semantics);
    the underlying NHWC tensor rides along so the fused loss / argmax kernels can take the fast path."""
    v = y_nhwc[..., :num_classes].permute(0, 3, 1, 2)
    v._ffa_nhwc = y_nhwc
    v._ffa_classes = num_classes
    return v


class HipCrossEntropyLoss(nn.Module):
    """Drop-in for nn.CrossEntropyLoss(weight=w) as built by FLAIRLosses
    (flair_hub/tasks/module_setup.py:150-161); also yields argmax(softmax(logits)) from the same pass."""

    def __init__(self, weight: Optional[torch.Tensor] = None, num_classes: Optional[int] = None):
        super().__init__()
        if weight is None:
            if num_classes is None:
                raise ValueError("HipCrossEntropyLoss needs class weights or a class count")
            weight = torch.ones(num_classes)
        self.register_buffer("weight", weight.float().clone())
        self.last = {}

    @staticmethod
    def prepare_targets(t: torch.Tensor) -> torch.Tensor:
        if t.ndim == 4:  # one-hot NCHW (flair_hub/tasks/tasks_module.py:153)
            return ops.onehot_to_index(t)
        return t if (t.dtype == torch.uint8 and t.is_contiguous()) else t.to(torch.uint8).contiguous()

    def forward(self, logits: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        k = self.weight.numel()
        nhwc = getattr(logits, "_ffa_nhwc", None)
        if nhwc is None:
            if logits.ndim != 4 or logits.shape[1] != k:
                raise ValueError(f"expected logits [B,{k},H,W], got {tuple(logits.shape)}")
            nhwc = to_nhwc(logits.float().contiguous(), torch.float32, LOGIT_PITCH)
        self.last = {}
        return _SoftmaxCE.apply(nhwc, self.prepare_targets(targets), self.weight, k, self.last)

    def last_prediction(self) -> Optional[torch.Tensor]:
        return self.last.get("pred")
