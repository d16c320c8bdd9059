"""ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product path under flair-for-aigle_amd/).

Plain torch fp32 / NCHW / eager restatement of the conv stack the reference obtains from the absent
third-party dependency segmentation-models-pytorch==0.4.0 (reference requirements.txt:12) through
flair_hub/models/monotemp_model.py:68-92:  smp.Unet(encoder_name='resnet34') split into
``.encoder`` and ``.decoder`` + ``.segmentation_head`` (monotemp_model.py:94-97).

PARITY STATUS: the reference repository holds no tests, golden vectors or fixtures for this path and
smp/timm are not installed here, so the ARCHITECTURE is restated from the published smp 0.4.0 /
torchvision sources and pinned only by
  * the parameter count smp publishes for Unet-ResNet34 (24,436,369 @ in=3, classes=1) and
  * the state-dict key names the reference itself hard-codes (flair_hub/models/checkpoint.py:225-228).
Round 2: the ENCODER is additionally pinned against an independent implementation that is installed in this image,
transformers.ResNetModel (layer_type 'basic', depths 3-4-6-3): with the weights mapped, the stem and the four stage
outputs agree to 2e-5 in evaluation mode and with training-mode BatchNorm
(tests/test_oracle_goldens.py::test_resnet34_encoder_oracle_matches_the_huggingface_implementation).  smp's UnetDecoder
(nearest x2 + concat + two conv-BN-ReLU per block, 3x3 head) has no installed counterpart and stays restated.
The ARITHMETIC is torch.nn.functional on CPU -- the very ATen ops smp would call -- so op-level
parity with this file is parity with the reference's accelerator=cpu path ("parity unpinned" by
reference-side vectors).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

# ---- storage-precision emulation (round 3) ------------------------------------------------------------------------
# ``storage_dtype=torch.bfloat16`` makes this restatement round every tensor the MI355X product materialises in bf16 --
# the input, every conv output (BatchNorm statistics are then taken from the rounded values, as the product's conv
# epilogue takes them from the registers it stores), every BatchNorm(+residual)+ReLU output, the logits, and in the
# backward pass the gradient arriving at each of those tensors -- while all arithmetic between two stores stays f32
# (the product accumulates in f32 on the MFMA and in its reductions).  Conv weights are rounded on use (the packed MFMA
# operand is bf16) but their gradient is NOT rounded (the product's weight gradients are f32 sums).  With the rounding
# points aligned, ReLU masks only differ where an f32 pre-activation lands within f32 round-off of zero, so parameter
# gradients can be compared one by one instead of by direction (tests/test_fullsize_gpu.py).


class _RoundStore(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        return x.to(dt).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).to(g.dtype), None


def _q(x, dt):
    return x if dt is None else _RoundStore.apply(x, dt)


def _qw(w, dt):
    """operand rounding only: forward sees the bf16 value, the gradient passes through unrounded"""
    return w if dt is None else w + (w.detach().to(dt).to(w.dtype) - w.detach())


def _conv(m: nn.Conv2d, x, dt):
    return _q(F.conv2d(x, _qw(m.weight, dt), m.bias, m.stride, m.padding), dt)


ENCODER_WIDTHS = (64, 128, 256, 512)
ENCODER_BLOCKS = (3, 4, 6, 3)
DECODER_CHANNELS = (256, 128, 64, 32, 16)


class BasicBlock(nn.Module):
    # torchvision.models.resnet.BasicBlock, which smp's ResNetEncoder inherits
    def __init__(self, cin: int, cout: int, stride: int, storage_dtype=None):
        super().__init__()
        self.sd = storage_dtype
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        dt = self.sd
        if dt is None:
            identity = x if self.downsample is None else self.downsample(x)
            out = F.relu(self.bn1(self.conv1(x)))
            out = self.bn2(self.conv2(out))
            return F.relu(out + identity)
        identity = x if self.downsample is None else _q(self.downsample[1](_conv(self.downsample[0], x, dt)), dt)
        out = _q(F.relu(self.bn1(_conv(self.conv1, x, dt))), dt)
        out = self.bn2(_conv(self.conv2, out, dt))
        return _q(F.relu(out + identity), dt)  # one store: BatchNorm + residual + ReLU are one pass in the product


class ResNet34Encoder(nn.Module):
    """smp.encoders.resnet.ResNetEncoder(depth=5): returns [x, stem, layer1, layer2, layer3, layer4]."""

    def __init__(self, in_channels: int = 3, storage_dtype=None):
        super().__init__()
        self.sd = storage_dtype
        self.out_channels = (in_channels, 64, 64, 128, 256, 512)
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (w, n) in enumerate(zip(ENCODER_WIDTHS, ENCODER_BLOCKS), start=1):
            blocks = []
            for b in range(n):
                blocks.append(BasicBlock(cin, w, 2 if (b == 0 and li > 1) else 1, storage_dtype))
                cin = w
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        for m in self.modules():  # torchvision's un-pretrained init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        dt = self.sd
        x = _q(x, dt)
        feats = [x]
        x = _q(F.relu(self.bn1(_conv(self.conv1, x, dt))), dt)
        feats.append(x)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
            feats.append(x)
        return feats


class DecoderBlock(nn.Module):
    # smp.decoders.unet.decoder.DecoderBlock with use_batchnorm=True, attention_type=None
    def __init__(self, cin: int, cskip: int, cout: int, storage_dtype=None):
        super().__init__()
        self.sd = storage_dtype
        self.conv1 = nn.Sequential(nn.Conv2d(cin + cskip, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout),
                                   nn.ReLU(inplace=True))
        self.conv2 = nn.Sequential(nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.BatchNorm2d(cout),
                                   nn.ReLU(inplace=True))

    def forward(self, x, skip=None):
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        dt = self.sd
        if dt is None:
            return self.conv2(self.conv1(x))
        x = _q(F.relu(self.conv1[1](_conv(self.conv1[0], x, dt))), dt)
        return _q(F.relu(self.conv2[1](_conv(self.conv2[0], x, dt))), dt)


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels, decoder_channels=DECODER_CHANNELS, storage_dtype=None):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]  # drop the input-resolution identity, deepest first
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.blocks = nn.ModuleList(DecoderBlock(i, s, o, storage_dtype) for i, s, o in zip(in_ch, skip_ch, decoder_channels))
        for m in self.modules():  # smp.base.initialization.initialize_decoder
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")

    def forward(self, *features):
        feats = list(features[1:])[::-1]
        x, skips = feats[0], feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class SegmentationHead(nn.Sequential):
    def __init__(self, cin: int, classes: int):
        super().__init__(nn.Conv2d(cin, classes, 3, padding=1), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self[0].weight)  # smp initialize_head
        nn.init.constant_(self[0].bias, 0)


class UnetResNet34(nn.Module):
    """What smp.create_model(arch='unet', encoder_name='resnet34', classes, in_channels) returns."""

    def __init__(self, in_channels: int = 3, classes: int = 1, storage_dtype=None):
        super().__init__()
        self.sd = storage_dtype
        self.encoder = ResNet34Encoder(in_channels, storage_dtype)
        self.decoder = UnetDecoder(self.encoder.out_channels, storage_dtype=storage_dtype)
        self.segmentation_head = SegmentationHead(DECODER_CHANNELS[-1], classes)

    def forward(self, x):
        y = self.decoder(*self.encoder(x))
        if self.sd is None:
            return self.segmentation_head(y)
        return _conv(self.segmentation_head[0], y, self.sd)


def count_parameters(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
