"""ResNet-34 encoder + U-Net decoder + segmentation head on libflairhip kernels.

This is the L0 conv stack the reference obtains from segmentation_models_pytorch 0.4.0
(flair_hub/models/monotemp_model.py:68-97: ``smp.create_model(arch='unet', encoder_name='resnet34')``
split into ``.encoder`` and ``.decoder`` + ``.segmentation_head``).  Module / parameter names follow
smp's so state dicts interchange; tensors between layers are NHWC in the compute dtype.

Layer table, channel widths and init: SURVEY.md section 8a / Appendix C.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import nn as hnn

FUSED_FORKS = os.environ.get("FFA_FUSED_FORKS", "1") != "0"  # A/B switch: let autograd sum the skip gradients
ENCODER_WIDTHS = (64, 128, 256, 512)
ENCODER_BLOCKS = (3, 4, 6, 3)
DECODER_CHANNELS = (256, 128, 64, 32, 16)


class BasicBlock(nn.Module):
    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = hnn.HipConv2d(cin, cout, 3, stride, 1)
        self.bn1 = hnn.HipBatchNorm2d(cout)
        self.conv2 = hnn.HipConv2d(cout, cout, 3, 1, 1)
        self.bn2 = hnn.HipBatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(hnn.HipConv2d(cin, cout, 1, stride, 0), hnn.HipBatchNorm2d(cout))

    def forward(self, x):
        if self.training and torch.is_grad_enabled():
            return hnn.basic_block(x, self)  # one autograd node: the fork at x costs no elementwise add in backward
        identity = x
        if self.downsample is not None:
            identity = hnn.conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        out = hnn.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        return hnn.conv_bn_act(out, self.conv2, self.bn2, relu=True, residual=identity)


class ResNet34Encoder(nn.Module):
    """forward(x_nhwc) -> [x, stem@H/2, layer1@H/4, layer2@H/8, layer3@H/16, layer4@H/32] (NHWC)."""

    def __init__(self, in_channels: int = 3):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = (in_channels, 64, 64, 128, 256, 512)
        self.conv1 = hnn.HipConv2d(in_channels, 64, 7, 2, 3)
        self.bn1 = hnn.HipBatchNorm2d(64)
        cin = 64
        for li, (w, n) in enumerate(zip(ENCODER_WIDTHS, ENCODER_BLOCKS), start=1):
            blocks = []
            for b in range(n):
                blocks.append(BasicBlock(cin, w, 2 if (b == 0 and li > 1) else 1))
                cin = w
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        for m in self.modules():  # torchvision's init for un-pretrained ResNets
            if isinstance(m, hnn.HipConv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        feats = [x]
        x = hnn.conv_bn_act(x, self.conv1, self.bn1, relu=True)
        if self.training and torch.is_grad_enabled() and FUSED_FORKS:
            # Every skip feature has two consumers (next stage + decoder).  The consumer on the encoder side hands
            # out the alias the decoder reads, so that its backward receives the decoder's gradient and adds it in a
            # kernel it runs anyway (max-pool backward / a dgrad epilogue) instead of autograd's elementwise sum.
            x, skip = hnn.max_pool_fork(x)
            feats.append(skip)
            for li in range(1, 5):
                blocks = getattr(self, f"layer{li}")
                for bi, blk in enumerate(blocks):
                    if bi == 0 and li > 1:
                        x, skip = hnn.basic_block(x, blk, fork=True)
                        feats[-1] = skip
                    else:
                        x = blk(x)
                feats.append(x)
            return feats
        feats.append(x)
        x = hnn.max_pool(x)
        for li in range(1, 5):
            x = getattr(self, f"layer{li}")(x)
            feats.append(x)
        return feats


class DecoderBlock(nn.Module):
    def __init__(self, cin: int, cskip: int, cout: int):
        super().__init__()
        # children "0" = conv, "1" = batch norm ("2" = ReLU holds no state): smp's Conv2dReLU layout
        self.conv1 = nn.Sequential(hnn.HipConv2d(cin + cskip, cout, 3, 1, 1), hnn.HipBatchNorm2d(cout))
        self.conv2 = nn.Sequential(hnn.HipConv2d(cout, cout, 3, 1, 1), hnn.HipBatchNorm2d(cout))

    def forward(self, x, skip: Optional[torch.Tensor] = None):
        if self.training and torch.is_grad_enabled():
            y = hnn.decoder_block(x, skip, self)  # one autograd node (two-source conv1, fused BN-backward sums)
            if y is not None:
                return y
        # nearest x2 + concat + conv1 + BN + ReLU; the concatenated tensor is only materialised when the two-source
        # kernels do not take the channel split
        x = hnn.up_conv_bn_act(x, skip, self.conv1[0], self.conv1[1])
        return hnn.conv_bn_act(x, self.conv2[0], self.conv2[1], relu=True)


class UnetDecoder(nn.Module):
    def __init__(self, encoder_channels: Sequence[int], decoder_channels: Sequence[int] = DECODER_CHANNELS):
        super().__init__()
        enc = list(encoder_channels[1:])[::-1]
        in_ch = [enc[0]] + list(decoder_channels[:-1])
        skip_ch = enc[1:] + [0]
        self.blocks = nn.ModuleList(DecoderBlock(i, s, o) for i, s, o in zip(in_ch, skip_ch, decoder_channels))
        for m in self.modules():  # smp initialize_decoder
            if isinstance(m, hnn.HipConv2d):
                nn.init.kaiming_uniform_(m.weight, mode="fan_in", nonlinearity="relu")

    def forward(self, *features: torch.Tensor) -> torch.Tensor:
        feats = list(features[1:])[::-1]
        x, skips = feats[0], feats[1:]
        for i, blk in enumerate(self.blocks):
            x = blk(x, skips[i] if i < len(skips) else None)
        return x


class SegmentationHead(nn.Sequential):
    """Conv2d(16, classes, 3, padding=1) with bias; children "1"/"2" are smp's Identity upsampling / activation."""

    def __init__(self, cin: int, classes: int):
        super().__init__(hnn.HipConv2d(cin, classes, 3, 1, 1, bias=True), nn.Identity(), nn.Identity())
        nn.init.xavier_uniform_(self[0].weight)
        nn.init.constant_(self[0].bias, 0)
        self.classes = classes

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return hnn.conv_bias(x, self[0])


class UnetResNet34(nn.Module):
    """Counterpart of smp.create_model('unet', 'resnet34', classes=..., in_channels=...)."""

    def __init__(self, in_channels: int = 3, classes: int = 1):
        super().__init__()
        self.encoder = ResNet34Encoder(in_channels)
        self.decoder = UnetDecoder(self.encoder.out_channels)
        self.segmentation_head = SegmentationHead(DECODER_CHANNELS[-1], classes)

    def forward(self, x_nhwc: torch.Tensor) -> torch.Tensor:
        return self.segmentation_head(self.decoder(*self.encoder(x_nhwc)))
