"""ORACLE -- test infrastructure only (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the product path under flair-for-aigle_amd/).

Plain torch fp32 / eager / CPU restatement of the reference's DEFAULT architecture `swin_*-upernet`
(/root/reference/configs/train/config_models.yaml:5, configs/config_model_zonal_segmentation.yaml:26), which the
reference obtains from two third-party packages that are neither vendored under /root/reference nor installed here:
  flair_hub/models/monotemp_model.py:64-92  smp.create_model(arch="upernet", encoder_name="tu-swin_...",
      classes, in_channels, img_size)   -> segmentation-models-pytorch==0.4.0 (requirements.txt:12)
      -> TimmUniversalEncoder -> timm.create_model(name, features_only=True, in_chans, img_size)   (timm unpinned)
  monotemp_model.py:94-97                    .encoder  /  DecoderWrapper(.decoder, .segmentation_head)

PARITY STATUS: **parity unpinned**.  The reference holds no test, golden vector or fixture for this path and timm / smp
are absent, so the architecture is restated from the published algorithm (Liu et al., "Swin Transformer", ICCV 2021;
timm's models/swin_transformer.py as of the 1.0 series -- dynamic padding to the window grid, PatchMerging at the START of
stages 1..3; smp 0.4.0 decoders/upernet/decoder.py + base/heads.py + encoders/timm_universal.py).  Pins available:
  * the parameter count the reference publishes for LC-A (aerial only, Swin-B + UPerNet): 89.4 M
    (/root/reference/README.md:413) -- tests/test_oracle_goldens.py checks this file's count against it;
  * the "0-channel dummy feature at stride 2" convention the reference itself relies on
    (flair_hub/models/flair_model.py:302-306, 506-518): encoder.out_channels = [in, 0, C, 2C, 4C, 8C];
  * `relative_position_bias_table` as a state-dict key of shape [(2 ws - 1)^2, heads]
    (flair_hub/models/checkpoint.py:33-56, 265-271);
  * the ENCODER against an independent implementation that IS installed here: transformers.SwinModel (HuggingFace's port
    of the original Microsoft code).  With the weights mapped (qkv split, timm's merge-at-stage-start vs HF's
    merge-at-stage-end) the four stage outputs agree to 2e-5 on a 224 px input, where no stage needs padding
    (tests/test_oracle_goldens.py::test_swin_oracle_matches_the_huggingface_implementation): window partition, the
    shifted-window mask, the relative-position-bias indexing, the MLP and the patch-merging order are pinned.  What stays
    unpinned is the padded case (timm rolls, then pads; HF pads, then rolls) and smp's UPerNet decoder.
The ARITHMETIC is torch.nn.functional on CPU, i.e. the ATen ops timm / smp themselves call.

State-dict key names follow timm's FeatureListNet flattening (`layers_0.blocks.0...`) under smp's `encoder.model.`,
and smp's module tree for the decoder / head.
"""
from __future__ import annotations

import math
import re
from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

SWIN_VARIANTS = {
    # name -> (embed_dim, depths, heads)
    "tiny": (96, (2, 2, 6, 2), (3, 6, 12, 24)),
    "small": (96, (2, 2, 18, 2), (3, 6, 12, 24)),
    "base": (128, (2, 2, 18, 2), (4, 8, 16, 32)),
    "large": (192, (2, 2, 18, 2), (6, 12, 24, 48)),
}


def parse_swin_name(name: str):
    """'swin_base_patch4_window12_384' -> (embed_dim, depths, heads, window, patch)"""
    m = re.fullmatch(r"(?:tu-)?swin_(tiny|small|base|large)_patch(\d+)_window(\d+)_(\d+)(?:\..*)?", name)
    if not m:
        raise KeyError(f"not a Swin-v1 encoder name: {name}")
    dim, depths, heads = SWIN_VARIANTS[m.group(1)]
    return dim, depths, heads, int(m.group(3)), int(m.group(2))


def window_partition(x, ws: int):
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C)


def window_reverse(windows, ws: int, H: int, W: int):
    C = windows.shape[-1]
    x = windows.view(-1, H // ws, W // ws, ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, H, W, C)


def relative_position_index(ws: int) -> torch.Tensor:
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij"))  # 2, ws, ws
    flat = torch.flatten(coords, 1)
    rel = flat[:, :, None] - flat[:, None, :]  # 2, N, N : coords[i] - coords[j]
    rel = rel.permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)  # N, N


def shifted_window_mask(Hp: int, Wp: int, ws: int, shift: int) -> torch.Tensor:
    """SW-MSA mask on the padded grid: [nW, N, N] of 0 / -100"""
    img = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for w in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img, ws).view(-1, ws * ws)
    mask = mw.unsqueeze(1) - mw.unsqueeze(2)
    return mask.masked_fill(mask != 0, -100.0).masked_fill(mask == 0, 0.0)


class WindowAttention(nn.Module):
    def __init__(self, dim: int, heads: int, ws: int):
        super().__init__()
        self.dim, self.heads, self.ws = dim, heads, ws
        self.scale = (dim // heads) ** -0.5
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        self.register_buffer("relative_position_index", relative_position_index(ws), persistent=False)
        self.qkv = nn.Linear(dim, dim * 3)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)

    def forward(self, x, mask=None):
        B_, N, C = x.shape
        qkv = self.qkv(x).reshape(B_, N, 3, self.heads, -1).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        bias = self.relative_position_bias_table[self.relative_position_index.view(-1)].view(N, N, -1)
        attn = attn + bias.permute(2, 0, 1).unsqueeze(0)
        if mask is not None:
            nW = mask.shape[0]
            attn = attn.view(-1, nW, self.heads, N, N) + mask.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, self.heads, N, N)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B_, N, -1)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class DropPath(nn.Module):
    """timm layers/drop.py: stochastic depth per sample, kept samples scaled by 1 / keep_prob (training mode only)"""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        return x * mask.div_(keep)


class SwinBlock(nn.Module):
    def __init__(self, dim: int, resolution: Tuple[int, int], heads: int, ws: int, shift: int, drop_path: float = 0.0):
        super().__init__()
        self.drop_path1 = DropPath(drop_path)
        self.drop_path2 = DropPath(drop_path)
        # a window never exceeds the map; a map that fits one window is not shifted
        self.ws = min(ws, min(resolution))
        self.shift = 0 if min(resolution) <= ws else shift
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, heads, self.ws)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, dim * 4)

    def _attn(self, x):
        B, H, W, C = x.shape
        ws, shift = self.ws, self.shift
        if shift:
            x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
        pad_h = (ws - H % ws) % ws
        pad_w = (ws - W % ws) % ws
        x = F.pad(x, (0, 0, 0, pad_w, 0, pad_h))
        Hp, Wp = H + pad_h, W + pad_w
        xw = window_partition(x, ws).view(-1, ws * ws, C)
        mask = shifted_window_mask(Hp, Wp, ws, shift).to(x.dtype) if shift else None
        aw = self.attn(xw, mask).view(-1, ws, ws, C)
        x = window_reverse(aw, ws, Hp, Wp)[:, :H, :W, :].contiguous()
        if shift:
            x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
        return x

    def forward(self, x):
        B, H, W, C = x.shape
        x = x + self.drop_path1(self._attn(self.norm1(x)))
        x = x.reshape(B, -1, C)
        x = x + self.drop_path2(self.mlp(self.norm2(x)))
        return x.reshape(B, H, W, C)


class PatchMerging(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.norm = nn.LayerNorm(4 * dim)
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)

    def forward(self, x):
        B, H, W, C = x.shape
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        _, H, W, _ = x.shape
        x = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)
        return self.reduction(self.norm(x))


class SwinStage(nn.Module):
    def __init__(self, dim_in: int, dim: int, resolution, depth: int, heads: int, ws: int, downsample: bool,
                 drop_path=None):
        super().__init__()
        self.downsample = PatchMerging(dim_in) if downsample else nn.Identity()
        dp = drop_path if drop_path is not None else [0.0] * depth
        self.blocks = nn.Sequential(*[
            SwinBlock(dim, resolution, heads, ws, 0 if i % 2 == 0 else ws // 2, dp[i]) for i in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class PatchEmbed(nn.Module):
    def __init__(self, in_chans: int, dim: int, patch: int):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, dim, patch, patch)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x):
        return self.norm(self.proj(x).permute(0, 2, 3, 1))  # NHWC


class SwinFeatures(nn.Module):
    """timm.create_model('swin_*', features_only=True, out_indices=(0,1,2,3)): FeatureListNet over
    patch_embed, layers_0..3 (flatten_sequential), NHWC feature maps."""

    def __init__(self, name: str, in_chans: int, img_size: int, drop_path_rate: float = 0.1):
        super().__init__()
        dim, depths, heads, ws, patch = parse_swin_name(name)
        self.patch_embed = PatchEmbed(in_chans, dim, patch)
        res = img_size // patch
        dims = [dim * 2 ** i for i in range(4)]
        self.channels = dims
        total = sum(depths)
        dpr = [drop_path_rate * i / max(total - 1, 1) for i in range(total)]  # torch.linspace(0, rate, total)
        for i in range(4):
            if i > 0:
                res = (res + 1) // 2
            stage = SwinStage(dims[i - 1] if i else dim, dims[i], (res, res), depths[i], heads[i], ws, i > 0,
                              dpr[sum(depths[:i]):sum(depths[:i + 1])])
            setattr(self, f"layers_{i}", stage)
        self.apply(self._init)

    @staticmethod
    def _init(m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)

    def forward(self, x) -> List[torch.Tensor]:
        x = self.patch_embed(x)
        feats = []
        for i in range(4):
            x = getattr(self, f"layers_{i}")(x)
            feats.append(x)
        return feats


class TimmUniversalEncoder(nn.Module):
    """smp 0.4.0 encoders/timm_universal.py for a transformer-style (stride-4 first) channel-last backbone:
    [x, 0-channel dummy at stride 2, f4, f8, f16, f32] in NCHW"""

    def __init__(self, name: str, in_channels: int, img_size: int, drop_path_rate: float = 0.1):
        super().__init__()
        self.model = SwinFeatures(name, in_channels, img_size, drop_path_rate)
        self.out_channels = [in_channels, 0] + list(self.model.channels)

    def forward(self, x):
        feats = [f.permute(0, 3, 1, 2).contiguous() for f in self.model(x)]
        B, _, H, W = x.shape
        dummy = torch.empty([B, 0, H // 2, W // 2], dtype=x.dtype, device=x.device)
        return [x, dummy] + feats


def conv_bn_relu(cin: int, cout: int, k: int, padding: int = 0) -> nn.Sequential:
    """smp base/modules.py Conv2dReLU(use_batchnorm=True)"""
    return nn.Sequential(nn.Conv2d(cin, cout, k, padding=padding, bias=False), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


class PSPModule(nn.Module):
    def __init__(self, cin: int, cout: int, sizes: Sequence[int] = (1, 2, 3, 6)):
        super().__init__()
        self.blocks = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(s), conv_bn_relu(cin, cin // len(sizes), 1)) for s in sizes])
        self.out_conv = conv_bn_relu(cin * 2, cout, 1)

    def forward(self, x):
        h, w = x.shape[2:]
        out = [x] + [F.interpolate(b(x), size=(h, w), mode="bilinear", align_corners=False) for b in self.blocks]
        return self.out_conv(torch.cat(out, dim=1))


class FPNBlock(nn.Module):
    def __init__(self, skip_channels: int, pyramid_channels: int):
        super().__init__()
        self.skip_conv = conv_bn_relu(skip_channels, pyramid_channels, 1) if skip_channels != 0 else nn.Identity()

    def forward(self, x, skip):
        _, channels, h, w = skip.shape
        x = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=False)
        if channels != 0:
            x = x + self.skip_conv(skip)
        return x


class UPerNetDecoder(nn.Module):
    def __init__(self, encoder_channels, pyramid_channels: int = 256, segmentation_channels: int = 64):
        super().__init__()
        ch = list(encoder_channels)[::-1]  # [8C, 4C, 2C, C, 0, in]
        self.psp = PSPModule(ch[0], pyramid_channels)
        self.fpn_stages = nn.ModuleList([FPNBlock(c, pyramid_channels) for c in ch[1:]])
        self.fpn_bottleneck = conv_bn_relu((len(ch) - 1) * pyramid_channels, segmentation_channels, 3, padding=1)

    def forward(self, *features):
        out_size = features[0].shape[2:]
        target = [s // 4 for s in out_size]
        feats = features[1:][::-1]  # head of the encoder first; the input image itself is dropped
        fpn = [self.psp(feats[0])]
        for f, stage in zip(feats[1:], self.fpn_stages):
            fpn.append(stage(fpn[-1], f))
        resized = [F.interpolate(f, size=target, mode="bilinear", align_corners=False) for f in fpn]
        return self.fpn_bottleneck(torch.cat(resized, dim=1))


class SegmentationHead(nn.Sequential):
    """smp base/heads.py: Conv2d(k=1) -> UpsamplingBilinear2d(scale 4) (align_corners=True) -> Identity activation"""

    def __init__(self, cin: int, classes: int, kernel_size: int = 1, upsampling: int = 4):
        super().__init__(nn.Conv2d(cin, classes, kernel_size, padding=kernel_size // 2),
                         nn.UpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity(),
                         nn.Identity())
        nn.init.xavier_uniform_(self[0].weight)
        nn.init.constant_(self[0].bias, 0)


class SwinUPerNet(nn.Module):
    """What smp.create_model(arch='upernet', encoder_name='tu-swin_...', classes, in_channels, img_size) returns."""

    def __init__(self, encoder_name: str, in_channels: int = 3, classes: int = 1, img_size: int = 512,
                 drop_path_rate: float = 0.1):
        super().__init__()
        self.encoder = TimmUniversalEncoder(encoder_name, in_channels, img_size, drop_path_rate)
        self.decoder = UPerNetDecoder(self.encoder.out_channels)
        self.segmentation_head = SegmentationHead(64, classes)

    def forward(self, x):
        return self.segmentation_head(self.decoder(*self.encoder(x)))


def count_parameters(m: nn.Module) -> int:
    return sum(p.numel() for p in m.parameters())
