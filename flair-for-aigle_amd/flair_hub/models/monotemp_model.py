"""Mono-temporal backbone selection -- HIP-backed counterpart of the reference's
flair_hub/models/monotemp_model.py:34-97 (FLAIR_Monotemp) and :7-31 (DecoderWrapper).

The reference asks segmentation_models_pytorch for ``smp.create_model(arch=<decoder>,
encoder_name=<encoder>, classes, in_channels[, img_size])`` and keeps either ``.encoder`` or
``.decoder`` + ``.segmentation_head``.  Here the same split is served by the libflairhip conv stack;
no third-party model zoo and no weight download is involved (smp's default ``encoder_weights=
"imagenet"`` fetch has no counterpart: weights come from the checkpoint or from the seeded init).

Supported ``models.monotemp_model.arch`` values: ``resnet34-unet`` (BASELINE configs 1-3).
Other encoder/decoder pairs of the reference (Swin + UPerNet) raise
NotImplementedError with the arch name -- they are scheduled after the U-Net path meets its bar.
"""
from __future__ import annotations

from typing import Any, Dict

import torch.nn as nn

from flairhip import unet

SUPPORTED_ARCHS = {"resnet34-unet"}


class DecoderWrapper(nn.Module):
    """decoder followed by segmentation head, called with the encoder's feature list unpacked."""

    def __init__(self, decoder: nn.Module, segmentation_head: nn.Module) -> None:
        super().__init__()
        self.decoder = decoder
        self.segmentation_head = segmentation_head

    def forward(self, *features: Any):
        return self.segmentation_head(self.decoder(*features))


def _split_arch(config: Dict[str, Any]):
    arch = config["models"]["monotemp_model"]["arch"]
    parts = arch.split("-")
    encoder, decoder = parts[0], parts[1]
    if encoder.startswith("tu_"):  # tolerate the timm-universal spelling the reference falls back to
        encoder = encoder[3:]
    return arch, encoder, decoder


class FLAIR_Monotemp(nn.Module):
    """Same constructor and attributes as the reference class: ``.seg_model`` is the encoder
    (``return_type='encoder'``, exposing ``.out_channels``) or the decoder + head wrapper."""

    def __init__(self, config: Dict[str, Any], channels: int = 3, classes: int = 19, img_size: int = 512,
                 return_type: str = "encoder") -> None:
        super().__init__()
        self.return_type = return_type
        assert self.return_type in ["encoder", "decoder"], 'return_type should be one of ["encoder", "decoder"]'
        arch, encoder, decoder = _split_arch(config)
        if f"{encoder}-{decoder}".lower() not in SUPPORTED_ARCHS:
            raise NotImplementedError(
                f"monotemp arch '{arch}' has no libflairhip implementation yet (available: {sorted(SUPPORTED_ARCHS)})")
        if return_type == "encoder":
            self.seg_model = unet.ResNet34Encoder(channels)
        else:
            # the reference builds a whole smp model with `channels` inputs and discards its encoder
            # (flair_model.py:153-159); only the channel table of the encoder is needed here
            enc_channels = (channels, 64, 64, 128, 256, 512)
            self.seg_model = DecoderWrapper(unet.UnetDecoder(enc_channels),
                                            unet.SegmentationHead(unet.DECODER_CHANNELS[-1], classes))
