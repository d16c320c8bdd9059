"""Mono-temporal backbone selection -- HIP-backed counterpart of the reference's
flair_hub/models/monotemp_model.py:34-97 (FLAIR_Monotemp) and :7-31 (DecoderWrapper).

The reference asks segmentation_models_pytorch for ``smp.create_model(arch=<decoder>,
encoder_name=<encoder>, classes, in_channels[, img_size])`` and keeps either ``.encoder`` or
``.decoder`` + ``.segmentation_head``.  Here the same split is served by the libflairhip conv stack;
no third-party model zoo and no weight download is involved (smp's default ``encoder_weights=
"imagenet"`` fetch has no counterpart: weights come from the checkpoint or from the seeded init).

Supported ``models.monotemp_model.arch`` values: ``resnet34-unet`` (BASELINE configs 1-3) and
``swin_{tiny,small,base,large}_patch4_window{7,12}_{224,384}-upernet`` (the reference's default arch and the fork's zonal
configuration; BASELINE config 4; evaluation in fp32/bf16 and bf16 training, see flairhip/swin.py).  Any other encoder/decoder pair
raises NotImplementedError with the arch name.
"""
from __future__ import annotations

from typing import Any, Dict

import torch.nn as nn

from flairhip import swin, unet

SUPPORTED_ARCHS = {"resnet34-unet", "swin_*-upernet"}


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


def _is_supported(encoder: str, decoder: str) -> bool:
    if f"{encoder}-{decoder}".lower() == "resnet34-unet":
        return True
    return decoder.lower() == "upernet" and swin.is_swin_name(encoder)


class FLAIR_Monotemp(nn.Module):
    """Same constructor and attributes as the reference class: ``.seg_model`` is the encoder
    (``return_type='encoder'``, exposing ``.out_channels``) or the decoder + head wrapper."""

    def __init__(self, config: Dict[str, Any], channels: int = 3, classes: int = 19, img_size: int = 512,
                 return_type: str = "encoder") -> None:
        super().__init__()
        self.return_type = return_type
        assert self.return_type in ["encoder", "decoder"], 'return_type should be one of ["encoder", "decoder"]'
        arch, encoder, decoder = _split_arch(config)
        if not _is_supported(encoder, decoder):
            raise NotImplementedError(
                f"monotemp arch '{arch}' has no libflairhip implementation yet (available: {sorted(SUPPORTED_ARCHS)})")
        if decoder.lower() == "upernet":
            # smp.create_model("upernet", "tu-swin_...", img_size=img_size): the Swin is built for this input size
            if return_type == "encoder":
                # timm's SwinTransformer default (stochastic depth 0.1); `drop_path_rate` under models.monotemp_model
                # is an optional addition of this build (smp.create_model would take it as a keyword too)
                dpr = float(config["models"]["monotemp_model"].get("drop_path_rate", 0.1))
                self.seg_model = swin.HipSwinEncoder(encoder, channels, img_size, drop_path_rate=dpr)
            else:
                dim = swin.parse_swin_name(encoder)[0]
                enc_channels = [channels, 0] + [dim * 2 ** i for i in range(4)]
                self.seg_model = DecoderWrapper(swin.HipUPerNetDecoder(enc_channels), swin.HipUPerNetHead(64, classes))
            return
        if return_type == "encoder":
            self.seg_model = unet.ResNet34Encoder(channels)
        else:
            # the reference builds a whole smp model with `channels` inputs and discards its encoder
            # (flair_model.py:153-159); only the channel table of the encoder is needed here
            enc_channels = (channels, 64, 64, 128, 256, 512)
            self.seg_model = DecoderWrapper(unet.UnetDecoder(enc_channels),
                                            unet.SegmentationHead(unet.DECODER_CHANNELS[-1], classes))
