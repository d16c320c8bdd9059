"""FLAIR_HUB_Model on libflairhip -- drop-in for the reference's flair_hub/models/flair_model.py
(class FLAIR_HUB_Model :16, forward :357-430, interpolate_map :318-327, FusionHandler :437-547).

Same constructor, attributes (``encoders``, ``fusion_handler``, ``main_decoders``, ``aux_decoders``,
``task_nclasses``, ``config``) and ``forward(batch, apply_mod_dropout=False) -> (logits_tasks,
logits_aux)`` contract.  Inside, tensors are NHWC in the compute dtype (``config['hardware']
['precision']``: 'bf16' default, 'fp32' for the 1e-4 parity mode) and every operator is a HIP kernel.

Scope (SURVEY.md section 8 + 8f rank 1): any number of mono-temporal modalities (AERIAL_RGBI, AERIAL-RLT_PAN,
DEM_ELEV, SPOT_RGBI) fused per stage by FusionHandler, one or several tasks, auxiliary per-modality decoders and
modality dropout.  Several Sentinel U-TAE branches without an aerial encoder average their class scores (case 3).
"""
from __future__ import annotations

import logging
import math
import random
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from flairhip import nn as hnn
from flairhip import ops
from flair_hub.models.monotemp_model import FLAIR_Monotemp

logger = logging.getLogger(__name__)

MONO_KEYS = ["AERIAL_RGBI", "AERIAL-RLT_PAN", "DEM_ELEV", "SPOT_RGBI"]
MULTI_KEYS = ["SENTINEL2_TS", "SENTINEL1-ASC_TS", "SENTINEL1-DESC_TS"]


def compute_dtype_of(config: dict) -> torch.dtype:
    name = str(config.get("hardware", {}).get("precision", "bf16")).lower()
    if name in ("bf16", "bfloat16", "bf16-mixed"):
        return torch.bfloat16
    if name in ("fp32", "32", "float32", "32-true"):
        return torch.float32
    raise ValueError(f"unsupported hardware.precision '{name}' (use 'bf16' or 'fp32')")


class FusionHandler(nn.Module):
    """Feature fusion across modalities (reference :437-547).

    One active modality: its feature list is returned untouched (:489-494).  Several mono-temporal modalities
    (:505-547): every stage of every modality is aligned to the first modality's stage size (bilinear,
    align_corners=False, only when the sizes differ), the stages are concatenated along channels and mixed by
    the per-stage 1x1 ``conv_f``.  Here the concat is never materialised: conv_f over the concat is evaluated as
    a chain of per-modality 1x1 convs through the conv epilogue's residual input (flairhip.nn.fusion_conv1x1).

    Stage 0 (the input-resolution identity feature) is not mixed: every decoder of this build drops it
    (smp's UnetDecoder.forward starts from ``features[1:]``), so running a 1x1 conv over B x H x W x (C_a + C_b)
    would be a full-resolution pass nobody reads.  The slot keeps the first modality's tensor; ``conv_f[0]``
    still exists so state dicts keep their keys."""

    def __init__(self, backbones_channels: List[int], target_fused_channels: List[int], mono_keys: List[str],
                 multi_keys: List[str]) -> None:
        super().__init__()
        self.mono_keys, self.multi_keys = mono_keys, multi_keys
        target = list(target_fused_channels)
        if len(target) > 2 and (target[0] == 0 or target[1] == 0):
            target = target[2:]
        self.conv_f = nn.ModuleList(
            hnn.HipConv2d(cin, cout, 1, 1, 0, bias=True) for cin, cout in zip(backbones_channels, target))
        self.stage_channels: Dict[str, List[int]] = {}  # real channels per stage of each modality (set by the model)

    def forward(self, feature_maps: dict, target_fm_maps):
        active = list(feature_maps.keys())
        mono = [k for k in active if k in self.mono_keys]
        multi = [k for k in active if k in self.multi_keys]
        if len(mono) == 1 and not multi:
            return feature_maps[mono[0]]
        if not mono and len(multi) == 1:
            return feature_maps[multi[0]]
        if not mono:
            # case 3 (:496-501): several time-series branches, no aerial encoder -> the mean of their class-score maps
            # (already resized to the label size, :390-392); one pass, f32 sum, one rounding
            if len(multi) > 4:
                raise NotImplementedError(f"mean over {len(multi)} time-series branches (ffa_mean_stack takes four)")
            return hnn.mean_stack([feature_maps[k] for k in multi])
        # case 4 (:504-547): a U-TAE branch contributes its decoder maps, coarse to fine, one per aerial stage (the
        # reference zips them in list order, flair_model.py:514-531); every map is resized to the stage it meets
        def strip(maps):  # the [input, 0-channel placeholder] pair of a transformer-style encoder (:508-518)
            maps = list(maps)
            if len(maps) > 2 and (maps[0].shape[-1] == 0 or maps[1].shape[-1] == 0):
                return maps[:2], maps[2:]
            return [], maps
        lead, target = strip(target_fm_maps)
        per_mod = {}
        for m in active:
            maps = strip(feature_maps[m])[1]
            if len(maps) != len(target):  # (:520-521)
                maps = [maps[0]] * (len(target) - len(maps)) + maps
            per_mod[m] = maps
        # with a placeholder pair every real stage is mixed; otherwise stage 0 (the input-resolution identity feature,
        # which every decoder of this build drops) keeps the first modality's tensor
        fused = list(lead) if lead else [target[0]]
        for s in range(0 if lead else 1, len(target)):
            th, tw = target[s].shape[1], target[s].shape[2]
            xs = [hnn.bilinear(per_mod[m][s], (th, tw)) for m in active]
            splits = [self.stage_channels[m][s] for m in active]
            fused.append(hnn.fusion_conv1x1(xs, splits, self.conv_f[s]))
        return fused


class FLAIR_HUB_Model(nn.Module):
    def __init__(self, config: dict, img_input_sizes: dict):
        super().__init__()
        self.config = config
        self.img_input_sizes = img_input_sizes
        self.mono_keys = list(MONO_KEYS)
        self.multi_keys = list(MULTI_KEYS)
        self.compute_dtype = compute_dtype_of(config)

        mods = config["modalities"]
        inputs = mods["inputs"]
        self.aux_losses = {m: v for m, v in mods.get("aux_loss", {}).items() if v and inputs.get(m, False)}
        self.tasks = len(config["labels"])
        self.task_nclasses = sum(len(config["labels_configs"][t]["value_name"]) for t in config["labels"])

        # channel count per modality (reference :70-87)
        self.channels_dict = {}
        for m in inputs:
            if m in ("AERIAL-RLT_PAN", "DEM_ELEV"):
                self.channels_dict[m] = 1
            else:
                self.channels_dict[m] = len(mods["inputs_channels"][m]) if m in mods.get("inputs_channels", {}) else 0
        if inputs.get("DEM_ELEV", False):
            pp = mods["pre_processings"]
            self.channels_dict["DEM_ELEV"] = 1 if (pp["calc_elevation"] and not pp["calc_elevation_stack_dsm"]) else 2

        self.encoders = nn.ModuleDict()
        for m in self.mono_keys:
            if inputs.get(m, False):
                self.encoders[m] = FLAIR_Monotemp(config, channels=self.channels_dict[m], classes=self.task_nclasses,
                                                  img_size=img_input_sizes[m], return_type="encoder")
        has_mono = len(self.encoders) > 0

        # Sentinel time-series branches (reference :101-134): U-TAE with FLAIR's fixed hyper-parameters; its class
        # score layer covers all tasks, and next to aerial encoders its widths follow their stage count
        active_multi = [k for k in self.multi_keys if inputs.get(k, False)]
        if active_multi:
            mt = config["models"]["multitemp_model"]
            if self.task_nclasses != mt["out_conv"][-1]:
                mt["out_conv"].append(self.task_nclasses)
            if has_mono:
                mono_ch = next(iter(self.encoders.values())).seg_model.out_channels
                mt["encoder_widths"] = self.adjust_fm_length(config, mono_ch)
                mt["decoder_widths"] = self.adjust_fm_length(config, mono_ch)
            from flairhip.utae import HipUTAE
            for m in active_multi:
                self.encoders[m] = HipUTAE(
                    input_dim=len(mods["inputs_channels"][m]), encoder_widths=mt["encoder_widths"],
                    decoder_widths=mt["decoder_widths"], out_conv=mt["out_conv"], str_conv_k=mt["str_conv_k"],
                    str_conv_s=mt["str_conv_s"], str_conv_p=mt["str_conv_p"], agg_mode=mt["agg_mode"],
                    encoder_norm=mt["encoder_norm"], n_head=mt["n_head"], d_model=mt["d_model"], d_k=mt["d_k"],
                    encoder=False, return_maps=True, pad_value=mt["pad_value"], padding_mode=mt["padding_mode"],
                    precision="bf16" if self.compute_dtype == torch.bfloat16 else "fp32")
        if not self.encoders:
            raise ValueError("no active input modality in config['modalities']['inputs']")

        def n_cls(task):
            return len(config["labels_configs"][task]["value_name"])

        if has_mono:
            # channels per stage of every modality (reference calc_backbones_channels :290-314): the aerial stages, and
            # for a U-TAE branch its decoder widths reversed (its maps come coarse to fine)
            def real_stages(ch):  # smp's transformer-style encoders lead with [in, 0]: both dropped (reference :302-306)
                ch = list(ch)
                return ch[2:] if len(ch) > 2 and (ch[0] == 0 or ch[1] == 0) else ch
            stage_channels = {m: real_stages(self.encoders[m].seg_model.out_channels) for m in self.encoders
                              if m in self.mono_keys}
            for m in active_multi:
                stage_channels[m] = list(config["models"]["multitemp_model"]["decoder_widths"])[::-1]
            total_per_stage = [sum(c) for c in zip(*stage_channels.values())]
            target = next(iter(self.encoders.values())).seg_model.out_channels
        else:
            stage_channels, total_per_stage, target = {}, [1], [1]  # the reference's dummy (:141-143)
        self.fusion_handler = FusionHandler(total_per_stage, target, self.mono_keys, self.multi_keys)
        self.fusion_handler.stage_channels = stage_channels

        self.main_decoders = nn.ModuleDict()
        for task in config["labels"]:
            # U-Net decoders over the fused aerial stages, or -- Sentinel only -- a 1x1 head over the U-TAE scores
            self.main_decoders[task] = FLAIR_Monotemp(config, channels=1, classes=n_cls(task), return_type="decoder") \
                if has_mono else hnn.HipConv2d(self.task_nclasses, n_cls(task), 1, 1, 0, bias=True)
        # one extra decoder per (aux-loss modality, task), fed by that modality's own features (reference :170-188)
        self.aux_decoders = nn.ModuleDict()
        for task in config["labels"]:
            for m in self.aux_losses:
                self.aux_decoders[f"{m}__{task}"] = FLAIR_Monotemp(
                    config, channels=1, classes=n_cls(task), return_type="decoder") if m in self.mono_keys \
                    else hnn.HipConv2d(self.task_nclasses, n_cls(task), 1, 1, 0, bias=True)
        self._pack_plan = hnn.PackPlan(self)
        self._log_parameter_table()

    # ---- helpers ---------------------------------------------------------------------------------

    def _log_parameter_table(self) -> None:
        arch = self.config["models"]["monotemp_model"]["arch"]
        total = 0
        for kind, group in (("backbone", self.encoders), ("aux loss decoder", self.aux_decoders),
                            ("task decoder", self.main_decoders)):
            for key, mod in group.items():
                n = sum(p.numel() for p in mod.parameters())
                total += n
                logger.info("| %-30s | %-28s | %-14s | %13s |", key, arch, kind, f"{n:,}")
        logger.info("| %-30s   %-28s   %-14s   %13s |", "Total parameters", "", "", f"{total:,}")

    @staticmethod
    def adjust_fm_length(config: dict, mono_temp_backbone_channels) -> List[int]:
        """U-TAE widths next to an aerial encoder (reference :196-214): as many stages as the encoder has (without
        its two zero-channel dummies), linearly spaced between the configured extremes and snapped to powers of two
        -- [64, 64, 64, 128] becomes [64, 64, 64, 128, 128, 128] for the six ResNet-34 stages"""
        import numpy as np
        ch = list(mono_temp_backbone_channels)
        if len(ch) > 2 and (ch[0] == 0 or ch[1] == 0):
            ch = ch[2:]
        widths = config["models"]["multitemp_model"]["encoder_widths"]
        spaced = np.linspace(min(widths) - 1, max(widths) + 1, len(ch)).astype(int)
        return [int(2 ** round(math.log(int(v), 2))) for v in spaced]

    def interpolate_map(self, x: torch.Tensor, size) -> torch.Tensor:
        """bilinear, align_corners=False (reference :318-327); ``size`` may be an int as in the reference's call."""
        hw = (size, size) if isinstance(size, int) else tuple(size)
        return hnn.bilinear(x, hw)

    def _input_nhwc(self, x: torch.Tensor, mod: str, norm=None) -> torch.Tensor:
        """batch tensor -> NHWC compute tensor.  Besides the reference's normalised f32 [B,C,H,W] tensors, raw raster
        samples (uint8 / uint16 / int16 / float32) are accepted when the batch carries '<MOD>_NORM' = f32 [2,C]
        (mean, std): the zonal dataset's (x - mean) / std then happens in the layout kernel on the device."""
        enc = self.encoders[mod].seg_model
        if x.ndim != 4 or x.shape[1] != enc.in_channels:
            raise ValueError(f"batch['{mod}'] must be [B,{enc.in_channels},H,W], got {tuple(x.shape)}")
        if x.shape[-1] % 32 or x.shape[-2] % 32:
            raise RuntimeError(f"input height and width must be divisible by 32, got {tuple(x.shape[-2:])}")
        if not x.is_cuda:
            raise RuntimeError("FLAIR_HUB_Model (libflairhip) runs on an MI355X only: move the batch to cuda")
        if x.dtype != torch.float32 or norm is not None:
            # raw raster samples (uint8 / uint16 / int16, or float32 next to a '<MOD>_NORM' entry)
            if norm is None:
                raise ValueError(f"batch['{mod}'] is {x.dtype}: batch['{mod}_NORM'] (f32 [2,C]: mean, std) is required")
            norm = norm.to(x.device, torch.float32)
            if x.dtype == torch.uint8:
                return ops.u8_nchw_to_nhwc(x.contiguous(), self.compute_dtype, norm[0].contiguous(),
                                           norm[1].contiguous(), getattr(enc, "input_pitch", ops.pad_channels(enc.in_channels)))
            return ops.raw_nchw_to_nhwc(x.contiguous(), self.compute_dtype, norm[0].contiguous(), norm[1].contiguous(),
                                        getattr(enc, "input_pitch", ops.pad_channels(enc.in_channels)))
        return hnn.to_nhwc(x, self.compute_dtype, getattr(enc, "input_pitch", ops.pad_channels(enc.in_channels)))

    def modality_dropout(self, feature_maps: Dict[str, list], modalities_dropout_dict: Dict[str, float]):
        """Reference :328-352: with probability ``modalities_dropout_dict[mod]`` the modality's feature maps are
        replaced by fresh Xavier-uniform noise (created per call, never trained).  Same host RNG draws
        (``torch.rand(1)``) as the reference; the noise itself is drawn on the device, so it matches the
        reference in distribution, not bit for bit.  Xavier bound for an NCHW [B,C,H,W] tensor:
        fan_in = C*H*W, fan_out = B*H*W (torch.nn.init._calculate_fan_in_and_fan_out)."""
        for key in feature_maps.keys():
            if torch.rand(1).item() < modalities_dropout_dict[key]:
                real = self.fusion_handler.stage_channels[key]
                maps = list(feature_maps[key])
                # a transformer encoder's [input, 0-channel placeholder] pair is not part of stage_channels (and no
                # decoder reads it): it stays as it is
                lead = 2 if len(maps) > 2 and (maps[0].shape[-1] == 0 or maps[1].shape[-1] == 0) else 0
                noise = maps[:lead]
                for t, c in zip(maps[lead:], real):
                    b, h, w, cp = t.shape
                    bound = math.sqrt(6.0 / float(c * h * w + b * h * w))
                    n = torch.empty_like(t).uniform_(-bound, bound)
                    if cp > c:
                        n[..., c:] = 0  # pad channels stay zero (layout invariant of every kernel)
                    noise.append(n)
                feature_maps[key] = noise
        return feature_maps

    # ---- forward ---------------------------------------------------------------------------------

    def forward(self, batch: dict, apply_mod_dropout: bool = False) -> Tuple[Dict[str, torch.Tensor], Dict]:
        labels = self.config["labels"]
        if self.training:
            self._pack_plan.refresh(self.compute_dtype)  # all conv operands re-packed in one launch per step
        fmaps: Dict[str, list] = {}
        first_mod = next(iter(self.encoders))
        # the reference learns the output size from the label tensor (:371); the zonal dataset fabricates a
        # zero label for that purpose -- fall back to the input size when no label rides along
        # (the reference passes the int shape[-1], i.e. assumes square tiles; both dims are kept here, which is
        # the same thing for the square tiles it is used with)
        if labels and labels[0] in batch:
            img_size = tuple(batch[labels[0]].shape[-2:])
        else:
            img_size = tuple(batch[first_mod].shape[-2:])

        logits_tasks: Dict[str, torch.Tensor] = {}
        logits_aux: Dict[str, torch.Tensor] = {}

        def decode(decoder, feats, task):
            y = self.interpolate_map(decoder.seg_model(*feats), img_size)
            return hnn.logits_view(y, len(self.config["labels_configs"][task]["value_name"]))

        def head1x1(conv, scores, task):  # the Sentinel-only model's nn.Conv2d(task_nclasses, classes, 1) (:153-166)
            if self.training and torch.is_grad_enabled():
                return hnn.logits_view(hnn.conv_bias(scores, conv),
                                       len(self.config["labels_configs"][task]["value_name"]))
            pw = conv.packed(scores.dtype, ring=False)
            bias = torch.zeros(conv.out_pitch, dtype=torch.float32, device=scores.device)
            bias[: conv.out_channels] = conv.bias.detach()
            y = ops.conv2d(scores, pw, 0, conv.out_pitch, bias=bias)
            return hnn.logits_view(y, len(self.config["labels_configs"][task]["value_name"]))

        scores: Dict[str, torch.Tensor] = {}  # class scores of the time-series branches (NHWC)
        for mod, encoder in self.encoders.items():
            if mod in self.mono_keys:
                fmaps[mod] = encoder.seg_model(self._input_nhwc(batch[mod], mod, batch.get(mod + "_NORM")))
                if self.aux_losses.get(mod):
                    for task in labels:
                        logits_aux[f"aux_{mod}_{task}"] = decode(self.aux_decoders[f"{mod}__{task}"], fmaps[mod], task)
            else:
                # U-TAE: class scores AND decoder maps (reference :388-404); dates ride along as '<SENSOR>_DATES'
                dates = batch.get(mod.replace("TS", "DATES"))
                if dates is None:
                    raise KeyError(f"batch['{mod.replace('TS', 'DATES')}'] (acquisition dates, [B, T]) is required")
                s_nhwc, maps, _ = encoder.forward_nhwc(batch[mod], dates)
                scores[mod] = self.interpolate_map(s_nhwc, img_size)
                fmaps[mod] = maps
                if self.aux_losses.get(mod):
                    for task in labels:
                        logits_aux[f"aux_{mod}_{task}"] = head1x1(self.aux_decoders[f"{mod}__{task}"], scores[mod], task)

        if apply_mod_dropout and len(self.encoders) > 1:
            fmaps = self.modality_dropout(fmaps, {key: random.uniform(0, 1) for key in fmaps.keys()})

        if any(m in self.mono_keys for m in self.encoders):
            fused = self.fusion_handler(fmaps, fmaps[first_mod])
            for task in labels:
                logits_tasks[task] = decode(self.main_decoders[task], fused, task)
        else:
            fused_scores = self.fusion_handler(scores, None)  # one time-series branch: its scores (:493-494)
            for task in labels:
                if len(labels) > 1:
                    logits_tasks[task] = head1x1(self.main_decoders[task], fused_scores, task)
                else:
                    logits_tasks[task] = hnn.logits_view(fused_scores, self.task_nclasses)
        return logits_tasks, logits_aux
