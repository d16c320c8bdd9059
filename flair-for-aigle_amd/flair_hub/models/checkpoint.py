"""Checkpoint loading -- counterpart of the reference's flair_hub/models/checkpoint.py (load_checkpoint
:176-290).  Host plumbing, kept because its state-dict KEY NAMES are part of the drop-in contract:
``[model.]encoders.<MOD>.seg_model.*``, ``[model.]main_decoders.<TASK>.seg_model.decoder.*``,
``[model.]main_decoders.<TASK>.seg_model.segmentation_head.0.{weight,bias}``, ``criterion.<TASK>.weight``.

Behaviour kept: .safetensors or torch formats; 'model.' prefix stripped when the module has none; a task head
whose class count differs from the config is re-initialised (Xavier weight, zero bias), as is any other
tensor whose shape disagrees -- except a Swin ``relative_position_bias_table``, which is resized (bicubic) to the
model's window (:33-56, :265-271); strict=False load; SystemExit on a bad path unless exit_on_fail=False.
Added: encoder keys of smp's timm-universal variant (``...seg_model.model.conv1.weight``) are accepted,
because the reference's constructor fallback (monotemp_model.py:67-92) can produce either spelling.
"""
from __future__ import annotations

import logging
import os
import re
from typing import Any, Dict

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)


def _fresh_like(t: torch.Tensor, key: str) -> torch.Tensor:
    p = torch.empty_like(t)
    if "weight" in key and p.ndim >= 2:
        nn.init.xavier_uniform_(p)
    else:
        nn.init.zeros_(p)
    return p


def interpolate_bias_table(ckpt_tensor: torch.Tensor, model_tensor: torch.Tensor) -> torch.Tensor:
    """Swin `relative_position_bias_table` [(2w-1)^2, heads] of a checkpoint trained with another window size:
    bicubic resize (align_corners=False) of the (2w-1) x (2w-1) table per head, as the reference does at load time
    (flair_hub/models/checkpoint.py:33-56).  Host-side, once per load."""
    old_len, heads = ckpt_tensor.shape
    new_len = model_tensor.shape[0]
    if old_len == new_len:
        return ckpt_tensor
    so, sn = int(old_len ** 0.5), int(new_len ** 0.5)
    if so * so != old_len or sn * sn != new_len:
        raise ValueError(f"relative position bias tables must be square: {old_len} -> {new_len}")
    t = ckpt_tensor.reshape(1, so, so, heads).permute(0, 3, 1, 2)
    t = torch.nn.functional.interpolate(t, size=(sn, sn), mode="bicubic", align_corners=False)
    return t.permute(0, 2, 3, 1).reshape(new_len, heads)


def _read(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    ckpt = torch.load(path, map_location="cpu")
    return ckpt.get("state_dict", ckpt)


def normalize_keys(state: Dict[str, torch.Tensor], model_keys) -> Dict[str, torch.Tensor]:
    model_keys = set(model_keys)
    has_prefix = any(k.startswith("model.") for k in state)
    if has_prefix and not any(k.startswith("model.") for k in model_keys):
        state = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state.items()}
    out = {}
    for k, v in state.items():
        if k not in model_keys and ".layers." in k:  # timm's un-flattened Swin spelling next to FeatureListNet's
            alt = re.sub(r"\.layers\.(\d+)\.", r".layers_\1.", k)
            if alt in model_keys:
                k = alt
        if k not in model_keys and ".seg_model.model." in k:
            alt = k.replace(".seg_model.model.", ".seg_model.", 1)
            if alt in model_keys:
                k = alt
        out[k] = v
    return out


def load_checkpoint(conf: Dict[str, Any], seg_module: nn.Module, exit_on_fail: bool = True) -> None:
    path = conf["paths"]["ckpt_model_path"]
    logger.info("loading checkpoint from: %s", path)
    if not path or not os.path.isfile(path):
        logger.info("invalid checkpoint path")
        if exit_on_fail:
            raise SystemExit()
        return
    model_dict = seg_module.state_dict()
    state = normalize_keys(_read(path), model_dict.keys())
    reinit = 0
    for task in conf["labels"]:
        n_classes = len(conf["labels_configs"][task]["value_name"])
        for prefix in ("model.", ""):
            wk = f"{prefix}main_decoders.{task}.seg_model.segmentation_head.0.weight"
            bk = wk[:-len("weight")] + "bias"
            if wk in model_dict and (wk not in state or state[wk].shape[0] != n_classes):
                state[wk] = _fresh_like(model_dict[wk], wk)
                state[bk] = _fresh_like(model_dict[bk], bk)
                reinit += 2
                logger.info("head of task '%s' re-initialised (%s classes)", task, n_classes)
    for k in list(state):
        if k in model_dict and state[k].shape != model_dict[k].shape:
            if "relative_position_bias_table" in k:
                try:
                    state[k] = interpolate_bias_table(state[k].float(), model_dict[k]).to(model_dict[k].dtype)
                    logger.info("interpolated %s: %s -> %s", k, tuple(state[k].shape), tuple(model_dict[k].shape))
                    continue
                except ValueError as e:
                    logger.info("interpolation failed for %s (%s): re-initialised", k, e)
            logger.info("shape mismatch for %s: checkpoint %s vs model %s -> re-initialised", k,
                        tuple(state[k].shape), tuple(model_dict[k].shape))
            state[k] = _fresh_like(model_dict[k], k) if model_dict[k].is_floating_point() else model_dict[k].clone()
            reinit += 1
    missing, unexpected = seg_module.load_state_dict(state, strict=False)
    logger.info("checkpoint loaded: %d tensors, %d re-initialised, %d missing, %d unexpected", len(state), reinit,
                len(missing), len(unexpected))
