"""Logits -> uint8 outputs -- counterpart of the reference's flair_zonal_detection/postprocess.py
(convert :9-30).  'argmax': first maximum over the class axis as uint8 with a leading singleton axis;
'class_prob': round-half-even of softmax * 255.  Both run as one HIP kernel over NHWC logits
(ffa_predict_u8); inside the tile loop the margin crop is fused into the same kernel so only 1 byte
per pixel leaves the GPU (the reference ships 76 B/pixel of f32 logits to the host and loops in numpy).
"""
from __future__ import annotations

import numpy as np
import torch

from flairhip import nn as hnn
from flairhip import ops


def _as_device_nhwc(img):
    """(C,H,W) numpy / torch logits -> [1,H,W,32] f32 NHWC on the GPU (+ class count)."""
    if isinstance(img, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
    else:
        t = img.detach().float()
    if t.ndim != 3:
        raise ValueError("Expected logits with shape (C, H, W)")
    if t.shape[0] > hnn.LOGIT_PITCH:
        raise ValueError(f"at most {hnn.LOGIT_PITCH} classes are supported, got {t.shape[0]}")
    return ops.nchw_to_nhwc(t.cuda()[None].contiguous(), torch.float32, hnn.LOGIT_PITCH), t.shape[0]


def convert(img, img_type: str):
    """Same contract as the reference: (C,H,W) logits -> uint8 (1,H,W) for 'argmax', (C,H,W) for 'class_prob';
    any other type raises ValueError.  numpy in -> numpy out, torch in -> torch (device) out."""
    if img_type not in ("class_prob", "argmax"):
        raise ValueError(f"Unknown output type: {img_type}")
    nhwc, k = _as_device_nhwc(img)
    out = ops.predict_u8(nhwc, k, img_type)  # [1,H,W] or [1,K,H,W]
    out = out if img_type == "argmax" else out[0]
    return out.cpu().numpy() if isinstance(img, np.ndarray) else out


def convert_to_cog(input_path: str, output_path: str) -> None:
    raise NotImplementedError("COG conversion is GDAL file plumbing outside the hot path (SURVEY.md section 2, row 14)")
