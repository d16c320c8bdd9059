"""Batch collation for samples with Sentinel time series of different lengths -- counterpart of the reference's
flair_hub/data/utils_data/padding.py (pad_tensor :34-45, pad_collate_flair :48-88).

The Sentinel fields of a batch ('<SENSOR>_TS' [T, C, H, W] and '<SENSOR>_DATES' [T]) are padded along the time axis
to the longest series of the batch with ``pad_value``; U-TAE recognises the padded dates by their all-``pad_value``
images (flairhip/utae.py, ffa_detect_pad_images).  Every other tensor is stacked, everything else is listed."""
from __future__ import annotations

from typing import Dict, List

import torch

TO_PAD_KEYS = ("SENTINEL2_TS", "SENTINEL2_DATES", "SENTINEL1-ASC_TS", "SENTINEL1-ASC_DATES", "SENTINEL1-DESC_TS",
               "SENTINEL1-DESC_DATES")


def pad_tensor(x: torch.Tensor, l: int, pad_value=0) -> torch.Tensor:
    """``x`` extended along its first axis to length ``l`` with ``pad_value``"""
    missing = l - x.shape[0]
    if missing <= 0:
        return x
    tail = torch.full((missing,) + tuple(x.shape[1:]), pad_value, dtype=x.dtype, device=x.device)
    return torch.cat([x, tail], dim=0)


def pad_collate_flair(sample_dict: List[Dict], pad_value=0) -> Dict:
    batch = {}
    for key in sample_dict[0].keys():
        values = [s[key] for s in sample_dict]
        if key in TO_PAD_KEYS:
            if all(len(v) == 0 for v in values):
                batch[key] = torch.empty((len(values), 0))
                continue
            longest = max(v.shape[0] for v in values if len(v) > 0)
            batch[key] = torch.stack([pad_tensor(v, longest, pad_value) if len(v) > 0
                                      else torch.zeros((longest,), dtype=v.dtype) for v in values], dim=0)
        elif isinstance(values[0], torch.Tensor):
            batch[key] = torch.stack(values, dim=0)
        else:
            batch[key] = values
    return batch
