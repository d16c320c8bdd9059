#!/usr/bin/env python
"""tests/golden/pad_collate.npz: inputs and outputs of the reference's own pad_collate_flair
(/root/reference/flair_hub/data/utils_data/padding.py:48-88, imports cleanly) for ragged Sentinel series."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from flair_hub.data.utils_data.padding import pad_collate_flair  # noqa: E402

g = torch.Generator().manual_seed(3)
lens = [4, 7, 2, 5]  # (a mixed batch with ONE empty series makes the reference itself fail in torch.stack)
samples = []
for i, t in enumerate(lens):
    samples.append({
        "SENTINEL2_TS": torch.randn(t, 3, 4, 4, generator=g) if t else torch.zeros(0),
        "SENTINEL2_DATES": torch.randint(0, 365, (t,), generator=g).float() if t else torch.zeros(0),
        "AERIAL_RGBI": torch.randn(2, 8, 8, generator=g),
        "ID": f"tile{i}",
    })
out = pad_collate_flair(samples, pad_value=0)
save = {f"in{i}_{k}": v.numpy() for i, s in enumerate(samples) for k, v in s.items() if torch.is_tensor(v)}
save.update({f"out_{k}": v.numpy() for k, v in out.items() if torch.is_tensor(v)})
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pad_collate.npz"), **save)
empty = pad_collate_flair([{"SENTINEL2_TS": torch.zeros(0), "SENTINEL2_DATES": torch.zeros(0)} for _ in range(3)])
save["empty_TS_shape"] = np.array(empty["SENTINEL2_TS"].shape)
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pad_collate.npz"), **save)
print({k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in out.items()}, tuple(empty["SENTINEL2_TS"].shape))
