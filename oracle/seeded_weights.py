"""ORACLE -- test infrastructure only.

Deterministic, non-trivial weights for parity tests: every tensor of a state dict is filled from a
torch.Generator seeded per key (order independent), so the reference-driven golden run
(tests/golden/gen_goldens.py), the CPU oracle and the HIP product all hold identical parameters without
a 98 MB checkpoint in the repository.  A checksum of the result is stored beside the goldens."""
from __future__ import annotations

import zlib

import torch


def fill_state_dict(state_dict, seed: int = 1234):
    out = {}
    for key in sorted(state_dict.keys()):
        ref = state_dict[key]
        g = torch.Generator().manual_seed(seed + (zlib.crc32(key.encode()) & 0x7FFFFFF))
        if key.endswith("num_batches_tracked"):
            v = torch.zeros_like(ref)
        elif ref.ndim == 4:
            fan_in = ref.shape[1] * ref.shape[2] * ref.shape[3]
            v = torch.randn(ref.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif key.endswith("running_var"):
            v = torch.rand(ref.shape, generator=g) * 0.5 + 0.75
        elif key.endswith("running_mean"):
            v = torch.randn(ref.shape, generator=g) * 0.1
        elif key.endswith(".weight"):  # BatchNorm gamma / criterion weights are 1-D '.weight'
            v = torch.rand(ref.shape, generator=g) * 0.5 + 0.75
        else:  # biases
            v = torch.randn(ref.shape, generator=g) * 0.1
        out[key] = v.to(ref.dtype)
    return out


def checksum(state_dict) -> float:
    return float(sum(v.double().abs().sum() for k, v in sorted(state_dict.items()) if v.is_floating_point()))


def fill_utae_state_dict(state_dict, seed: int = 4321):
    """Same idea for the U-TAE branch (flair_hub/models/multitemp_model.py): its Conv1d / Linear weights are 3-D / 2-D,
    GroupNorm / BatchNorm affine weights 1-D, the attention's master query 'Q' 2-D without a '.weight' suffix."""
    out = {}
    for key in sorted(state_dict.keys()):
        ref = state_dict[key]
        g = torch.Generator().manual_seed(seed + (zlib.crc32(key.encode()) & 0x7FFFFFF))
        if key.endswith("num_batches_tracked"):
            v = torch.zeros_like(ref)
        elif key.endswith("running_var"):
            v = torch.rand(ref.shape, generator=g) * 0.5 + 0.75
        elif key.endswith("running_mean"):
            v = torch.randn(ref.shape, generator=g) * 0.1
        elif key.endswith(".Q"):
            v = torch.randn(ref.shape, generator=g) * 0.7
        elif key.endswith(".weight") and ref.ndim >= 2:
            fan_in = ref[0].numel()
            if ".up.0." in key:  # ConvTranspose2d weight is [in, out, k, k]
                fan_in = ref.shape[0] * ref.shape[2] * ref.shape[3]
            v = torch.randn(ref.shape, generator=g) * (2.0 / fan_in) ** 0.5
        elif key.endswith(".weight"):  # norm gammas
            v = torch.rand(ref.shape, generator=g) * 0.5 + 0.75
        else:  # biases, norm betas
            v = torch.randn(ref.shape, generator=g) * 0.1
        out[key] = v.to(ref.dtype)
    return out


def fill_swin_state_dict(state_dict, seed: int = 2468):
    """Swin-Transformer + UPerNet models: LayerNorm / BatchNorm affine weights 1-D, nn.Linear weights 2-D, convolutions
    4-D, `relative_position_bias_table` 2-D without a '.weight' suffix.  Everything else as fill_state_dict."""
    out = {}
    for key in sorted(state_dict.keys()):
        ref = state_dict[key]
        g = torch.Generator().manual_seed(seed + (zlib.crc32(key.encode()) & 0x7FFFFFF))
        if key.endswith("num_batches_tracked"):
            v = torch.zeros_like(ref)
        elif key.endswith("relative_position_bias_table"):
            v = torch.randn(ref.shape, generator=g) * 0.5
        elif key.endswith("running_var"):
            v = torch.rand(ref.shape, generator=g) * 0.5 + 0.75
        elif key.endswith("running_mean"):
            v = torch.randn(ref.shape, generator=g) * 0.1
        elif key.endswith(".weight") and ref.ndim >= 2:
            v = torch.randn(ref.shape, generator=g) * (1.0 / ref[0].numel()) ** 0.5
        elif key.endswith(".weight"):
            v = torch.rand(ref.shape, generator=g) * 0.5 + 0.75
        else:
            v = torch.randn(ref.shape, generator=g) * 0.1
        out[key] = v.to(ref.dtype)
    return out
