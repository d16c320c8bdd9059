"""ORACLE -- test infrastructure only (imported by tests/ and nothing else).

CPU restatement of the reference's U-TAE Sentinel branch in evaluation mode, as a function of a state dict with the
reference's own key names: /root/reference/flair_hub/models/multitemp_model.py
  UTAE.forward :132-166, LTAE2d.forward :237-284, PositionalEncoder :287-313, MultiHeadAttention.forward :337-373,
  ScaledDotProductAttention.forward :388-403, TemporallySharedBlock.smart_forward :420-447, ConvLayer :452-497,
  DownConvBlock :526-564, UpConvBlock :568-599, Temporal_Aggregator (att_group) :603-662.
Pinned: tests/test_oracle_goldens.py compares it with tests/golden/utae_eval.npz, the outputs of the reference's own
UTAE class on the same seeded weights (tests/golden/gen_utae_goldens.py)."""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

EPS = 1e-5


def _conv_reflect(x, sd, key, k=3, p=1, s=1):
    """nn.Conv2d(kernel k, padding p, padding_mode='reflect') + bias  (:473-482)"""
    if p:
        x = F.pad(x, (p, p, p, p), mode="reflect")
    return F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=s)


def _bn_eval(x, sd, key):
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"], sd[key + ".bias"],
                        training=False, eps=EPS)


def _conv_layer(x, sd, prefix, n_layers, norm, k=3, p=1, s=1, n_groups=4):
    """ConvLayer: (conv -> norm -> ReLU) per layer (last_relu=True everywhere in UTAE)  (:452-497)"""
    for i in range(n_layers):
        x = _conv_reflect(x, sd, f"{prefix}.conv.{3 * i}", k, p, s)
        if norm == "group":
            x = F.group_norm(x, n_groups, sd[f"{prefix}.conv.{3 * i + 1}.weight"], sd[f"{prefix}.conv.{3 * i + 1}.bias"], EPS)
        else:
            x = _bn_eval(x, sd, f"{prefix}.conv.{3 * i + 1}")
        x = F.relu(x)
    return x


def _shared(fn, x, pad_value=0.0):
    """TemporallySharedBlock.smart_forward: fold (b, t), run the block on the non-padded dates only, padded dates
    (every value == pad_value) come out as pad_value  (:420-447)"""
    b, t, c, h, w = x.shape
    flat = x.reshape(b * t, c, h, w)
    pad = (flat == pad_value).all(-1).all(-1).all(-1)
    out_valid = fn(flat[~pad])
    out = torch.full((b * t,) + tuple(out_valid.shape[1:]), float(pad_value), dtype=out_valid.dtype)
    out[~pad] = out_valid
    return out.reshape(b, t, *out.shape[1:])


def positional_encoding(pos, d=16, T=1000, repeat=16):
    """PositionalEncoder  (:287-313): pos [B, T] -> [B, T, d * repeat]"""
    denom = torch.pow(torch.tensor(float(T)), 2 * torch.div(torch.arange(d).float(), 2, rounding_mode="floor") / d)
    tab = pos[:, :, None] / denom[None, None, :]
    tab = tab.clone()
    tab[:, :, 0::2] = torch.sin(tab[:, :, 0::2])
    tab[:, :, 1::2] = torch.cos(tab[:, :, 1::2])
    return torch.cat([tab] * repeat, dim=-1)


def ltae(x, pos, pad_mask, sd, pre="temporal_encoder", n_head=16, d_k=4):
    """LTAE2d.forward (eval: dropouts are identities)  (:237-284, :337-403): x [B, T, C, h, w] ->
    (out [B, C_out, h, w], attn [n_head, B, T, h, w])"""
    B, T, C, h, w = x.shape
    seq = x.permute(0, 3, 4, 1, 2).reshape(B * h * w, T, C)
    pm = pad_mask[:, None, None, :].expand(B, h, w, T).reshape(B * h * w, T)
    out = F.group_norm(seq.permute(0, 2, 1), n_head, sd[pre + ".in_norm.weight"], sd[pre + ".in_norm.bias"], EPS)
    out = F.conv1d(out, sd[pre + ".inconv.weight"], sd[pre + ".inconv.bias"]).permute(0, 2, 1)  # [N, T, 256]
    d_model = out.shape[-1]
    pe = positional_encoding(pos, d_model // n_head, 1000, n_head)  # [B, T, 256]
    out = out + pe[:, None, None].expand(B, h, w, T, d_model).reshape(B * h * w, T, d_model)
    # multi-head attention with one learnt query per head
    k = F.linear(out, sd[pre + ".attention_heads.fc1_k.weight"], sd[pre + ".attention_heads.fc1_k.bias"])
    k = k.view(-1, T, n_head, d_k).permute(2, 0, 1, 3)  # [head, N, T, d_k]
    q = sd[pre + ".attention_heads.Q"]  # [head, d_k]
    scores = torch.einsum("hd,hntd->hnt", q, k) / math.sqrt(d_k)
    scores = scores.masked_fill(pm[None], -1e3)
    attn = torch.softmax(scores, dim=-1)  # [head, N, T]
    v = out.view(-1, T, n_head, d_model // n_head).permute(2, 0, 1, 3)  # [head, N, T, 16]
    o = torch.einsum("hnt,hntc->hnc", attn, v).permute(1, 0, 2).reshape(-1, d_model)  # heads concatenated
    o = F.linear(o, sd[pre + ".mlp.0.weight"], sd[pre + ".mlp.0.bias"])
    o = F.relu(F.batch_norm(o, sd[pre + ".mlp.1.running_mean"], sd[pre + ".mlp.1.running_var"], sd[pre + ".mlp.1.weight"],
                            sd[pre + ".mlp.1.bias"], training=False, eps=EPS))
    o = F.group_norm(o, n_head, sd[pre + ".out_norm.weight"], sd[pre + ".out_norm.bias"], EPS)
    o = o.view(B, h, w, -1).permute(0, 3, 1, 2)
    attn = attn.view(n_head, B, h, w, T).permute(0, 1, 4, 2, 3)
    return o, attn


def aggregate_att_group(x, pad_mask, attn):
    """Temporal_Aggregator(mode='att_group')  (:609-628, :640-654): x [B, T, C, H, W], attn [heads, B, T, h, w]"""
    n_heads, b, t, h, w = attn.shape
    a = attn.reshape(n_heads * b, t, h, w)
    if x.shape[-2] > w:
        a = F.interpolate(a, size=x.shape[-2:], mode="bilinear", align_corners=False)
    else:
        a = F.avg_pool2d(a, kernel_size=w // x.shape[-2])
    a = a.view(n_heads, b, t, *x.shape[-2:])
    if pad_mask.any():
        a = a * (~pad_mask).float()[None, :, :, None, None]
    out = torch.stack(x.chunk(n_heads, dim=2))  # h x B x T x C/h x H x W
    out = (a[:, :, :, None] * out).sum(dim=2)
    return torch.cat([g for g in out], dim=1)


def utae_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, pos: torch.Tensor, n_stages: int = 4, k: int = 3,
                 s: int = 1, p: int = 1, n_head: int = 16, d_k: int = 4, pad_value: float = 0.0):
    """UTAE.forward with return_maps (eval)  (:132-166) -> (logits, maps, attn)"""
    if s != 1:
        raise NotImplementedError("restated for the stride-1 configuration FLAIR hard-codes (model_utils.py:55-71)")
    pad_mask = (x == pad_value).all(-1).all(-1).all(-1)  # B x T
    fmaps: List[torch.Tensor] = [_shared(lambda z: _conv_layer(z, sd, "in_conv.conv", 2, "group"), x, pad_value)]
    for i in range(n_stages - 1):
        def down(z, i=i):
            z = _conv_layer(z, sd, f"down_blocks.{i}.down", 1, "group", k, p, s)
            z = _conv_layer(z, sd, f"down_blocks.{i}.conv1", 1, "group")
            return z + _conv_layer(z, sd, f"down_blocks.{i}.conv2", 1, "group")
        fmaps.append(_shared(down, fmaps[-1], pad_value))
    out, attn = ltae(fmaps[-1], pos, pad_mask, sd, n_head=n_head, d_k=d_k)
    maps = [out]
    for i in range(n_stages - 1):
        skip = aggregate_att_group(fmaps[-(i + 2)], pad_mask, attn)
        pre = f"up_blocks.{i}"
        up = F.conv_transpose2d(out, sd[pre + ".up.0.weight"], sd[pre + ".up.0.bias"], stride=s, padding=p)
        up = F.relu(_bn_eval(up, sd, pre + ".up.1"))
        sk = F.relu(_bn_eval(F.conv2d(skip, sd[pre + ".skip_conv.0.weight"], sd[pre + ".skip_conv.0.bias"]), sd,
                             pre + ".skip_conv.1"))
        out = _conv_layer(torch.cat([up, sk], dim=1), sd, pre + ".conv1", 1, "batch")
        out = out + _conv_layer(out, sd, pre + ".conv2", 1, "batch")
        maps.append(out)
    logits = _conv_layer(out, sd, "out_conv.conv", 2, "batch")
    return logits, maps, attn
