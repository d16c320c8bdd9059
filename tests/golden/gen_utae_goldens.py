#!/usr/bin/env python
"""Golden fixtures for the U-TAE Sentinel branch, produced by RUNNING THE REFERENCE'S OWN UTAE
(/root/reference/flair_hub/models/multitemp_model.py imports cleanly: torch + numpy only) in the build container.

Weights come from oracle/seeded_weights.fill_utae_state_dict (a deterministic function of key and shape), so the
fixture holds only the inputs and the reference's outputs: tests/golden/utae_eval.npz.
Two scenarios with the hard-coded parameters of flair_zonal_detection/model_utils.py:55-71
(widths 64,64,64,128 / 32,32,64,128, str_conv 3/1/1, att_group, group norm, 16 heads, d_model 256, d_k 4, reflect):
  a: no padded date      b: two padded (all-zero) dates in sample 1, one in sample 0
Usage:  python tests/golden/gen_utae_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from flair_hub.models.multitemp_model import UTAE  # noqa: E402  (the reference's own class)
from oracle.seeded_weights import fill_utae_state_dict  # noqa: E402

PARAMS = dict(encoder_widths=[64, 64, 64, 128], decoder_widths=[32, 32, 64, 128], out_conv=[32, 19], str_conv_k=3,
              str_conv_s=1, str_conv_p=1, agg_mode="att_group", encoder_norm="group", n_head=16, d_model=256, d_k=4,
              encoder=False, return_maps=True, pad_value=0, padding_mode="reflect")


def main():
    torch.manual_seed(0)
    net = UTAE(input_dim=10, **PARAMS)
    net.load_state_dict(fill_utae_state_dict(net.state_dict()))
    net.eval()
    out = {}
    g = torch.Generator().manual_seed(11)
    for tag, (B, T, H, W, pads) in {"a": (2, 5, 10, 10, []), "b": (2, 6, 12, 10, [(0, 5), (1, 4), (1, 5)])}.items():
        x = torch.randn(B, T, 10, H, W, generator=g)
        pos = torch.sort(torch.randint(0, 365, (B, T), generator=g), dim=1).values.float()
        for b, t in pads:
            x[b, t] = 0.0
        with torch.no_grad():
            logits, maps = net(x, batch_positions=pos)
            net.return_maps = False
            logits2, att = net(x, batch_positions=pos, return_att=True)
            net.return_maps = True
        assert torch.equal(logits, logits2)
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_pos"] = pos.numpy()
        out[f"{tag}_logits"] = logits.numpy()
        out[f"{tag}_att"] = att.numpy()
        for i, m in enumerate(maps):
            out[f"{tag}_map{i}"] = m.numpy()
        print(tag, tuple(logits.shape), [tuple(m.shape) for m in maps], tuple(att.shape),
              float(logits.abs().max()), float(logits.std()))
    np.savez_compressed(os.path.join(HERE, "utae_eval.npz"), **out)
    print("keys:", len(net.state_dict()), "params:", sum(p.numel() for p in net.parameters()))
    gen_training(net)


def gen_training(net):
    """tests/golden/utae_train.npz: one training-mode forward + backward of the reference's UTAE (BatchNorm batch
    statistics; the two nn.Dropout probabilities set to 0 on the INSTANCE so that the step is deterministic; padded dates
    present) -- loss, class scores, and for every parameter the gradient's norm plus the whole gradient (small tensors) or
    a strided sample of it."""
    net.load_state_dict(fill_utae_state_dict(net.state_dict()))
    net.train()
    net.temporal_encoder.dropout.p = 0.0
    net.temporal_encoder.attention_heads.attention.dropout.p = 0.0
    g = torch.Generator().manual_seed(21)
    B, T, H, W = 3, 5, 10, 10
    x = torch.randn(B, T, 10, H, W, generator=g)
    for b, t in [(0, 4), (2, 3), (2, 4)]:
        x[b, t] = 0.0
    pos = torch.sort(torch.randint(0, 365, (B, T), generator=g), dim=1).values.float()
    tgt = torch.randint(0, 19, (B, H, W), generator=g)
    for p in net.parameters():
        p.grad = None
    logits, maps = net(x, batch_positions=pos)
    loss = torch.nn.functional.cross_entropy(logits, tgt)
    loss.backward()
    out = {"x": x.numpy(), "pos": pos.numpy(), "target": tgt.numpy().astype(np.uint8), "logits": logits.detach().numpy(),
           "loss": np.float64(loss.item())}
    names = []
    for k, p in net.named_parameters():
        gr = p.grad.detach().flatten()
        names.append(k)
        out["norm__" + k] = np.float64(gr.double().norm().item())
        out["grad__" + k] = (gr if gr.numel() <= 4096 else gr[:: max(1, gr.numel() // 2048)]).numpy()
    # BatchNorm running statistics after the step (momentum update of the batch statistics)
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out["stat__" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "utae_train.npz"), **out)
    print("train: loss", loss.item(), "params", len(names), "fixture keys", len(out))


if __name__ == "__main__":
    main()
