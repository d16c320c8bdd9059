"""Multi-modality / multi-task path on the GPU: two mono-temporal encoders fused
per stage by FusionHandler (bilinear alignment + 1x1 mix evaluated without the channel concat), two task
decoders with task weights, an auxiliary decoder, modality dropout.

Checked against (a) tests/golden/fusion_two_mod.npz -- what the REFERENCE's own FLAIR_HUB_Model /
SegmentationTask computed on CPU for the same seeded weights and inputs -- and (b) oracle/fusion_glue.py on
other shapes.  fp32 mode: logits within 1e-4 (relative to the logit scale), loss within 2e-5 relative.
"""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import MOD, ROOT, TASK

pytestmark = pytest.mark.gpu

GOLD = os.path.join(ROOT, "tests", "golden")
LPIS, DEM = "ALL_LABEL-LPIS", "DEM_ELEV"


def _product(precision, sizes, **cfg_kw):
    from flairhip.configs import fusion_unet_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    from oracle.seeded_weights import fill_state_dict
    cfg = fusion_unet_config(precision=precision, **cfg_kw)
    task = build_segmentation_module(cfg, sizes, "train")
    task.model.load_state_dict(fill_state_dict(task.model.state_dict()))
    return task.cuda(), cfg


def _golden_batch(d):
    tc = torch.from_numpy(d["t_cosia"]).long()
    return {MOD: torch.from_numpy(d["x_aerial"]).cuda(), DEM: torch.from_numpy(d["x_dem"]).cuda(),
            TASK: F.one_hot(tc, 19).permute(0, 3, 1, 2).float().contiguous().cuda(),
            LPIS: torch.from_numpy(d["t_lpis"]).cuda()}


def _close(got, ref, tol):
    return np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max())


def test_fp32_fusion_matches_reference_golden(cuda):
    d = np.load(os.path.join(GOLD, "fusion_two_mod.npz"))
    info = json.load(open(os.path.join(GOLD, "fusion_two_mod.json")))
    task, cfg = _product("fp32", {MOD: 96, DEM: 64})
    batch = _golden_batch(d)
    task.eval()
    with torch.no_grad():
        lt, la = task.model(batch)
    assert sorted(lt.keys()) == info["logit_keys"] and sorted(la.keys()) == info["aux_keys"]
    assert _close(lt[TASK].float().cpu().numpy(), d["logits_cosia"], 1e-4)
    assert _close(lt[LPIS][:1].float().cpu().numpy(), d["logits_lpis"], 1e-4)
    assert _close(la["aux_AERIAL_RGBI_" + TASK][:1].float().cpu().numpy(), d["logits_aux_cosia"], 1e-4)

    task.train()
    loss, preds, targets = task.step(batch, training=True)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss = float.fromhex(info["train_loss"])
    assert abs(loss.item() - ref_loss) <= 2e-5 * ref_loss
    assert (preds[TASK].cpu().numpy() == d["preds_train_cosia"]).mean() > 0.9995
    assert (preds[LPIS].cpu().numpy() == d["preds_train_lpis"]).mean() > 0.9995
    named = dict(task.model.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == info["unused_parameters"]
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - info["grad_norm"]) <= 5e-3 * info["grad_norm"]
    for k in [f[len("grad__"):] for f in d.files if f.startswith("grad__")]:
        ref = d["grad__" + k]
        rel = np.linalg.norm(named[k].grad.cpu().numpy() - ref) / np.linalg.norm(ref)
        # the DEM branch ends in 2x2-pixel maps: BatchNorm over 8 samples amplifies 1e-7 differences in the batch
        # statistics (summation order) to the percent level in the gradients of the layers above it
        assert rel <= (5e-2 if "DEM_ELEV" in k else 1e-2), f"{k}: relative grad error {rel}"


def test_fp32_fusion_matches_oracle_same_size_modalities(cuda):
    """both modalities at the tile size (no bilinear alignment), rectangular tile, batch 1"""
    from oracle.fusion_glue import FlairHubOracle, step_loss
    from oracle.seeded_weights import fill_state_dict
    task, cfg = _product("fp32", {MOD: 96, DEM: 96}, aux_loss=False, lpis_weight=2.0)
    oracle = FlairHubOracle(cfg)
    oracle.load_state_dict(fill_state_dict(oracle.state_dict()))
    g = torch.Generator().manual_seed(5)
    batch = {MOD: torch.randn(1, 5, 96, 96, generator=g), DEM: torch.randn(1, 2, 96, 96, generator=g),
             TASK: torch.randint(0, 19, (1, 96, 96), generator=g), LPIS: torch.randint(0, 23, (1, 96, 96), generator=g)}
    oracle.train()
    ref_loss, ref_preds, ref_logits = step_loss(oracle, batch)
    task.train()
    loss, preds, _ = task.step({k: v.cuda() for k, v in batch.items()}, training=True)
    assert abs(loss.item() - ref_loss.item()) <= 2e-5 * abs(ref_loss.item())
    for t in (TASK, LPIS):
        assert (preds[t].cpu().long() == ref_preds[t]).float().mean().item() > 0.9995


def test_fusion_conv1x1_equals_conv_over_concat(cuda):
    from flairhip import nn as hnn
    from flairhip import ops
    g = torch.Generator().manual_seed(2)
    conv = hnn.HipConv2d(64 + 24, 40, 1, 1, 0, bias=True).cuda()
    with torch.no_grad():
        conv.bias.copy_(torch.randn(40, generator=g))
    xa = torch.randn(2, 64, 12, 20, generator=g)
    xb = torch.randn(2, 24, 12, 20, generator=g)
    wref = conv.weight.detach().cpu().clone().requires_grad_(True)
    bref = conv.bias.detach().cpu().clone().requires_grad_(True)
    xa_r, xb_r = xa.clone().requires_grad_(True), xb.clone().requires_grad_(True)
    ref = F.conv2d(torch.cat([xa_r, xb_r], 1), wref, bref)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)

    def nhwc(t, pitch):
        o = torch.zeros(t.shape[0], t.shape[2], t.shape[3], pitch)
        o[..., : t.shape[1]] = t.permute(0, 2, 3, 1)
        return o.cuda().requires_grad_(True)

    a, b = nhwc(xa, ops.pad_channels(64)), nhwc(xb, ops.pad_channels(24))
    y = hnn.fusion_conv1x1([a, b], [64, 24], conv)
    assert y.shape == (2, 12, 20, ops.pad_channels(40))
    got = y[..., :40].permute(0, 3, 1, 2).float().cpu()
    assert (got - ref.detach()).abs().max().item() <= 1e-4 * ref.abs().max().item()
    dyn = torch.zeros_like(y)
    dyn[..., :40] = dy.permute(0, 2, 3, 1).cuda()
    y.backward(dyn)
    for got_g, ref_g in ((conv.weight.grad.cpu(), wref.grad), (conv.bias.grad.cpu(), bref.grad),
                         (a.grad[..., :64].permute(0, 3, 1, 2).cpu(), xa_r.grad),
                         (b.grad[..., :24].permute(0, 3, 1, 2).cpu(), xb_r.grad)):
        assert ((got_g - ref_g).norm() / ref_g.norm()).item() <= 1e-4


def test_bf16_fusion_trains_with_modality_dropout(cuda):
    task, cfg = _product("bf16", {MOD: 64, DEM: 32}, modality_dropout=0.5)
    assert task.mod_dropout
    g = torch.Generator().manual_seed(3)
    batch = {MOD: torch.randn(4, 5, 64, 64, generator=g).cuda(), DEM: torch.randn(4, 2, 32, 32, generator=g).cuda(),
             TASK: torch.randint(0, 19, (4, 64, 64), generator=g).cuda(),
             LPIS: torch.randint(0, 23, (4, 64, 64), generator=g).cuda()}
    task.train()
    opt = torch.optim.AdamW(task.model.parameters(), lr=1e-3)
    torch.manual_seed(0)
    losses = []
    for _ in range(10):
        loss, _, _ = task.step(batch, training=True)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(l == l for l in losses), losses
    task.eval()
    with torch.no_grad():  # evaluation never drops a modality
        l0, _, _ = task.step(batch)
        l1, _, _ = task.step(batch)
    assert l0.item() == l1.item()
    assert min(losses[-3:]) < losses[0], losses


def test_modality_dropout_replaces_features_with_xavier_noise(cuda):
    task, cfg = _product("fp32", {MOD: 64, DEM: 32})
    model = task.model
    from flairhip import ops
    feats = {DEM: [torch.ones(2, 32 >> s, 32 >> s, ops.pad_channels(c), device="cuda") for s, c in
                   zip(range(6), (2, 64, 64, 128, 256, 512))]}
    out = model.modality_dropout(dict(feats), {DEM: 2.0})  # probability > 1: always dropped
    for t, c in zip(out[DEM], model.fusion_handler.stage_channels[DEM]):
        b, h, w, cp = t.shape
        bound = (6.0 / (c * h * w + b * h * w)) ** 0.5
        assert t[..., :c].abs().max().item() <= bound and t[..., :c].abs().max().item() > 0.5 * bound
        assert cp == c or float(t[..., c:].abs().max()) == 0.0
    kept = model.modality_dropout(dict(feats), {DEM: -1.0})  # probability < 0: never dropped
    assert all(a is b for a, b in zip(kept[DEM], feats[DEM]))
