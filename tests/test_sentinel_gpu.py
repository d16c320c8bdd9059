"""FLAIR_HUB_Model with a Sentinel-2 time-series (U-TAE) branch against tests/golden/sentinel.{npz,json}: outputs of
the reference's own FLAIR_HUB_Model (flair_hub/models/flair_model.py:101-134,388-430,486-547, run by
tests/golden/gen_goldens.py sentinel) in evaluation mode.
  s1: SENTINEL2_TS alone, two tasks -> U-TAE scores over 42 classes, one 1x1 head per task, one padded date
  s2: AERIAL_RGBI + SENTINEL2_TS -> U-TAE widths adjusted to the six aerial stages, per-stage 1x1 fusion"""
import json
import os

import numpy as np
import pytest
import torch

from helpers import ROOT

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")


def _fill(sd):
    from oracle.seeded_weights import fill_state_dict, fill_utae_state_dict
    utae = {k: v for k, v in sd.items() if k.startswith("encoders.SENTINEL")}
    rest = {k: v for k, v in sd.items() if k not in utae}
    out = fill_state_dict(rest) if rest else {}
    out.update(fill_utae_state_dict(utae))
    return out


def test_sentinel_only_model_matches_the_reference(cuda):
    from flairhip.configs import fusion_unet_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    d = np.load(os.path.join(GOLD, "sentinel.npz"))
    info = json.load(open(os.path.join(GOLD, "sentinel.json")))["s1"]
    cfg = fusion_unet_config(precision="fp32", aux_loss=False)
    cfg["modalities"]["inputs"] = {m: (m == "SENTINEL2_TS") for m in cfg["modalities"]["inputs"]}
    cfg["modalities"]["inputs_channels"]["SENTINEL2_TS"] = list(range(1, 11))
    cfg["modalities"]["aux_loss"] = {m: False for m in cfg["modalities"]["aux_loss"]}
    task = build_segmentation_module(cfg, {"SENTINEL2_TS": 10}, "train")
    assert cfg["models"]["multitemp_model"]["out_conv"] == info["multitemp_model"]["out_conv"] == [32, 19, 42]
    assert sorted(task.model.state_dict().keys()) == info["state_dict_keys"]
    task.model.load_state_dict(_fill(task.model.state_dict()))
    task = task.to(cuda).eval()
    batch = {"SENTINEL2_TS": torch.tensor(d["s1_x"]).to(cuda), "SENTINEL2_DATES": torch.tensor(d["s1_pos"]).to(cuda),
             "AERIAL_LABEL-COSIA": torch.zeros(2, 19, 40, 40, device=cuda),
             "ALL_LABEL-LPIS": torch.zeros(2, 40, 40, dtype=torch.long, device=cuda)}
    with torch.no_grad():
        lt, la = task.model(batch)
    assert sorted(lt.keys()) == info["logit_keys"] and sorted(la.keys()) == info["aux_keys"]
    for key, ref in (("AERIAL_LABEL-COSIA", d["s1_logits_cosia"]), ("ALL_LABEL-LPIS", d["s1_logits_lpis"])):
        got = lt[key].float().cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key
        assert (got.argmax(1) == ref.argmax(1)).mean() >= 0.999


def test_aerial_plus_sentinel_fusion_matches_the_reference(cuda):
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    d = np.load(os.path.join(GOLD, "sentinel.npz"))
    info = json.load(open(os.path.join(GOLD, "sentinel.json")))["s2"]
    cfg = unet_resnet34_config(in_channels=5, precision="fp32")
    cfg["modalities"]["inputs"]["SENTINEL2_TS"] = True
    cfg["modalities"]["inputs_channels"]["SENTINEL2_TS"] = list(range(1, 11))
    task = build_segmentation_module(cfg, {"AERIAL_RGBI": 64, "SENTINEL2_TS": 10}, "train")
    mt = cfg["models"]["multitemp_model"]
    assert mt["encoder_widths"] == info["multitemp_model"]["encoder_widths"] == [64, 64, 64, 128, 128, 128]
    sd = task.model.state_dict()
    assert sorted(sd.keys()) == info["state_dict_keys"]
    for k, shape in info["state_dict_shapes"].items():
        assert list(sd[k].shape) == shape, k
    task.model.load_state_dict(_fill(sd))
    task = task.to(cuda).eval()
    batch = {"AERIAL_RGBI": torch.tensor(d["s2_x_aerial"]).to(cuda), "SENTINEL2_TS": torch.tensor(d["s2_x"]).to(cuda),
             "SENTINEL2_DATES": torch.tensor(d["s2_pos"]).to(cuda),
             "AERIAL_LABEL-COSIA": torch.zeros(2, 19, 64, 64, device=cuda)}
    with torch.no_grad():
        lt, la = task.model(batch)
    assert sorted(lt.keys()) == info["logit_keys"] and not la
    got, ref = lt["AERIAL_LABEL-COSIA"].float().cpu().numpy(), d["s2_logits"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert (got.argmax(1) == ref.argmax(1)).mean() >= 0.999


def test_aerial_plus_sentinel_training_step_matches_the_reference(cuda):
    """sentinel.{npz,json} s2_train: the reference's own SegmentationTask.step + backward on the aerial U-Net fused per
    stage with the U-TAE branch (BatchNorm batch statistics, a padded date, U-TAE dropouts at p = 0): loss, total and
    per-parameter gradient norms, sampled gradients, the set of parameters without gradient"""
    import torch.nn.functional as F
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    d = np.load(os.path.join(GOLD, "sentinel.npz"))
    info = json.load(open(os.path.join(GOLD, "sentinel.json")))["s2_train"]
    cfg = unet_resnet34_config(in_channels=5, precision="fp32")
    cfg["modalities"]["inputs"]["SENTINEL2_TS"] = True
    cfg["modalities"]["inputs_channels"]["SENTINEL2_TS"] = list(range(1, 11))
    task = build_segmentation_module(cfg, {"AERIAL_RGBI": 64, "SENTINEL2_TS": 10}, "train")
    task.model.load_state_dict(_fill(task.model.state_dict()))
    task = task.to(cuda).train()
    utae = task.model.encoders["SENTINEL2_TS"]
    utae.mlp_dropout = utae.attn_dropout = 0.0
    tc = torch.tensor(d["s2t_target"]).long()
    batch = {"AERIAL_RGBI": torch.tensor(d["s2t_x_aerial"]).to(cuda), "SENTINEL2_TS": torch.tensor(d["s2t_x"]).to(cuda),
             "SENTINEL2_DATES": torch.tensor(d["s2t_pos"]).to(cuda),
             "AERIAL_LABEL-COSIA": F.one_hot(tc, 19).permute(0, 3, 1, 2).float().contiguous().to(cuda)}
    loss, preds, _ = task.step(batch, training=True)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss = float.fromhex(info["loss"])
    assert abs(loss.item() - ref_loss) <= 5e-5 * ref_loss
    named = dict(task.model.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == info["unused_parameters"]
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - info["grad_norm"]) <= 1e-2 * info["grad_norm"]
    off = [(k, n, named[k].grad.double().norm().item()) for k, n in info["grad_norms"].items()
           if n > 1e-6 and abs(named[k].grad.double().norm().item() - n) > 5e-2 * n]
    assert not off, off[:8]
    for k in [f[len("s2t_grad__"):] for f in d.files if f.startswith("s2t_grad__")]:
        ref = d["s2t_grad__" + k]
        rel = np.linalg.norm(named[k].grad.cpu().numpy() - ref) / np.linalg.norm(ref)
        assert rel <= 2e-2, f"{k}: relative gradient error {rel}"
