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


def _two_branch_task(cuda):
    from flairhip.configs import fusion_unet_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    S2, S1 = "SENTINEL2_TS", "SENTINEL1-ASC_TS"
    cfg = fusion_unet_config(precision="fp32", aux_loss=False)
    cfg["modalities"]["inputs"] = {m: False for m in cfg["modalities"]["inputs"]}
    cfg["modalities"]["inputs"][S2] = True
    cfg["modalities"]["inputs"][S1] = True
    cfg["modalities"]["inputs_channels"][S2] = list(range(1, 11))
    cfg["modalities"]["inputs_channels"][S1] = [1, 2]
    cfg["modalities"]["aux_loss"] = {m: False for m in cfg["modalities"]["aux_loss"]}
    task = build_segmentation_module(cfg, {S2: 10, S1: 10}, "train")
    return task, cfg


def test_several_sentinel_branches_average_their_scores_like_the_reference(cuda):
    """FusionHandler case 3 (reference flair_model.py:496-501; round-2 review "missing" #3): Sentinel-2 + Sentinel-1
    branches, no aerial encoder -> mean of the resized class-score maps -> one 1x1 head per task.
    tests/golden/sentinel_mean.{npz,json} = the reference's own FLAIR_HUB_Model / SegmentationTask (gen_goldens.py
    sentinel_mean): evaluation logits, then one training step (loss, gradient norms, sampled gradients, unused set)."""
    import torch.nn.functional as F
    d = np.load(os.path.join(GOLD, "sentinel_mean.npz"))
    info = json.load(open(os.path.join(GOLD, "sentinel_mean.json")))
    S2, S1 = "SENTINEL2_TS", "SENTINEL1-ASC_TS"
    task, cfg = _two_branch_task(cuda)
    assert sorted(task.model.state_dict().keys()) == info["state_dict_keys"]
    assert cfg["models"]["multitemp_model"]["out_conv"] == info["multitemp_model"]["out_conv"]
    task.model.load_state_dict(_fill(task.model.state_dict()))
    task = task.to(cuda).eval()
    tc, tl = torch.tensor(d["t_cosia"]).long(), torch.tensor(d["t_lpis"]).long()
    batch = {S2: torch.tensor(d["x_s2"]).to(cuda), "SENTINEL2_DATES": torch.tensor(d["pos_s2"]).to(cuda),
             S1: torch.tensor(d["x_s1"]).to(cuda), "SENTINEL1-ASC_DATES": torch.tensor(d["pos_s1"]).to(cuda),
             "AERIAL_LABEL-COSIA": F.one_hot(tc, 19).permute(0, 3, 1, 2).float().contiguous().to(cuda),
             "ALL_LABEL-LPIS": tl.to(cuda)}
    with torch.no_grad():
        lt, la = task.model(batch)
    assert sorted(lt.keys()) == info["logit_keys"] and sorted(la.keys()) == info["aux_keys"]
    for key, ref in (("AERIAL_LABEL-COSIA", d["logits_cosia"]), ("ALL_LABEL-LPIS", d["logits_lpis"])):
        got = lt[key].float().cpu().numpy()
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), key
        assert (got.argmax(1) == ref.argmax(1)).mean() >= 0.999
    # ---- training step ----
    task.train()
    for m in (S2, S1):
        task.model.encoders[m].mlp_dropout = task.model.encoders[m].attn_dropout = 0.0
    loss, preds, _ = task.step(batch, training=True)
    loss.backward()
    torch.cuda.synchronize()
    tr = info["train"]
    ref_loss = float.fromhex(tr["loss"])
    assert abs(loss.item() - ref_loss) <= 5e-5 * ref_loss
    named = dict(task.model.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == tr["unused_parameters"]
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - tr["grad_norm"]) <= 1e-2 * tr["grad_norm"]
    off = [(k, n, named[k].grad.double().norm().item()) for k, n in tr["grad_norms"].items()
           if n > 1e-6 and abs(named[k].grad.double().norm().item() - n) > 5e-2 * n]
    assert not off, off[:8]
    for k in [f[len("grad__"):] for f in d.files if f.startswith("grad__")]:
        ref = d["grad__" + k]
        rel = np.linalg.norm(named[k].grad.cpu().numpy() - ref) / np.linalg.norm(ref)
        assert rel <= 2e-2, f"{k}: relative gradient error {rel}"


def test_mean_stack_kernel(cuda):
    from flairhip import ops
    g = torch.Generator().manual_seed(4)
    for dt in (torch.float32, torch.bfloat16):
        xs = [torch.randn(3, 10, 12, 48, generator=g).to(cuda).to(dt) for _ in range(3)]
        got = ops.mean_stack(xs)
        want = (torch.stack([x.float() for x in xs]).sum(0) / 3).to(dt)
        torch.cuda.synchronize()
        assert (got.float() - want.float()).abs().max().item() <= (1e-6 if dt == torch.float32 else 2 ** -7) * 4
        assert torch.equal(ops.mean_stack(xs[:1], divisor=4).float(), (xs[0].float() / 4).to(dt).float())
    with pytest.raises(ValueError):
        ops.mean_stack(xs * 2)


@pytest.mark.parametrize("filter_clouds", [False, True])
def test_zonal_run_with_a_sentinel_time_series(cuda, tmp_path, filter_clouds):
    """flair_zonal_detection with SENTINEL2_TS enabled next to the aerial mosaic (reference dataset.py:100-104,121-169):
    the band stack is read per tile (bilinear, boundless), reshaped to [T, 10, h, w], optionally cloud-filtered per tile
    (nearest-neighbour read of the 2-band-per-date mask stack), and the day offsets ride along as SENTINEL2_DATES; the
    whole run_inference output equals the model applied by hand to the same batches"""
    import yaml
    from helpers import MOD, TASK
    from flair_zonal_detection.dataset import pad_series_collate
    from flair_zonal_detection.inference import (compute_patch_sizes, prep_config, prep_dataset, run_inference,
                                                  generate_patches_from_reference)
    from flair_zonal_detection.model_utils import build_inference_model
    from flair_zonal_detection.raster import ArrayRaster
    from flairhip import ops
    from oracle.tile_bookkeeping import write_window
    rng = np.random.default_rng(9)
    H, W, res, patch, margin, T = 300, 400, 0.2, 128, 16, 5   # 80 m x 60 m: the modalities must share their bounds
    left, top = 651992.36, 6860417.84
    aerial = ArrayRaster(rng.integers(0, 255, (3, H, W)).astype(np.uint8), left, top, res)
    # 10 m Sentinel stack (T dates x 10 bands), cloud / snow masks at 20 m
    s2 = ArrayRaster(rng.normal(0.3, 0.2, (T * 10, 6, 8)).astype(np.float32), left, top, 10.0)
    msk = np.zeros((T * 2, 3, 4), np.uint8)
    msk[2 * 1 + 1] = 100          # date 1: cloudy everywhere
    msk[2 * 3 + 1, :1] = 100      # date 3: cloudy in the northern third -> filtered for some tiles only
    mask = ArrayRaster(msk, left, top, 20.0)
    dates = tmp_path / "dates.txt"
    dates.write_text("\n".join(["20210301", "20210420", "20210610", "20210815", "20211005"]) + "\n")
    cfg = yaml.safe_load(open(os.path.join(GOLD, "zonal_config.yaml")))
    cfg.update({"output_path": str(tmp_path), "output_name": "z", "img_pixels_detection": patch, "margin": margin,
                "output_px_meters": res, "output_type": "argmax", "batch_size": 3, "num_worker": 0,
                "hardware": {"precision": "fp32"}, "hip_graph": not filter_clouds})
    cfg["modalities"]["inputs"]["SENTINEL2_TS"] = True
    cfg["modalities"][MOD].update({"input_img_path": aerial, "channels": [1, 2, 3],
                                   "normalization": {"type": "custom", "means": [105.66, 111.35, 102.18],
                                                     "stds": [52.23, 45.62, 44.30]}})
    cfg["modalities"]["SENTINEL2_TS"].update({"input_img_path": s2, "channels": list(range(1, 11)),
                                              "dates_txt": str(dates), "filter_clouds": filter_clouds,
                                              "filter_clouds_img_path": mask, "temporal_average": False})
    cfg["tasks"] = [{"name": TASK, "active": True, "class_names": {i: f"c{i}" for i in range(19)}}]
    # seeded checkpoint for the fused model (aerial U-Net + U-TAE): a placeholder file first, the configuration
    # validator wants the path to exist
    cfg["model_weights"] = str(tmp_path / "w.ckpt")
    torch.save({"state_dict": {}}, cfg["model_weights"])
    conf = prep_config(dict(cfg))
    sizes = compute_patch_sizes(conf)
    assert sizes["SENTINEL2_TS"] == 3  # 128 px x 0.2 m = 25.6 m of 10 m pixels
    from flair_zonal_detection.model_utils import prepare_model_config
    from flair_hub.models.flair_model import FLAIR_HUB_Model
    probe = FLAIR_HUB_Model(prepare_model_config(conf), sizes)
    sd = _fill(probe.state_dict())
    torch.save({"state_dict": {"model." + k: v for k, v in sd.items()}}, cfg["model_weights"])
    got = run_inference(cfg)[TASK].data

    # the same batches by hand
    conf = prep_config(dict(cfg))
    ref_img = aerial
    tiles = generate_patches_from_reference(conf, ref_img, None)
    model = build_inference_model(conf, compute_patch_sizes(conf)).to(cuda)
    ds = prep_dataset(conf, tiles, compute_patch_sizes(conf))
    item = ds[0]
    assert item["SENTINEL2_TS"].shape[1:] == (10, 3, 3) and item["SENTINEL2_DATES"].shape == item["SENTINEL2_TS"].shape[:1]
    lengths = sorted({int(ds[i]["SENTINEL2_TS"].shape[0]) for i in range(len(ds))})
    assert lengths == ([3, 4] if filter_clouds else [5])  # date 1 always dropped, date 3 for the northern tiles
    canvas = np.zeros_like(got)
    bounds = tuple(aerial.bounds)
    lefts, tops = np.asarray(tiles["left"]), np.asarray(tiles["top"])
    for s in range(0, len(ds), 3):
        batch = pad_series_collate([ds[i] for i in range(s, min(s + 3, len(ds)))])
        idx = batch.pop("index").flatten().tolist()
        inputs = {k: v.to(cuda) for k, v in batch.items()}
        for m in ds.modalities:
            if ds.delivers_raw(m):
                inputs[m + "_NORM"] = torch.tensor(np.stack(ds.norm_vectors(m)), dtype=torch.float32, device=cuda)
        with torch.no_grad():
            lt, _ = model(inputs)
        lg = lt[TASK]
        pred = ops.predict_u8(lg._ffa_nhwc, lg._ffa_classes, "argmax", crop=(margin, margin, patch - 2 * margin,
                                                                              patch - 2 * margin)).cpu().numpy()
        for j, ti in enumerate(idx):
            col, row, w, h, skip = write_window(lefts[ti], tops[ti], bounds, res, pred.shape[-2], pred.shape[-1])
            if not skip:
                canvas[:, row:row + h, col:col + w] = pred[j][:h, :w]
    assert (got == canvas).mean() >= 0.9999
    assert got.any()
