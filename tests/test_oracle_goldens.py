"""CPU suite (no GPU): pins the oracle AND the host-side product code against fixtures produced by the
reference's own code (tests/golden/gen_goldens.py ran /root/reference with stand-ins for its absent
third-party imports).  Bit-exact for the float64 tile bookkeeping, exact for integer outputs."""
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

from helpers import MOD, ROOT, TASK

GOLD = os.path.join(ROOT, "tests", "golden")


def fh(s):
    return float.fromhex(s)


class RasterInfo:
    """attribute-only raster (no pixel array) for the slicing API"""

    def __init__(self, left, top, res, height, width):
        from flair_zonal_detection.raster import BoundingBox
        self.res = (res, res)
        self.shape = (height, width)
        self.height, self.width = height, width
        self.crs = "EPSG:2154"
        # rasterio.transform.array_bounds arithmetic
        self.bounds = BoundingBox(left, (height * -res) + top, (width * res) + left, top)


@pytest.fixture(scope="module")
def grids():
    return json.load(open(os.path.join(GOLD, "slicing_grids.json")))


def _zone_of(sc):
    r = sc["raster"]
    ras = RasterInfo(r["left"], r["top"], r["res"], r["height"], r["width"])
    if sc["zone"] is None:
        return ras, None
    return ras, tuple(fh(v) for v in sc["zone"])


def test_goldens_cover_the_edge_cases(grids):
    assert grids["bdortho_5km_full"]["n_tiles"] == 58 * 58  # 25000 px / 432 px stride, clamped last row / column
    assert grids["no_overlap"]["n_tiles"] == 0
    assert grids["small_single_tile"]["n_tiles"] == 1


@pytest.mark.parametrize("name", ["bdortho_5km_full", "odd_origin_crop", "small_single_tile", "exact_multiple",
                                  "coarse_res_margin0", "res_rounding_1p5"])
def test_oracle_slicing_is_bit_exact(grids, name):
    from oracle.tile_bookkeeping import slice_tiles
    sc = grids[name]
    ras, zone = _zone_of(sc)
    tiles = slice_tiles(zone, tuple(ras.bounds), sc["patch"], sc["margin"], sc["reference_resolution"])
    assert len(tiles) == sc["n_tiles"]
    for got, exp in zip(tiles, sc["tiles"]):
        assert got["id"] == exp["id"]
        for k in ("left", "bottom", "right", "top"):
            assert got[k].hex() == exp[k], (name, exp["id"], k)
        if "box" in exp:
            assert [v.hex() for v in got["box"]] == exp["box"]


@pytest.mark.parametrize("name", ["bdortho_5km_full", "odd_origin_crop", "small_single_tile", "exact_multiple",
                                  "coarse_res_margin0", "res_rounding_1p5", "no_overlap"])
def test_product_slicing_is_bit_exact(grids, name, lib):
    """libflairhip's ffa_slice_grid behind flair_zonal_detection.slicing.generate_patches_from_reference
    (host code: runs without a GPU)."""
    from flair_zonal_detection.slicing import generate_patches_from_reference
    sc = grids[name]
    r = sc["raster"]
    ras = RasterInfo(r["left"], r["top"], r["res"], r["height"], r["width"])
    cfg = {"img_pixels_detection": sc["patch"], "margin": sc["margin"], "output_path": "/tmp", "output_name": "golden",
           "reference_modality": MOD, "reference_resolution": sc["reference_resolution"]}
    if sc["no_overlap"]:
        zone = (r["left"] - 1000.0, r["top"] + 500.0, r["left"] - 900.0, r["top"] + 600.0)
    elif sc["crop"] is None:
        zone = None
    else:  # a geozone whose bounding box falls inside the crop window's outermost pixels
        c0, r0, c1, r1 = sc["crop"]
        res = r["res"]
        zone = (r["left"] + (c0 + 0.4) * res, r["top"] - (r1 - 0.4) * res, r["left"] + (c1 - 0.4) * res,
                r["top"] - (r0 + 0.4) * res)
    df = generate_patches_from_reference(cfg, ras, zone)
    assert len(df) == sc["n_tiles"]
    if not len(df):
        return
    assert list(df.columns)[:12] == ["id", "input_id", "output_id", "job_done", "left", "bottom", "right", "top",
                                     "left_o", "bottom_o", "right_o", "top_o"]
    assert [df["left_o"][0].hex(), df["bottom_o"][0].hex(), df["right_o"][0].hex(), df["top_o"][0].hex()] == sc["zone"]
    for i, exp in enumerate(sc["tiles"]):
        row = df.iloc[i]
        assert row["id"] == exp["id"]
        for k in ("left", "bottom", "right", "top"):
            assert float(row[k]).hex() == exp[k], (name, exp["id"], k)
        if "box" in exp:
            g = row["geometry"]
            b = g.bounds if hasattr(g, "bounds") else g
            assert [float(v).hex() for v in b] == exp["box"]


def test_write_windows_match_reference(grids, lib):
    from flairhip import ops
    from oracle.tile_bookkeeping import write_window
    cases = json.load(open(os.path.join(GOLD, "write_windows.json")))
    n = 0
    for case in cases:
        sc = grids[case["scenario"]]
        r = sc["raster"]
        ras = RasterInfo(r["left"], r["top"], r["res"], r["height"], r["width"])
        out_res = fh(case["out_res"])
        keep = sc["patch"] - 2 * sc["margin"]
        scale = sc["reference_resolution"] / out_res
        pred = keep if abs(scale - 1) < 1e-9 else int(round(keep * scale))
        written = iter(zip(case["windows"], case["shapes"]))
        for idx in case["tile_indices"]:
            t = sc["tiles"][idx]
            o = write_window(fh(t["left"]), fh(t["top"]), tuple(ras.bounds), out_res, pred, pred)
            w = ops.write_window(fh(t["left"]), fh(t["top"]), tuple(ras.bounds), out_res, pred, pred)
            assert (w.col_off, w.row_off, w.width, w.height, bool(w.skip)) == o
            if o[4]:
                continue
            win, shape = next(written)
            assert list(o[:4]) == win and shape == [o[3], o[2]]
            n += 1
    assert n > 300


def test_oracle_convert_matches_reference():
    from oracle.tile_bookkeeping import convert
    d = np.load(os.path.join(GOLD, "convert.npz"))
    assert np.array_equal(convert(d["logits"], "argmax"), d["argmax"])
    assert np.array_equal(convert(d["logits"], "class_prob"), d["class_prob"])
    with pytest.raises(ValueError):
        convert(d["logits"], "logits")


@pytest.fixture(scope="module")
def glue():
    return json.load(open(os.path.join(GOLD, "glue.json")))


def test_state_dict_keys_and_loss_weights_match_reference(glue):
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import FLAIRLosses, build_segmentation_module
    from oracle.tile_bookkeeping import flair_loss_weights
    cfg = unet_resnet34_config(in_channels=5, precision="fp32")
    task = build_segmentation_module(cfg, {MOD: 64}, "train")
    assert sorted(task.state_dict().keys()) == glue["state_dict_keys"]
    w = FLAIRLosses(cfg).get_default_weights(TASK)
    assert [float(v) for v in w] == glue["loss_weights"]
    assert list(flair_loss_weights(cfg["labels_configs"][TASK])) == glue["loss_weights"]


def test_zonal_config_expansion_matches_reference(glue):
    import yaml
    from flair_zonal_detection.inference import initialize_geometry_and_resolutions
    from flair_zonal_detection.model_utils import compute_patch_sizes, prepare_model_config
    from oracle.tile_bookkeeping import patch_size
    z = glue["zonal"]
    cfg = yaml.safe_load(open(os.path.join(GOLD, "zonal_config.yaml")))
    ras = RasterInfo(651992.36, 6860417.84, 0.2, 6173, 7311)
    cfg["modalities"][MOD]["input_img_path"] = ras
    cfg = initialize_geometry_and_resolutions(cfg)
    for k in ("reference_resolution", "reference_modality", "tile_size_m", "margin_size_m", "image_bounds"):
        assert cfg[k] == z[k], k
    assert compute_patch_sizes(cfg) == z["patch_sizes"]
    assert patch_size(512, 0.2, 0.2) == z["patch_sizes"][MOD]
    m = prepare_model_config(cfg)
    assert m["labels"] == z["labels"]
    assert len(m["labels_configs"][TASK]["value_name"]) == z["n_classes"]
    assert m["modalities"]["inputs_channels"] == z["inputs_channels"]
    assert m["modalities"]["aux_loss"] == z["aux_loss"]
    assert m["modalities"]["pre_processings"] == z["pre_processings"]
    assert m["models"]["monotemp_model"] == z["monotemp_model"]
    assert m["paths"]["ckpt_model_path"] == z["ckpt_model_path"]


def test_oracle_reproduces_reference_glue_outputs(glue):
    """oracle conv stack + seeded weights == what the reference's FLAIR_HUB_Model / SegmentationTask computed
    around the same conv stack (eval logits, predict_step argmax)."""
    import torch.nn.functional as F
    from helpers import oracle_to_product_keys
    from oracle.seeded_weights import checksum, fill_state_dict
    from oracle.unet_resnet34 import UnetResNet34
    d = np.load(os.path.join(GOLD, "glue_unet64.npz"))
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    cfg = unet_resnet34_config(in_channels=5, precision="fp32")
    task = build_segmentation_module(cfg, {MOD: 64}, "train")
    sd = fill_state_dict(task.model.state_dict())
    assert abs(checksum(sd) - glue["weights_checksum"]) <= 1e-6 * glue["weights_checksum"]
    oracle = UnetResNet34(5, 19)
    to_oracle = oracle_to_product_keys({k: k for k in oracle.state_dict()})  # product key -> oracle key
    oracle.load_state_dict({to_oracle[k]: v for k, v in sd.items() if k in to_oracle})
    x, t = torch.from_numpy(d["x"]), torch.from_numpy(d["t"]).long()
    oracle.eval()
    with torch.no_grad():
        logits = oracle(x)
    assert np.abs(logits.numpy() - d["logits_eval"]).max() <= 1e-5
    assert np.array_equal(logits.argmax(1).numpy().astype(np.uint8), d["preds_eval"])
    oracle.train()
    out = oracle(x)
    loss = F.cross_entropy(out, t, weight=torch.tensor(glue["loss_weights"]))
    assert abs(loss.item() - fh(glue["train_loss"])) <= 1e-5
    assert (out.argmax(1).numpy().astype(np.uint8) == d["preds_train"]).mean() > 0.9999


@pytest.fixture(scope="module")
def fusion():
    return json.load(open(os.path.join(GOLD, "fusion_two_mod.json")))


def fusion_batch(d, device="cpu"):
    tc = torch.from_numpy(d["t_cosia"]).long()
    return {"AERIAL_RGBI": torch.from_numpy(d["x_aerial"]).to(device), "DEM_ELEV": torch.from_numpy(d["x_dem"]).to(device),
            TASK: torch.nn.functional.one_hot(tc, 19).permute(0, 3, 1, 2).float().to(device),
            "ALL_LABEL-LPIS": torch.from_numpy(d["t_lpis"]).long().to(device)}


def test_fusion_product_surface_matches_reference(fusion):
    """two modalities + two tasks + an auxiliary decoder: the product builds the modules, criterion keys and
    state-dict keys the reference builds"""
    from flairhip.configs import fusion_unet_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    task = build_segmentation_module(fusion_unet_config(precision="fp32"), {MOD: 96, "DEM_ELEV": 64}, "train")
    assert sorted(task.model.state_dict().keys()) == fusion["state_dict_keys"]
    assert sorted(task.criterion.keys()) == fusion["criterion_keys"]
    assert sorted(task.model.aux_decoders.keys()) == ["AERIAL_RGBI__AERIAL_LABEL-COSIA", "AERIAL_RGBI__ALL_LABEL-LPIS"]


def test_oracle_reproduces_reference_fusion_outputs(fusion):
    """oracle/fusion_glue.py == the reference's FLAIR_HUB_Model (FusionHandler case 4, aux decoders) and
    SegmentationTask.step (task-weighted loss sum) run on the same seeded weights and inputs"""
    from flairhip.configs import fusion_unet_config
    from oracle.fusion_glue import FlairHubOracle, step_loss
    from oracle.seeded_weights import checksum, fill_state_dict
    d = np.load(os.path.join(GOLD, "fusion_two_mod.npz"))
    oracle = FlairHubOracle(fusion_unet_config(precision="fp32"))
    assert sorted(oracle.state_dict().keys()) == fusion["state_dict_keys"]
    oracle.load_state_dict(fill_state_dict(oracle.state_dict()))
    assert abs(checksum(oracle.state_dict()) - fusion["weights_checksum"]) <= 1e-6 * fusion["weights_checksum"]
    batch = fusion_batch(d)
    oracle.eval()
    with torch.no_grad():
        lt, la = oracle(batch)
    assert sorted(lt.keys()) == fusion["logit_keys"] and sorted(la.keys()) == fusion["aux_keys"]
    scale = np.abs(d["logits_cosia"]).max()
    assert np.abs(lt[TASK].numpy() - d["logits_cosia"]).max() <= 1e-5 * max(1.0, scale)
    assert np.abs(lt["ALL_LABEL-LPIS"][:1].numpy() - d["logits_lpis"]).max() <= 1e-5 * max(1.0, np.abs(d["logits_lpis"]).max())
    assert np.abs(la["aux_AERIAL_RGBI_" + TASK][:1].numpy() - d["logits_aux_cosia"]).max() <= 1e-5 * max(1.0, scale)
    oracle.train()
    loss, preds, _ = step_loss(oracle, batch)
    assert abs(loss.item() - fh(fusion["train_loss"])) <= 1e-5 * fh(fusion["train_loss"])
    assert (preds[TASK].numpy().astype(np.uint8) == d["preds_train_cosia"]).mean() > 0.9999
    assert (preds["ALL_LABEL-LPIS"].numpy().astype(np.uint8) == d["preds_train_lpis"]).mean() > 0.9999
    loss.backward()
    named = dict(oracle.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == fusion["unused_parameters"]
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - fusion["grad_norm"]) <= 1e-3 * fusion["grad_norm"]
    for k in [f[len("grad__"):] for f in d.files if f.startswith("grad__")]:
        ref = d["grad__" + k]
        assert np.abs(named[k].grad.numpy() - ref).max() <= 1e-3 * np.abs(ref).max(), k


def test_library_exports_every_declared_symbol(lib):
    from flairhip import lib as L
    header = open(os.path.join(ROOT, "include", "flairhip.h")).read()
    declared = set(re.findall(r"\b(ffa_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/flairhip.h but not exported"
        assert name in L.SIGNATURES, f"{name} has no ctypes signature"
    assert set(L.SIGNATURES) <= declared
    assert lib.ffa_target_arch() == b"gfx950"


def test_argument_errors_are_reported_not_thrown(lib):
    from flairhip import lib as L
    rc = lib.ffa_conv2d(0, None, None, None, None, None, 1, 8, 8, 16, 8, 8, 16, 32, 32, 3, 3, 1, 1, 1, 0, None)
    assert rc == -1 and b"null" in lib.ffa_last_error()
    assert lib.ffa_slice_grid(0.0, 0.0, 10.0, 10.0, 0.0, 0.0, 64, 32, 0.2, None, 0) == -1
    with pytest.raises(L.FlairHipError):
        L.check(rc, "conv2d")


def test_tile_batcher_equals_dataset_plus_default_collate():
    """the zonal loop's pinned-buffer batcher hands over exactly the tiles the Dataset / DataLoader path produces
    (raw uint8, device-normalise mode), including the ragged last batch and tiles hanging over the raster edge"""
    import pandas as pd
    from torch.utils.data import DataLoader
    from flair_zonal_detection.dataset import MultiModalSlicedDataset, TileBatcher
    from flair_zonal_detection.raster import ArrayRaster
    g = np.random.default_rng(5)
    ras = ArrayRaster(g.integers(0, 255, (4, 90, 120), dtype=np.uint8), 1000.0, 2000.0, 0.5)
    P = 32
    boxes = []
    for r0 in (-8, 20, 70):          # first tile hangs over the top edge, last over the bottom
        for c0 in (-5, 40, 100):
            left, top = 1000.0 + c0 * 0.5, 2000.0 - r0 * 0.5
            boxes.append((left, top - P * 0.5, left + P * 0.5, top))
    df = pd.DataFrame({"geometry": boxes, "left": [b[0] for b in boxes], "top": [b[3] for b in boxes],
                       "id": [str(i) for i in range(len(boxes))]})
    mods = {MOD: {"input_img_path": ras, "channels": [1, 2, 4],
                  "normalization": {"type": "custom", "means": [1.0, 2.0, 3.0], "stds": [4.0, 5.0, 6.0]}}}
    ds = MultiModalSlicedDataset(df, mods, {MOD: P}, "05-15", {"labels": [], "labels_configs": {}}, device_normalize=True)
    assert TileBatcher.supports(ds)
    got = [{k: v.clone() for k, v in b.items()} for b in TileBatcher(ds, 4)]  # buffers are reused two batches later
    want = list(DataLoader(ds, batch_size=4))
    assert len(got) == len(want) == 3 and got[-1][MOD].shape[0] == 1
    for a, b in zip(got, want):
        assert a[MOD].dtype == torch.uint8 and torch.equal(a[MOD], b[MOD]) and torch.equal(a["index"], b["index"])
    mean, std = ds.norm_vectors(MOD)
    assert list(mean) == [1.0, 2.0, 3.0] and list(std) == [4.0, 5.0, 6.0]
    plain = MultiModalSlicedDataset(df, mods, {MOD: P}, "05-15", {"labels": [], "labels_configs": {}})
    ref = (want[0][MOD][1].double() - torch.tensor(mean).double()[:, None, None]) / torch.tensor(std).double()[:, None, None]
    assert torch.allclose(plain[1][MOD].double(), ref, atol=1e-6)  # host normalisation path unchanged


def test_configure_optimizers_variants_mirror_the_reference():
    """tasks_module.py:344-391: AdamW / Adam / SGD and the three scheduler modes, with the reference's constants
    (OneCycleLR div_factor 1000 without momentum cycling, plateau factor 0.5 / cooldown 4 / min_lr 1e-7,
    cycle_then_plateau = warm-up cycle of int(fraction * total) steps followed by the plateau scheduler)"""
    import torch
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    sched = torch.optim.lr_scheduler

    def task_for(**hyper):
        cfg = unet_resnet34_config(in_channels=5, precision="bf16", total_steps=100)
        cfg["hyperparams"].update(hyper)
        return build_segmentation_module(cfg, {"AERIAL_RGBI": 512}, "train"), cfg

    task, cfg = task_for()  # defaults: adamw + one_cycle_lr
    out = task.configure_optimizers()
    opt, sc = out["optimizer"], out["lr_scheduler"]
    assert isinstance(opt, torch.optim.AdamW) and sc["interval"] == "step" and isinstance(sc["scheduler"], sched.OneCycleLR)
    g = opt.param_groups[0]
    assert g["weight_decay"] == 0.01 and tuple(g["betas"]) == (0.9, 0.999) and not g.get("fused")  # CPU parameters
    assert abs(g["lr"] - 5e-5 / 1000) < 1e-12 and sc["scheduler"].total_steps == 100  # starts at max_lr / div_factor
    assert len(g["params"]) == len(list(task.model.parameters()))

    task, _ = task_for(optimizer="sgd", scheduler="reduce_on_plateau", plateau_patience=7)
    out = task.configure_optimizers()
    assert isinstance(out["optimizer"], torch.optim.SGD) and out["optimizer"].param_groups[0]["lr"] == 5e-5
    rs = out["lr_scheduler"]
    assert rs["monitor"] == "val_loss" and rs["interval"] == "epoch" and isinstance(rs["scheduler"], sched.ReduceLROnPlateau)
    assert (rs["scheduler"].factor, rs["scheduler"].patience, rs["scheduler"].cooldown) == (0.5, 7, 4)
    assert rs["scheduler"].min_lrs == [1e-7]

    task, _ = task_for(optimizer="adam", scheduler="cycle_then_plateau", warmup_fraction=0.2)
    out = task.configure_optimizers()
    assert isinstance(out["optimizer"], torch.optim.Adam) and not isinstance(out["optimizer"], torch.optim.AdamW)
    assert set(out) == {"optimizer"} and task._warmup_scheduler.total_steps == 20
    assert isinstance(task._plateau_scheduler, sched.ReduceLROnPlateau) and task._plateau_scheduler.patience == 10

    task, _ = task_for(scheduler=None)
    assert isinstance(task.configure_optimizers(), torch.optim.AdamW)  # bare optimizer, as the reference returns it
    task, _ = task_for(optimizer="lamb")
    with pytest.raises(ValueError):
        task.configure_optimizers()


def test_nearest_zoom_index_equals_scipy():
    """flair_zonal_detection.inference._zoom_index against scipy.ndimage.zoom(order=0) itself (what the reference's
    resample_prediction calls, inference.py:212-226), including scipy's constant-0 last position on some sizes;
    the scales are ref_res / output_px_meters pairs a config can produce (0.2 m -> 0.25, 0.3, 0.15, 0.4, 0.5, 0.8 m)."""
    from scipy.ndimage import zoom
    from flair_zonal_detection.inference import _zoom_index
    scales = [0.5, 0.8, 2 / 3, 4 / 3, 1.25, 2, 4, 0.4, 0.25, 0.2 / 0.25, 0.2 / 0.3, 0.2 / 0.15, 0.2 / 0.5, 0.2 / 0.8, 3.0]
    hit_constant = 0
    for n in list(range(2, 700, 5)) + [226, 256, 432, 448, 512]:
        for z in scales:
            if int(round(n * z)) < 1:
                continue
            ref = zoom(np.arange(1, n + 1, dtype=np.int64), z, order=0)  # values 1..n: the constant 0 is visible
            got = np.array([0 if k < 0 else k + 1 for k in _zoom_index(n, z)])
            assert got.shape == ref.shape and np.array_equal(got, ref), (n, z)
            hit_constant += int((ref == 0).any())
    assert hit_constant > 0  # the quirk is exercised


def test_utae_oracle_matches_the_references_own_utae():
    """oracle/utae.py (functional CPU restatement) against the outputs of the reference's UTAE class itself on the
    same seeded weights: logits, every decoder feature map and the attention masks, without and with padded dates."""
    import sys
    import types
    from oracle.seeded_weights import fill_utae_state_dict
    from oracle.utae import utae_forward
    d = np.load(os.path.join(GOLD, "utae_eval.npz"))
    shapes = _utae_state_shapes()
    sd = fill_utae_state_dict({k: torch.zeros(s) for k, s in shapes.items()})
    for tag in ("a", "b"):
        x, pos = torch.tensor(d[f"{tag}_x"]), torch.tensor(d[f"{tag}_pos"])
        logits, maps, att = utae_forward(sd, x, pos)
        assert np.abs(logits.numpy() - d[f"{tag}_logits"]).max() <= 2e-5
        assert np.abs(att.numpy() - d[f"{tag}_att"]).max() <= 1e-6
        for i, m in enumerate(maps):
            assert np.abs(m.numpy() - d[f"{tag}_map{i}"]).max() <= 2e-5, i


def _utae_state_shapes(input_dim=10, enc=(64, 64, 64, 128), dec=(32, 32, 64, 128), out_conv=(32, 19), n_head=16,
                       d_model=256, d_k=4):
    """parameter / buffer names and shapes of the reference's UTAE (multitemp_model.py:73-130), spelled out so that the
    tests need no reference import"""
    s = {}

    def conv(pre, ci, co, k=3):
        s[pre + ".weight"], s[pre + ".bias"] = (co, ci, k, k), (co,)

    def gn(pre, c):
        s[pre + ".weight"], s[pre + ".bias"] = (c,), (c,)

    def bn(pre, c):
        gn(pre, c)
        s[pre + ".running_mean"], s[pre + ".running_var"], s[pre + ".num_batches_tracked"] = (c,), (c,), ()

    conv("in_conv.conv.conv.0", input_dim, enc[0]); gn("in_conv.conv.conv.1", enc[0])
    conv("in_conv.conv.conv.3", enc[0], enc[0]); gn("in_conv.conv.conv.4", enc[0])
    for i in range(len(enc) - 1):
        conv(f"down_blocks.{i}.down.conv.0", enc[i], enc[i]); gn(f"down_blocks.{i}.down.conv.1", enc[i])
        conv(f"down_blocks.{i}.conv1.conv.0", enc[i], enc[i + 1]); gn(f"down_blocks.{i}.conv1.conv.1", enc[i + 1])
        conv(f"down_blocks.{i}.conv2.conv.0", enc[i + 1], enc[i + 1]); gn(f"down_blocks.{i}.conv2.conv.1", enc[i + 1])
    for j, i in enumerate(range(len(enc) - 1, 0, -1)):
        d_in, d_out, d_skip = dec[i], dec[i - 1], enc[i - 1]
        pre = f"up_blocks.{j}"
        conv(pre + ".skip_conv.0", d_skip, d_skip, 1); bn(pre + ".skip_conv.1", d_skip)
        s[pre + ".up.0.weight"], s[pre + ".up.0.bias"] = (d_in, d_out, 3, 3), (d_out,)
        bn(pre + ".up.1", d_out)
        conv(pre + ".conv1.conv.0", d_out + d_skip, d_out); bn(pre + ".conv1.conv.1", d_out)
        conv(pre + ".conv2.conv.0", d_out, d_out); bn(pre + ".conv2.conv.1", d_out)
    t = "temporal_encoder"
    s[t + ".inconv.weight"], s[t + ".inconv.bias"] = (d_model, enc[-1], 1), (d_model,)
    s[t + ".attention_heads.Q"] = (n_head, d_k)
    s[t + ".attention_heads.fc1_k.weight"], s[t + ".attention_heads.fc1_k.bias"] = (n_head * d_k, d_model), (n_head * d_k,)
    gn(t + ".in_norm", enc[-1]); gn(t + ".out_norm", enc[-1])
    s[t + ".mlp.0.weight"], s[t + ".mlp.0.bias"] = (enc[-1], d_model), (enc[-1],)
    bn(t + ".mlp.1", enc[-1])
    conv("out_conv.conv.conv.0", dec[0], out_conv[0]); bn("out_conv.conv.conv.1", out_conv[0])
    conv("out_conv.conv.conv.3", out_conv[0], out_conv[1]); bn("out_conv.conv.conv.4", out_conv[1])
    return s


def test_pad_collate_matches_the_reference():
    """flair_hub.data.utils_data.padding.pad_collate_flair against the reference's own function
    (tests/golden/gen_padding_golden.py): ragged series, an all-empty batch, stacked tensors, listed strings"""
    from flair_hub.data.utils_data.padding import pad_collate_flair
    d = np.load(os.path.join(GOLD, "pad_collate.npz"))
    samples = []
    for i in range(4):
        samples.append({"SENTINEL2_TS": torch.tensor(d[f"in{i}_SENTINEL2_TS"]),
                        "SENTINEL2_DATES": torch.tensor(d[f"in{i}_SENTINEL2_DATES"]),
                        "AERIAL_RGBI": torch.tensor(d[f"in{i}_AERIAL_RGBI"]), "ID": f"tile{i}"})
    out = pad_collate_flair(samples, pad_value=0)
    assert out["ID"] == [f"tile{i}" for i in range(4)]
    for k in ("SENTINEL2_TS", "SENTINEL2_DATES", "AERIAL_RGBI"):
        assert np.array_equal(out[k].numpy(), d[f"out_{k}"]), k
    empty = pad_collate_flair([{"SENTINEL2_TS": torch.zeros(0), "SENTINEL2_DATES": torch.zeros(0)} for _ in range(3)])
    assert tuple(empty["SENTINEL2_TS"].shape) == tuple(d["empty_TS_shape"])


# ---- Swin-Transformer + UPerNet: what can be pinned without timm / smp ------------------------

def test_swin_upernet_oracle_parameter_count_matches_the_published_model():
    """/root/reference/README.md:413: LC-A (aerial only, swin_base_patch4_window12_384 + UPerNet) has 89.4 M parameters"""
    import torch
    from oracle.swin_upernet import SwinUPerNet, count_parameters
    with torch.device("meta"):
        n3 = count_parameters(SwinUPerNet("swin_base_patch4_window12_384", 3, 19, 512))
        n5 = count_parameters(SwinUPerNet("swin_base_patch4_window12_384", 5, 19, 512))
    assert round(n3 / 1e6, 1) == 89.4 and round(n5 / 1e6, 1) == 89.4
    o = SwinUPerNet("swin_tiny_patch4_window7_224", 5, 19, 64)
    assert o.encoder.out_channels == [5, 0, 96, 192, 384, 768]  # smp's placeholder convention (flair_model.py:302-306)
    assert tuple(o.state_dict()["encoder.model.layers_3.blocks.1.attn.relative_position_bias_table"].shape) == (9, 24)


def test_bias_table_interpolation_matches_the_reference():
    """tests/golden/swin_two_mod.npz table*: the reference's checkpoint.interpolate_bias_table (:33-56)"""
    import numpy as np
    import torch
    from flair_hub.models.checkpoint import interpolate_bias_table
    d = np.load(os.path.join(GOLD, "swin_two_mod.npz"))
    for i in range(3):
        src, ref = torch.from_numpy(d[f"table{i}_in"]), d[f"table{i}_out"]
        got = interpolate_bias_table(src, torch.zeros(ref.shape)).numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref)


def test_checkpoint_loader_resizes_swin_bias_tables(tmp_path):
    """a checkpoint trained at another window size loads: tables resized, everything else copied, `layers.N` accepted"""
    import torch
    from safetensors.torch import save_file
    from flair_hub.models.checkpoint import interpolate_bias_table, load_checkpoint
    from flairhip.swin import SwinUPerNet
    src = SwinUPerNet("swin_tiny_patch4_window7_224", 3, 19, 256)   # stage maps 64 / 32 / 16 / 8: window 7 everywhere
    dst = SwinUPerNet("swin_tiny_patch4_window7_224", 3, 19, 128)   # last stage map 4 -> window 4, table 49 x 24
    sd = {"model." + k.replace("layers_", "layers."): v.clone() for k, v in src.state_dict().items()}
    path = str(tmp_path / "ckpt.safetensors")
    save_file(sd, path)
    conf = {"paths": {"ckpt_model_path": path}, "labels": [], "labels_configs": {}}
    load_checkpoint(conf, dst)
    key = "encoder.model.layers_3.blocks.0.attn.relative_position_bias_table"
    a, b = src.state_dict()[key], dst.state_dict()[key]
    assert a.shape == (169, 24) and b.shape == (49, 24)
    assert torch.equal(b, interpolate_bias_table(a, b))
    k2 = "encoder.model.layers_1.blocks.0.attn.qkv.weight"
    assert torch.equal(src.state_dict()[k2], dst.state_dict()[k2])


def test_swin_oracle_matches_the_huggingface_implementation():
    """An independent third-party implementation of the same published architecture is installed in this image
    (transformers.SwinModel, a port of the original Microsoft code; timm itself is absent): on an input whose stage maps
    are multiples of the window (224 px, window 7 -> 56 / 28 / 14 / 7: no padding anywhere, where timm pads after the roll
    and HF before it) oracle/swin_upernet.py's encoder must reproduce its four stage outputs -- patch embedding, W-MSA /
    SW-MSA with the shifted-window mask and the relative-position-bias indexing, MLP, the patch-merging order."""
    transformers = pytest.importorskip("transformers")
    import torch
    from oracle.seeded_weights import fill_swin_state_dict
    from oracle.swin_upernet import TimmUniversalEncoder
    torch.manual_seed(0)
    enc = TimmUniversalEncoder("swin_tiny_patch4_window7_224", 5, 224, drop_path_rate=0.0).eval()
    enc.load_state_dict(fill_swin_state_dict(enc.state_dict()))
    cfg = transformers.SwinConfig(image_size=224, patch_size=4, num_channels=5, embed_dim=96, depths=[2, 2, 6, 2],
                                  num_heads=[3, 6, 12, 24], window_size=7, drop_path_rate=0.0, hidden_act="gelu",
                                  layer_norm_eps=1e-5, qkv_bias=True)
    hf = transformers.SwinModel(cfg, add_pooling_layer=False).eval()
    hf_keys = set(hf.state_dict().keys())
    split_names = ("q_proj", "k_proj", "v_proj") if any(".q_proj." in k for k in hf_keys) else None
    sd = {}
    for k, v in enc.state_dict().items():
        k = k[len("model."):]
        if k.startswith("patch_embed.proj"):
            sd[k.replace("patch_embed.proj", "embeddings.patch_embeddings.projection")] = v
        elif k.startswith("patch_embed.norm"):
            sd[k.replace("patch_embed.norm", "embeddings.norm")] = v
        else:
            m = re.match(r"layers_(\d+)\.(.*)", k)
            i, rest = int(m.group(1)), m.group(2)
            if rest.startswith("downsample."):  # timm merges at the START of stage i, HF at the END of stage i - 1
                sd[f"encoder.layers.{i - 1}.{rest}"] = v
                continue
            rest = rest.replace("norm1", "layernorm_before").replace("norm2", "layernorm_after")
            rest = rest.replace("attn.relative_position_bias_table",
                                "attention.relative_position_bias.relative_position_bias_table"
                                if split_names else "attention.self.relative_position_bias_table")
            rest = rest.replace("attn.proj", "attention.o_proj" if split_names else "attention.output.dense")
            if ".attn.qkv." in "." + rest:
                C = v.shape[0] // 3
                names = split_names or ("attention.self.query", "attention.self.key", "attention.self.value")
                for j, nm in enumerate(names):
                    tgt = rest.replace("attn.qkv", ("attention." + nm) if split_names else nm)
                    sd[f"encoder.layers.{i}.{tgt}"] = v[j * C:(j + 1) * C]
                continue
            sd[f"encoder.layers.{i}.{rest}"] = v
    missing, unexpected = hf.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("relative_position_index" in k or k.startswith("layernorm.") for k in missing), missing
    feats = {}
    for i, layer in enumerate(hf.encoder.layers):
        layer.blocks[-1].register_forward_hook(lambda mod, inp, out, i=i: feats.__setitem__(i, out[0] if isinstance(out, tuple) else out))
    x = torch.randn(1, 5, 224, 224, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        ours = enc(x)[2:]
        hf(x)
    for i, f in enumerate(ours):
        B, C, H, W = f.shape
        ref = feats[i].reshape(B, H, W, C).permute(0, 3, 1, 2)
        err = (f - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err <= 2e-5, (i, err)


@pytest.mark.parametrize("train_mode", [False, True])
def test_resnet34_encoder_oracle_matches_the_huggingface_implementation(train_mode):
    """the conv stack's oracle is restated from smp 0.4.0 / torchvision, both absent.  An independent
    implementation of the same published encoder IS installed: transformers.ResNetModel (layer_type 'basic', depths
    3-4-6-3, widths 64-128-256-512 = ResNet-34).  With the weights mapped, oracle/unet_resnet34.ResNet34Encoder must
    reproduce its stem and four stage outputs -- the 7x7 stride-2 stem, max-pool 3x3 / 2 / 1, BasicBlock wiring, where the
    stride and the 1x1 projection shortcut sit -- in evaluation mode and with training-mode BatchNorm statistics."""
    transformers = pytest.importorskip("transformers")
    import torch
    from oracle.seeded_weights import fill_state_dict
    from oracle.unet_resnet34 import ResNet34Encoder
    enc = ResNet34Encoder(5)
    enc.load_state_dict(fill_state_dict(enc.state_dict()))
    cfg = transformers.ResNetConfig(num_channels=5, embedding_size=64, hidden_sizes=[64, 128, 256, 512],
                                    depths=[3, 4, 6, 3], layer_type="basic", hidden_act="relu",
                                    downsample_in_first_stage=False)
    hf = transformers.ResNetModel(cfg)
    sd = {}
    for k, v in enc.state_dict().items():
        parts = k.split(".")
        if parts[0] in ("conv1", "bn1"):
            sd["embedder.embedder." + ("convolution." if parts[0] == "conv1" else "normalization.") + parts[-1]] = v
            continue
        stage, layer = int(parts[0][len("layer"):]) - 1, int(parts[1])
        base = f"encoder.stages.{stage}.layers.{layer}."
        if parts[2] == "downsample":
            sd[base + "shortcut." + ("convolution." if parts[3] == "0" else "normalization.") + parts[-1]] = v
        else:
            idx = int(parts[2][-1]) - 1  # conv1 / bn1 -> layer.0, conv2 / bn2 -> layer.1
            sd[base + f"layer.{idx}." + ("convolution." if parts[2].startswith("conv") else "normalization.") + parts[-1]] = v
    hf.load_state_dict(sd, strict=True)
    enc.train(train_mode)
    hf.train(train_mode)
    x = torch.randn(3, 5, 96, 64, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        ours = enc(x)                       # [x, stem @ 1/2, layer1 @ 1/4, ..., layer4 @ 1/32]
        ref = hf(x, output_hidden_states=True).hidden_states  # (after embedder incl. max-pool, stage 1..4)
    assert len(ref) == 5
    for i in range(1, 5):
        a, b = ours[i + 1], ref[i]
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, b.abs().max().item()), i
    # the stem: HF's first hidden state is after its max-pool; pooling the oracle's stride-2 feature must give the same
    pooled = torch.nn.functional.max_pool2d(ours[1], 3, 2, 1)
    assert (pooled - ref[0]).abs().max().item() <= 2e-5 * max(1.0, ref[0].abs().max().item())


def test_sentinel_patch_helpers_match_the_reference():
    """tests/golden/sentinel_utils.npz: the reference's own reshape_sentinel / filter_time_series / temporal_average"""
    import datetime
    import numpy as np
    from flair_hub.data.utils_data.sentinel import filter_time_series, reshape_sentinel, temporal_average
    d = np.load(os.path.join(GOLD, "sentinel_utils.npz"))
    assert np.array_equal(reshape_sentinel(d["reshape_in"], 10), d["reshape_out"])
    assert np.array_equal(filter_time_series(d["filter_in"]), d["filter_out"]) and d["filter_out"].any()
    assert np.array_equal(filter_time_series(d["filter2_in"]), d["filter2_out"]) and d["filter2_out"].sum() == 1
    dates = [datetime.datetime.strptime(str(v), "%Y%m%d") for v in d["avg_days"]]
    for tag, period in (("m", "monthly"), ("s", "semi-monthly")):
        a, off = temporal_average(d["avg_in"], dates, period=period, ref_date="05-15")
        assert a.shape == d[f"avg_{tag}_out"].shape and np.array_equal(off, d[f"avg_{tag}_days"])
        assert np.allclose(a, d[f"avg_{tag}_out"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("train_mode", [False, True])
def test_upernet_oracle_blocks_against_the_huggingface_upernet(train_mode):
    """Round-2 review, weak #1: the UPerNet half of oracle/swin_upernet.py is restated from memory of smp 0.4.0 and no
    installed package holds smp's decoder.  transformers ships the OTHER published UPerNet (mmsegmentation's UPerHead,
    transformers.UperNetForSemanticSegmentation); its head cannot be weight-mapped onto smp's as a whole:
      * PSP: HF's branches are `channels` wide (512) and the bottleneck over [x | 4 branches] is a 3x3 conv; smp's
        branches are in_channels / 4 wide and its out_conv over the same concat order is a 1x1 conv;
      * FPN: HF runs a 3x3 `fpn_conv` on every lateral after the top-down sums; smp has none (1x1 skip_conv, upsample
        the coarser map to the skip's size, add);
      * fusion: HF resizes every level to the finest LATERAL (stride 4 for Swin) with the config's align_corners and
        classifies there; smp resizes to input / 4 with align_corners=False, fuses with a 3x3 conv to 64 channels, and
        its SegmentationHead upsamples x4 with align_corners=True after a 1x1 classifier.
    What the two share, and what is pinned here against HF's modules with the oracle's weights mapped in:
      1. a pyramid-pooling branch = AdaptiveAvgPool2d(scale) -> 1x1 conv (no bias) -> BatchNorm2d -> ReLU, in that order
         (HF UperNetPyramidPoolingBlock == oracle PSPModule.blocks[i]), in evaluation and in training mode;
      2. the concat order of the PSP input, [x, branch(1), branch(2), branch(3), branch(6)] (HF psp_forward), with the
         branches resized bilinearly, align_corners=False, to x's size;
      3. one top-down step: lateral(skip) + bilinear(coarser -> skip's size, align_corners=False)
         (HF UperNetHead.forward's update of laterals[i - 1] == oracle FPNBlock).
    Everything else of the decoder (branch width, 1x1 out_conv, no fpn_convs, the 64-channel 3x3 fusion, the x4
    align_corners=True head) stays "parity unpinned" (DESIGN.md section 2)."""
    pytest.importorskip("transformers")
    import torch
    import torch.nn.functional as F
    from transformers.models.upernet import modeling_upernet as hf
    from oracle.swin_upernet import FPNBlock, PSPModule
    g = torch.Generator().manual_seed(11)

    def fill(mod):
        with torch.no_grad():
            for k, v in mod.state_dict().items():
                if v.dtype.is_floating_point:
                    v.copy_(torch.rand(v.shape, generator=g) + 0.5 if ("running_var" in k or k.endswith("1.weight"))
                            else torch.randn(v.shape, generator=g) * 0.3)

    cin, cout = 48, 24
    psp = PSPModule(cin, cout)
    fill(psp)
    psp.train(train_mode)
    x = torch.randn(3, cin, 12, 12, generator=g)
    branch_outs = []
    for blk, scale in zip(psp.blocks, (1, 2, 3, 6)):
        ref = hf.UperNetPyramidPoolingBlock(scale, cin, cin // 4)
        # oracle branch: Sequential(AdaptiveAvgPool2d, Sequential(conv, bn, relu)); HF: [pool, ConvModule(conv, batch_norm)]
        sd = blk.state_dict()
        ref.load_state_dict({"1.conv.weight": sd["1.0.weight"], "1.batch_norm.weight": sd["1.1.weight"],
                             "1.batch_norm.bias": sd["1.1.bias"], "1.batch_norm.running_mean": sd["1.1.running_mean"],
                             "1.batch_norm.running_var": sd["1.1.running_var"],
                             "1.batch_norm.num_batches_tracked": sd["1.1.num_batches_tracked"]}, strict=True)
        ref.train(train_mode)
        with torch.no_grad():
            a, b = blk(x), ref(x)
        assert a.shape == b.shape == (3, cin // 4, scale, scale)
        assert torch.allclose(a, b, rtol=0, atol=1e-6), scale
        branch_outs.append(b)
    # 2. the tensor the PSP's last conv sees: HF's concat order, the oracle's resize convention
    with torch.no_grad():
        cat = torch.cat([x] + [F.interpolate(b, size=x.shape[2:], mode="bilinear", align_corners=False)
                               for b in branch_outs], dim=1)
        want = psp.out_conv(cat)
        psp.train(train_mode)
        got = psp(x) if not train_mode else None  # (train mode would update the branch statistics a second time)
    if got is not None:
        assert torch.allclose(got, want, rtol=0, atol=1e-6)

    # 3. top-down step
    fpn = FPNBlock(40, cout)
    fill(fpn)
    fpn.train(train_mode)
    lat = hf.UperNetConvModule(40, cout, kernel_size=1)
    sd = fpn.skip_conv.state_dict()
    lat.load_state_dict({"conv.weight": sd["0.weight"], "batch_norm.weight": sd["1.weight"], "batch_norm.bias": sd["1.bias"],
                         "batch_norm.running_mean": sd["1.running_mean"], "batch_norm.running_var": sd["1.running_var"],
                         "batch_norm.num_batches_tracked": sd["1.num_batches_tracked"]}, strict=True)
    lat.train(train_mode)
    coarse = torch.randn(3, cout, 6, 6, generator=g)
    skip = torch.randn(3, 40, 12, 12, generator=g)
    with torch.no_grad():
        a = fpn(coarse, skip)
        b = lat(skip) + F.interpolate(coarse, size=skip.shape[2:], mode="bilinear", align_corners=False)
    assert torch.allclose(a, b, rtol=0, atol=1e-6)
