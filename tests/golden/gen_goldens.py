#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE'S OWN CODE
(/root/reference, read-only) in the build container.

The reference has no tests and its third-party dependencies (rasterio, geopandas, shapely, pytorch_lightning,
torchmetrics, segmentation_models_pytorch, skimage) are not installed, so the modules are imported with
small stand-in modules injected into sys.modules for those imports ONLY inside this script:
  * rasterio.open -> a fake dataset with .profile/.shape/.transform/.bounds; rasterio.mask.mask -> the crop
    window of the scenario; rasterio.transform.array_bounds -> the affine formula rasterio uses
  * geopandas.GeoDataFrame -> records list; shapely.geometry.box -> bounds tuple
  * pytorch_lightning.LightningModule -> nn.Module with .log/.device; torchmetrics -> no-op metrics
  * segmentation_models_pytorch.create_model -> oracle/unet_resnet34.UnetResNet34 or oracle/swin_upernet.SwinUPerNet
    (the network itself is third-party code absent from the reference tree; only the reference's GLUE around it is
    pinned here)
Only data (inputs + the reference's outputs) is written; no reference source is copied.  The fixtures are
committed; this script is not run on the GPU box (the reference does not travel).

Usage:  python tests/golden/gen_goldens.py [fusion | sentinel | sentinel_mean | swin]
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# ---- stand-ins for the absent third-party modules ------------------------------------------------

class FakeTransform(tuple):
    """(a, b, c, d, e, f) affine, x = a*col + b*row + c, y = d*col + e*row + f"""


def array_bounds(height, width, transform):
    a, b, c, d, e, f = transform
    w, n = c, f
    e_, s = (width * a + height * b) + c, (width * d + height * e) + f
    return w, s, e_, n


class FakeBounds(tuple):
    left = property(lambda s: s[0])
    bottom = property(lambda s: s[1])
    right = property(lambda s: s[2])
    top = property(lambda s: s[3])


class FakeRaster:
    def __init__(self, left, top, res, height, width, crop=None):
        self.transform = FakeTransform((res, 0.0, left, 0.0, -res, top))
        self.shape = (height, width)
        self.height, self.width = height, width
        self.res = (res, res)
        self.profile = {"crs": "EPSG:2154"}
        self.crs = "EPSG:2154"
        self.crop = crop  # (c0, r0, c1, r1) pixel window of raster INTERSECT geozone
        self.bounds = FakeBounds(array_bounds(height, width, self.transform))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass


_current_raster = {}


def fake_open(path, *a, **k):
    return _current_raster[path]


def fake_mask(src, geometries, crop=True):
    if src.crop is None:
        raise ValueError("Input shapes do not overlap raster.")
    c0, r0, c1, r1 = src.crop
    a, b, c, d, e, f = src.transform
    out_transform = FakeTransform((a, b, (a * c0 + b * r0) + c, d, e, (d * c0 + e * r0) + f))
    return np.zeros((1, r1 - r0, c1 - c0), dtype=np.uint8), out_transform


class FakeGDF:
    def __init__(self, rows=None, crs=None, geometry=None):
        self.rows = list(rows or [])

    def __len__(self):
        return len(self.rows)


class _Rows:
    def __init__(self, rows):
        self.rows = rows

    @property
    def iloc(self):
        return self

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            return self.rows[int(i)]
        return _Rows([self.rows[int(j)] for j in i])


def install_stubs():
    rio = _mod("rasterio", open=fake_open)
    rio.mask = _mod("rasterio.mask", mask=fake_mask)
    rio.transform = _mod("rasterio.transform", array_bounds=array_bounds, rowcol=None, from_origin=None)
    rio.io = _mod("rasterio.io", DatasetReader=object)
    rio.windows = _mod("rasterio.windows", Window=lambda col_off, row_off, width, height: (col_off, row_off, width, height),
                       from_bounds=None)
    rio.features = _mod("rasterio.features", shapes=None)
    rio.enums = _mod("rasterio.enums", Resampling=types.SimpleNamespace(bilinear=1, nearest=0))
    rio.shutil = _mod("rasterio.shutil", copy=None)
    _mod("geopandas", GeoDataFrame=FakeGDF)
    sh = _mod("shapely")
    sh.geometry = _mod("shapely.geometry", box=lambda *a: tuple(a), Polygon=object, shape=None, mapping=None)
    _mod("skimage", img_as_float=None)
    _mod("tqdm", tqdm=lambda it, **k: it)

    class LightningModule(torch.nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        def log(self, *a, **k):
            pass

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, LightningDataModule=object)
    pl.utilities = _mod("pytorch_lightning.utilities")
    pl.utilities.rank_zero = _mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=lambda f: f)

    class _NoMetric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def update(self, *a, **k):
            pass

        def compute(self):
            return torch.tensor(0.0)

        def reset(self):
            pass

    tm = _mod("torchmetrics")
    tm.classification = _mod("torchmetrics.classification", MulticlassJaccardIndex=_NoMetric)
    tm.aggregation = _mod("torchmetrics.aggregation", MeanMetric=_NoMetric)

    from oracle.unet_resnet34 import UnetResNet34

    def create_model(arch, encoder_name, classes, in_channels, img_size=None, **kw):
        if arch == "upernet":
            # smp's own encoder table has no Swin: the reference's first call raises KeyError and it retries with the
            # timm-universal prefix (monotemp_model.py:67-83)
            if not encoder_name.startswith("tu-"):
                raise KeyError(encoder_name)
            from oracle.swin_upernet import SwinUPerNet
            return SwinUPerNet(encoder_name, in_channels, classes, img_size if img_size is not None else 512)
        if img_size is not None:
            raise TypeError("img_size")  # what smp 0.4.0's Unet constructor does (SURVEY.md Appendix C caveat)
        assert arch == "unet" and encoder_name.replace("tu-", "") == "resnet34", (arch, encoder_name)
        return UnetResNet34(in_channels, classes)

    _mod("segmentation_models_pytorch", create_model=create_model)
    sys.path.insert(0, REF)


def hexf(v):
    return float(v).hex()


# ---- scenarios ----------------------------------------------------------------------------------

SLICING_SCENARIOS = [
    # name, left, top, res, H, W, crop (c0,r0,c1,r1) or None = whole raster, patch, margin
    ("bdortho_5km_full", 651000.0, 6865000.0, 0.2, 25000, 25000, None, 512, 40),
    ("odd_origin_crop", 651992.36, 6860417.84, 0.2, 6173, 7311, (117, 301, 5120, 6002), 512, 40),
    ("small_single_tile", 1000.5, 2000.25, 0.2, 400, 380, None, 512, 40),
    ("exact_multiple", 0.0, 86.4 * 4, 0.2, 432 * 4, 432 * 3, None, 512, 40),
    ("coarse_res_margin0", 300000.0, 6500000.0, 0.5, 3000, 2111, (10, 0, 2100, 2950), 256, 0),
    ("res_rounding_1p5", 12345.678, 98765.432, 1.5, 1777, 901, None, 128, 16),
    ("no_overlap", 0.0, 100.0, 0.2, 500, 500, "none", 512, 40),
]


def gen_slicing(out):
    import flair_zonal_detection.slicing as ref_slicing
    for name, left, top, res, H, W, crop, patch, margin in SLICING_SCENARIOS:
        ras = FakeRaster(left, top, res, H, W, None if crop == "none" else (crop or (0, 0, W, H)))
        _current_raster["img"] = ras
        cfg = {"img_pixels_detection": patch, "margin": margin, "output_path": "/tmp", "output_name": "golden",
               "reference_modality": "AERIAL_RGBI", "reference_resolution": round(res, 5)}
        gdf = ref_slicing.generate_patches_from_reference(cfg, "img", geozone_contour_geometries=[object()])
        small = len(gdf.rows) <= 200
        tiles = []
        for r in gdf.rows:
            t = {"id": r["id"], "left": hexf(r["left"]), "bottom": hexf(r["bottom"]), "right": hexf(r["right"]),
                 "top": hexf(r["top"])}
            if small:  # shapely.geometry.box(x_min, y_max, x_max, y_min) argument order -> (x0, y0, x1, y1)
                t["box"] = [hexf(v) for v in (r["geometry"][0], r["geometry"][3], r["geometry"][2], r["geometry"][1])]
            tiles.append(t)
        zone = [hexf(gdf.rows[0][k]) for k in ("left_o", "bottom_o", "right_o", "top_o")] if gdf.rows else None
        out[name] = {"raster": {"left": left, "top": top, "res": res, "height": H, "width": W}, "zone": zone,
                     "crop": None if crop in (None, "none") else list(crop), "no_overlap": crop == "none",
                     "patch": patch, "margin": margin, "reference_resolution": round(res, 5),
                     "n_tiles": len(tiles), "tiles": tiles}
        print(f"  slicing {name}: {len(tiles)} tiles")


def gen_windows(slicing, out):
    """inference_and_write of the reference with a fake model / loader / output raster: records the windows."""
    import flair_zonal_detection.inference as ref_inf

    class Recorder:
        def __init__(self):
            self.calls = []

        def write(self, arr, band, window=None):
            self.calls.append((list(window), list(arr.shape), int(band)))

        def close(self):
            pass

    for name in ("odd_origin_crop", "small_single_tile", "res_rounding_1p5", "coarse_res_margin0"):
        sc = slicing[name]
        r = sc["raster"]
        ras = FakeRaster(r["left"], r["top"], r["res"], r["height"], r["width"])
        rows = [{"id": t["id"], "left": float.fromhex(t["left"]), "top": float.fromhex(t["top"])} for t in sc["tiles"]]
        tiles_gdf = _Rows(rows)
        patch, margin = sc["patch"], sc["margin"]
        for out_res in (sc["reference_resolution"], sc["reference_resolution"] * 2):
            idx = list(range(len(rows)))[:64]

            class Model(torch.nn.Module):
                def forward(self, inputs):
                    b = inputs["AERIAL_RGBI"].shape[0]
                    return {"T": torch.zeros(b, 3, patch, patch)}, {}

            loader = [{"AERIAL_RGBI": torch.zeros(len(idx), 1, 4, 4), "index": torch.tensor(idx)[:, None]}]
            rec = Recorder()
            cfg = {"device": "cpu", "margin": margin, "img_pixels_detection": patch, "output_type": "argmax",
                   "reference_resolution": sc["reference_resolution"], "output_px_meters": out_res}
            ref_inf.inference_and_write(Model(), loader, tiles_gdf, cfg, {"T": rec}, ras)
            out.append({"scenario": name, "out_res": hexf(out_res), "tile_indices": idx,
                        "windows": [c[0] for c in rec.calls], "shapes": [c[1] for c in rec.calls]})
            print(f"  windows {name} out_res={out_res}: {len(rec.calls)} writes")


def gen_convert(path):
    import flair_zonal_detection.postprocess as ref_pp
    g = np.random.default_rng(7)
    z = (g.standard_normal((19, 24, 20)) * 3).astype(np.float32)
    z[:, 0, 0] = 0.5  # ties -> first index
    z[3, 1, 1] = z[7, 1, 1] = 9.0
    np.savez_compressed(path, logits=z, argmax=ref_pp.convert(z, "argmax"), class_prob=ref_pp.convert(z, "class_prob"))
    try:
        ref_pp.convert(z, "logits")
        raised = False
    except ValueError:
        raised = True
    assert raised
    print("  convert: argmax / class_prob written")


def gen_glue(path_json, path_npz):
    """FLAIR_HUB_Model.forward + SegmentationTask.step + FLAIRLosses of the reference, on CPU, with the oracle
    conv stack standing in for smp; plus compute_patch_sizes / prepare_model_config outputs."""
    from flair_hub.tasks.module_setup import FLAIRLosses, build_segmentation_module
    import flair_zonal_detection.model_utils as ref_mu
    from flair_zonal_detection.inference import initialize_geometry_and_resolutions
    cfgs = _load_cfgs()

    cfg = cfgs.unet_resnet34_config(in_channels=5, precision="fp32")
    cfg["models"]["monotemp_model"]["arch"] = "resnet34-unet"
    w = FLAIRLosses(cfg).get_default_weights("AERIAL_LABEL-COSIA")
    from oracle.seeded_weights import checksum, fill_state_dict
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {"AERIAL_RGBI": 64}, stage="train")
    task.model.load_state_dict(fill_state_dict(task.model.state_dict()))
    wsum = checksum(task.model.state_dict())
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 64, 64, generator=g)
    t = torch.randint(0, 19, (2, 64, 64), generator=g)
    onehot = torch.nn.functional.one_hot(t, 19).permute(0, 3, 1, 2).float()
    task.eval()  # eval outputs first: the training step below moves the BatchNorm running statistics
    with torch.no_grad():
        logits_eval = task.model({"AERIAL_RGBI": x, "AERIAL_LABEL-COSIA": onehot})[0]["AERIAL_LABEL-COSIA"]
        pred_step = task.predict_step({"AERIAL_RGBI": x, "AERIAL_LABEL-COSIA": onehot}, 0)
    task.train()
    loss, preds, targets = task.step({"AERIAL_RGBI": x, "AERIAL_LABEL-COSIA": onehot}, training=True)
    loss.backward()
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in task.model.parameters() if p.grad is not None))
    np.savez_compressed(path_npz, x=x.numpy(), t=t.numpy().astype(np.uint8),
                        logits_eval=logits_eval.numpy(), preds_train=preds["AERIAL_LABEL-COSIA"].numpy().astype(np.uint8),
                        preds_eval=pred_step["preds_AERIAL_LABEL-COSIA"].numpy().astype(np.uint8))

    # zonal config expansion
    zonal = __import__("yaml").safe_load(open(os.path.join(REF, "configs/config_model_zonal_segmentation.yaml")))
    zonal["monotemp_arch"] = "resnet34-unet"
    zonal["model_weights"] = "/nonexistent/weights.safetensors"
    __import__("yaml").safe_dump(zonal, open(os.path.join(HERE, "zonal_config.yaml"), "w"), sort_keys=False)  # the input of this run
    _current_raster["/path/to/input_image.tif"] = FakeRaster(651992.36, 6860417.84, 0.2, 6173, 7311)
    zonal = initialize_geometry_and_resolutions(zonal)
    sizes = ref_mu.compute_patch_sizes(zonal)
    mcfg = ref_mu.prepare_model_config(zonal)
    info = {
        "loss_weights": [float(v) for v in w],
        "train_loss": hexf(loss.item()), "grad_norm": float(gn), "weights_checksum": wsum,
        "state_dict_keys": sorted(task.state_dict().keys()),
        "zonal": {"reference_resolution": zonal["reference_resolution"], "reference_modality": zonal["reference_modality"],
                  "tile_size_m": zonal["tile_size_m"], "margin_size_m": zonal["margin_size_m"],
                  "image_bounds": zonal["image_bounds"], "patch_sizes": sizes,
                  "labels": mcfg["labels"], "n_classes": len(mcfg["labels_configs"]["AERIAL_LABEL-COSIA"]["value_name"]),
                  "inputs_channels": mcfg["modalities"]["inputs_channels"], "aux_loss": mcfg["modalities"]["aux_loss"],
                  "pre_processings": mcfg["modalities"]["pre_processings"],
                  "monotemp_model": mcfg["models"]["monotemp_model"], "ckpt_model_path": mcfg["paths"]["ckpt_model_path"]},
    }
    json.dump(info, open(path_json, "w"), indent=1)
    print(f"  glue: loss {loss.item():.6f} grad-norm {float(gn):.6f}, {len(info['state_dict_keys'])} state-dict keys")


def _load_cfgs():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_cfgs", os.path.join(ROOT, "flair-for-aigle_amd", "flairhip", "configs.py"))
    cfgs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cfgs)
    return cfgs


FUSION_SAMPLED_GRADS = ["fusion_handler.conv_f.3.weight", "fusion_handler.conv_f.5.bias",
                        "encoders.DEM_ELEV.seg_model.layer2.0.conv1.weight",
                        "encoders.AERIAL_RGBI.seg_model.conv1.weight",
                        "main_decoders.ALL_LABEL-LPIS.seg_model.segmentation_head.0.weight"]


def gen_fusion(path_json, path_npz):
    """Two modalities (aerial 96x96x5 + DEM 64x64x2, so every stage is bilinearly aligned by a factor 1.5; smaller
    DEM tiles would leave 1x1 pixels x 2 samples for the last BatchNorm's batch statistics, which is numerically
    meaningless in any implementation), two tasks with task weights 1 / 0.5, an auxiliary aerial decoder: the reference's FLAIR_HUB_Model.forward (FusionHandler
    case 4) + SegmentationTask.step on CPU, oracle conv stack standing in for smp."""
    from flair_hub.tasks.module_setup import build_segmentation_module
    from oracle.seeded_weights import checksum, fill_state_dict
    cfg = _load_cfgs().fusion_unet_config(precision="fp32")
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {"AERIAL_RGBI": 96, "DEM_ELEV": 64}, stage="train")
    task.model.load_state_dict(fill_state_dict(task.model.state_dict()))
    wsum = checksum(task.model.state_dict())  # before the training step moves the BatchNorm running statistics
    g = torch.Generator().manual_seed(23)
    xa = torch.randn(2, 5, 96, 96, generator=g)
    xd = torch.randn(2, 2, 64, 64, generator=g)
    tc = torch.randint(0, 19, (2, 96, 96), generator=g)
    tl = torch.randint(0, 23, (2, 96, 96), generator=g)
    batch = {"AERIAL_RGBI": xa, "DEM_ELEV": xd,
             "AERIAL_LABEL-COSIA": torch.nn.functional.one_hot(tc, 19).permute(0, 3, 1, 2).float(),  # one-hot
             "ALL_LABEL-LPIS": tl}                                                                    # index
    task.eval()
    with torch.no_grad():
        lt, la = task.model(batch)
    task.train()
    loss, preds, _ = task.step(batch, training=True)
    loss.backward()
    named = dict(task.model.named_parameters())
    gn = torch.sqrt(sum((p.grad ** 2).sum() for p in named.values() if p.grad is not None))
    np.savez_compressed(
        path_npz, x_aerial=xa.numpy(), x_dem=xd.numpy(), t_cosia=tc.numpy().astype(np.uint8),
        t_lpis=tl.numpy().astype(np.uint8),
        logits_cosia=lt["AERIAL_LABEL-COSIA"].numpy(), logits_lpis=lt["ALL_LABEL-LPIS"][:1].numpy(),  # sample 0 only
        logits_aux_cosia=la["aux_AERIAL_RGBI_AERIAL_LABEL-COSIA"][:1].numpy(),                      # (fixture size)
        preds_train_cosia=preds["AERIAL_LABEL-COSIA"].numpy().astype(np.uint8),
        preds_train_lpis=preds["ALL_LABEL-LPIS"].numpy().astype(np.uint8),
        **{"grad__" + k: named[k].grad.numpy() for k in FUSION_SAMPLED_GRADS})
    info = {"train_loss": hexf(loss.item()), "grad_norm": float(gn), "weights_checksum": wsum,
            "logit_keys": sorted(lt.keys()), "aux_keys": sorted(la.keys()),
            "criterion_keys": sorted(task.criterion.keys()),
            "unused_parameters": sorted(k for k, p in named.items() if p.grad is None),
            "state_dict_keys": sorted(task.model.state_dict().keys())}
    json.dump(info, open(path_json, "w"), indent=1)
    print(f"  fusion: loss {loss.item():.6f} grad-norm {float(gn):.6f}, aux keys {info['aux_keys']}, "
          f"{len(info['unused_parameters'])} parameters without gradient")


def _fill_mixed(sd):
    """seeded weights: the U-TAE rules for the Sentinel encoder's tensors, the U-Net rules for everything else"""
    from oracle.seeded_weights import fill_state_dict, fill_utae_state_dict
    utae = {k: v for k, v in sd.items() if k.startswith("encoders.SENTINEL")}
    rest = {k: v for k, v in sd.items() if k not in utae}
    out = fill_state_dict(rest) if rest else {}
    out.update(fill_utae_state_dict(utae))
    # 1x1 task heads of the Sentinel-only model (nn.Conv2d weights are 4-D: fill_state_dict handles them)
    return out


def gen_sentinel(path_json, path_npz):
    """The reference's FLAIR_HUB_Model with a Sentinel-2 time-series branch, evaluation-mode forward on CPU:
      s1: SENTINEL2_TS only, two tasks (COSIA 19 + LPIS 23 classes -> U-TAE out_conv grows to 42, one 1x1 head per
          task: flair_model.py:101-104,153-166,420-424), one padded date
      s2: AERIAL_RGBI (64 x 64 x 5) + SENTINEL2_TS (10 x 10 x 10, T = 4), one task: U-TAE widths adjusted to the six
          aerial stages (:106-112,196-214), FusionHandler case 4 (:504-547)"""
    from flair_hub.tasks.module_setup import build_segmentation_module
    cfgs = _load_cfgs()
    g = torch.Generator().manual_seed(31)
    out, info = {}, {}
    # ---- s1 ----
    cfg = cfgs.fusion_unet_config(precision="fp32", aux_loss=False)
    cfg["modalities"]["inputs"] = {m: (m == "SENTINEL2_TS") for m in cfg["modalities"]["inputs"]}
    cfg["modalities"]["inputs_channels"]["SENTINEL2_TS"] = list(range(1, 11))
    cfg["modalities"]["aux_loss"] = {m: False for m in cfg["modalities"]["aux_loss"]}
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {"SENTINEL2_TS": 10}, stage="train")
    task.model.load_state_dict(_fill_mixed(task.model.state_dict()))
    xs = torch.randn(2, 5, 10, 10, 10, generator=g)
    xs[1, 4] = 0.0
    pos = torch.sort(torch.randint(0, 365, (2, 5), generator=g), dim=1).values.float()
    batch = {"SENTINEL2_TS": xs, "SENTINEL2_DATES": pos, "AERIAL_LABEL-COSIA": torch.zeros(2, 19, 40, 40),
             "ALL_LABEL-LPIS": torch.zeros(2, 40, 40, dtype=torch.long)}
    task.eval()
    with torch.no_grad():
        lt, la = task.model(batch)
    out.update(s1_x=xs.numpy(), s1_pos=pos.numpy(), s1_logits_cosia=lt["AERIAL_LABEL-COSIA"].numpy(),
               s1_logits_lpis=lt["ALL_LABEL-LPIS"].numpy())
    info["s1"] = {"logit_keys": sorted(lt.keys()), "aux_keys": sorted(la.keys()),
                  "state_dict_keys": sorted(task.model.state_dict().keys()),
                  "multitemp_model": {k: cfg["models"]["multitemp_model"][k] for k in ("encoder_widths", "decoder_widths", "out_conv")}}
    print("  sentinel s1:", {k: tuple(v.shape) for k, v in lt.items()}, info["s1"]["multitemp_model"])
    # ---- s2 ----
    cfg = cfgs.unet_resnet34_config(in_channels=5, precision="fp32")
    cfg["modalities"]["inputs"]["SENTINEL2_TS"] = True
    cfg["modalities"]["inputs_channels"]["SENTINEL2_TS"] = list(range(1, 11))
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {"AERIAL_RGBI": 64, "SENTINEL2_TS": 10}, stage="train")
    task.model.load_state_dict(_fill_mixed(task.model.state_dict()))
    xa = torch.randn(2, 5, 64, 64, generator=g)
    xs = torch.randn(2, 4, 10, 10, 10, generator=g)
    pos = torch.sort(torch.randint(0, 365, (2, 4), generator=g), dim=1).values.float()
    batch = {"AERIAL_RGBI": xa, "SENTINEL2_TS": xs, "SENTINEL2_DATES": pos,
             "AERIAL_LABEL-COSIA": torch.zeros(2, 19, 64, 64)}
    task.eval()
    with torch.no_grad():
        lt, la = task.model(batch)
    out.update(s2_x_aerial=xa.numpy(), s2_x=xs.numpy(), s2_pos=pos.numpy(), s2_logits=lt["AERIAL_LABEL-COSIA"].numpy())
    info["s2"] = {"logit_keys": sorted(lt.keys()), "aux_keys": sorted(la.keys()),
                  "state_dict_keys": sorted(task.model.state_dict().keys()),
                  "state_dict_shapes": {k: list(v.shape) for k, v in task.model.state_dict().items()
                                        if k.startswith(("encoders.SENTINEL2_TS", "fusion_handler"))},
                  "multitemp_model": {k: cfg["models"]["multitemp_model"][k] for k in ("encoder_widths", "decoder_widths", "out_conv")}}
    print("  sentinel s2:", {k: tuple(v.shape) for k, v in lt.items()}, info["s2"]["multitemp_model"])
    # ---- s2 training step: aerial U-Net + Sentinel branch fused per stage, SegmentationTask.step + backward with the two
    # nn.Dropout probabilities of the U-TAE set to 0 on the instance (deterministic step) ----
    g2 = torch.Generator().manual_seed(47)
    task.model.load_state_dict(_fill_mixed(task.model.state_dict()))
    te = task.model.encoders["SENTINEL2_TS"].temporal_encoder
    te.dropout.p = 0.0
    te.attention_heads.attention.dropout.p = 0.0
    xa = torch.randn(3, 5, 64, 64, generator=g2)
    xs = torch.randn(3, 4, 10, 10, 10, generator=g2)
    xs[2, 3] = 0.0
    pos = torch.sort(torch.randint(0, 365, (3, 4), generator=g2), dim=1).values.float()
    tc = torch.randint(0, 19, (3, 64, 64), generator=g2)
    batch = {"AERIAL_RGBI": xa, "SENTINEL2_TS": xs, "SENTINEL2_DATES": pos,
             "AERIAL_LABEL-COSIA": torch.nn.functional.one_hot(tc, 19).permute(0, 3, 1, 2).float()}
    task.train()
    loss, preds, _ = task.step(batch, training=True)
    loss.backward()
    named = dict(task.model.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None))
    sampled = ["encoders.SENTINEL2_TS.in_conv.conv.conv.0.weight", "encoders.SENTINEL2_TS.temporal_encoder.attention_heads.Q",
               "encoders.SENTINEL2_TS.temporal_encoder.inconv.weight", "encoders.SENTINEL2_TS.up_blocks.0.up.0.weight",
               "encoders.SENTINEL2_TS.down_blocks.2.conv2.conv.1.weight", "fusion_handler.conv_f.3.weight",
               "encoders.AERIAL_RGBI.seg_model.layer1.0.conv1.weight"]
    out.update(s2t_x_aerial=xa.numpy(), s2t_x=xs.numpy(), s2t_pos=pos.numpy(), s2t_target=tc.numpy().astype(np.uint8),
               **{"s2t_grad__" + k: named[k].grad.numpy() for k in sampled})
    info["s2_train"] = {"loss": hexf(loss.item()), "grad_norm": float(gn),
                        "grad_norms": {k: float(p.grad.double().norm()) for k, p in named.items() if p.grad is not None},
                        "unused_parameters": sorted(k for k, p in named.items() if p.grad is None)}
    print(f"  sentinel s2 train: loss {loss.item():.6f} grad-norm {float(gn):.6f}, "
          f"{len(info['s2_train']['unused_parameters'])} parameters without gradient")
    np.savez_compressed(path_npz, **out)
    json.dump(info, open(path_json, "w"), indent=1)


def gen_sentinel_mean(path_json, path_npz):
    """FusionHandler case 3 (flair_model.py:496-501): SEVERAL time-series branches and no aerial encoder -- the
    reference averages their class-score maps (each bilinearly resized to the label size, :390-392) and, with two tasks,
    puts one 1x1 head per task on the mean (:420-424).  SENTINEL2_TS (10 bands, T = 5, one padded date) +
    SENTINEL1-ASC_TS (2 bands, T = 3): evaluation forward, then one SegmentationTask.step + backward with the U-TAE
    dropouts set to 0 on the instances."""
    from flair_hub.tasks.module_setup import build_segmentation_module
    cfgs = _load_cfgs()
    g = torch.Generator().manual_seed(59)
    S2, S1 = "SENTINEL2_TS", "SENTINEL1-ASC_TS"
    cfg = cfgs.fusion_unet_config(precision="fp32", aux_loss=False)
    cfg["modalities"]["inputs"] = {m: False for m in cfg["modalities"]["inputs"]}
    cfg["modalities"]["inputs"][S2] = True
    cfg["modalities"]["inputs"][S1] = True
    cfg["modalities"]["inputs_channels"][S2] = list(range(1, 11))
    cfg["modalities"]["inputs_channels"][S1] = [1, 2]
    cfg["modalities"]["aux_loss"] = {m: False for m in cfg["modalities"]["aux_loss"]}
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {S2: 10, S1: 10}, stage="train")
    task.model.load_state_dict(_fill_mixed(task.model.state_dict()))
    x2 = torch.randn(2, 5, 10, 10, 10, generator=g)
    x2[1, 4] = 0.0
    p2 = torch.sort(torch.randint(0, 365, (2, 5), generator=g), dim=1).values.float()
    x1 = torch.randn(2, 3, 2, 10, 10, generator=g)
    p1 = torch.sort(torch.randint(0, 365, (2, 3), generator=g), dim=1).values.float()
    tc = torch.randint(0, 19, (2, 40, 40), generator=g)
    tl = torch.randint(0, 23, (2, 40, 40), generator=g)
    batch = {S2: x2, S2.replace("TS", "DATES"): p2, S1: x1, S1.replace("TS", "DATES"): p1,
             "AERIAL_LABEL-COSIA": torch.nn.functional.one_hot(tc, 19).permute(0, 3, 1, 2).float(), "ALL_LABEL-LPIS": tl}
    task.eval()
    with torch.no_grad():
        lt, la = task.model(batch)
    out = dict(x_s2=x2.numpy(), pos_s2=p2.numpy(), x_s1=x1.numpy(), pos_s1=p1.numpy(),
               t_cosia=tc.numpy().astype(np.uint8), t_lpis=tl.numpy().astype(np.uint8),
               logits_cosia=lt["AERIAL_LABEL-COSIA"].numpy(), logits_lpis=lt["ALL_LABEL-LPIS"].numpy())
    for m in (S2, S1):
        te = task.model.encoders[m].temporal_encoder
        te.dropout.p = 0.0
        te.attention_heads.attention.dropout.p = 0.0
    task.train()
    loss, preds, _ = task.step(batch, training=True)
    loss.backward()
    named = dict(task.model.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None))
    sampled = [f"encoders.{S2}.in_conv.conv.conv.0.weight", f"encoders.{S1}.in_conv.conv.conv.0.weight",
               f"encoders.{S1}.temporal_encoder.attention_heads.Q", "main_decoders.ALL_LABEL-LPIS.weight",
               "main_decoders.AERIAL_LABEL-COSIA.bias"]
    out.update({"grad__" + k: named[k].grad.numpy() for k in sampled})
    info = {"logit_keys": sorted(lt.keys()), "aux_keys": sorted(la.keys()),
            "state_dict_keys": sorted(task.model.state_dict().keys()),
            "multitemp_model": {k: cfg["models"]["multitemp_model"][k] for k in ("encoder_widths", "decoder_widths", "out_conv")},
            "train": {"loss": hexf(loss.item()), "grad_norm": float(gn),
                      "grad_norms": {k: float(p.grad.double().norm()) for k, p in named.items() if p.grad is not None},
                      "unused_parameters": sorted(k for k, p in named.items() if p.grad is None)}}
    np.savez_compressed(path_npz, **out)
    json.dump(info, open(path_json, "w"), indent=1)
    print("  sentinel mean:", {k: tuple(v.shape) for k, v in lt.items()}, f"train loss {loss.item():.6f} grad-norm {float(gn):.6f}, "
          f"{len(info['train']['unused_parameters'])} parameters without gradient")


SWIN_SAMPLED_GRADS = ["fusion_handler.conv_f.0.weight", "fusion_handler.conv_f.1.weight",
                      "encoders.DEM_ELEV.seg_model.model.layers_1.blocks.1.attn.relative_position_bias_table",
                      "encoders.DEM_ELEV.seg_model.model.patch_embed.proj.weight",
                      "encoders.AERIAL_RGBI.seg_model.model.layers_1.blocks.0.mlp.fc1.weight",
                      "encoders.AERIAL_RGBI.seg_model.model.layers_0.blocks.1.attn.qkv.bias",
                      "main_decoders.AERIAL_LABEL-COSIA.seg_model.decoder.fpn_stages.1.skip_conv.0.weight",
                      "main_decoders.ALL_LABEL-LPIS.seg_model.segmentation_head.0.weight"]


def gen_swin(path_json, path_npz):
    """The reference's glue around a transformer-style encoder (smp's [input, 0-channel placeholder, f4, f8, f16, f32]
    feature list): FLAIR_HUB_Model.forward in evaluation mode with two Swin-T encoders (aerial 96x96x5 + DEM 64x64x2:
    every stage aligned by a factor 1.5), FusionHandler's placeholder stripping (flair_model.py:506-545), two UPerNet
    task decoders and an auxiliary aerial decoder -- plus checkpoint.interpolate_bias_table (:33-56) on seeded tables.
    oracle/swin_upernet.py stands in for smp / timm (absent); the GLUE is the reference's own code."""
    from flair_hub.models.checkpoint import interpolate_bias_table
    from flair_hub.tasks.module_setup import build_segmentation_module
    from oracle.seeded_weights import checksum, fill_swin_state_dict
    cfg = _load_cfgs().fusion_unet_config(precision="fp32")
    cfg["models"]["monotemp_model"]["arch"] = "swin_tiny_patch4_window7_224-upernet"
    torch.manual_seed(2025)
    task = build_segmentation_module(cfg, {"AERIAL_RGBI": 96, "DEM_ELEV": 64}, stage="train")
    task.model.load_state_dict(fill_swin_state_dict(task.model.state_dict()))
    g = torch.Generator().manual_seed(29)
    xa = torch.randn(2, 5, 96, 96, generator=g)
    xd = torch.randn(2, 2, 64, 64, generator=g)
    batch = {"AERIAL_RGBI": xa, "DEM_ELEV": xd,
             "AERIAL_LABEL-COSIA": torch.zeros(2, 19, 96, 96), "ALL_LABEL-LPIS": torch.zeros(2, 96, 96, dtype=torch.long)}
    task.eval()
    with torch.no_grad():
        lt, la = task.model(batch)
    out = dict(x_aerial=xa.numpy(), x_dem=xd.numpy(), logits_cosia=lt["AERIAL_LABEL-COSIA"].numpy(),
               logits_lpis=lt["ALL_LABEL-LPIS"][:1].numpy(),
               logits_aux_cosia=la["aux_AERIAL_RGBI_AERIAL_LABEL-COSIA"][:1].numpy())
    wsum = checksum(task.model.state_dict())  # before the training step moves the BatchNorm running statistics
    # one training step of the reference's SegmentationTask (stochastic depth off: the draws of timm's DropPath are not
    # reproducible across implementations; its arithmetic is tested separately): loss, predictions, gradients
    for m in task.model.modules():
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
    gt = torch.Generator().manual_seed(31)  # its own stream: the bias tables below keep their draws from `g`
    tc = torch.randint(0, 19, (2, 96, 96), generator=gt)
    tl = torch.randint(0, 23, (2, 96, 96), generator=gt)
    tbatch = {"AERIAL_RGBI": xa, "DEM_ELEV": xd,
              "AERIAL_LABEL-COSIA": torch.nn.functional.one_hot(tc, 19).permute(0, 3, 1, 2).float(), "ALL_LABEL-LPIS": tl}
    task.train()
    loss, preds, _ = task.step(tbatch, training=True)
    loss.backward()
    named = dict(task.model.named_parameters())
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None))
    out.update(t_cosia=tc.numpy().astype(np.uint8), t_lpis=tl.numpy().astype(np.uint8),
               preds_train_cosia=preds["AERIAL_LABEL-COSIA"].numpy().astype(np.uint8),
               preds_train_lpis=preds["ALL_LABEL-LPIS"].numpy().astype(np.uint8),
               **{"grad__" + k: named[k].grad.numpy() for k in SWIN_SAMPLED_GRADS})
    for i, (n_old, n_new, heads) in enumerate([(23, 15, 4), (13, 23, 3), (13, 13, 6)]):
        src = torch.randn(n_old * n_old, heads, generator=g)
        out[f"table{i}_in"] = src.numpy()
        out[f"table{i}_out"] = interpolate_bias_table(src, torch.zeros(n_new * n_new, heads)).numpy()
    np.savez_compressed(path_npz, **out)
    info = {"weights_checksum": wsum, "logit_keys": sorted(lt.keys()),
            "aux_keys": sorted(la.keys()), "state_dict_keys": sorted(task.model.state_dict().keys()),
            "encoder_out_channels": list(task.model.encoders["AERIAL_RGBI"].seg_model.out_channels),
            "train": {"loss": hexf(loss.item()), "grad_norm": float(gn),
                      "grad_norms": {k: float(p.grad.double().norm()) for k, p in named.items() if p.grad is not None},
                      "unused_parameters": sorted(k for k, p in named.items() if p.grad is None)}}
    json.dump(info, open(path_json, "w"), indent=1)
    print("  swin:", {k: tuple(v.shape) for k, v in lt.items()}, len(info["state_dict_keys"]), "keys; train loss",
          f"{loss.item():.6f} grad-norm {float(gn):.6f}, {len(info['train']['unused_parameters'])} parameters without gradient")


def main():
    install_stubs()
    if sys.argv[1:] == ["swin"]:
        return gen_swin(os.path.join(HERE, "swin_two_mod.json"), os.path.join(HERE, "swin_two_mod.npz"))
    if sys.argv[1:] == ["sentinel_mean"]:
        return gen_sentinel_mean(os.path.join(HERE, "sentinel_mean.json"), os.path.join(HERE, "sentinel_mean.npz"))
    if sys.argv[1:] == ["sentinel"]:
        return gen_sentinel(os.path.join(HERE, "sentinel.json"), os.path.join(HERE, "sentinel.npz"))
    if sys.argv[1:] == ["fusion"]:  # regenerate only the multi-modality fixture
        return gen_fusion(os.path.join(HERE, "fusion_two_mod.json"), os.path.join(HERE, "fusion_two_mod.npz"))
    slicing = {}
    gen_slicing(slicing)
    json.dump(slicing, open(os.path.join(HERE, "slicing_grids.json"), "w"))
    windows = []
    gen_windows(slicing, windows)
    json.dump(windows, open(os.path.join(HERE, "write_windows.json"), "w"))
    gen_convert(os.path.join(HERE, "convert.npz"))
    gen_glue(os.path.join(HERE, "glue.json"), os.path.join(HERE, "glue_unet64.npz"))
    gen_fusion(os.path.join(HERE, "fusion_two_mod.json"), os.path.join(HERE, "fusion_two_mod.npz"))
    gen_sentinel(os.path.join(HERE, "sentinel.json"), os.path.join(HERE, "sentinel.npz"))


if __name__ == "__main__":
    main()
