"""Model construction for zonal inference -- counterpart of the reference's
flair_zonal_detection/model_utils.py (compute_patch_sizes :19-35, prepare_model_config :38-109,
build_inference_model :112-119)."""
from __future__ import annotations

import logging
from copy import deepcopy
from typing import Any, Dict

from flair_hub.models.checkpoint import load_checkpoint
from flair_hub.models.flair_model import FLAIR_HUB_Model
from flair_zonal_detection.raster import open_raster

logger = logging.getLogger(__name__)


def get_resolution(path) -> float:
    src = open_raster(path)
    try:
        return abs(src.res[0])
    finally:
        if isinstance(path, (str, bytes)):
            src.close()


def compute_patch_sizes(config: Dict[str, Any]) -> Dict[str, int]:
    """Patch size in pixels per modality: img_pixels_detection scaled by modality / reference resolution."""
    target_res = config["reference_resolution"]
    sizes = {}
    for mod, active in config["modalities"]["inputs"].items():
        if not active:
            continue
        scale = get_resolution(config["modalities"][mod]["input_img_path"]) / target_res
        sizes[mod] = int(round(config["img_pixels_detection"] / scale))
    logger.info("patch sizes: %s", sizes)
    return sizes


def prepare_model_config(config: Dict[str, Any]) -> Dict[str, Any]:
    """Expand the zonal YAML into the model-config layout FLAIR_HUB_Model expects (same defaults as the reference)."""
    cfg = deepcopy(config)
    cfg.setdefault("models", {})
    if "monotemp_arch" in config:
        cfg["models"]["monotemp_model"] = {"arch": config["monotemp_arch"], "new_channels_init_mode": "random"}
    if "multitemp_model_ref_date" in config:
        cfg["models"]["multitemp_model"] = {
            "ref_date": config["multitemp_model_ref_date"], "encoder_widths": [64, 64, 64, 128],
            "decoder_widths": [32, 32, 64, 128], "out_conv": [32, 19], "str_conv_k": 3, "str_conv_s": 1,
            "str_conv_p": 1, "agg_mode": "att_group", "encoder_norm": "group", "n_head": 16, "d_model": 256,
            "d_k": 4, "pad_value": 0, "padding_mode": "reflect",
        }
    active = [t for t in cfg["tasks"] if t.get("active", False)]
    cfg.setdefault("labels", [t["name"] for t in active])
    cfg.setdefault("labels_configs", {t["name"]: {"value_name": list(t["class_names"].values())} for t in active})
    mods = cfg["modalities"]
    mods.setdefault("inputs_channels", {m: mods.get(m, {}).get("channels", []) for m in mods["inputs"]})
    mods.setdefault("aux_loss", {m: False for m in mods["inputs"]})
    dem = mods.get("DEM_ELEV", {})
    mods.setdefault("pre_processings", {
        "calc_elevation": dem.get("calc_elevation", False),
        "calc_elevation_stack_dsm": dem.get("calc_elevation_stack_dsm", False),
        "filter_sentinel2": False, "filter_sentinel2_max_cloud": 100, "filter_sentinel2_max_snow": 100,
        "filter_sentinel2_max_frac_cover": 1.0, "temporal_average_sentinel2": False,
        "temporal_average_sentinel1": False, "use_augmentation": False,
    })
    cfg.setdefault("paths", {})["ckpt_model_path"] = config["model_weights"]
    return cfg


def build_inference_model(config: Dict[str, Any], patch_sizes: Dict[str, int]) -> FLAIR_HUB_Model:
    model_cfg = prepare_model_config(config)
    model = FLAIR_HUB_Model(config=model_cfg, img_input_sizes=patch_sizes)
    load_checkpoint(model_cfg, model)
    return model.eval()
