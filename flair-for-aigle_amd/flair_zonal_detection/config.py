"""Zonal-inference configuration -- counterpart of the reference's flair_zonal_detection/config.py
(load_config :6, validate_config :14-29, recaps :33-92).  Same required keys and the same error types."""
from __future__ import annotations

import logging
import os

import yaml

logger = logging.getLogger(__name__)

REQUIRED_KEYS = ["output_path", "output_name", "model_weights", "img_pixels_detection", "margin", "modalities",
                 "tasks", "output_px_meters"]


def load_config(path: str) -> dict:
    with open(path, "r") as f:
        return yaml.safe_load(f)


def validate_config(config: dict) -> None:
    for key in REQUIRED_KEYS:
        if key not in config:
            raise ValueError(f"Missing required config key: {key}")
    if not os.path.isfile(config["model_weights"]):
        raise FileNotFoundError(f"Model weights not found at: {config['model_weights']}")
    os.makedirs(config["output_path"], exist_ok=True)


def config_recap_1(config: dict) -> None:
    mods = ", ".join(m for m, on in config["modalities"]["inputs"].items() if on)
    tasks = ", ".join(t["name"] for t in config["tasks"] if t["active"])
    logger.info("FLAIR-HUB zone detection | output %s/%s.tif | modalities %s | tasks %s | output type %s | "
                "checkpoint %s | batch %s", config["output_path"], config["output_name"], mods, tasks,
                config.get("output_type", "argmax"), config["model_weights"], config.get("batch_size"))


def config_recap_2(config: dict) -> None:
    res = config["reference_resolution"]
    shape = config.get("image_shape_px", {})
    if shape:
        logger.info("image %s x %s px (%.2f m x %.2f m)", shape["height"], shape["width"], shape["height"] * res,
                    shape["width"] * res)
    logger.info("reference resolution %s m/px, output %s m/px, patch %s px (%.2f m), margin %s px (%.2f m)", res,
                config["output_px_meters"], config["img_pixels_detection"], config["img_pixels_detection"] * res,
                config["margin"], config["margin"] * res)
