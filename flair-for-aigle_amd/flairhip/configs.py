"""Config dictionaries with the reference's key layout (configs/train/*.yaml merged by
flair_hub/utils/config_io.py:11-37) for the BASELINE workloads: aerial-only U-Net (ResNet-34 encoder),
19 COSIA classes, classes 15-18 weighted 0 (configs/train/config_supervision.yaml:28-34), AdamW 5e-5 /
wd 0.01 / betas (0.9, 0.999), OneCycleLR (configs/train/config_task.yaml:28-41)."""
from __future__ import annotations

from copy import deepcopy

COSIA_CLASSES = {
    0: "building", 1: "greenhouse", 2: "swimming_pool", 3: "impervious surface", 4: "pervious surface",
    5: "bare soil", 6: "water", 7: "snow", 8: "herbaceous vegetation", 9: "agricultural land", 10: "plowed land",
    11: "vineyard", 12: "deciduous", 13: "coniferous", 14: "brushwood", 15: "clear cut", 16: "ligneous",
    17: "mixed", 18: "undefined",
}

_ALL_MODS = ["AERIAL_RGBI", "AERIAL-RLT_PAN", "DEM_ELEV", "SPOT_RGBI", "SENTINEL2_TS", "SENTINEL1-ASC_TS",
             "SENTINEL1-DESC_TS"]


def unet_resnet34_config(in_channels: int = 5, precision: str = "bf16", batch_size: int = 32,
                         task: str = "AERIAL_LABEL-COSIA", total_steps: int = 1000) -> dict:
    cfg = {
        "labels": [task],
        "labels_configs": {
            task: {
                "task_weight": 1,
                "value_name": deepcopy(COSIA_CLASSES),
                "value_weights": {"default": 1, "default_exceptions": {15: 0, 16: 0, 17: 0, 18: 0},
                                  "per_modality_exceptions": {}},
            }
        },
        "modalities": {
            "inputs": {m: (m == "AERIAL_RGBI") for m in _ALL_MODS},
            "inputs_channels": {"AERIAL_RGBI": list(range(1, in_channels + 1))},
            "aux_loss": {m: False for m in _ALL_MODS},
            "aux_loss_weight": {},
            "modality_dropout": {m: 0 for m in _ALL_MODS},
            "pre_processings": {"calc_elevation": False, "calc_elevation_stack_dsm": False, "use_augmentation": False},
        },
        "models": {
            "monotemp_model": {"arch": "resnet34-unet", "new_channels_init_mode": "random"},
            # carried for config compatibility (the reference reads these keys even without a Sentinel branch,
            # flair_model.py:106-113); unused by the mono-temporal path
            "multitemp_model": {"ref_date": "05-15", "encoder_widths": [64, 64, 64, 128],
                                "decoder_widths": [32, 32, 64, 128], "out_conv": [32, 19], "str_conv_k": 3,
                                "str_conv_s": 1, "str_conv_p": 1, "agg_mode": "att_group", "encoder_norm": "group",
                                "n_head": 16, "d_model": 256, "d_k": 4, "pad_value": 0, "padding_mode": "reflect"},
        },
        "hyperparams": {
            "num_epochs": 1, "batch_size": batch_size, "seed": 2025, "learning_rate": 5e-5, "optimizer": "adamw",
            "optim_weight_decay": 0.01, "optim_betas": [0.9, 0.999], "scheduler": "one_cycle_lr",
            "warmup_fraction": 0.2, "plateau_patience": 10, "total_steps": total_steps,
        },
        "hardware": {"accelerator": "gpu", "num_nodes": 1, "gpus_per_node": 1, "strategy": "auto", "num_workers": 0,
                     "precision": precision},
        "saving": {"ckpt_monitor": "val_miou", "ckpt_monitor_mode": "max"},
        "paths": {},
    }
    return cfg


LPIS_CLASSES = {
    0: "grasses", 1: "wheat", 2: "barley", 3: "maize", 4: "other cereals", 5: "rice", 6: "flax/hemp/tobacco",
    7: "sunflower", 8: "rapeseed", 9: "other oilseed crops", 10: "soy", 11: "other protein crops",
    12: "fodder legumes", 13: "beetroots", 14: "potatoes", 15: "other arable crops", 16: "vineyard",
    17: "olive groves", 18: "fruits orchards", 19: "nut orchards", 20: "other permanent crops", 21: "mixed crops",
    22: "background",
}


def fusion_unet_config(precision: str = "bf16", batch_size: int = 8, aux_loss: bool = True,
                       lpis_weight: float = 0.5, modality_dropout: float = 0.0) -> dict:
    """Two mono-temporal modalities (aerial 5 channels + DEM elevation 2 channels) fused per stage, two tasks
    (COSIA 19 classes, LPIS 23 classes: configs/train/config_supervision.yaml:41-75) and an auxiliary aerial
    decoder -- the conv part of BASELINE.json's multi-modal configuration."""
    cfg = unet_resnet34_config(in_channels=5, precision=precision, batch_size=batch_size)
    cfg["labels"] = ["AERIAL_LABEL-COSIA", "ALL_LABEL-LPIS"]
    cfg["labels_configs"]["ALL_LABEL-LPIS"] = {
        "task_weight": lpis_weight, "label_channel_nomenclature": 1, "value_name": deepcopy(LPIS_CLASSES),
        "value_weights": {"default": 1, "default_exceptions": None, "per_modality_exceptions": {}},
    }
    mods = cfg["modalities"]
    mods["inputs"]["DEM_ELEV"] = True
    mods["aux_loss"]["AERIAL_RGBI"] = bool(aux_loss)
    mods["aux_loss_weight"] = {"AERIAL_RGBI": 1.0}
    mods["modality_dropout"]["DEM_ELEV"] = modality_dropout
    return cfg
