"""Module construction -- counterpart of the reference's flair_hub/tasks/module_setup.py
(build_segmentation_module :48-82, FLAIRLosses :119-218).

FLAIRLosses keeps the reference's weight rules (default weight for every class, ``default_exceptions``
overrides -- e.g. classes 15-18 weigh 0 in configs/train/config_supervision.yaml:28-34 -- and
``per_modality_exceptions`` for auxiliary losses); the loss objects are HipCrossEntropyLoss, the fused
softmax + weighted-NLL (+ argmax) kernel, instead of nn.CrossEntropyLoss(weight=w).
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
from torch import nn

from flairhip.nn import HipCrossEntropyLoss
from flair_hub.models.flair_model import FLAIR_HUB_Model
from flair_hub.tasks.tasks_module import SegmentationTask


def build_segmentation_module(config: Dict[str, Any], in_img_sizes: Any, stage: str = "train") -> SegmentationTask:
    assert stage in ["train", "predict"], "stage must be either 'train' or 'predict'"
    model = FLAIR_HUB_Model(config, in_img_sizes)
    if stage == "train":
        return SegmentationTask(model=model, config=config, criterion=FLAIRLosses(config).get_losses())
    return SegmentationTask(model=model, config=config)


def get_input_img_sizes(config: Dict[str, Any], dm: Any, stage: str = "fit") -> Dict[str, int]:
    """Peek one batch of the data module to learn H = W per modality (reference :86-112)."""
    assert stage in {"fit", "predict"}, f"Unsupported stage '{stage}'"
    dm.setup(stage)
    loader = dm.train_dataloader() if stage == "fit" else dm.predict_dataloader()
    first = next(iter(loader))
    return {m: first[m][0].shape[-1] for m, on in config["modalities"]["inputs"].items() if on and m in first}


class FLAIRLosses:
    def __init__(self, config: Dict[str, dict]) -> None:
        self.config = config
        self.default_weights: Dict[str, torch.Tensor] = {}
        self.losses: nn.ModuleDict = self._build_losses()

    def _build_losses(self) -> nn.ModuleDict:
        losses = nn.ModuleDict()
        mods = self.config["modalities"]
        for task in self.config["labels"]:
            cfg = self.config["labels_configs"][task]
            losses[task] = self._create_task_loss(task, cfg)
            for modality, on in mods.get("aux_loss", {}).items():
                if on and mods["inputs"].get(modality, False):
                    losses[f"aux_{modality}_{task}"] = self._create_aux_loss(task, modality, cfg)
        return losses

    def _compute_default_weights(self, task_config: Dict[str, dict]) -> torch.Tensor:
        vw = task_config["value_weights"]
        w = torch.FloatTensor([vw["default"]] * len(task_config["value_name"]))
        for cls, value in (vw.get("default_exceptions") or {}).items():
            w[cls] = value
        return w

    def _create_task_loss(self, task_name: str, task_config: Dict[str, dict]) -> nn.Module:
        w = self._compute_default_weights(task_config)
        self.default_weights[task_name] = w
        return HipCrossEntropyLoss(weight=w)

    def _create_aux_loss(self, task_name: str, modality: str, task_config: Dict[str, dict]) -> nn.Module:
        w = self.default_weights[task_name].clone()
        exceptions = (task_config["value_weights"].get("per_modality_exceptions") or {}).get(modality)
        for cls, value in (exceptions or {}).items():
            w[cls] = value
        return HipCrossEntropyLoss(weight=w)

    def get_losses(self) -> nn.ModuleDict:
        return self.losses

    def get_default_weights(self, task_name: Optional[str] = None):
        return self.default_weights if task_name is None else self.default_weights.get(task_name, None)
