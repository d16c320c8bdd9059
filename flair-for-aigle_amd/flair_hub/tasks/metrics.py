"""Small device-side stand-ins for the two torchmetrics classes the reference's SegmentationTask
uses (flair_hub/tasks/tasks_module.py:63-93): MulticlassJaccardIndex and MeanMetric.  Metric plumbing is
outside the measured hot path; these exist so the task module keeps its logging behaviour without
torchmetrics (not installed here).  State is a K x K confusion matrix kept on the device."""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn


class MulticlassJaccardIndex(nn.Module):
    def __init__(self, num_classes: int, average: Optional[str] = "macro"):
        super().__init__()
        self.num_classes, self.average = num_classes, average
        self.register_buffer("confmat", torch.zeros(num_classes, num_classes, dtype=torch.long), persistent=False)

    @torch.no_grad()
    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        k = self.num_classes
        if (preds.is_cuda and preds.dtype == torch.uint8 and target.dtype == torch.uint8 and k <= 32
                and preds.is_contiguous() and target.is_contiguous() and self.confmat.is_cuda):
            from flairhip import ops  # block-local LDS histogram kernel: exact, no host synchronisation
            ops.confusion_matrix_update(self.confmat, preds, target)
            return
        idx = target.reshape(-1).long() * k + preds.reshape(-1).long()
        self.confmat += torch.bincount(idx, minlength=k * k).view(k, k).to(self.confmat.device)

    def compute(self) -> torch.Tensor:
        cm = self.confmat.float()
        tp = cm.diag()
        union = cm.sum(0) + cm.sum(1) - tp
        iou = tp / union  # NaN where a class is absent, as torchmetrics' average=None reports
        if self.average is None or self.average == "none":
            return iou
        iou = torch.nan_to_num(iou, nan=0.0)
        if self.average == "weighted":
            support = cm.sum(1)
            return (iou * support).sum() / support.sum().clamp(min=1)
        return iou[union > 0].mean() if (union > 0).any() else iou.sum() * 0

    def reset(self) -> None:
        self.confmat.zero_()


class MeanMetric(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("total", torch.zeros((), dtype=torch.float64), persistent=False)
        self.register_buffer("count", torch.zeros((), dtype=torch.float64), persistent=False)

    @torch.no_grad()
    def update(self, value) -> None:
        v = value.detach() if torch.is_tensor(value) else torch.tensor(float(value))
        self.total += v.to(self.total.device, torch.float64).sum()
        self.count += v.numel()

    def compute(self) -> torch.Tensor:
        return (self.total / self.count.clamp(min=1)).float()

    def reset(self) -> None:
        self.total.zero_()
        self.count.zero_()
