"""SegmentationTask -- counterpart of the reference's LightningModule
(flair_hub/tasks/tasks_module.py:11; step :133-167, predict_step :337-342, configure_optimizers :344-391).

pytorch_lightning is not a dependency here: the class is a plain nn.Module that exposes the same hook
methods (forward / step / training_step / validation_step / predict_step / configure_optimizers and the
epoch hooks); flair_hub.tasks.trainers.HipTrainer drives them the way Lightning's Trainer would.

What changes on the hot path (same results, fewer passes over B x K x H x W):
  * loss, class-weight sum and argmax(softmax(logits)) come from ONE kernel pass over the logits
    (reference: cross_entropy + softmax + argmax = three passes, tasks_module.py:155,158)
  * one-hot targets are reduced to uint8 indices by a kernel (reference: torch.argmax, :153); index
    targets are accepted directly
  * the NaN/Inf check of :156/:204 costs a host sync per step in the reference; here the flag stays on the
    device and is inspected at epoch end unless hyperparams.check_loss_every_step is set.
"""
from __future__ import annotations

import logging
from typing import Any, Dict

import torch
import torch.nn as nn

from flairhip import ops
from flairhip.nn import HipCrossEntropyLoss
from flair_hub.tasks.metrics import MeanMetric, MulticlassJaccardIndex

logger = logging.getLogger(__name__)


class SegmentationTask(nn.Module):
    def __init__(self, model, config: Dict[str, Any], criterion=None):
        super().__init__()
        self.model = model
        self.config = config
        self.criterion = criterion
        self.trainer = None
        self.global_step = 0
        self._logged: Dict[str, Any] = {}

        self._scheduler_type = None
        self._using_plateau = False
        self._warmup_scheduler = None
        self._plateau_scheduler = None
        self._lr_scheduler = None

        mods = config["modalities"]
        self.mod_dropout = any(v > 0 for v in mods.get("modality_dropout", {}).values())
        self.aux_loss_modalities = [m for m, on in mods.get("aux_loss", {}).items()
                                    if on and mods["inputs"].get(m, False)]
        self.aux_loss_weight = mods.get("aux_loss_weight", {})

        labels, lcfg = config["labels"], config["labels_configs"]

        def jaccard(avg):
            return nn.ModuleDict({t.replace(".", "_"): MulticlassJaccardIndex(len(lcfg[t]["value_name"]), average=avg)
                                  for t in labels})

        self.train_metrics, self.val_metrics, self.val_iou = jaccard("weighted"), jaccard("weighted"), jaccard(None)
        self.train_loss, self.val_loss = MeanMetric(), MeanMetric()
        self.register_buffer("_invalid_loss_flag", torch.zeros((), dtype=torch.int32), persistent=False)

    # ---- Lightning-like surface ------------------------------------------------------------------

    @property
    def device(self) -> torch.device:
        return next(self.model.parameters()).device

    def log(self, name: str, value, **kwargs) -> None:
        self._logged[name] = value

    def lr_schedulers(self):
        return self._lr_scheduler

    # ---- forward / step --------------------------------------------------------------------------

    def forward(self, batch, apply_mod_dropout: bool = False):
        return self.model(batch, apply_mod_dropout)

    def step(self, batch, training: bool = False):
        apply_mod_dropout = self.mod_dropout if training else False
        dict_logits_task, dict_logits_aux = self.forward(batch, apply_mod_dropout)

        loss_sum = 0
        all_preds, all_targets = {}, {}
        for task, logits in dict_logits_task.items():
            targets = HipCrossEntropyLoss.prepare_targets(batch[task].to(self.device))
            crit = self.criterion[task]
            main_loss = crit(logits, targets)
            self._check_for_invalid_loss(main_loss, task)
            main_preds = crit.last_prediction() if isinstance(crit, HipCrossEntropyLoss) else None
            if main_preds is None:
                main_preds = torch.argmax(logits.detach(), dim=1)
            aux_loss = self._compute_aux_loss(dict_logits_aux, task, targets)
            task_weight = self.config["labels_configs"][task].get("task_weight", 1.0)
            loss_sum = loss_sum + task_weight * (main_loss + aux_loss)
            all_preds[task] = main_preds
            all_targets[task] = targets
        return loss_sum, all_preds, all_targets

    def _compute_aux_loss(self, dict_logits_aux, task, targets):
        # The reference tests `task in dict_logits_aux` against keys 'aux_<mod>_<task>' (tasks_module.py:180),
        # which never matches, so its auxiliary term is always 0 (the aux decoders are built and run, their logits
        # just never reach the loss).
        return 0.0

    def _check_for_invalid_loss(self, loss, task, is_aux: bool = False):
        bad = (~torch.isfinite(loss.detach())).to(torch.int32).reshape(())
        self._invalid_loss_flag += bad
        if self.config.get("hyperparams", {}).get("check_loss_every_step", False):
            self._report_invalid_loss()

    def _report_invalid_loss(self):
        if int(self._invalid_loss_flag.item()):
            logger.info("NaN or Inf detected in a training/validation loss")
            self._invalid_loss_flag.zero_()

    def training_step(self, batch, batch_idx):
        loss, all_preds, all_targets = self.step(batch, training=True)
        self.train_loss.update(loss)
        for task in all_preds:
            self.train_metrics[task.replace(".", "_")].update(all_preds[task], all_targets[task])
        return loss

    def on_train_batch_end(self, outputs, batch, batch_idx):
        if self._scheduler_type == "cycle_then_plateau" and not self._using_plateau:
            if self.global_step < self._warmup_scheduler.total_steps:
                self._warmup_scheduler.step()
            if self.global_step == self._warmup_scheduler.total_steps:
                self._using_plateau = True

    def _sync_metrics(self, metrics) -> None:
        """Sum the metric state over the data-parallel ranks before it is read -- the reference gets this from
        torchmetrics' state synchronisation and ``sync_dist=True`` (flair_hub/tasks/tasks_module.py:215-236,296-300).
        Without it every rank would log (and schedule its learning rate on) the score of its own shard."""
        from flairhip.distributed import all_reduce_sum_
        tensors = []
        for m in metrics:
            tensors += [m.confmat] if hasattr(m, "confmat") else [m.total, m.count]
        all_reduce_sum_(tensors)

    def on_train_epoch_end(self):
        self._report_invalid_loss()
        self._sync_metrics(list(self.train_metrics.values()) + [self.train_loss])
        for task, metric in self.train_metrics.items():
            self.log(f"train_miou_{task.split('-')[-1]}", metric.compute())
            metric.reset()
        self.log("train_loss", self.train_loss.compute())
        self.train_loss.reset()

    def validation_step(self, batch, batch_idx):
        loss, all_preds, all_targets = self.step(batch, training=False)
        self.val_loss.update(loss)
        for task in all_preds:
            key = task.replace(".", "_")
            self.val_metrics[key].update(all_preds[task], all_targets[task])
            self.val_iou[key].update(all_preds[task], all_targets[task])
        return loss

    def on_validation_epoch_end(self):
        self._report_invalid_loss()
        self._sync_metrics(list(self.val_metrics.values()) + list(self.val_iou.values()) + [self.val_loss])
        self.log("val_loss", self.val_loss.compute())
        mious = []
        for task in self.val_metrics:
            miou = self.val_metrics[task].compute()
            mious.append(float(miou))
            self.log(f"val_miou_{task.split('-')[-1]}", miou)
            self.val_metrics[task].reset()
            self.val_iou[task].reset()
        self.log("val_miou", sum(mious) / max(len(mious), 1))
        val_loss = self._logged.get("val_loss")
        self.val_loss.reset()
        if self._scheduler_type == "cycle_then_plateau" and self._using_plateau and val_loss is not None:
            self._plateau_scheduler.step(val_loss)

    def predict_step(self, batch, batch_idx=0, dataloader_idx=0):
        """{'preds_<task>': argmax(softmax(logits))} -- uint8 class maps straight from the logits kernel."""
        dict_logits_task, _ = self.forward(batch, apply_mod_dropout=False)
        out = {}
        for task, logits in dict_logits_task.items():
            nhwc = getattr(logits, "_ffa_nhwc", None)
            if nhwc is not None:
                out[f"preds_{task}"] = ops.predict_u8(nhwc, logits._ffa_classes, "argmax")
            else:
                out[f"preds_{task}"] = torch.argmax(logits, dim=1)
        return out

    # ---- optimisation ----------------------------------------------------------------------------

    def configure_optimizers(self):
        cfg = self.config["hyperparams"]
        optimizer = self._init_optimizer(cfg)
        total_steps = self.trainer.estimated_stepping_batches if self.trainer is not None else cfg.get("total_steps")
        scheduler_type = cfg.get("scheduler", None)
        warmup_fraction = cfg.get("warmup_fraction", 0.0)
        self._scheduler_type = scheduler_type
        sched = torch.optim.lr_scheduler

        if scheduler_type == "reduce_on_plateau":
            scheduler = sched.ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=cfg["plateau_patience"],
                                                cooldown=4, min_lr=1e-7)
            return {"optimizer": optimizer,
                    "lr_scheduler": {"scheduler": scheduler, "monitor": "val_loss", "interval": "epoch"}}
        if scheduler_type == "one_cycle_lr":
            scheduler = sched.OneCycleLR(optimizer, max_lr=cfg["learning_rate"], total_steps=total_steps,
                                         pct_start=warmup_fraction, cycle_momentum=False, div_factor=1000)
            return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step"}}
        if scheduler_type == "cycle_then_plateau":
            warmup_steps = int(warmup_fraction * total_steps)
            self._warmup_scheduler = sched.OneCycleLR(optimizer, max_lr=cfg["learning_rate"], total_steps=warmup_steps,
                                                      pct_start=1.0, cycle_momentum=False, div_factor=1000,
                                                      final_div_factor=1)
            self._plateau_scheduler = sched.ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=10,
                                                              cooldown=4, min_lr=1e-7)
            return {"optimizer": optimizer}
        return optimizer

    def _init_optimizer(self, cfg):
        optim_type = cfg["optimizer"]
        params = self.model.parameters()
        lr = cfg["learning_rate"]
        if optim_type == "sgd":
            return torch.optim.SGD(params, lr=lr)
        if optim_type in ["adam", "adamw"]:
            # same update rule as the reference's default construction (tasks_module.py:385-389).  On the GPU:
            # flairhip.optim.HipAdamW / HipAdam -- one launch over all parameter tensors (csrc/optim.hip; torch's fused
            # multi-tensor kernel took four launches at half the bandwidth: 0.26 -> 0.14 ms per step), same state layout
            # as torch's, hipGraph-capturable.  ``hyperparams.torch_optimizer: true`` keeps torch's fused implementation.
            params = list(params)
            on_gpu = bool(params) and all(p.is_cuda and p.dtype == torch.float32 for p in params)
            if on_gpu and not cfg.get("torch_optimizer", False):
                from flairhip.optim import HipAdam, HipAdamW
                cls = HipAdamW if optim_type == "adamw" else HipAdam
                return cls(params, lr=lr, weight_decay=cfg["optim_weight_decay"], betas=tuple(cfg["optim_betas"]))
            cls = torch.optim.AdamW if optim_type == "adamw" else torch.optim.Adam
            return cls(params, lr=lr, weight_decay=cfg["optim_weight_decay"], betas=tuple(cfg["optim_betas"]),
                       fused=on_gpu)
        raise ValueError(f"Unsupported optimizer type: {optim_type}")
