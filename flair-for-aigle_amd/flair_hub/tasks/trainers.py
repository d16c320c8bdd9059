"""Training / prediction drivers -- counterpart of the reference's flair_hub/tasks/trainers.py
(train :35-108 builds a pytorch_lightning.Trainer and calls fit/validate; predict :111-136).

Lightning is replaced by HipTrainer: one process per GPU, the SegmentationTask hooks are called in
Lightning's order, gradients are averaged across ranks by flairhip.distributed.GradSync (RCCL over
xGMI, overlapped with backward) in place of Lightning's DDP strategy.  Callbacks, loggers and
checkpoint policies of the reference are orchestration outside the hot path; what is kept is the
best-checkpoint-on-val_miou save so that stages can reload it.
"""
from __future__ import annotations

import logging
import os
from typing import Any, Dict, Iterable, Optional

import torch
import torch.distributed as dist

from flairhip.distributed import GradSync, ShardedLoader, ensure_process_group

logger = logging.getLogger(__name__)


def check_batchnorm_and_batch_size(config: Dict[str, Any], seg_module: torch.nn.Module) -> None:
    """Abort on BatchNorm with batch size 1 (reference :17-32): batch statistics of one sample are degenerate."""
    from flairhip.nn import HipBatchNorm2d
    if config["hyperparams"]["batch_size"] == 1 and any(isinstance(m, HipBatchNorm2d) for m in seg_module.modules()):
        raise SystemExit("BatchNorm layers need a batch size > 1 for training")


def _to_device(batch: dict, device) -> dict:
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


class DevicePrefetcher:
    """Host batches -> device batches, one batch ahead on a copy stream: while the device runs step k the host fetches
    batch k+1 from the loader and its H2D copy proceeds next to the compute kernels (the reference schema moves 25 MB
    per tile -- f32 imagery + f32 one-hot labels -- i.e. ~15 ms of PCIe time per 32-tile batch, as long as the step
    itself).  Pinned loader output (DataLoader(pin_memory=True)) makes the copies asynchronous; pageable tensors
    still work, their copies just block the host."""

    def __init__(self, loader: Iterable, device: torch.device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device)

    def __len__(self) -> int:
        return len(self.loader)

    def _stage(self, host: Optional[dict]):
        if host is None:
            return None
        with torch.cuda.stream(self.stream):
            dev = _to_device(host, self.device)
            done = torch.cuda.Event()
            done.record(self.stream)
        return dev, done

    def __iter__(self):
        it = iter(self.loader)
        cur = self._stage(next(it, None))
        while cur is not None:
            batch, done = cur
            main = torch.cuda.current_stream(self.device)
            main.wait_event(done)
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)  # allocated on the copy stream, consumed on the compute stream
            yield batch  # the consumer enqueues step k and comes back for more
            cur = self._stage(next(it, None))  # fetch + copy of batch k+1 overlap step k on the device


class HipTrainer:
    """The subset of pytorch_lightning.Trainer the reference relies on."""

    def __init__(self, accelerator: str = "gpu", devices: int = 1, strategy: str = "auto", num_nodes: int = 1,
                 max_epochs: int = 1, default_root_dir: Optional[str] = None, monitor: str = "val_miou",
                 monitor_mode: str = "max", max_steps: Optional[int] = None, hip_graph: bool = False, **unused):
        if accelerator not in ("gpu", "cuda", "auto"):
            raise RuntimeError("HipTrainer drives the MI355X path only (accelerator='gpu'); the CPU restatement "
                               "lives in oracle/ and is test infrastructure")
        self.max_epochs, self.max_steps = max_epochs, max_steps
        # forward + loss + backward + optimizer captured once and replayed per batch (flairhip.graph.GraphedTrainStep;
        # several ranks: graph(forward + backward) -> bucketed all-reduce -> graph(optimizer)); batches of another
        # shape and modality dropout fall back to eager steps
        self.hip_graph = bool(hip_graph)
        self.default_root_dir = default_root_dir
        self.monitor, self.monitor_mode = monitor, monitor_mode
        local = int(os.environ.get("LOCAL_RANK", "0"))
        # FFA_DIST_BACKEND=gloo rehearses several ranks on a one-GPU box: they then share cuda:0
        ndev = max(torch.cuda.device_count(), 1)
        self.device = torch.device("cuda", local % ndev if os.environ.get("FFA_DIST_BACKEND") == "gloo" else local)
        # Lightning's DDP strategy sets the process group up itself (reference trainers.py:81-91 only names the
        # strategy); under torchrun (WORLD_SIZE > 1) so does this trainer, before anything collective happens
        ensure_process_group(self.device)
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        if devices not in (None, "auto") and int(devices) * int(num_nodes) > 1 and self.world_size == 1:
            raise RuntimeError(f"HipTrainer: {devices} device(s) x {num_nodes} node(s) requested but this is a single "
                               "process: launch one process per GPU with `python -m torch.distributed.run "
                               "--nproc-per-node N ...` (the trainer never spawns ranks itself)")
        self.seed = int(unused.get("seed", 0))
        self.estimated_stepping_batches = 0
        self.callback_metrics: Dict[str, Any] = {}
        self.best_model_path = None
        self.optimizers = []

    # ---- fit -------------------------------------------------------------------------------------

    def _shard(self, loader, shuffle=None, drop_last=True):
        """the rank's share of a loader (what Lightning's DistributedSampler injection does under DDP)"""
        if loader is None or self.world_size == 1 or isinstance(loader, ShardedLoader):
            return loader
        return ShardedLoader(loader, self.rank, self.world_size, shuffle=shuffle, seed=self.seed, drop_last=drop_last)

    def fit(self, model, datamodule=None, train_dataloaders: Optional[Iterable] = None,
            val_dataloaders: Optional[Iterable] = None) -> None:
        torch.cuda.set_device(self.device)
        model.to(self.device)
        model.trainer = self
        if datamodule is not None:
            datamodule.setup("fit")
            train_dataloaders = datamodule.train_dataloader()
            val_dataloaders = datamodule.val_dataloader() if hasattr(datamodule, "val_dataloader") else None
        train_dataloaders = self._shard(train_dataloaders, drop_last=True)
        val_dataloaders = self._shard(val_dataloaders, shuffle=False, drop_last=False)
        steps_per_epoch = len(train_dataloaders)
        self.estimated_stepping_batches = self.max_steps or steps_per_epoch * self.max_epochs

        opt_cfg = model.configure_optimizers()
        optimizer = opt_cfg["optimizer"] if isinstance(opt_cfg, dict) else opt_cfg
        sched_cfg = opt_cfg.get("lr_scheduler") if isinstance(opt_cfg, dict) else None
        scheduler = sched_cfg["scheduler"] if sched_cfg else None
        interval = sched_cfg.get("interval", "epoch") if sched_cfg else None
        model._lr_scheduler = scheduler
        self.optimizers = [optimizer]
        use_graph = (self.hip_graph and not getattr(model, "mod_dropout", False) and
                     isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)))
        graph_ddp = use_graph and self.world_size > 1
        # exact_unused: a parameter no rank produced a gradient for keeps grad = None, as under the reference's
        # ddp_find_unused_parameters_true and as in the single-GPU path (AdamW then skips it instead of decaying it);
        # graph mode runs without autograd hooks: the finished gradients are handed to reduce_grads()
        # graph mode reduces after backward (nothing to overlap with): one collective over all gradients
        sync = GradSync(model, hooks=not graph_ddp, exact_unused=True, **({"bucket_bytes": 1 << 30} if graph_ddp else {}))

        def reduce_now():
            ps = [p for p in model.parameters() if p.grad is not None]
            sync.reduce_grads(ps, [p.grad for p in ps])

        best = None
        done = False
        graphed, graph_sig, loss = None, None, None
        for epoch in range(self.max_epochs):
            model.train()
            if isinstance(train_dataloaders, ShardedLoader):
                train_dataloaders.set_epoch(epoch)  # a new shared permutation per epoch, as Lightning does
            for i, batch in enumerate(DevicePrefetcher(train_dataloaders, self.device)):
                if use_graph and graphed is None and model.global_step >= 2:
                    # two ordinary steps first: they size every workspace, fill the weight-pack plan and create the
                    # optimizer state, so the capture itself needs no warm-up steps that would move the weights
                    from flairhip.graph import GraphedTrainStep
                    # nothing may keep the previous step's autograd graph alive: its AccumulateGrad nodes are bound
                    # to the stream they were created on, and running them from the capture stream aborts the capture
                    loss = None
                    graphed = GraphedTrainStep(model, optimizer, {k: v for k, v in batch.items() if torch.is_tensor(v)},
                                               warmup_steps=0, grad_reduce=sync.reduce_grads if graph_ddp else None)
                    graph_sig = {k: (tuple(v.shape), v.dtype) for k, v in batch.items() if torch.is_tensor(v)}
                if graphed is not None and graph_sig == {k: (tuple(v.shape), v.dtype) for k, v in batch.items()
                                                         if torch.is_tensor(v)}:
                    loss = graphed({k: v for k, v in batch.items() if torch.is_tensor(v)})
                else:
                    loss = model.training_step(batch, i)
                    optimizer.zero_grad(set_to_none=True)
                    loss.backward()
                    if graph_ddp:
                        reduce_now()
                    else:
                        sync.finish()
                    optimizer.step()
                if scheduler is not None and interval == "step":
                    scheduler.step()
                model.global_step += 1
                model.on_train_batch_end(loss, batch, i)
                if self.max_steps and model.global_step >= self.max_steps:
                    done = True
                    break
            model.on_train_epoch_end()
            if val_dataloaders is not None:
                self.validate(model, dataloaders=val_dataloaders)
                score = self.callback_metrics.get(self.monitor)
                if score is not None:
                    score = float(score)
                    better = best is None or (score > best if self.monitor_mode == "max" else score < best)
                    if better:
                        best = score
                        self._save_best(model, epoch, score)
                if scheduler is not None and interval == "epoch":
                    monitor = sched_cfg.get("monitor")
                    scheduler.step(self.callback_metrics[monitor]) if monitor else scheduler.step()
            if done:
                break
        sync.remove()

    def _save_best(self, model, epoch: int, score: float) -> None:
        if self.rank != 0 or not self.default_root_dir:
            return
        os.makedirs(self.default_root_dir, exist_ok=True)
        path = os.path.join(self.default_root_dir, f"ckpt-epoch={epoch:02d}-{self.monitor}={score:.2f}.ckpt")
        torch.save({"state_dict": model.state_dict(), "epoch": epoch}, path)
        self.best_model_path = path

    # ---- validate / predict ----------------------------------------------------------------------

    @torch.no_grad()
    def validate(self, model, datamodule=None, dataloaders: Optional[Iterable] = None):
        model.to(self.device)
        model.trainer = self
        if datamodule is not None:
            datamodule.setup("validate")
            dataloaders = datamodule.val_dataloader()
        dataloaders = self._shard(dataloaders, shuffle=False, drop_last=False)
        model.eval()
        for i, batch in enumerate(dataloaders):
            model.validation_step(_to_device(batch, self.device), i)
        model.on_validation_epoch_end()
        self.callback_metrics.update({k: (float(v) if torch.is_tensor(v) else v) for k, v in model._logged.items()})
        return [dict(self.callback_metrics)]

    @torch.no_grad()
    def predict(self, model, datamodule=None, dataloaders: Optional[Iterable] = None, return_predictions: bool = True):
        model.to(self.device)
        model.trainer = self
        if datamodule is not None:
            datamodule.setup("predict")
            dataloaders = datamodule.predict_dataloader()
        dataloaders = self._shard(dataloaders, shuffle=False, drop_last=False)  # every rank predicts its own share
        model.eval()
        outs = []
        for i, batch in enumerate(dataloaders):
            out = model.predict_step(_to_device(batch, self.device), i)
            if return_predictions:
                outs.append(out)
        return outs


def train(config: Dict[str, Any], data_module, seg_module, out_dir: str) -> HipTrainer:
    """Same role as the reference's train(): build the trainer from config['hardware'] / ['hyperparams'], fit,
    then run a final validation."""
    check_batchnorm_and_batch_size(config, seg_module)
    hw = config.get("hardware", {})
    trainer = HipTrainer(accelerator=hw.get("accelerator", "gpu"), devices=hw.get("gpus_per_node", 1),
                         strategy=hw.get("strategy", "auto"), num_nodes=hw.get("num_nodes", 1),
                         max_epochs=config["hyperparams"]["num_epochs"], default_root_dir=out_dir,
                         monitor=config.get("saving", {}).get("ckpt_monitor", "val_miou"),
                         monitor_mode=config.get("saving", {}).get("ckpt_monitor_mode", "max"),
                         hip_graph=bool(hw.get("hip_graph", False)))
    trainer.fit(seg_module, datamodule=data_module)
    trainer.validate(seg_module, datamodule=data_module)
    return trainer


def predict(config: Dict[str, Any], data_module, seg_module, out_dir: str):
    hw = config.get("hardware", {})
    trainer = HipTrainer(accelerator=hw.get("accelerator", "gpu"), devices=hw.get("gpus_per_node", 1),
                         strategy=hw.get("strategy", "auto"), num_nodes=hw.get("num_nodes", 1))
    return trainer.predict(seg_module, datamodule=data_module, return_predictions=True)
