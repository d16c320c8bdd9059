"""N > 1 on the real product: two ranks of the HIP U-Net sharing the one GPU of the test box (gloo backend -- RCCL
refuses two ranks on one device, and the product has no CPU path for the gloo CPU test to drive).  HipTrainer sets
up the process group, shards the global batches, GradSync averages the gradients; the replicas must stay bit-identical
and equal the single-process emulation "mean of the per-rank gradients, one optimizer step" (the Lightning DDP
semantics of the reference: flair_hub/tasks/trainers.py:81-91, DistributedSampler + drop_last
flair_hub/tasks/module_setup.py:40)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

from helpers import ROOT, MOD, TASK

pytestmark = pytest.mark.gpu

STEPS, B_RANK, TILE, WORLD = 3, 2, 64, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _global_batches():
    g = torch.Generator().manual_seed(5)
    return [{MOD: torch.randn(WORLD * B_RANK, 5, TILE, TILE, generator=g),
             TASK: torch.randint(0, 19, (WORLD * B_RANK, TILE, TILE), generator=g, dtype=torch.uint8)}
            for _ in range(STEPS)]


def _build(seed_shift=0, arch=None):
    from flairhip.configs import unet_resnet34_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    cfg = unet_resnet34_config(in_channels=5, precision="bf16", batch_size=B_RANK, total_steps=STEPS)
    if arch:
        cfg["models"]["monotemp_model"].update({"arch": arch, "drop_path_rate": 0.0})  # no per-rank random masks
    torch.manual_seed(cfg["hyperparams"]["seed"] + seed_shift)
    return build_segmentation_module(cfg, {MOD: TILE}, "train")


def _worker(rank, port, out_dir, arch=None):
    for p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(WORLD), FFA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from flair_hub.tasks.trainers import HipTrainer
    task = _build(seed_shift=rank, arch=arch)  # ranks start from DIFFERENT weights: rank 0's must win
    trainer = HipTrainer(max_epochs=1, max_steps=STEPS)  # process group + sharding happen inside
    assert trainer.world_size == WORLD and trainer.rank == rank
    trainer.fit(task, train_dataloaders=_global_batches())
    torch.cuda.synchronize()
    torch.save({k: v.detach().cpu() for k, v in task.model.named_parameters()}, os.path.join(out_dir, f"w{rank}.pt"))
    import torch.distributed as dist
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_mean_gradient_step(cuda, tmp_path):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    w0, w1 = (torch.load(tmp_path / f"w{r}.pt") for r in range(WORLD))
    for k in w0:
        assert torch.equal(w0[k], w1[k]), f"replicas diverged at {k}"

    # single-process emulation with rank 0's start weights
    task = _build().to(cuda)
    task.train()

    class _T:  # what configure_optimizers reads from the trainer
        estimated_stepping_batches = STEPS
    task.trainer = _T()
    opt_cfg = task.configure_optimizers()
    opt, sched = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    params = dict(task.model.named_parameters())
    for i, gb in enumerate(_global_batches()):
        sums = {}
        for r in range(WORLD):
            sl = {k: v[r * B_RANK:(r + 1) * B_RANK].to(cuda) for k, v in gb.items()}
            opt.zero_grad(set_to_none=True)
            task.training_step(sl, i).backward()
            for k, p in params.items():
                if p.grad is not None:
                    sums[k] = p.grad.clone() if k not in sums else sums[k] + p.grad
        for k, p in params.items():
            # a parameter no rank produced a gradient for keeps grad = None (HipTrainer runs GradSync with exact_unused:
            # the reference's ddp_find_unused_parameters_true), so AdamW neither decays it nor creates state for it
            p.grad = sums[k] * (1.0 / WORLD) if k in sums else None
        opt.step()
        sched.step()
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in params.items():
        d = (p.detach().cpu() - w0[k]).abs().max().item()
        worst = max(worst, d / (w0[k].abs().max().item() + 1e-12))
        assert torch.equal(p.detach().cpu(), w0[k]), f"{k}: data-parallel step differs from the mean-gradient step ({d})"


def _rccl_worker(rank, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from flairhip.distributed import GradSync
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)  # backend "nccl" IS RCCL on ROCm
    results = {}
    for mode in ("plain", "rccl"):
        task = _build().to(dev)
        task.train()

        class _T:
            estimated_stepping_batches = STEPS
        task.trainer = _T()
        opt = task.configure_optimizers()["optimizer"]
        sync = GradSync(task.model, always_sync=(mode == "rccl"))
        for i, gb in enumerate(_global_batches()):
            b = {k: v[:B_RANK].to(dev) for k, v in gb.items()}
            loss = task.training_step(b, i)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            sync.finish()
            opt.step()
        if mode == "rccl":
            assert sync._buckets is not None and len(sync._buckets) >= 3  # 24.4 M f32 parameters in 32 MiB buckets
            # from step 1 on the weight gradients were written straight into the buckets: p.grad aliases its slot
            w = task.model.encoders[MOD].seg_model.layer2[0].conv1.weight
            assert w.grad.data_ptr() == sync._view(w).data_ptr()
        torch.cuda.synchronize()
        results[mode] = {k: v.detach().cpu() for k, v in task.model.named_parameters()}
        sync.remove()
    torch.save(results, os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


def test_bucket_protocol_over_rccl_on_one_gpu(cuda, tmp_path):
    """The collectives of the data-parallel step through the backend the product ships with: backend "nccl" = RCCL.
    One GPU admits one RCCL rank, so the group has a single member (the sum over one rank is the identity): what
    this executes for real is RCCL communicator set-up, the broadcast of the bucket order, the bucketed all-reduce
    kernels launched from the autograd hooks underneath backward, the in-bucket gradient views and the stream-side
    waits -- and the result must equal the same steps without any synchronisation, bit for bit."""
    mp.spawn(_rccl_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / "rccl.pt")
    for k, v in res["plain"].items():
        if k.startswith("fusion_handler."):
            continue  # gradient-less: the synchronised run zero-fills them, AdamW then applies weight decay alone
        assert torch.equal(v, res["rccl"][k]), k


def _rccl_graph_worker(rank, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "flair-for-aigle_amd")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from flairhip.distributed import GradSync
    from flairhip.graph import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    results = {}
    for mode in ("graph", "graph+rccl"):
        task = _build().to(dev)
        task.train()

        class _T:
            estimated_stepping_batches = STEPS + 4
        task.trainer = _T()
        opt = task.configure_optimizers()["optimizer"]
        batches = [{k: v[:B_RANK].to(dev) for k, v in gb.items()} for gb in _global_batches()]
        sync = GradSync(task.model, hooks=False, always_sync=(mode == "graph+rccl"), exact_unused=True)
        stepper = GraphedTrainStep(task, opt, batches[0], warmup_steps=2, grad_reduce=sync.reduce_grads)
        losses = [stepper(b).item() for b in batches]
        torch.cuda.synchronize()
        results[mode] = ({k: v.detach().cpu() for k, v in task.model.named_parameters()}, losses)
    torch.save(results, os.path.join(out_dir, "rccl_graph.pt"))
    dist.destroy_process_group()


def test_the_multi_gpu_program_over_rccl_on_one_gpu(cuda, tmp_path):
    """What `bench.py --gpus N` and HipTrainer run for N > 1 -- hipGraph(forward + backward) -> GradSync.reduce_grads
    (bucketed all-reduce on RCCL's stream, stream-side waits, gradients already inside the buckets) -> hipGraph(AdamW) --
    in a one-rank RCCL group: equal, bit for bit, to the same two graphs without the collectives."""
    mp.spawn(_rccl_graph_worker, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(tmp_path / "rccl_graph.pt")
    (wa, la), (wb, lb) = res["graph"], res["graph+rccl"]
    assert la == lb
    for k in wa:
        assert torch.equal(wa[k], wb[k]), k


SWIN = "swin_tiny_patch4_window7_224-upernet"


def test_two_ranks_train_swin_upernet_like_the_mean_gradient_step(cuda, tmp_path):
    """the same flow for the transformer architecture: parameters that never receive a gradient (smp's unused FPNBlock of
    the input image) take part in the buckets with zeros but keep grad = None, the replicas stay bit-identical, and the result equals the
    mean-gradient step up to the summation order of the attention backward's LDS atomics"""
    mp.spawn(_worker, args=(_free_port(), str(tmp_path), SWIN), nprocs=WORLD, join=True)
    w0, w1 = (torch.load(tmp_path / f"w{r}.pt") for r in range(WORLD))
    for k in w0:
        assert torch.equal(w0[k], w1[k]), f"replicas diverged at {k}"
    task = _build(arch=SWIN).to(cuda)
    task.train()

    class _T:
        estimated_stepping_batches = STEPS
    task.trainer = _T()
    opt_cfg = task.configure_optimizers()
    opt, sched = opt_cfg["optimizer"], opt_cfg["lr_scheduler"]["scheduler"]
    params = dict(task.model.named_parameters())
    start = {k: p.detach().clone() for k, p in params.items()}
    for i, gb in enumerate(_global_batches()):
        sums = {}
        for r in range(WORLD):
            sl = {k: v[r * B_RANK:(r + 1) * B_RANK].to(cuda) for k, v in gb.items()}
            opt.zero_grad(set_to_none=True)
            task.training_step(sl, i).backward()
            for k, p in params.items():
                if p.grad is not None:
                    sums[k] = p.grad.clone() if k not in sums else sums[k] + p.grad
        for k, p in params.items():
            # a parameter no rank produced a gradient for keeps grad = None (HipTrainer runs GradSync with exact_unused:
            # the reference's ddp_find_unused_parameters_true), so AdamW neither decays it nor creates state for it
            p.grad = sums[k] * (1.0 / WORLD) if k in sums else None
        opt.step()
        sched.step()
    torch.cuda.synchronize()
    unused = [k for k in params if "fpn_stages.4" in k]
    assert unused, "the architecture is expected to hold gradient-less parameters"
    for k in unused:  # untouched by four steps of AdamW with weight decay, on every replica
        assert torch.equal(w0[k], start[k].cpu()), k
    for k, p in params.items():
        moved = (p.detach() - start[k]).abs().max().item()
        d = (p.detach().cpu() - w0[k]).abs().max().item()
        assert d <= 0.05 * moved + 1e-7, f"{k}: data-parallel step differs from the mean-gradient step ({d} vs a move of {moved})"
