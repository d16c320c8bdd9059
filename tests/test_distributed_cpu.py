"""N > 1 path on CPU: flairhip.distributed.GradSync with the gloo backend, world size 2.

The bucketing / overlap / unused-parameter logic is device-agnostic torch.distributed code, so it is
exercised here without a GPU: every rank must end a step with the MEAN of the per-rank gradients, for a
model that has a parameter which never receives a gradient (the reference needs
'ddp_find_unused_parameters_true' for that), over several steps (step 0 learns the arrival order, later steps
launch each bucket from the gradient hooks while backward is still running)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT  # noqa: F401  (sets sys.path)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(40, 300)
        self.b = torch.nn.Linear(300, 300)
        self.c = torch.nn.Linear(300, 7)
        self.unused = torch.nn.Linear(5, 5)  # never part of the graph

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _data(rank, step):
    g = torch.Generator().manual_seed(100 * rank + step)
    return torch.randn(16, 40, generator=g), torch.randn(16, 7, generator=g)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))
    from flairhip.distributed import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234 + rank)  # ranks start with DIFFERENT weights: GradSync must broadcast rank 0's
    net = Net()
    sync = GradSync(net, bucket_bytes=100 * 1024, exact_unused=True)  # several buckets
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    for step in range(4):
        x, y = _data(rank, step)
        loss = ((net(x) - y) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        sync.finish()
        if step == 3:
            torch.save({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                       os.path.join(out_dir, f"grads_{rank}.pt"))
        opt.step()
    torch.save(net.state_dict(), os.path.join(out_dir, f"weights_{rank}.pt"))
    assert net.unused.weight.grad is None
    assert len(sync._buckets) >= 3
    dist.destroy_process_group()


def test_gradsync_gloo_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w0, w1 = (torch.load(tmp_path / f"weights_{r}.pt") for r in range(world))
    for k in w0:
        assert torch.equal(w0[k], w1[k]), f"replicas diverged at {k}"

    # single-process reference: same start weights (rank 0's), mean of the per-rank gradients each step
    torch.manual_seed(1234)
    net = Net()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    for step in range(4):
        grads = None
        for r in range(world):
            x, y = _data(r, step)
            net.zero_grad(set_to_none=True)
            ((net(x) - y) ** 2).mean().backward()
            g = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
            grads = g if grads is None else {k: grads[k] + g[k] for k in g}
        for k, p in net.named_parameters():
            p.grad = grads[k] / world if k in grads else None
        if step == 3:
            got = torch.load(tmp_path / "grads_0.pt")
            assert set(got) == set(grads)
            for k in grads:
                assert torch.allclose(got[k], grads[k] / world, rtol=1e-5, atol=1e-7), k
        opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(w0[k], v, rtol=1e-4, atol=1e-6), k


# --------------------------------------------------------------------------------------------------
# ranks whose backward differs from step to step (the model's modality dropout draws per rank:
# flair_hub/models/flair_model.py:343-352): the sequence of collectives must not depend on it


class BranchNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.enc_a = torch.nn.Linear(20, 200)
        self.enc_b = torch.nn.Linear(20, 200)
        self.head = torch.nn.Linear(200, 5)
        self.never = torch.nn.Linear(3, 3)

    def forward(self, x, use_a, use_b):
        h = 0
        if use_a:
            h = h + torch.relu(self.enc_a(x))
        if use_b:
            h = h + torch.relu(self.enc_b(x))
        return self.head(h)


# (use_a, use_b) per (step, rank): different sub-modules skipped on different ranks in different steps, including
# step 0 (where the bucket layout is learnt) and a branch that first gets a gradient after step 0
BRANCHES = [[(True, False), (False, True)], [(True, True), (True, False)], [(False, True), (False, True)],
            [(True, False), (True, True)]]


def _branch_worker(rank, world, port, out_dir, exact):
    sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))
    from flairhip.distributed import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(77 + rank)
    net = BranchNet()
    sync = GradSync(net, bucket_bytes=8 * 1024, exact_unused=exact)
    opt = torch.optim.AdamW(net.parameters(), lr=0.01, weight_decay=0.1)
    never0 = net.never.weight.detach().clone()
    for step, per_rank in enumerate(BRANCHES):
        ua, ub = per_rank[rank]
        g = torch.Generator().manual_seed(10 * step + rank)
        x, y = torch.randn(8, 20, generator=g), torch.randn(8, 5, generator=g)
        loss = ((net(x, ua, ub) - y) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        sync.finish()
        if exact:
            assert net.never.weight.grad is None  # no rank produced one: left alone, as DDP does
        else:
            assert net.never.weight.grad is not None and float(net.never.weight.grad.abs().max()) == 0.0
        opt.step()
        torch.save(net.state_dict(), os.path.join(out_dir, f"w_{step}_{rank}.pt"))
    if exact:
        assert torch.equal(net.never.weight, never0)  # AdamW skipped it (no weight decay on a gradient-less tensor)
    dist.destroy_process_group()


@pytest.mark.parametrize("exact", [False, True], ids=["zero-fill", "exact-unused"])
def test_gradsync_ranks_skip_different_submodules(tmp_path, exact):
    world = 2
    mp.spawn(_branch_worker, args=(world, _free_port(), str(tmp_path), exact), nprocs=world, join=True)
    for step in range(len(BRANCHES)):
        w0, w1 = (torch.load(tmp_path / f"w_{step}_{r}.pt") for r in range(world))
        for k in w0:
            assert torch.equal(w0[k], w1[k]), f"replicas diverged at step {step}, {k}"
    # reference: mean over ranks with a missing gradient counted as zero
    torch.manual_seed(77)
    net = BranchNet()
    opt = torch.optim.AdamW(net.parameters(), lr=0.01, weight_decay=0.1)
    for step, per_rank in enumerate(BRANCHES):
        sums = {}
        for r in range(world):
            g = torch.Generator().manual_seed(10 * step + r)
            x, y = torch.randn(8, 20, generator=g), torch.randn(8, 5, generator=g)
            net.zero_grad(set_to_none=True)
            ((net(x, *per_rank[r]) - y) ** 2).mean().backward()
            for k, p in net.named_parameters():
                if p.grad is not None:
                    sums[k] = sums.get(k, 0) + p.grad.clone()
        for k, p in net.named_parameters():
            if k in sums:
                p.grad = sums[k] / world
            else:
                p.grad = None if exact else torch.zeros_like(p)
        opt.step()
        got = torch.load(tmp_path / f"w_{step}_0.pt")
        for k, v in net.state_dict().items():
            assert torch.allclose(got[k], v, rtol=1e-5, atol=1e-7), (step, k)


def test_sharded_loader_covers_the_data_once():
    sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))
    from flairhip.distributed import ShardedLoader
    from torch.utils.data import DataLoader, TensorDataset
    world = 4
    data = torch.arange(103)
    # (a) DataLoader over a map-style dataset: DistributedSampler injection, shared permutation per epoch
    seen = {0: [], 1: []}
    for epoch in (0, 1):
        for r in range(world):
            ld = ShardedLoader(DataLoader(TensorDataset(data), batch_size=5, shuffle=True, drop_last=True), r, world,
                               seed=3)
            ld.set_epoch(epoch)
            items = [int(v) for (b,) in ld for v in b]
            assert len(items) == (103 // world // 5) * 5  # drop_last at both levels, same count on every rank
            seen[epoch].append(items)
        flat = [v for items in seen[epoch] for v in items]
        assert len(set(flat)) == len(flat)  # no sample on two ranks
    assert seen[0] != seen[1]  # set_epoch reshuffles
    # (b) generic iterable of global batches: contiguous slices per rank
    batches = [{"x": torch.arange(i * 8, i * 8 + 8), "ids": [str(j) for j in range(8)], "meta": 1} for i in range(3)]
    batches.append({"x": torch.arange(24, 30), "ids": [str(j) for j in range(6)], "meta": 1})  # ragged last batch
    for drop_last, want in ((True, 24), (False, 30)):
        got = []
        for r in range(world):
            for b in ShardedLoader(batches, r, world, drop_last=drop_last):
                assert len(b["ids"]) == b["x"].numel() and b["meta"] == 1
                got += b["x"].tolist()
        assert sorted(got) == list(range(want))


def test_sharded_loader_forwards_loader_arguments_and_counts_kept_batches():
    """round-2 advisor finding: the rebuilt DataLoader keeps generator / worker arguments, a custom batch_sampler is
    refused (not silently un-collated), and len() in slice mode counts the batches that are actually yielded"""
    sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))
    from flairhip.distributed import ShardedLoader
    from torch.utils.data import BatchSampler, DataLoader, SequentialSampler, TensorDataset
    ds = TensorDataset(torch.arange(40))
    g = torch.Generator().manual_seed(5)
    ld = ShardedLoader(DataLoader(ds, batch_size=4, shuffle=True, generator=g, num_workers=0), 1, 2, seed=1)
    assert ld.inner.generator is g and ld.inner.batch_size == 4 and len(ld) == len(list(ld)) == 5
    with pytest.raises(TypeError, match="batch_sampler"):
        ShardedLoader(DataLoader(ds, batch_sampler=BatchSampler(SequentialSampler(ds), 4, False)), 0, 2)

    class Global:  # an iterable of global batches that says how it batches (what DataLoader exposes)
        def __init__(self, n, bs, drop_last):
            self.dataset, self.batch_size, self.drop_last = range(n), bs, drop_last

        def __len__(self):
            return len(self.dataset) // self.batch_size + (0 if self.drop_last or len(self.dataset) % self.batch_size == 0 else 1)

        def __iter__(self):
            n, bs = len(self.dataset), self.batch_size
            for i in range(0, n - (n % bs if self.drop_last else 0), bs):
                yield {"x": torch.arange(i, min(i + bs, n))}

    for n, bs, inner_drop, world in ((30, 8, False, 4), (30, 8, True, 4), (32, 8, False, 4), (30, 6, False, 4), (28, 8, False, 4)):
        for drop_last in (True, False):
            for r in range(world):
                sl = ShardedLoader(Global(n, bs, inner_drop), r, world, drop_last=drop_last)
                assert len(sl) == len(list(sl)), (n, bs, inner_drop, drop_last, r)


# --------------------------------------------------------------------------------------------------
# the hook-less protocol of the hipGraph data-parallel step (GraphedTrainStep(grad_reduce=GradSync.reduce_grads)):
# finished gradients are handed over after backward; with exact_unused a parameter that no rank produced a gradient
# for keeps grad = None, so AdamW neither decays it nor creates state for it (the reference's
# ddp_find_unused_parameters_true behaviour, flair_hub/tasks/trainers.py:81-91)


def _reduce_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "flair-for-aigle_amd"))
    from flairhip.distributed import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(77 + rank)
    net = Net()
    sync = GradSync(net, bucket_bytes=100 * 1024, hooks=False, exact_unused=True)
    unused0 = net.unused.weight.detach().clone()  # after the broadcast from rank 0
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2, weight_decay=0.1)
    for step in range(4):
        x, y = _data(rank, step)
        loss = ((net(x) - y) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        ps = [p for p in net.parameters() if p.grad is not None]
        sync.reduce_grads(ps, [p.grad for p in ps])
        assert net.unused.weight.grad is None and net.unused.bias.grad is None
        if step == 3:
            torch.save({k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None},
                       os.path.join(out_dir, f"rgrads_{rank}.pt"))
        opt.step()
    assert torch.equal(net.unused.weight, unused0), "AdamW touched a parameter nobody produced a gradient for"
    assert net.unused.weight not in opt.state
    torch.save(net.state_dict(), os.path.join(out_dir, f"rweights_{rank}.pt"))
    dist.destroy_process_group()


def test_reduce_grads_hookless_protocol_leaves_unused_parameters_alone(tmp_path):
    world = 2
    mp.spawn(_reduce_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    w0, w1 = (torch.load(tmp_path / f"rweights_{r}.pt") for r in range(world))
    for k in w0:
        assert torch.equal(w0[k], w1[k]), f"replicas diverged at {k}"
    g0, g1 = (torch.load(tmp_path / f"rgrads_{r}.pt") for r in range(world))
    assert set(g0) == set(g1) and not any(k.startswith("unused") for k in g0)
    # mean of the per-rank gradients at the common weights of step 3 is what both ranks hold
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
