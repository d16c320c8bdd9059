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
    sync = GradSync(net, bucket_bytes=100 * 1024)  # several buckets
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
