"""Whole-step hipGraph capture.

A batch-32 training step is ~900 kernel launches (conv / BN / pool / loss forward and backward, optimizer);
the kernels sum to 19.8 ms but eager dispatch leaves ~1 ms of gaps, and at the zonal loop's batch 8 the step is
launch-bound outright.  Every libflairhip entry point launches on the caller's stream without allocating or
synchronising, so the whole step -- SegmentationTask.training_step (forward, fused loss, metric update),
backward and the optimizer -- can be captured once into a HIP graph and replayed per batch: static shapes,
static buffers, one launch per step.  The learning-rate schedule stays on the host and writes the next value
into the optimizer's device-resident ``lr`` tensor between replays.

Capture hazard (found the hard way: segfault in hipStreamEndCapture): tensors that keep an EARLIER step's autograd
graph alive (a stored loss) keep its AccumulateGrad nodes alive, and those run on the stream they were created on --
from inside the capture that is an illegal cross-stream dependency.  Drop such references before constructing a
GraphedTrainStep (HipTrainer does; the warm-up steps here never keep their loss).

With world size > 1 (``grad_reduce``) the same captured kernels run as TWO graphs with the collectives between them:
graph A = forward + loss + backward (weight gradients written straight into GradSync's flat buckets), then the bucketed
all-reduce (RCCL, eager calls on RCCL's stream, the compute stream waits for them stream-side), then graph B = the
optimizer step.  One rank and several ranks therefore execute the same kernels from the same captures; the eager step
with the all-reduces issued from autograd hooks (overlapped with backward) stays available as the fallback.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from . import nn as hnn


def _capture_mode() -> str:
    """cudaStreamCaptureMode of the captures in this module.  With a process group alive, ProcessGroupNCCL's watchdog
    THREAD polls the completion events of earlier collectives (hipEventQuery); under the default "global" mode any such
    call from any thread while this thread captures is an error and takes the process down ("operation not permitted
    when stream is capturing" -- hit on the one-rank RCCL rehearsal of the data-parallel graph step, a race on when the
    watchdog last looked).  "thread_local" keeps the safety net for the capturing thread and leaves other threads alone."""
    import torch.distributed as dist
    return "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"


def make_capturable(optimizer: torch.optim.Optimizer) -> None:
    """Adam / AdamW: device-resident step counters and learning rate, as graph capture requires."""
    for g in optimizer.param_groups:
        if "capturable" in g:
            g["capturable"] = True
        if not torch.is_tensor(g["lr"]):
            dev = g["params"][0].device
            g["lr"] = torch.tensor(float(g["lr"]), dtype=torch.float32, device=dev)
        if "initial_lr" in g and torch.is_tensor(g["initial_lr"]):
            g["initial_lr"] = float(g["initial_lr"])


class GraphedTrainStep:
    """step(batch) -> loss tensor (static buffer, valid until the next call).

    ``grad_reduce(params, grads)`` (e.g. flairhip.distributed.GradSync(hooks=False).reduce_grads) switches to the
    data-parallel form: forward + loss + backward are one graph, the optimizer step a second one, and every replay of
    the first is followed by the gradient reduction -- collectives stay outside the captures."""

    def __init__(self, task, optimizer: torch.optim.Optimizer, example_batch: Dict[str, torch.Tensor],
                 warmup_steps: int = 3, after_step: Optional[Callable[[], None]] = None,
                 grad_reduce: Optional[Callable] = None):
        self.task, self.optimizer, self.after_step, self.grad_reduce = task, optimizer, after_step, grad_reduce
        self.static_batch = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in example_batch.items()}
        self._bns = [m for m in task.modules() if isinstance(m, hnn.HipBatchNorm2d)]
        make_capturable(optimizer)

        # warm-up on a side stream: sizes every workspace, fills the weight-pack plan, creates optimizer state
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup_steps):
                self._eager_step(i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.warmup_steps = warmup_steps

        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        pending = [bn._pending_batches for bn in self._bns]
        # every packed operand counts as stale from here: the pack kernels MUST be part of the captured step (a plan
        # that happens to be fresh at capture time would otherwise leave the replays training on frozen operands)
        hnn.bump_state_epoch()
        with torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            loss = task.training_step(self.static_batch, 0)
            loss.backward()
            if grad_reduce is None:
                optimizer.step()
        self.loss = loss
        for bn, n in zip(self._bns, pending):  # capture ran the Python but not the kernels
            bn._pending_batches = n
        self.opt_graph = None
        if grad_reduce is not None:  # the gradient tensors the replay rewrites in place
            self.params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            self.static_grads = [p.grad for p in self.params]
            # point every p.grad at its bucket view (the reduction runs on whatever the capture left in the gradient
            # buffers -- no kernel of graph A has run yet, the values are never used) and capture the optimizer step on
            # those tensors: the same addresses in every later step
            grad_reduce(self.params, self.static_grads)
            self.opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.opt_graph, capture_error_mode=_capture_mode()):
                optimizer.step()

    def _eager_step(self, i: int):
        loss = self.task.training_step(self.static_batch, i)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        if self.grad_reduce is not None:
            params = [p for g in self.optimizer.param_groups for p in g["params"] if p.grad is not None]
            self.grad_reduce(params, [p.grad for p in params])
        self.optimizer.step()
        if self.after_step is not None:
            self.after_step()
        return loss

    def __call__(self, batch: Dict[str, torch.Tensor]) -> torch.Tensor:
        for k, v in batch.items():
            if torch.is_tensor(v):
                dst = self.static_batch[k]
                if dst.data_ptr() != v.data_ptr():
                    dst.copy_(v, non_blocking=True)
        self.graph.replay()
        # weights and BatchNorm statistics moved on the device behind Python's back: version-keyed caches (packed
        # operands used by eager steps, eval-mode BatchNorm folds used by validation) must not survive the replay
        hnn.bump_state_epoch()
        if self.grad_reduce is not None:
            self.grad_reduce(self.params, self.static_grads)
            self.opt_graph.replay()
            hnn.bump_state_epoch()
        for bn in self._bns:  # host-side bookkeeping the replay skips
            bn.note_batch()
        if self.after_step is not None:
            self.after_step()
        return self.loss


class GraphedCall:
    """fn(inputs: dict of tensors) -> dict of tensors, captured once into a HIP graph and replayed with new input
    values copied into the static buffers.  For inference loops with a fixed batch shape (the zonal tile loop: eval
    forward + margin-crop/argmax is ~120 kernel launches for 1.4 ms of GPU work at batch 8).  The returned tensors
    are static buffers, valid until the next call."""

    def __init__(self, fn, example_inputs: Dict[str, torch.Tensor], warmup: int = 2):
        self.static_in = {k: v.clone() for k, v in example_inputs.items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                fn(self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=_capture_mode()):
            self.static_out = fn(self.static_in)

    def __call__(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        for k, dst in self.static_in.items():
            src = inputs[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_out
