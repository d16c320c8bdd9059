"""Data-parallel gradient synchronisation over RCCL / xGMI -- the replacement for the Lightning DDP
strategy the reference selects at flair_hub/tasks/trainers.py:81-91 (``strategy='auto' |
'ddp_find_unused_parameters_true'``) -- plus the rank-sharded batch feed Lightning adds with its
DistributedSampler (drop_last=True: flair_hub/tasks/module_setup.py:40).

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm; "gloo" in CPU tests).  Every
rank holds a full replica; the only exchange of a step is the mean of the parameter gradients
(24.4 M f32 = 97.8 MB for the U-Net / ResNet-34).  Gradients live in a few large contiguous f32
buckets; each bucket's all-reduce is issued as soon as its last gradient has been written, so the
collective runs on RCCL's stream underneath the rest of backward.  xGMI is point-to-point (7 links per
GPU), so few large messages beat many small ones: the default bucket is 32 MiB (3 buckets for this
model), not DDP's 25 MB tuned for NVSwitch rings.

Layout and protocol (every rank issues the SAME collectives in the SAME order in every step, whatever
its own backward did -- the property ``find_unused_parameters`` gives the reference):
  * the buckets hold ALL parameters that require a gradient.  Their order is the order in which rank 0
    saw gradients arrive in the first step (parameters that got none go last); rank 0 broadcasts that
    order, so a rank whose first step differed -- modality dropout draws per rank
    (flair_hub/models/flair_model.py:343-352) -- still builds the identical layout;
  * buckets are launched strictly in index order; a bucket is complete when every one of its
    parameters has reported a gradient in this step; ``finish()`` zero-fills the slots of parameters
    that got no gradient and launches what is left, in order.  A parameter that starts (or stops)
    receiving gradients in a later step therefore changes only how much of the exchange overlaps
    backward, never the sequence of collectives;
  * a parameter without a local gradient takes part with zeros and ends the step with the mean of the other ranks'
    gradients in ``p.grad`` -- zeros when no rank had one.  ``exact_unused=True`` adds one small all-reduce of
    "had a gradient" flags per step and leaves ``p.grad`` untouched (None) for parameters no rank produced a gradient
    for, exactly what DDP's find_unused_parameters does (AdamW then skips them instead of applying weight decay
    alone); it costs a host read per step whenever some parameter had no local gradient;
  * ``p.grad`` is a view into the bucket.  Producers that know about the bucket (the weight-gradient
    kernels: ``grad_buffer(p)``) write there directly and autograd adopts the tensor they return, so a
    step moves no gradient bytes besides the collective itself; any other gradient is copied into its
    slot when its hook fires (BatchNorm affine parameters, biases: a few KB).

BatchNorm statistics stay per rank (the reference does not use SyncBN); parameters and buffers are
broadcast from rank 0 once at construction because the reference loads checkpoints on rank 0 only
(flair_hub/models/checkpoint.py:176 @rank_zero_only).
"""
from __future__ import annotations

from typing import Dict, Iterable, Iterator, List, Optional

import torch
import torch.distributed as dist


def grad_buffer(p: torch.Tensor) -> Optional[torch.Tensor]:
    """The tensor a gradient producer should write ``p``'s gradient into (a bucket view), or None."""
    return getattr(p, "_ffa_grad_buf", None)


class _Bucket:
    __slots__ = ("flat", "slots", "pending", "work", "launched")

    def __init__(self, flat, slots):
        self.flat, self.slots, self.pending, self.work, self.launched = flat, slots, len(slots), None, False


class GradSync:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 32 << 20, process_group=None,
                 broadcast_from_rank0: bool = True, hooks: bool = True, exact_unused: bool = False,
                 always_sync: bool = False):
        self.group = process_group
        # always_sync: run the bucket protocol (collectives included) even in a one-rank group -- a single GPU can
        # then exercise the RCCL backend end to end (tests/test_distributed_gpu.py); normally a lone rank skips it
        self.always_sync = always_sync
        self.exact_unused = exact_unused
        self._flags: Optional[torch.Tensor] = None
        self._given_key, self._given_any = None, None  # reduce_grads: cached 'some rank has a gradient' mask
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params: List[torch.nn.Parameter] = [p for p in module.parameters() if p.requires_grad]
        self._index = {p: i for i, p in enumerate(self.params)}
        self.bucket_bytes = bucket_bytes
        self._fired_order: List[int] = []      # step 0: arrival order (parameter indices)
        self._fired = set()                    # parameters that reported a gradient in the current step
        self._buckets: Optional[List[_Bucket]] = None
        self._where: Dict[torch.nn.Parameter, tuple] = {}
        self._next = 0                         # next bucket to launch (strict index order)
        # hooks=False: no autograd hooks; the caller hands the finished gradients to reduce_grads() (the hipGraph
        # step: forward + backward are replayed as one graph, which cannot contain the collectives)
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params] if hooks else []
        if self.world > 1 and broadcast_from_rank0:
            with torch.no_grad():
                for t in list(module.parameters()) + [b for b in module.buffers() if b.is_floating_point()]:
                    dist.broadcast(t.data, 0, group=process_group)

    # ---- bucket construction -----------------------------------------------------------------------

    def _shared_order(self, local_order: List[int]) -> List[int]:
        """rank 0's arrival order followed by the parameters it saw no gradient for -- identical on every rank"""
        seen = set(local_order)
        order = list(local_order) + [i for i in range(len(self.params)) if i not in seen]
        if self.world > 1 or self.always_sync:
            dev = self.params[0].device
            t = torch.tensor(order, dtype=torch.int64, device=dev if dev.type == "cuda" else "cpu")
            dist.broadcast(t, 0, group=self.group)
            order = [int(v) for v in t.tolist()]
        return order

    def _build_buckets(self, local_order: List[int]) -> None:
        order = self._shared_order(local_order)
        groups, cur, cur_bytes = [], [], 0
        for i in order:
            p = self.params[i]
            nbytes = p.numel() * 4
            if cur and cur_bytes + nbytes > self.bucket_bytes:
                groups.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            groups.append(cur)
        self._buckets = []
        for plist in groups:
            total = sum(p.numel() for p in plist)
            flat = torch.zeros(total, dtype=torch.float32, device=plist[0].device)
            slots, off = [], 0
            for p in plist:
                slots.append((p, off))
                self._where[p] = (len(self._buckets), off)
                if p.dtype == torch.float32:
                    p._ffa_grad_buf = flat[off: off + p.numel()].view_as(p)
                off += p.numel()
            self._buckets.append(_Bucket(flat, slots))

    def _view(self, p) -> torch.Tensor:
        bi, off = self._where[p]
        return self._buckets[bi].flat[off: off + p.numel()].view_as(p)

    # ---- per-step protocol -------------------------------------------------------------------------

    def _adopt(self, p) -> None:
        """make p.grad the bucket view, moving the values there if the producer wrote elsewhere"""
        view = self._view(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if self.world == 1 and not self.always_sync:
            return
        if self._buckets is None:
            self._fired_order.append(self._index[p])
            return
        b = self._buckets[self._where[p][0]]
        if p in self._fired and not b.launched:  # a second backward pass accumulated into the view already
            return
        if b.launched:
            raise RuntimeError("GradSync: a gradient arrived after its bucket was reduced -- call finish() once per "
                               "optimizer step, after the last backward of the step")
        self._adopt(p)
        self._fired.add(p)
        b.pending -= 1
        self._launch_ready()

    def _launch_ready(self) -> None:
        while self._next < len(self._buckets) and self._buckets[self._next].pending == 0:
            self._launch(self._buckets[self._next])
            self._next += 1

    def _launch(self, b: _Bucket) -> None:
        b.launched = True
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        """Call after backward, before optimizer.step(): completes the step's collectives (zero-filling the slots of
        parameters that produced no gradient), waits for them (stream-side on RCCL, no host stall) and turns the
        sums into means.  Afterwards every parameter of the module has its mean gradient in ``p.grad``."""
        if self.world == 1 and not self.always_sync:
            return
        if self._buckets is None:  # first step: learn the arrival order, no overlap yet
            self._build_buckets(self._fired_order)
            self._fired_order = []
            for p in self.params:
                if p.grad is not None:
                    self._adopt(p)
                    self._fired.add(p)
            for b in self._buckets:
                b.pending = 0
        for b in self._buckets[self._next:]:
            for p, off in b.slots:
                if p not in self._fired:
                    b.flat[off: off + p.numel()].zero_()
            self._launch(b)
        missing = [p for p in self.params if p not in self._fired]
        flag_work = None
        if self.exact_unused:  # always issued, always last: the sequence of collectives never depends on the data
            if self._flags is None:
                self._flags = torch.zeros(len(self.params), dtype=torch.float32, device=self.params[0].device)
            self._flags.zero_()
            if self._fired:
                self._flags[torch.tensor([self._index[p] for p in self._fired], device=self._flags.device)] = 1.0
            flag_work = dist.all_reduce(self._flags, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        for b in self._buckets:
            b.work.wait()
            b.work = None
            b.flat.mul_(1.0 / self.world)
            b.pending, b.launched = len(b.slots), False
        if flag_work is not None:
            flag_work.wait()
        if missing:
            # another rank may have produced a gradient for these: the replicas must apply the same update
            somewhere = self._flags.tolist() if self.exact_unused else None
            for p in missing:
                if somewhere is None or somewhere[self._index[p]] > 0:
                    p.grad = self._view(p)
        self._fired = set()
        self._next = 0

    def reduce_grads(self, params, grads) -> None:
        """Mean over ranks of ``grads`` (one tensor per parameter of ``params``, e.g. the static gradient tensors of a
        replayed hipGraph), through the same flat buckets: copy in (a no-op for gradients that already live in their
        bucket view), one all-reduce per bucket, scale, and point every ``p.grad`` at its bucket view for the optimizer.

        RCCL runs the bucket collectives in issue order on its own stream, so all of them are launched before the first
        wait (the compute stream waits stream-side, the host never blocks).  gloo (rehearsals: several ranks sharing
        one GPU) stages device tensors through host buffers in its worker threads, and a collective handed to it while
        the device is still busy with the replayed graph took 0.25-15 s (round 2's "13 s/step"; tools/ddp_graph_probe.py:
        30 ms when the stream is idle at the call, 4.7 s when it is not, one bucket at a time or three), so for gloo on
        device tensors the stream is drained first and the buckets go one by one.

        ``exact_unused``: parameters outside ``params`` keep ``grad = None`` unless another rank handed one in (the set
        is exchanged once per distinct ``params`` set, not per step: under a replayed graph it never changes)."""
        if self._buckets is None:
            self._build_buckets([self._index[p] for p in params])
        given = set()
        for p, g in zip(params, grads):
            view = self._view(p)
            if g.data_ptr() != view.data_ptr():
                view.copy_(g)
            given.add(p)
        # slots of parameters without a gradient take part with zeros: merged into contiguous runs (they sit together at
        # the end of the arrival order: one fill instead of one per parameter), the run list cached per set of givers
        zkey = (len(given), frozenset(self._index[p] for p in given)) if len(given) < len(self.params) else None
        if zkey is None:
            self._zero_runs = (None, [])
        elif getattr(self, "_zero_runs", (None, None))[0] != zkey:
            runs = []
            for bi, b in enumerate(self._buckets):
                cur = None
                for p, off in b.slots:
                    if p not in given:
                        if cur is not None and cur[1] == off:
                            cur[1] = off + p.numel()
                        else:
                            cur = [off, off + p.numel()]
                            runs.append((bi, cur))
                    else:
                        cur = None
            self._zero_runs = (zkey, runs)
        for bi, (lo, hi) in self._zero_runs[1]:
            self._buckets[bi].flat[lo:hi].zero_()
        if self.world > 1 or self.always_sync:
            backend = dist.get_backend(self.group)
            serial = backend == "gloo"
            # RCCL averages inside the collective (ncclAvg: sum, then one division in f32 -- the same value as the
            # multiplication by 1 / world for the power-of-two world sizes of a node): no scaling pass over the 98 MB
            avg = backend == "nccl"
            if serial and self._buckets[0].flat.is_cuda:
                torch.cuda.current_stream().synchronize()
            works = []
            for b in self._buckets:
                w = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group,
                                    async_op=True)
                if serial:
                    w.wait()
                works.append(w)
            for b, w in zip(self._buckets, works):
                if not serial:
                    w.wait()
                if not avg:
                    b.flat.mul_(1.0 / self.world)
        somewhere = None
        if self.exact_unused and len(given) < len(self.params):
            key = frozenset(self._index[p] for p in given)
            if self._given_key != key:
                flags = torch.zeros(len(self.params), dtype=torch.float32, device=self.params[0].device)
                if given:
                    flags[torch.tensor(sorted(key), device=flags.device)] = 1.0
                if self.world > 1 or self.always_sync:
                    dist.all_reduce(flags, op=dist.ReduceOp.SUM, group=self.group)
                self._given_key, self._given_any = key, [v > 0 for v in flags.tolist()]
            somewhere = self._given_any
        for p in self.params:
            if somewhere is None or somewhere[self._index[p]]:
                p.grad = self._view(p)
            elif p not in given:
                p.grad = None

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        for p in self.params:
            if hasattr(p, "_ffa_grad_buf"):
                del p._ffa_grad_buf


# ------------------------------------------------------------------------------------------------------
# rank-sharded batch feed


def world_info():
    """(rank, world size) of the default process group, (0, 1) without one"""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def ensure_process_group(device: Optional[torch.device] = None) -> bool:
    """Initialise the default process group from the torchrun environment when WORLD_SIZE > 1 and nobody did yet
    (backend: FFA_DIST_BACKEND, default "nccl" = RCCL).  Returns True when a group exists afterwards."""
    import os
    if dist.is_initialized():
        return True
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return False
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver (RCCL needs it)
    backend = os.environ.get("FFA_DIST_BACKEND", "nccl")
    if backend == "nccl" and device is not None:
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
    return True


class ShardedLoader:
    """What Lightning's DDP strategy does to the DataLoaders (DistributedSampler injection), for the loaders
    HipTrainer is given.

    * a ``torch.utils.data.DataLoader`` over a map-style dataset is rebuilt around a
      ``DistributedSampler(shuffle=<the loader shuffled>, seed=<shared>, drop_last=<the loader's>)``: rank r sees
      every world-th sample of the shared permutation, ``set_epoch`` is called per epoch (Lightning does the same);
    * a loader that is already sharded (its sampler is a DistributedSampler) is used as it is;
    * any other iterable of batch dicts is taken to yield GLOBAL batches: rank r keeps rows
      [r * B / world, (r + 1) * B / world) of every tensor (lists likewise).  With ``drop_last`` (training, the
      reference's setting) a batch whose leading size is not a multiple of the world size is dropped, so every rank
      runs the same number of steps; without it (validation / prediction) the ragged batch is split unevenly.
    """

    def __init__(self, loader: Iterable, rank: int, world: int, shuffle: Optional[bool] = None, seed: int = 0,
                 drop_last: bool = True):
        self.rank, self.world, self.drop_last = rank, world, drop_last
        self.epoch = 0
        self.sampler = None
        self.inner = loader
        self.mode = "slice"
        if world <= 1:
            self.mode = "plain"
            return
        from torch.utils.data import DataLoader, DistributedSampler, RandomSampler
        if isinstance(loader, DataLoader) and loader.batch_sampler is not None and hasattr(loader.dataset, "__len__"):
            if isinstance(loader.sampler, DistributedSampler):
                self.mode, self.sampler = "plain", loader.sampler
                return
            if loader.batch_size is None:
                # a user batch_sampler (or batch_size=None): its batches cannot be re-derived from a DistributedSampler
                # without knowing its constructor (Lightning re-instantiates the known ones and raises for the rest)
                raise TypeError("ShardedLoader: a DataLoader with a custom batch_sampler / batch_size=None cannot be "
                                "sharded automatically; build it around a DistributedSampler(num_replicas=world, rank=rank) "
                                "yourself (it is then used as it is) or hand over an iterable of global batches")
            shuf = isinstance(loader.sampler, RandomSampler) if shuffle is None else shuffle
            self.sampler = DistributedSampler(loader.dataset, num_replicas=world, rank=rank, shuffle=shuf, seed=seed,
                                              drop_last=loader.drop_last)
            extra = {}
            if loader.num_workers > 0:  # only valid together with worker processes
                extra["prefetch_factor"] = loader.prefetch_factor
                extra["multiprocessing_context"] = loader.multiprocessing_context
            if getattr(loader, "pin_memory_device", ""):
                extra["pin_memory_device"] = loader.pin_memory_device
            self.inner = DataLoader(loader.dataset, batch_size=loader.batch_size, sampler=self.sampler,
                                    num_workers=loader.num_workers, collate_fn=loader.collate_fn,
                                    pin_memory=loader.pin_memory, drop_last=loader.drop_last,
                                    timeout=loader.timeout, worker_init_fn=loader.worker_init_fn,
                                    persistent_workers=loader.persistent_workers, generator=loader.generator, **extra)
            self.mode = "plain"

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch
        if self.sampler is not None:
            self.sampler.set_epoch(epoch)

    def __len__(self) -> int:
        """batches this rank will yield.  Slice mode: global batches whose leading size does not divide by the world size
        are skipped under ``drop_last`` -- counted exactly when the inner iterable says how it batches (``batch_size``,
        ``drop_last`` and a sized ``dataset``, as a DataLoader over an iterable-free dataset does), otherwise
        ``len(inner)`` (an upper bound by at most the one ragged batch)."""
        n = len(self.inner)
        if self.mode != "slice":
            return n
        bs, ds = getattr(self.inner, "batch_size", None), getattr(self.inner, "dataset", None)
        if not isinstance(bs, int) or bs <= 0 or ds is None or not hasattr(ds, "__len__"):
            return n
        full, rem = divmod(len(ds), bs)
        ragged = 1 if (rem and not getattr(self.inner, "drop_last", False)) else 0
        if self.drop_last:
            return (full if bs % self.world == 0 else 0) + (ragged if rem % self.world == 0 else 0)
        kept = lambda m: 1 if self._bounds(m)[1] > self._bounds(m)[0] else 0
        return full * kept(bs) + (kept(rem) if ragged else 0)

    def _bounds(self, n: int):
        per = n // self.world if self.drop_last else (n + self.world - 1) // self.world
        lo = min(self.rank * per, n)
        return lo, min(lo + per, n)

    def __iter__(self) -> Iterator:
        if self.mode == "plain":
            yield from self.inner
            return
        for batch in self.inner:
            n = None
            for v in batch.values():
                if torch.is_tensor(v) and v.dim() > 0:
                    n = v.shape[0]
                    break
            if n is None:
                raise ValueError("ShardedLoader: batch without a tensor to take the batch size from")
            if self.drop_last and n % self.world != 0:
                continue  # training: every rank must run the same number of steps (each one is a collective)
            lo, hi = self._bounds(n)
            if hi <= lo:
                continue  # evaluation: the ragged last batch gave this rank nothing
            yield {k: (v[lo:hi] if (torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n) or
                       (isinstance(v, (list, tuple)) and len(v) == n) else v) for k, v in batch.items()}


def all_reduce_sum_(tensors: Iterable[torch.Tensor]) -> None:
    """in-place sum over ranks of metric state (confusion matrices, loss totals): what torchmetrics' state sync and
    ``sync_dist=True`` do in the reference (flair_hub/tasks/tasks_module.py:215-236, 296-300)"""
    rank, world = world_info()
    if world <= 1:
        return
    for t in tensors:
        if dist.get_backend() == "gloo" and t.is_cuda:  # gloo rehearsals on a GPU box: reduce through the host
            h = t.detach().cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
