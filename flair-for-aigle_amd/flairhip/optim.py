"""HipAdamW / HipAdam: the optimizer step of the training hot path as ONE kernel launch over every parameter tensor
(csrc/optim.hip, ``ffa_adamw_multi``: three launches for the U-Net's 186 tensors).

Drop-in for ``torch.optim.AdamW`` / ``torch.optim.Adam`` as the reference constructs them
(flair_hub/tasks/tasks_module.py:385-389: lr, betas, weight_decay; amsgrad / foreach / differentiable are not used there
and are refused here).  Same update rule and operation order as torch's fused implementation and the SAME state layout
(``step`` -- a device f32 scalar per parameter --, ``exp_avg``, ``exp_avg_sq``), so ``state_dict()`` /
``load_state_dict()`` are interchangeable with torch's optimizer and LR schedulers see the usual ``param_groups``.
``lr`` may be a float or a device tensor (the captured step keeps it on the device: flairhip.graph.make_capturable);
everything the step launches is capturable into a hipGraph.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Tuple

import torch

from . import lib as _l


class HipAdamW(torch.optim.Optimizer):
    decoupled = True  # AdamW: p -= lr * wd * p;  Adam: g += wd * p

    def __init__(self, params: Iterable, lr=1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, amsgrad: bool = False, maximize: bool = False, capturable: bool = True):
        if amsgrad:
            raise NotImplementedError("HipAdamW: amsgrad is not implemented (the reference does not use it)")
        if not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or eps < 0 or weight_decay < 0:
            raise ValueError("HipAdamW: invalid hyper-parameter")
        # the keys torch's Adam / AdamW keep in their param_groups (2.10: AdamW = Adam with decoupled_weight_decay), so that
        # a state dict moves between the two implementations without changing the update rule
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False,
                                      maximize=maximize, capturable=capturable, foreach=None, fused=None,
                                      differentiable=False, decoupled_weight_decay=self.decoupled))
        self._tables = {}  # group index -> (pointer key, ctypes arrays): rebuilt only when a tensor moved

    def _arrays(self, gi, params):
        st = self.state
        key = tuple((p.data_ptr(), p.grad.data_ptr(), st[p]["exp_avg"].data_ptr(), st[p]["exp_avg_sq"].data_ptr(),
                     st[p]["step"].data_ptr(), p.numel()) for p in params)
        hit = self._tables.get(gi)
        if hit is None or hit[0] != key:
            n = len(params)
            cols = [(C.c_void_p * n)(*[k[j] for k in key]) for j in range(5)]
            hit = (key, cols, (C.c_longlong * n)(*[k[5] for k in key]))
            self._tables[gi] = hit
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _l.load()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            for p in params:
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise TypeError("HipAdamW: f32 parameters and gradients on the GPU (there is no CPU path)")
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise ValueError("HipAdamW: parameters and gradients must be contiguous")
                s = self.state[p]
                if not s:
                    s["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            torch._foreach_add_([self.state[p]["step"] for p in params], 1.0)  # one launch; the kernel reads the new count
            lr = group["lr"]
            if not torch.is_tensor(lr):  # eager use with a host-side learning rate: a fill kernel, no host-to-device copy
                lr = torch.full((), float(lr), dtype=torch.float32, device=params[0].device)
            elif lr.device != params[0].device or lr.dtype != torch.float32:
                lr = lr.to(params[0].device, torch.float32)
            _, cols, numel = self._arrays(gi, params)
            b1, b2 = group["betas"]
            _l.check(lib.ffa_adamw_multi(len(params), *cols, numel, lr.data_ptr(), float(b1), float(b2), float(group["eps"]),
                                         float(group["weight_decay"]), 1 if self.decoupled else 0,
                                         1 if group["maximize"] else 0, torch.cuda.current_stream().cuda_stream),
                      "adamw_multi")
        return loss


class HipAdam(HipAdamW):
    decoupled = False

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **kw)
