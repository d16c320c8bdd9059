"""INTEGRATION.md's ctypes stub is executable documentation: run it as written (from the repository root) and compare
with torch's weighted cross-entropy, so that the stub cannot drift from include/flairhip.h."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_ctypes_stub_of_the_integration_guide_runs(cuda, monkeypatch):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    stub = [b for b in blocks if "def fused_ce" in b]
    assert len(stub) == 1
    monkeypatch.chdir(ROOT)  # the stub opens the library by its path relative to the repository root
    ns = {}
    exec(compile(stub[0], "INTEGRATION.md", "exec"), ns)
    g = torch.Generator().manual_seed(3)
    logits = torch.zeros(2, 24, 40, 32)
    logits[..., :19] = torch.randn(2, 24, 40, 19, generator=g) * 2
    target = torch.randint(0, 19, (2, 24, 40), generator=g).to(torch.uint8)
    w = torch.tensor([1.0] * 15 + [0.0] * 4)
    loss, pred = ns["fused_ce"](logits.to(cuda), target.to(cuda), w.to(cuda))
    ref = F.cross_entropy(logits[..., :19].permute(0, 3, 1, 2), target.long(), weight=w)
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert torch.equal(pred.cpu(), logits[..., :19].argmax(-1).to(torch.uint8))
