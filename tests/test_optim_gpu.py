"""flairhip.optim.HipAdamW / HipAdam (csrc/optim.hip, ffa_adamw_multi) against torch.optim.AdamW / Adam -- the optimizer
the reference constructs at flair_hub/tasks/tasks_module.py:385-389 -- on the GPU box: same trajectory, same state."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(cuda, seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 5, 7, 7), (64,), (128, 64, 3, 3), (19,), (4097,), (3, 4096), (1,), (256, 128, 3, 3), (33, 17)]
    shapes += [(7,)] * 80  # more tensors than one launch's descriptor table holds (72)
    ps = [torch.randn(*s, generator=g).to(cuda).requires_grad_() for s in shapes]
    # a parameter that is a misaligned view (element offset 1 of a larger buffer): the kernel's scalar path
    buf = torch.randn(5001, generator=g).to(cuda)
    ps.append(buf[1:].detach().requires_grad_())
    return ps


@pytest.mark.parametrize("kind", ["adamw", "adam"])
def test_trajectory_and_state_match_torch(cuda, kind):
    from flairhip.optim import HipAdam, HipAdamW
    ours_p, ref_p = _params(cuda, 1), _params(cuda, 1)
    kw = dict(lr=3e-3, betas=(0.9, 0.95), weight_decay=0.02 if kind == "adamw" else 0.01)
    ours = (HipAdamW if kind == "adamw" else HipAdam)(ours_p, **kw)
    ref = (torch.optim.AdamW if kind == "adamw" else torch.optim.Adam)(ref_p, fused=True, **kw)
    g = torch.Generator().manual_seed(9)
    for step in range(6):
        lr = 3e-3 * (1.0 + 0.3 * step)
        for opt in (ours, ref):
            for grp in opt.param_groups:
                grp["lr"] = lr
        for a, b in zip(ours_p, ref_p):
            if step == 2 and a.numel() == 19:
                a.grad = b.grad = None  # a parameter that skips a step keeps its own step count
                continue
            gr = torch.randn(a.shape, generator=g).to(cuda)
            a.grad, b.grad = gr.clone(), gr.clone()
        ours.step()
        ref.step()
    torch.cuda.synchronize()
    worst = 0.0
    for a, b in zip(ours_p, ref_p):
        sa, sb = ours.state[a], ref.state[b]
        assert float(sa["step"]) == float(sb["step"])
        for x, y in ((a, b), (sa["exp_avg"], sb["exp_avg"]), (sa["exp_avg_sq"], sb["exp_avg_sq"])):
            d = (x.detach() - y.detach()).abs().max().item()
            worst = max(worst, d / (y.detach().abs().max().item() + 1e-12))
    assert worst <= 2e-6, worst  # same formula and order as torch's fused kernel; at most an ulp of libm difference


def test_state_dict_round_trips_through_torchs_optimizer(cuda):
    from flairhip.optim import HipAdamW
    ps, qs = _params(cuda, 2)[:6], _params(cuda, 2)[:6]
    ours = HipAdamW(ps, lr=1e-3, weight_decay=0.01)
    for p in ps:
        p.grad = torch.ones_like(p)
    ours.step()
    ref = torch.optim.AdamW(qs, lr=1e-3, weight_decay=0.01, fused=True, capturable=True)
    import copy
    ref.load_state_dict(copy.deepcopy(ours.state_dict()))  # (torch keeps the tensors it is handed: no aliasing wanted)
    with torch.no_grad():
        for p, q in zip(ps, qs):
            q.copy_(p)
    for p, q in zip(ps, qs):
        p.grad, q.grad = torch.full_like(p, 0.5), torch.full_like(q, 0.5)
    ours.step()
    ref.step()
    torch.cuda.synchronize()
    for p, q in zip(ps, qs):
        assert (p - q).detach().abs().max().item() <= 2e-6 * float(q.detach().abs().max())


def test_rejects_what_it_does_not_implement(cuda):
    from flairhip.optim import HipAdamW
    with pytest.raises(NotImplementedError):
        HipAdamW([torch.zeros(3, device=cuda, requires_grad=True)], amsgrad=True)
    p = torch.zeros(3, requires_grad=True)  # CPU tensor: no CPU path
    p.grad = torch.ones(3)
    with pytest.raises(TypeError):
        HipAdamW([p]).step()
