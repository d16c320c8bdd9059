"""U-TAE Sentinel branch on libflairhip (flairhip/utae.py + csrc/temporal.hip) against
  * tests/golden/utae_eval.npz -- outputs of the REFERENCE'S OWN UTAE class (flair_hub/models/multitemp_model.py,
    run by tests/golden/gen_utae_goldens.py) on seeded weights: logits, every decoder map, attention masks, without
    and with padded dates;
  * oracle/utae.py (the CPU restatement pinned by the same golden) on further shapes;
  * torch.nn.functional for the individual kernels."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import ROOT
from test_oracle_goldens import _utae_state_shapes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
PARAMS = dict(encoder_widths=[64, 64, 64, 128], decoder_widths=[32, 32, 64, 128], out_conv=[32, 19], str_conv_k=3,
              str_conv_s=1, str_conv_p=1, agg_mode="att_group", encoder_norm="group", n_head=16, d_model=256, d_k=4,
              return_maps=True, pad_value=0, padding_mode="reflect")


def _model(cuda, precision):
    from flairhip.utae import HipUTAE
    from oracle.seeded_weights import fill_utae_state_dict
    sd = fill_utae_state_dict({k: torch.zeros(s) for k, s in _utae_state_shapes().items()})
    net = HipUTAE(10, precision=precision, **PARAMS)
    net.load_state_dict(sd)  # the reference's key names and shapes, strict
    return net.to(cuda).eval(), sd


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_utae_matches_the_references_own_outputs(cuda, precision):
    net, _ = _model(cuda, precision)
    d = np.load(os.path.join(GOLD, "utae_eval.npz"))
    for tag in ("a", "b"):
        x, pos = torch.tensor(d[f"{tag}_x"]).to(cuda), torch.tensor(d[f"{tag}_pos"]).to(cuda)
        with torch.no_grad():
            logits, maps = net(x, batch_positions=pos)
            net.return_maps = False
            _, att = net(x, batch_positions=pos, return_att=True)
            net.return_maps = True
        ref = d[f"{tag}_logits"]
        scale = np.abs(ref).max()
        tol = 1e-4 * max(1.0, scale) if precision == "fp32" else 0.04 * scale
        assert logits.shape == ref.shape and logits.dtype == torch.float32
        assert np.abs(logits.cpu().numpy() - ref).max() <= tol, tag
        assert np.abs(att.cpu().numpy() - d[f"{tag}_att"]).max() <= (1e-5 if precision == "fp32" else 8e-2)
        for i, m in enumerate(maps):
            r = d[f"{tag}_map{i}"]
            t = 1e-4 * max(1.0, np.abs(r).max()) if precision == "fp32" else 0.04 * np.abs(r).max()
            assert m.shape == r.shape and np.abs(m.cpu().numpy() - r).max() <= t, (tag, i)
        if precision == "fp32":
            agree = (logits.argmax(1).cpu().numpy() == ref.argmax(1)).mean()
            assert agree >= 0.999


def test_utae_other_shapes_against_the_oracle(cuda):
    from oracle.utae import utae_forward
    net, sd = _model(cuda, "fp32")
    g = torch.Generator().manual_seed(5)
    for (B, T, H, W, pads) in [(1, 3, 8, 14, []), (3, 9, 10, 10, [(2, 8), (2, 7), (0, 8)]), (2, 2, 16, 6, [(1, 1)])]:
        x = torch.randn(B, T, 10, H, W, generator=g)
        pos = torch.sort(torch.randint(0, 365, (B, T), generator=g), dim=1).values.float()
        for b, t in pads:
            x[b, t] = 0.0
        ref_logits, ref_maps, ref_att = utae_forward(sd, x, pos)
        with torch.no_grad():
            logits, maps = net(x.to(cuda), batch_positions=pos.to(cuda))
        assert (logits.cpu() - ref_logits).abs().max().item() <= 1e-4 * max(1.0, ref_logits.abs().max().item())
        for m, r in zip(maps, ref_maps):
            assert (m.cpu() - r).abs().max().item() <= 1e-4 * max(1.0, r.abs().max().item())


def test_utae_refuses_what_it_does_not_cover(cuda):
    from flairhip.utae import HipUTAE
    with pytest.raises(NotImplementedError):
        HipUTAE(10)  # the original U-TAE defaults (strided 4/2/1 convolutions) are not FLAIR's configuration
    net, _ = _model(cuda, "fp32")
    net.train()
    with torch.no_grad(), pytest.raises(NotImplementedError):  # training mode outside a training step
        net(torch.randn(1, 2, 10, 8, 8, device=cuda), batch_positions=torch.zeros(1, 2, device=cuda))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_group_norm_and_reflect_pad_kernels(cuda, dtype):
    from flairhip import ops
    g = torch.Generator().manual_seed(2)
    N, C, H, W = 5, 64, 7, 11
    x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    res = torch.randn(N, C, H, W, generator=g)
    xq, rq = x.to(dtype).float(), res.to(dtype).float()
    xd = xq.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    rd = rq.permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    tol = 1e-5 if dtype == torch.float32 else 2 ** -6
    got = ops.group_norm(xd, gamma.to(cuda), beta.to(cuda), 4, relu=True, residual=rd)
    ref = rq + F.relu(F.group_norm(xq, 4, gamma, beta, 1e-5))
    assert (got.float().cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    pad = ops.reflect_pad1(xd)
    assert torch.equal(pad.float().cpu().permute(0, 3, 1, 2), F.pad(xq, (1, 1, 1, 1), mode="reflect"))
    # per-pixel sequences: [B*T, h, w, C], statistics over the T dates x C/16 channels of every pixel
    B, T, h, w, C2 = 2, 5, 3, 4, 128
    s = torch.randn(B, T, C2, h, w, generator=g)
    g2, b2 = torch.rand(C2, generator=g) + 0.5, torch.randn(C2, generator=g)
    sq = s.to(dtype).float()
    sd_ = sq.reshape(B * T, C2, h, w).permute(0, 2, 3, 1).contiguous().to(dtype).to(cuda)
    got = ops.group_norm_seq(sd_, B, T, g2.to(cuda), b2.to(cuda), 16)
    seq = sq.permute(0, 3, 4, 2, 1).reshape(B * h * w, C2, T)  # [pixels, C, T] as LTAE2d builds it
    ref = F.group_norm(seq, 16, g2, b2, 1e-5).reshape(B, h, w, C2, T).permute(0, 4, 1, 2, 3).reshape(B * T, h, w, C2)
    assert (got.float().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_utae_training_step_matches_the_references_own_autograd(cuda, precision):
    """tests/golden/utae_train.npz: training-mode forward + backward of the REFERENCE'S OWN UTAE (BatchNorm batch
    statistics, padded dates, both dropouts at p = 0): loss, class scores, every parameter gradient (norm + values) and
    the BatchNorm running statistics after the step"""
    from flairhip import nn as hnn
    net, _ = _model(cuda, precision)
    net.train()
    net.mlp_dropout = net.attn_dropout = 0.0
    d = np.load(os.path.join(GOLD, "utae_train.npz"))
    x, pos = torch.tensor(d["x"]).to(cuda), torch.tensor(d["pos"]).to(cuda)
    tgt = torch.tensor(d["target"]).to(cuda)
    logits_nhwc, maps, attn = net.forward_nhwc(x, pos)
    logits = hnn.logits_view(logits_nhwc, 19)
    crit = hnn.HipCrossEntropyLoss(num_classes=19).to(cuda)
    loss = crit(logits, tgt)
    loss.backward()
    torch.cuda.synchronize()
    f32 = precision == "fp32"
    ref_logits = d["logits"]
    got = logits.detach().float().cpu().numpy()
    assert np.abs(got - ref_logits).max() <= (2e-4 if f32 else 0.12) * max(1.0, np.abs(ref_logits).max())
    assert abs(loss.item() - float(d["loss"])) <= (1e-4 if f32 else 2e-2) * float(d["loss"])
    bad, coss = [], []
    for k, p in net.named_parameters():
        assert p.grad is not None, k
        g = p.grad.detach().float().cpu().flatten()
        ref_norm = float(d["norm__" + k])
        ref = torch.tensor(d["grad__" + k])
        sample = g if g.numel() <= 4096 else g[:: max(1, g.numel() // 2048)]
        assert sample.shape == ref.shape, k
        if ref_norm < 1e-7:  # biases in front of a training-mode BatchNorm: exact zero in exact arithmetic
            assert g.norm().item() <= (1e-4 if f32 else 2e-2), k
            continue
        ratio = g.double().norm().item() / ref_norm
        cos = F.cosine_similarity(sample, ref, dim=0).item() if ref.norm() > 0 else 1.0
        lim_r, lim_c = ((0.98, 1.02), 0.999) if f32 else ((0.7, 1.4), 0.6)  # bf16 storage through ~20 GroupNorm layers on 10 x 10 maps; fp32 pins the arithmetic
        coss.append(cos)
        if not (lim_r[0] <= ratio <= lim_r[1]) or cos < lim_c:
            bad.append((k, round(ratio, 4), round(cos, 4)))
    assert not bad, bad[:10]
    assert sum(coss) / len(coss) >= (0.9995 if f32 else 0.8)
    sd = net.state_dict()
    for k in d.files:
        if k.startswith("stat__"):
            ref = d[k]
            got = sd[k[len("stat__"):]].float().cpu().numpy()
            assert np.abs(got - ref).max() <= (1e-4 if f32 else 5e-2) * max(1.0, np.abs(ref).max()), k


def test_utae_dropout_and_no_grad_behaviour(cuda):
    net, _ = _model(cuda, "bf16")
    net.train()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 10, 8, 8, generator=g).to(cuda)
    pos = torch.tensor([[1.0, 40.0, 90.0, 200.0]] * 2).to(cuda)
    a = net.forward_nhwc(x, pos)[0]
    b = net.forward_nhwc(x, pos)[0]
    assert not torch.equal(a, b)  # the reference's two dropouts draw new masks per call
    a.float().sum().backward()
    assert all(p.grad is not None for p in net.parameters())
    with torch.no_grad(), pytest.raises(NotImplementedError):
        net.forward_nhwc(x, pos)  # training mode without autograd is neither a training step nor an evaluation
