"""conv7x7_stem_kernel (csrc/conv7x7_stem.hip: the ResNet stem -- 7x7 stride-2 pad-3, <= 8 real input channels at pitch 16
-> 64 channels -- as four adjacent taps x 8 channels per K = 32 MFMA step) against torch's CPU fp32 conv2d on the
bf16-rounded operands (torchvision's `conv1` as smp's ResNetEncoder keeps it, reached from
flair_hub/models/monotemp_model.py:68-92) and against conv_igemm_kernel<7, 7, 2> on the same operands: ragged tiles, odd
sizes, several tiles per persistent block, statistics epilogue, the evaluation form (folded BatchNorm scale, bias, ReLU)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def to_nhwc(x_nchw, dev, cp):
    B, C, H, W = x_nchw.shape
    out = torch.zeros(B, H, W, cp, dtype=torch.float32)
    out[..., :C] = x_nchw.permute(0, 2, 3, 1)
    return out.to(BF).to(dev).contiguous()


def from_nhwc(x, C):
    return x[..., :C].float().cpu().permute(0, 3, 1, 2).contiguous()


def rq(x):
    return x.to(BF).float()


# cin, B, H, W, grid cap (0 = default)
CASES = [(5, 2, 64, 96, 0), (5, 3, 50, 70, 2), (3, 2, 17, 33, 0), (8, 1, 128, 64, 3), (1, 2, 31, 31, 1), (5, 2, 256, 256, 0),
         (5, 1, 7, 9, 0), (4, 3, 8, 8, 0), (5, 1, 1, 1, 0), (2, 2, 2, 130, 0)]


@pytest.mark.parametrize("case", CASES, ids=[f"c{c[0]}_b{c[1]}_{c[2]}x{c[3]}_g{c[4]}" for c in CASES])
def test_stem_conv_matches_torch_and_igemm(cuda, monkeypatch, case):
    from flairhip import ops, lib as L
    cin, B, H, W, cap = case
    if cap:
        monkeypatch.setenv("FFA_STEM_GRID", str(cap))
    g = torch.Generator().manual_seed(cin * 7 + H + W)
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(64, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
    xd = to_nhwc(x, cuda, 16)
    pw = ops.pack_conv_weight(w.to(cuda), BF, 2, 16)
    assert pw.bco & L.BCO_STEM, "the stem layer did not get the stem layout"
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    rows = ops.conv_stat_rows(B, Ho, Wo, pw)
    st = torch.zeros(rows * 2 * 64, dtype=torch.float32, device=cuda)
    got = ops.conv2d(xd, pw, 3, 64, stats=st)
    assert got.shape == (B, Ho, Wo, 64)
    ref = F.conv2d(rq(x), rq(w), None, 2, 3)
    torch.cuda.synchronize()
    err = (from_nhwc(got, 64) - ref).abs().max().item()
    assert err <= float(ref.abs().max()) * 2 ** -7, err  # one bf16 rounding of the output
    # statistics = sums / sums of squares of the STORED (rounded) values
    parts = st.view(rows, 2, 64).double().sum(0).cpu()
    vals = got.float().cpu().double().reshape(-1, 64)
    assert torch.allclose(parts[0], vals.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(parts[1], (vals * vals).sum(0), rtol=1e-5, atol=1e-3)
    # the generic kernel on the same operands: same products, different summation order
    monkeypatch.setenv("FFA_STEM", "0")
    pw_old = ops.pack_conv_weight(w.to(cuda), BF, 2, 16)
    assert not (pw_old.bco & L.BCO_STEM)
    old = ops.conv2d(xd, pw_old, 3, 64)
    torch.cuda.synchronize()
    assert (got.float() - old.float()).abs().max().item() <= float(ref.abs().max()) * 2 ** -6
    monkeypatch.delenv("FFA_STEM")
    again = ops.conv2d(xd, pw, 3, 64)
    torch.cuda.synchronize()
    assert torch.equal(got, again)


def test_stem_evaluation_form_with_folded_scale_bias_and_relu(cuda):
    from flairhip import ops, lib as L
    g = torch.Generator().manual_seed(8)
    B, cin, H, W = 2, 5, 96, 80
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(64, cin, 7, 7, generator=g) / (cin * 49) ** 0.5
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g)
    pw = ops.pack_conv_weight(w.to(cuda), BF, 2, 16, scale=scale.to(cuda))
    assert pw.bco & L.BCO_STEM
    got = ops.conv2d(to_nhwc(x, cuda, 16), pw, 3, 64, bias=shift.to(cuda), relu=True)
    ref = F.relu(F.conv2d(rq(x), rq(w * scale[:, None, None, None]), shift, 2, 3))
    torch.cuda.synchronize()
    assert (from_nhwc(got, 64) - ref).abs().max().item() <= float(ref.abs().max()) * 2 ** -7


def test_an_input_with_more_than_eight_channels_keeps_the_generic_kernel(cuda):
    from flairhip import ops, lib as L
    w = torch.randn(64, 10, 7, 7).to(cuda)
    assert not (ops.pack_conv_weight(w, BF, 2, 16).bco & L.BCO_STEM)


WG_CASES = [(5, 2, 64, 96), (5, 3, 50, 70), (3, 2, 17, 33), (8, 1, 128, 64), (5, 2, 256, 256), (5, 1, 7, 9), (4, 3, 8, 8),
            (2, 2, 2, 130)]


@pytest.mark.parametrize("cin,B,H,W", WG_CASES, ids=[f"c{c[0]}_b{c[1]}_{c[2]}x{c[3]}" for c in WG_CASES])
def test_stem_weight_gradient_matches_torch_and_the_older_kernel(cuda, monkeypatch, cin, B, H, W):
    """stem_wgrad8_kernel (two adjacent taps x 8 channels per 16 MFMA columns; every halo fragment read once for the four
    tile rows it serves) against torch.nn.grad.conv2d_weight on the bf16-rounded operands and against stem_wgrad_kernel"""
    from flairhip import ops
    g = torch.Generator().manual_seed(cin + H * 3 + W)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(B, cin, H, W, generator=g)
    dy = torch.randn(B, 64, Ho, Wo, generator=g)
    xd, dyd = to_nhwc(x, cuda, 16), to_nhwc(dy, cuda, 64)
    got = ops.conv_wgrad(xd, dyd, 64, cin, 7, 7, 2, 3)
    monkeypatch.setenv("FFA_STEM_WGRAD8", "0")
    old = ops.conv_wgrad(xd, dyd, 64, cin, 7, 7, 2, 3)
    ref = torch.nn.grad.conv2d_weight(rq(x), (64, cin, 7, 7), rq(dy), stride=2, padding=3)
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    tol = 2e-4 * scale * max(1.0, (B * Ho * Wo / 2000) ** 0.5)
    assert (got.cpu() - ref).abs().max().item() <= tol
    assert (got - old).abs().max().item() <= tol
    monkeypatch.delenv("FFA_STEM_WGRAD8")
    again = ops.conv_wgrad(xd, dyd, 64, cin, 7, 7, 2, 3)
    torch.cuda.synchronize()
    assert torch.equal(got, again)
