"""conv3x3_thin_kernel (csrc/conv3x3_thin.hip: <= 32-channel bf16 layers on v_mfma_f32_16x16x32_bf16, weights in
registers, LDS-DMA halo ring) against torch's CPU fp32 conv2d -- the ATen op smp's decoder tail and SegmentationHead
reach from flair_hub/models/monotemp_model.py:68-92 -- and against conv_igemm on the same operands: forward and dgrad
operands, the statistics epilogue, bias / residual / ReLU, the nearest-x2 two-source form, the pooled split epilogue,
ragged tiles and several tiles per persistent block."""
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


# cin, cout, H, W, grid cap (0 = default)
CASES = [(16, 16, 32, 64, 0), (16, 16, 40, 72, 3), (16, 19, 32, 32, 0), (32, 32, 16, 64, 0), (32, 32, 24, 40, 2),
         (32, 16, 16, 32, 0), (19, 16, 48, 96, 4), (30, 30, 20, 34, 0), (16, 32, 33, 47, 5)]


@pytest.mark.parametrize("case", CASES, ids=[f"c{c[0]}-{c[1]}_{c[2]}x{c[3]}_g{c[4]}" for c in CASES])
def test_thin_conv_matches_torch_and_igemm(cuda, monkeypatch, case):
    from flairhip import ops, lib as L
    cin, cout, H, W, cap = case
    if cap:
        monkeypatch.setenv("FFA_THIN_GRID", str(cap))
    g = torch.Generator().manual_seed(cin * 11 + cout + H)
    B = 3
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    cip = 32 if cin == 19 else ops.pad_channels(cin)   # a 19-class gradient arrives with the logits pitch
    cop = 32 if cout == 19 else ops.pad_channels(cout)
    xd = to_nhwc(x, cuda, cip)
    pw = ops.pack_conv_weight(w.to(cuda), BF, 1, cip)
    assert pw.bco & L.BCO_THIN, "eligible layer did not get the thin layout"
    pw_ig = ops.pack_conv_weight(w.to(cuda), BF, 1, cip, allow_thin=False)
    assert not (pw_ig.bco & L.BCO_THIN)
    bd = torch.zeros(cop, device=cuda)
    bd[:cout] = bias.to(cuda)
    rd = to_nhwc(res, cuda, cop)

    y = ops.conv2d(xd, pw, 1, cop)
    y_ig = ops.conv2d(xd, pw_ig, 1, cop)
    ref = F.conv2d(rq(x), rq(w), None, padding=1)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    tol = scale * 2 ** -7
    assert (from_nhwc(y, cout) - ref).abs().max().item() <= tol
    assert (y.float() - y_ig.float()).abs().max().item() <= tol  # same operands, other kernel: f32 summation order
    if cop > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0

    # epilogue: bias + residual + relu, statistics of the stored tensor
    rows = ops.conv_stat_rows(B, H, W, pw)
    st = torch.full((rows * 2 * cop,), float("nan"), device=cuda)
    y2 = ops.conv2d(xd, pw, 1, cop, bias=bd, residual=rd, relu=True, stats=st)
    ref2 = (ref + bias.view(1, -1, 1, 1) + rq(res)).relu()
    torch.cuda.synchronize()
    assert (from_nhwc(y2, cout) - ref2).abs().max().item() <= tol * 2
    part = st.view(rows, 2, cop).double().sum(0).cpu()
    stored = y2.double().cpu().view(-1, cop)
    assert torch.allclose(part[0], stored.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[1], (stored * stored).sum(0), rtol=1e-5, atol=1e-3)
    y3 = ops.conv2d(xd, pw, 1, cop, bias=bd, residual=rd, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)  # deterministic

    # dgrad operand (transposed, mirrored taps) through the same kernel
    dy = torch.randn(B, cout, H, W, generator=g)
    dyd = to_nhwc(dy, cuda, cop)
    pwt = ops.pack_conv_weight(w.to(cuda), BF, 1, cop, transpose=True)
    assert pwt.bco & L.BCO_THIN
    dx = ops.conv2d(dyd, pwt, 1, cip)
    refd = torch.nn.grad.conv2d_input(x.shape, rq(w), rq(dy), padding=1)
    torch.cuda.synchronize()
    assert (from_nhwc(dx, cin) - refd).abs().max().item() <= float(refd.abs().max()) * 2 ** -7


@pytest.mark.parametrize("c1,cout,Hl,Wl,cap", [(32, 16, 16, 32, 0), (32, 16, 20, 24, 3), (16, 16, 24, 16, 0)])
def test_thin_two_source_and_pooled_forms(cuda, monkeypatch, c1, cout, Hl, Wl, cap):
    """conv over nearest_x2(lo) without the upsampled tensor, and its adjoint: the dgrad whose output is 2x2 sum-pooled"""
    from flairhip import ops, lib as L
    if cap:
        monkeypatch.setenv("FFA_THIN_GRID", str(cap))
    g = torch.Generator().manual_seed(c1 + cout + Hl)
    B = 2
    lo = torch.randn(B, c1, Hl, Wl, generator=g)
    w = (torch.randn(cout, c1, 3, 3, generator=g) / (c1 * 9) ** 0.5)
    lod = to_nhwc(lo, cuda, c1)
    cop = ops.pad_channels(cout)
    pw = ops.pack_conv_weight(w.to(cuda), BF, 1, c1)
    assert pw.bco & L.BCO_THIN
    rows = ops.conv_stat_rows(B, 2 * Hl, 2 * Wl, pw)
    st = torch.full((rows * 2 * cop,), float("nan"), device=cuda)
    y = ops.conv2d_upcat(lod, None, pw, cop, stats=st)
    assert y is not None
    up = F.interpolate(rq(lo), scale_factor=2, mode="nearest")
    ref = F.conv2d(up, rq(w), None, padding=1)
    torch.cuda.synchronize()
    assert (from_nhwc(y, cout) - ref).abs().max().item() <= float(ref.abs().max()) * 2 ** -7
    part = st.view(rows, 2, cop).double().sum(0).cpu()
    stored = y.double().cpu().view(-1, cop)
    assert torch.allclose(part[0], stored.sum(0), rtol=1e-5, atol=1e-3)
    # adjoint: d lo = 2x2 sums of the full-resolution input gradient
    dy = torch.randn(B, cout, 2 * Hl, 2 * Wl, generator=g)
    dyd = to_nhwc(dy, cuda, cop)
    pwt = ops.pack_conv_weight(w.to(cuda), BF, 1, cop, transpose=True)
    assert pwt.bco & L.BCO_THIN
    pair = ops.conv2d_dgrad_upcat(dyd, pwt, c1, 0)
    assert pair is not None
    dlo, dskip = pair
    assert dskip is None
    dup = torch.nn.grad.conv2d_input(up.shape, rq(w), rq(dy), padding=1)
    refl = F.avg_pool2d(dup, 2) * 4
    torch.cuda.synchronize()
    assert (from_nhwc(dlo, c1) - refl).abs().max().item() <= float(refl.abs().max()) * 2 ** -7


WG_CASES = [(16, 16, 32, 64), (16, 16, 21, 45), (16, 19, 40, 40), (32, 32, 24, 40), (32, 16, 16, 96), (30, 30, 17, 33)]


@pytest.mark.parametrize("cin,cout,H,W", WG_CASES, ids=[f"c{c[0]}-{c[1]}_{c[2]}x{c[3]}" for c in WG_CASES])
def test_thin_wgrad_matches_torch_and_the_general_kernel(cuda, monkeypatch, cin, cout, H, W):
    """conv3x3_thin_wgrad_kernel (16x16x32 MFMA on transposed LDS reads, LDS-DMA staging, one slab per persistent
    block) against torch.nn.grad.conv2d_weight on the bf16-rounded operands and against conv_wgrad_kernel"""
    from flairhip import ops
    g = torch.Generator().manual_seed(cin * 5 + cout + W)
    B = 3
    x = torch.randn(B, cin, H, W, generator=g)
    dy = torch.randn(B, cout, H, W, generator=g)
    cip = ops.pad_channels(cin)
    cop = 32 if cout == 19 else ops.pad_channels(cout)
    xd, dyd = to_nhwc(x, cuda, cip), to_nhwc(dy, cuda, cop)
    got = ops.conv_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
    monkeypatch.setenv("FFA_THIN_WGRAD", "0")
    old = ops.conv_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
    ref = torch.nn.grad.conv2d_weight(rq(x), (cout, cin, 3, 3), rq(dy), padding=1)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    assert (got.cpu() - ref).abs().max().item() <= 2e-4 * scale * max(1.0, (B * H * W / 2000) ** 0.5)
    assert (got - old).abs().max().item() <= 2e-4 * scale * max(1.0, (B * H * W / 2000) ** 0.5)
    monkeypatch.delenv("FFA_THIN_WGRAD")
    again = ops.conv_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
    torch.cuda.synchronize()
    assert torch.equal(got, again)  # fixed summation order


WG64_CASES = [(64, 64, 3, 32, 64), (64, 128, 2, 40, 48), (128, 64, 2, 17, 33), (128, 128, 5, 64, 64), (256, 64, 1, 8, 96),
              # maps narrower than 32: the 16 x 16-tile instantiation (two image rows per 32-pixel k-step)
              (256, 256, 3, 16, 16), (64, 128, 2, 12, 20), (128, 128, 4, 8, 8), (512, 64, 2, 33, 16), (64, 64, 1, 5, 31)]


@pytest.mark.parametrize("cin,cout,B,H,W", WG64_CASES, ids=[f"c{c[0]}-{c[1]}_b{c[2]}_{c[3]}x{c[4]}" for c in WG64_CASES])
def test_wgrad64_matches_torch_and_the_general_kernel(cuda, monkeypatch, cin, cout, B, H, W):
    """conv3x3_wgrad64_kernel (64 x 64-channel blocks on maps >= 32 wide: LDS-DMA double buffer, transposed LDS reads,
    two k groups merged through LDS) against torch.nn.grad.conv2d_weight on the bf16-rounded operands and against
    conv_wgrad_kernel<2, 2, 2> (FFA_WGRAD64=0); ragged tiles, several splits per channel block, fixed summation order"""
    from flairhip import ops
    g = torch.Generator().manual_seed(cin + 3 * cout + W)
    x = torch.randn(B, cin, H, W, generator=g)
    dy = torch.randn(B, cout, H, W, generator=g)
    xd, dyd = to_nhwc(x, cuda, cin), to_nhwc(dy, cuda, cout)
    got = ops.conv_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
    monkeypatch.setenv("FFA_WGRAD64", "0")
    old = ops.conv_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
    ref = torch.nn.grad.conv2d_weight(rq(x), (cout, cin, 3, 3), rq(dy), padding=1)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    tol = 2e-4 * scale * max(1.0, (B * H * W / 2000) ** 0.5)
    assert (got.cpu() - ref).abs().max().item() <= tol
    assert (got - old).abs().max().item() <= tol
    monkeypatch.delenv("FFA_WGRAD64")
    again = ops.conv_wgrad(xd, dyd, cout, cin, 3, 3, 1, 1)
    torch.cuda.synchronize()
    assert torch.equal(got, again)


def test_thin_wgrad_of_the_upsampled_input(cuda):
    from flairhip import ops
    g = torch.Generator().manual_seed(77)
    B, c1, cout, Hl, Wl = 2, 32, 16, 20, 24
    lo = torch.randn(B, c1, Hl, Wl, generator=g)
    dy = torch.randn(B, cout, 2 * Hl, 2 * Wl, generator=g)
    lod, dyd = to_nhwc(lo, cuda, c1), to_nhwc(dy, cuda, ops.pad_channels(cout))
    got = ops.conv_wgrad_upcat(lod, None, dyd, cout)
    assert got is not None
    up = F.interpolate(rq(lo), scale_factor=2, mode="nearest")
    ref = torch.nn.grad.conv2d_weight(up, (cout, c1, 3, 3), rq(dy), padding=1)
    torch.cuda.synchronize()
    assert (got.cpu() - ref).abs().max().item() <= 5e-4 * float(ref.abs().max())


@pytest.mark.parametrize("cin,cout,H,W,up", [(16, 16, 32, 64, False), (32, 32, 24, 40, False), (16, 19, 40, 40, False),
                                              (32, 16, 20, 24, True), (64, 64, 24, 40, False), (128, 64, 16, 32, False)])
def test_normalise_on_load_equals_the_materialised_tensor(cuda, monkeypatch, cin, cout, H, W, up):
    """ffa_conv2d_pro / ffa_conv_wgrad_pro: conv and weight gradient of relu(x * sc + sh) with the normalisation done on
    the staged input (thin kernels: LDS fix-up of the DMA'd halo; ring16: the same; conv_wgrad_kernel: at the LDS store)
    == the same kernels run on the tensor ffa_bn_apply writes, bit for bit (zero padding applied after the
    normalisation: positive shifts would show otherwise)."""
    from flairhip import ops
    monkeypatch.setenv("FFA_THIN_GRID", "6")
    monkeypatch.setenv("FFA_RING_GRID", "8")
    monkeypatch.setenv("FFA_WGRAD64", "0")  # the prologue lives in conv_wgrad_kernel: compare like with like
    g = torch.Generator().manual_seed(cin + cout + H)
    B = 2
    cip = ops.pad_channels(cin)
    cop = 32 if cout == 19 else ops.pad_channels(cout)
    Hs, Ws = (H // 2, W // 2) if up else (H, W)
    x = to_nhwc(torch.randn(B, cin, Hs, Ws, generator=g), cuda, cip)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(cuda)
    sc = torch.zeros(cip, device=cuda)
    sh = torch.zeros(cip, device=cuda)
    sc[:cin] = (torch.rand(cin, generator=g) + 0.5).to(cuda)
    sh[:cin] = (torch.randn(cin, generator=g) * 0.5 + 0.3).to(cuda)
    dy = to_nhwc(torch.randn(B, cout, H, W, generator=g), cuda, cop)
    pw = ops.pack_conv_weight(w, BF, 1, cip)
    assert ops.pro_supported(pw, BF, up)
    xn = ops.bn_apply(x, sc, sh, relu=True)
    rows = ops.conv_stat_rows(B, H, W, pw)
    st_a = torch.zeros(rows * 2 * cop, device=cuda)
    st_b = torch.zeros(rows * 2 * cop, device=cuda)
    if up:
        ref = ops.conv2d_upcat(xn, None, pw, cop, stats=st_a)
        ref_w = ops.conv_wgrad_upcat(xn, None, dy, cout)
    else:
        ref = ops.conv2d(xn, pw, 1, cop, stats=st_a)
        ref_w = ops.conv_wgrad(xn, dy, cout, cin, 3, 3, 1, 1)
    got = ops.conv2d_pro(x, pw, cop, sc, sh, stats=st_b, up=up)
    got_w = ops.conv_wgrad_pro(x, dy, cout, cin, sc, sh, up=up)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert torch.equal(st_a, st_b)
    assert torch.equal(got_w, ref_w)
