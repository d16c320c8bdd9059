"""conv3x3_ring_kernel (LDS-DMA weight ring, csrc/conv3x3_ring.hip) against torch's CPU fp32 conv2d -- the ATen op
smp's Conv2dReLU / torchvision's BasicBlock reach from flair_hub/models/monotemp_model.py:68-92 -- and against the
conv_igemm kernels on the same operands; every tile configuration, the persistent multi-tile walk, the statistics
epilogue and the fused BatchNorm + ReLU prologue."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def to_nhwc(x_nchw, dtype, dev, cp=None):
    B, C, H, W = x_nchw.shape
    cp = cp or (C + 15) // 16 * 16
    out = torch.zeros(B, H, W, cp, dtype=torch.float32)
    out[..., :C] = x_nchw.permute(0, 2, 3, 1)
    return out.to(dtype).to(dev).contiguous()


def from_nhwc(x, C):
    return x[..., :C].float().cpu().permute(0, 3, 1, 2).contiguous()


def rq(x, dtype):
    return x.to(dtype).float()


CASES = [
    # cin, cout, H, W, cfg (FFA_RING_CFG: 0 = the library's choice), grid cap (0 = default)
    (64, 64, 16, 32, 0, 0),
    (64, 64, 40, 72, 0, 8),       # ragged tiles, several tiles per persistent block
    (128, 128, 24, 64, 0, 16),
    (192, 64, 32, 32, 0, 0),      # three 64-channel... six chunks
    (96, 128, 16, 48, 1, 8),      # odd number of chunks: the halo buffers swap roles between tiles
    (256, 256, 32, 32, 0, 0),
    (512, 512, 16, 16, 0, 0),     # 16x16 tiles
    (64, 192, 12, 20, 2, 8),      # 16x16 tiles forced on a wide map, ragged
    (128, 128, 32, 64, 4, 8),     # 128 co x 8x32 px blocks, eight waves
    (64, 128, 20, 40, 4, 8),
    (128, 256, 16, 32, 4, 8),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[f"c{c[0]}-{c[1]}_{c[2]}x{c[3]}_cfg{c[4]}" for c in CASES])
def test_ring_conv_matches_torch_and_igemm(cuda, monkeypatch, dtype, case):
    from flairhip import ops, lib as L
    cin, cout, H, W, cfg, cap = case
    monkeypatch.setenv("FFA_RING", "1")
    if cfg:
        monkeypatch.setenv("FFA_RING_CFG", str(cfg))
    if cap:
        monkeypatch.setenv("FFA_RING_GRID", str(cap))
    g = torch.Generator().manual_seed(cin * 7 + cout + H)
    B = 3
    x = torch.randn(B, cin, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, H, W, generator=g)
    xd = to_nhwc(x, dtype, cuda)
    cop = ops.pad_channels(cout)
    pw = ops.pack_conv_weight(w.to(cuda), dtype, 1, xd.shape[-1])
    assert pw.bco & L.BCO_RING, "eligible layer did not get the ring layout"
    pw_ig = ops.pack_conv_weight(w.to(cuda), dtype, 1, xd.shape[-1], allow_ring=False)
    bd = torch.zeros(cop, device=cuda)
    bd[:cout] = bias.to(cuda)
    rd = to_nhwc(res, dtype, cuda, cop)

    # plain
    y = ops.conv2d(xd, pw, 1, cop)
    y_ig = ops.conv2d(xd, pw_ig, 1, cop)
    ref = F.conv2d(rq(x, dtype), rq(w, dtype), None, padding=1)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    tol = 1e-4 * max(1.0, scale) if dtype == torch.float32 else scale * 2 ** -7
    assert (from_nhwc(y, cout) - ref).abs().max().item() <= tol
    # same operands, other kernel: only the f32 summation order differs
    tol2 = 2e-5 * max(1.0, scale) if dtype == torch.float32 else scale * 2 ** -7
    assert (y.float() - y_ig.float()).abs().max().item() <= tol2
    if cop > cout:
        assert float(y[..., cout:].float().abs().max()) == 0.0

    # epilogue: bias + residual + relu, and the statistics of the stored tensor
    rows = ops.conv_stat_rows(B, H, W, pw)
    st = torch.full((rows * 2 * cop,), float("nan"), device=cuda)
    y2 = ops.conv2d(xd, pw, 1, cop, bias=bd, residual=rd, relu=True, stats=st)
    ref2 = (ref + bias.view(1, -1, 1, 1) + rq(res, dtype)).relu()
    torch.cuda.synchronize()
    assert (from_nhwc(y2, cout) - ref2).abs().max().item() <= tol * 2
    part = st.view(rows, 2, cop).double().sum(0).cpu()
    stored = y2.double().cpu().view(-1, cop)
    assert torch.allclose(part[0], stored.sum(0), rtol=1e-5, atol=1e-3)
    assert torch.allclose(part[1], (stored * stored).sum(0), rtol=1e-5, atol=1e-3)
    # deterministic
    y3 = ops.conv2d(xd, pw, 1, cop, bias=bd, residual=rd, relu=True)
    torch.cuda.synchronize()
    assert torch.equal(y2, y3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("cin,cout,H,W,cfg", [(64, 64, 24, 40, 0), (128, 64, 16, 16, 0), (128, 128, 32, 32, 4)])
def test_ring_prologue_equals_materialised_batchnorm_relu(cuda, monkeypatch, dtype, cin, cout, H, W, cfg):
    """conv(relu(x * sc + sh)) with the normalisation done while the halo is staged == the same kernel run on the
    tensor ffa_bn_apply writes, bit for bit (same fma, same rounding to the storage type, zero padding applied after
    the normalisation)."""
    from flairhip import ops
    monkeypatch.setenv("FFA_RING", "1")
    if cfg:
        monkeypatch.setenv("FFA_RING_CFG", str(cfg))
    monkeypatch.setenv("FFA_RING_GRID", "8")
    g = torch.Generator().manual_seed(cin + cout)
    B = 2
    x = to_nhwc(torch.randn(B, cin, H, W, generator=g), dtype, cuda)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(cuda)
    sc = (torch.rand(cin, generator=g) + 0.5).to(cuda)
    sh = (torch.randn(cin, generator=g) * 0.5 + 0.3).to(cuda)  # positive shifts: relu(shift) != 0 would show in the padding
    cop = ops.pad_channels(cout)
    pw = ops.pack_conv_weight(w, dtype, 1, cin)
    xn = ops.bn_apply(x, sc, sh, relu=True)
    ref = ops.conv3x3_ring(xn, pw, cop)
    got = ops.conv3x3_ring(x, pw, cop, pro_scale=sc, pro_shift=sh)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


def test_kernel_timing_session_times_the_ring_launch_itself(cuda):
    """ffa_ktime_begin / _end (bench.py's roofline pass): inside a session the ring16 launch carries its own start / stop
    events; the result is unchanged, one timing per launch comes back with the kernel's tag, and a session cannot nest."""
    import ctypes as C
    from flairhip import lib as L, ops
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 32, 64, 128, generator=g).to(cuda).to(torch.bfloat16)
    w = (torch.randn(128, 128, 3, 3, generator=g) / 34.0).to(cuda)
    pw = ops.pack_conv_weight(w, torch.bfloat16, 1, 128)
    ref = ops.conv2d(x, pw, 1, 128)
    L.check(lib.ffa_ktime_begin(8), "ktime_begin")
    assert lib.ffa_ktime_begin(8) != 0  # no nesting
    got = [ops.conv2d(x, pw, 1, 128) for _ in range(3)]
    ms, tags = (C.c_float * 8)(), (C.c_int * 8)()
    n = lib.ffa_ktime_end(ms, tags, 8)
    assert n == 3 and all(tags[i] == 1 for i in range(3)) and all(0.0 < ms[i] < 5.0 for i in range(3))
    assert all(torch.equal(t, ref) for t in got)
    assert lib.ffa_ktime_end(ms, tags, 8) < 0  # no session open any more
    assert torch.equal(ops.conv2d(x, pw, 1, 128), ref)
