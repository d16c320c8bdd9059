"""Swin-Transformer + UPerNet (BASELINE config 4; the fork's zonal configuration) on the MI355X
against the CPU oracle oracle/swin_upernet.py (torch fp32 restatement of timm's Swin + smp's UPerNet; parity unpinned,
see its header) and, per kernel, against the torch.nn.functional op it replaces."""
import math

import pytest
import torch
import torch.nn.functional as F

import helpers  # noqa: F401  (sys.path)

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _bf(x):
    return x.to(torch.bfloat16)


# ---------------------------------------------------------------------------------------------------- token GEMM

@pytest.mark.parametrize("M,K,N,act,res", [
    (256, 96, 96, 0, False), (256, 96, 288, 0, False), (300, 128, 384, 0, True), (128, 384, 96, 1, False),
    (1024, 512, 2048, 1, False), (1024, 2048, 512, 0, True), (72, 32, 8, 0, False), (4096, 160, 136, 1, False),
    (129, 64, 264, 0, True),
])
def test_linear_matches_torch(M, K, N, act, res):
    from flairhip import ops
    g = torch.Generator().manual_seed(M * 7 + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    xb, wb = _bf(x), _bf(w)
    ref = F.linear(xb.float(), wb.float(), b)
    if act:
        ref = F.gelu(ref)
    if res:
        ref = _bf(ref).float() + _bf(r).float()  # the GEMM result is rounded to bf16 before the residual add
    y = ops.linear(xb.to(DEV), wb.to(DEV), b.to(DEV), act=act, residual=None if r is None else _bf(r).to(DEV))
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err
    # no-bias, in-place residual
    if res:
        rr = _bf(r).to(DEV).clone()
        y2 = ops.linear(xb.to(DEV), wb.to(DEV), None, residual=rr, out=rr)
        ref2 = _bf(F.linear(xb.float(), wb.float())).float() + _bf(r).float()
        assert (y2.float().cpu() - ref2).abs().max().item() <= 2e-2 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("M,K,N,act,res", [
    (1024, 512, 512, 0, False), (2048, 2048, 512, 0, True), (1024, 512, 2048, 1, False), (512, 256, 768, 0, True),
    (300, 256, 256, 1, True), (8192, 1024, 1024, 0, False), (1000, 4096, 1024, 0, True),
])
def test_linear_256_tile_kernel_matches_torch(M, K, N, act, res):
    """shapes that the dispatcher sends to gemm256_bf16_kernel (LDS-DMA ring, 256 x 256 block tile)"""
    from flairhip import ops
    g = torch.Generator().manual_seed(M + 3 * K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g) if res else None
    xb, wb = _bf(x), _bf(w)
    ref = F.linear(xb.float(), wb.float(), b)
    if act:
        ref = F.gelu(ref)
    if res:
        ref = _bf(ref).float() + _bf(r).float()
    for _ in range(3):  # the ring has no per-call state: repeated launches agree bit for bit
        y = ops.linear(xb.to(DEV), wb.to(DEV), b.to(DEV), act=act, residual=None if r is None else _bf(r).to(DEV))
        if _ == 0:
            first = y.clone()
        assert torch.equal(y, first)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err


def test_linear_rejects_bad_shapes():
    from flairhip import ops
    from flairhip.lib import FlairHipError
    x = torch.zeros(64, 48, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(64, 48, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(FlairHipError):
        ops.linear(x, w)  # K not a multiple of 32
    with pytest.raises(ValueError):
        ops.linear(x.float(), w)  # operands of different dtypes
    assert ops.linear(x.float(), w.float()).abs().max().item() == 0.0  # f32 parity kernel: any K


# ---------------------------------------------------------------------------------------------------- small kernels

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [96, 128, 1024, 4096])
def test_layer_norm(dtype, C):
    from flairhip import ops
    g = torch.Generator().manual_seed(C)
    x = (torch.randn(2, 5, 7, C, generator=g) * 2 + 0.5).to(dtype)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    ref = F.layer_norm(x.float(), (C,), w, b, 1e-5)
    y = ops.layer_norm(x.to(DEV), w.to(DEV), b.to(DEV)).float().cpu()
    tol = 2e-5 if dtype == torch.float32 else 4e-2
    assert (y - ref).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_merge_norm(dtype):
    from flairhip import ops
    from oracle.swin_upernet import PatchMerging
    g = torch.Generator().manual_seed(3)
    C = 96
    x = torch.randn(2, 8, 12, C, generator=g).to(dtype)
    pm = PatchMerging(C)
    pm.norm.weight.data = torch.rand(4 * C, generator=g) + 0.5
    pm.norm.bias.data = torch.randn(4 * C, generator=g)
    xf = x.float()
    B, H, W, _ = xf.shape
    gathered = xf.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)
    ref = pm.norm(gathered).detach()
    y = ops.patch_merge_norm(x.to(DEV), pm.norm.weight.data.to(DEV), pm.norm.bias.data.to(DEV)).float().cpu()
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() <= (2e-5 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_space_to_depth_and_gelu_and_pool(dtype):
    from flairhip import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 8, 12, 16, generator=g).to(dtype)
    y = ops.space_to_depth(x.to(DEV), 4).cpu()
    ref = x.reshape(2, 2, 4, 3, 4, 16).permute(0, 1, 3, 2, 4, 5).reshape(2, 2, 3, 256)
    assert torch.equal(y, ref)
    z = ops.gelu(x.to(DEV)).float().cpu()
    assert (z - F.gelu(x.float())).abs().max().item() <= (1e-6 if dtype == torch.float32 else 2e-2)
    big = torch.randn(2, 16, 16, 32, generator=g).to(dtype)
    for s in (1, 2, 3, 6):
        p = ops.adaptive_avg_pool(big.to(DEV), s).float().cpu()
        ref = F.adaptive_avg_pool2d(big.float().permute(0, 3, 1, 2), s).permute(0, 2, 3, 1)
        assert (p - ref).abs().max().item() <= (1e-5 if dtype == torch.float32 else 1e-2), s


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("hw_in,hw_out", [((4, 4), (16, 16)), ((6, 6), (16, 16)), ((16, 16), (16, 16)),
                                           ((32, 32), (16, 16)), ((1, 1), (16, 16)), ((8, 12), (32, 48))])
def test_bilinear_slice(dtype, align, hw_in, hw_out):
    from flairhip import ops
    g = torch.Generator().manual_seed(11)
    C = 16
    x = torch.randn(2, *hw_in, C, generator=g).to(dtype)
    add = torch.randn(2, *hw_out, C, generator=g).to(dtype)
    ref = F.interpolate(x.float().permute(0, 3, 1, 2), size=hw_out, mode="bilinear", align_corners=align)
    ref = ref.permute(0, 2, 3, 1)
    wide = torch.full((2, *hw_out, 48), 7.0, dtype=dtype, device=DEV)
    ops.bilinear_slice(x.to(DEV), hw_out, out=wide, offset=16, align_corners=align)
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    got = wide.float().cpu()
    assert (got[..., 16:32] - ref).abs().max().item() <= tol
    assert torch.all(got[..., :16] == 7.0) and torch.all(got[..., 32:] == 7.0)
    y = ops.bilinear_slice(x.to(DEV), hw_out, addend=add.to(DEV), align_corners=align).float().cpu()
    assert (y - (ref + add.float())).abs().max().item() <= 2 * tol


# ---------------------------------------------------------------------------------------------------- window attention

def _attention_reference(qkv, bias, table, heads, ws, shift):
    """timm's SwinTransformerBlock._attn between the two projections, on a given qkv tensor: padding tokens project to
    the qkv bias (norm1's output is zero-padded before the projection)."""
    from oracle.swin_upernet import relative_position_index, shifted_window_mask, window_partition, window_reverse
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    x = qkv
    if shift:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    Hp, Wp = H + ph, W + pw
    full = bias.view(1, 1, 1, C3).expand(B, Hp, Wp, C3).clone()
    full[:, :H, :W] = x
    xw = window_partition(full, ws).view(-1, ws * ws, C3)
    N = ws * ws
    q, k, v = xw.reshape(-1, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4).unbind(0)
    attn = (q * (C // heads) ** -0.5) @ k.transpose(-2, -1)
    rpb = table[relative_position_index(ws).view(-1)].view(N, N, -1).permute(2, 0, 1)
    attn = attn + rpb.unsqueeze(0)
    if shift:
        mask = shifted_window_mask(Hp, Wp, ws, shift)
        nW = mask.shape[0]
        attn = (attn.view(-1, nW, heads, N, N) + mask.unsqueeze(1).unsqueeze(0)).view(-1, heads, N, N)
    out = (attn.softmax(-1) @ v).transpose(1, 2).reshape(-1, ws, ws, C)
    out = window_reverse(out, ws, Hp, Wp)[:, :H, :W]
    if shift:
        out = torch.roll(out, shifts=(shift, shift), dims=(1, 2))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("H,W,heads,ws,shift", [
    (14, 14, 3, 7, 0), (14, 14, 3, 7, 3), (16, 16, 3, 7, 3), (16, 20, 4, 7, 0), (8, 8, 6, 8, 0), (4, 4, 3, 4, 0),
    (24, 24, 4, 12, 6), (32, 32, 4, 12, 6), (32, 32, 4, 12, 0), (16, 16, 8, 12, 6), (12, 12, 32, 12, 0),
])
def test_window_attention(dtype, H, W, heads, ws, shift):
    from flairhip import ops
    g = torch.Generator().manual_seed(H * 31 + ws + shift)
    C = heads * 32
    qkv = torch.randn(2, H, W, 3 * C, generator=g).to(dtype)
    bias = torch.randn(3 * C, generator=g)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    bias_used = bias if dtype == torch.float32 else _bf(bias).float()
    ref = _attention_reference(qkv.float(), bias_used, table, heads, ws, shift)
    y = ops.window_attention(qkv.to(DEV), bias.to(DEV), table.to(DEV), heads, ws, shift, 32 ** -0.5).float().cpu()
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    assert (y - ref).abs().max().item() <= tol


# ---------------------------------------------------------------------------------------------------- whole model

def _randomise(oracle, seed):
    g = torch.Generator().manual_seed(seed)
    for m in oracle.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_var.data = torch.rand(m.bias.shape, generator=g) * 0.5 + 0.75
        elif isinstance(m, torch.nn.LayerNorm):
            m.weight.data = torch.rand(m.weight.shape, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
        elif isinstance(m, torch.nn.Linear):
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (1.0 / math.sqrt(m.weight.shape[1]))
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
    for n, p in oracle.named_parameters():
        if n.endswith("relative_position_bias_table"):
            p.data = torch.randn(p.shape, generator=g) * 0.5
    oracle.segmentation_head[0].bias.data = torch.randn(oracle.segmentation_head[0].bias.shape, generator=g) * 0.1


@pytest.mark.parametrize("name,ch,size", [
    ("swin_tiny_patch4_window7_224", 5, 128),       # maps 32 / 16 / 8 / 4: padding 32 -> 35, 16 -> 21, 8 -> 14, window 4
    ("swin_base_patch4_window12_384", 3, 256),      # maps 64 / 32 / 16 / 8: 64 -> 72, 32 -> 36, 16 -> 24, window 8
])
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_swin_upernet_matches_oracle(name, ch, size, precision):
    from flairhip import ops
    from flairhip.swin import SwinUPerNet
    from oracle.swin_upernet import SwinUPerNet as OracleNet
    torch.manual_seed(1)
    oracle = OracleNet(name, ch, 19, size).eval()
    _randomise(oracle, 7)
    model = SwinUPerNet(name, ch, 19, size)
    missing, unexpected = model.load_state_dict(oracle.state_dict(), strict=True)
    assert not missing and not unexpected
    model = model.to(DEV).eval()
    dtype = torch.float32 if precision == "fp32" else torch.bfloat16
    x = torch.randn(2, ch, size, size, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = oracle(x)
        feats_ref = oracle.encoder(x)
        xn = ops.nchw_to_nhwc(x.to(DEV), dtype, ops.pad_channels(ch))
        feats = model.encoder(xn)
        y = model(xn)
    assert [f.shape[-1] for f in feats] == [ops.pad_channels(ch)] + oracle.encoder.out_channels[1:]
    assert feats[1].shape == (2, size // 2, size // 2, 0)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()
    for f, fr in zip(feats[2:], feats_ref[2:]):
        got = f.float().cpu().permute(0, 3, 1, 2)
        assert got.shape == fr.shape
        assert rel(got, fr) <= (2e-5 if precision == "fp32" else 3e-2)
    logits = y[..., :19].float().cpu().permute(0, 3, 1, 2)
    assert logits.shape == ref.shape
    if precision == "fp32":
        assert (logits - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
        assert torch.equal(logits.argmax(1), ref.argmax(1)) or (logits.argmax(1) == ref.argmax(1)).float().mean() > 0.9999
    else:
        assert rel(logits, ref) <= 4e-2
        assert (logits.argmax(1) == ref.argmax(1)).float().mean().item() >= 0.97


def test_swin_state_dict_accepts_timm_spelling_and_cpu_tensors_are_refused():
    from flairhip.swin import SwinUPerNet
    m = SwinUPerNet("swin_tiny_patch4_window7_224", 3, 5, 64)
    sd = {k.replace("layers_", "layers."): v for k, v in m.state_dict().items()}
    assert any(".layers." in k for k in sd)
    m2 = SwinUPerNet("swin_tiny_patch4_window7_224", 3, 5, 64)
    m2.load_state_dict(sd, strict=True)
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    with pytest.raises(RuntimeError):  # no CPU path in the product
        m2.train().encoder(torch.zeros(1, 64, 64, 16))


# ---------------------------------------------------------------------------------------------------- FLAIR_HUB_Model glue

def _flair_swin(precision, sizes):
    from flairhip.configs import fusion_unet_config
    from flair_hub.tasks.module_setup import build_segmentation_module
    from oracle.seeded_weights import fill_swin_state_dict
    cfg = fusion_unet_config(precision=precision)
    cfg["models"]["monotemp_model"]["arch"] = "swin_tiny_patch4_window7_224-upernet"
    cfg["models"]["monotemp_model"]["drop_path_rate"] = 0.0  # the golden's training step has stochastic depth off
    task = build_segmentation_module(cfg, sizes, "train")
    task.model.load_state_dict(fill_swin_state_dict(task.model.state_dict()))
    return task.cuda().eval(), cfg


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_swin_encoders_fused_training_step_matches_the_reference_task(precision):
    """tests/golden/swin_two_mod.*, "train": one step of the REFERENCE's SegmentationTask on that model (training-mode
    BatchNorm, two task losses with weights 1 / 0.5, the auxiliary decoders outside the loss): loss, predictions, the
    set of parameters without gradient, every parameter's gradient norm and eight sampled gradients"""
    import json
    import os
    import numpy as np
    from helpers import MOD, ROOT, TASK
    gold = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(gold, "swin_two_mod.npz"))
    info = json.load(open(os.path.join(gold, "swin_two_mod.json")))["train"]
    task, cfg = _flair_swin(precision, {MOD: 96, "DEM_ELEV": 64})
    tc = torch.from_numpy(d["t_cosia"]).long()
    batch = {MOD: torch.from_numpy(d["x_aerial"]).cuda(), "DEM_ELEV": torch.from_numpy(d["x_dem"]).cuda(),
             TASK: F.one_hot(tc, 19).permute(0, 3, 1, 2).float().cuda(),
             "ALL_LABEL-LPIS": torch.from_numpy(d["t_lpis"]).long().cuda()}
    task.train()
    loss, preds, _ = task.step(batch, training=True)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss = float.fromhex(info["loss"])
    named = dict(task.model.named_parameters())
    assert sorted(k for k, p in named.items() if p.grad is None) == info["unused_parameters"]
    fp32 = precision == "fp32"
    assert abs(loss.item() - ref_loss) <= (2e-5 if fp32 else 2e-2) * ref_loss
    agree = 0.9995 if fp32 else 0.9
    assert (preds[TASK].cpu().numpy() == d["preds_train_cosia"]).mean() > agree
    assert (preds["ALL_LABEL-LPIS"].cpu().numpy() == d["preds_train_lpis"]).mean() > agree
    gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in named.values() if p.grad is not None)).item()
    assert abs(gn - info["grad_norm"]) <= (5e-3 if fp32 else 0.15) * info["grad_norm"]
    if fp32:
        # per-parameter norms: the DEM branch ends in a 2 x 2 map, its PSP BatchNorms see 2 ... 8 samples per channel and
        # amplify f32 summation-order differences (the U-Net fusion fixture shows the same, tests/test_fusion_gpu.py)
        off = [(k, named[k].grad.double().norm().item(), v) for k, v in info["grad_norms"].items()
               if v > 1e-4 and abs(named[k].grad.double().norm().item() - v) > 5e-2 * v]
        assert not off, off[:8]
        # biases in front of a training-mode BatchNorm (fusion convolutions, the encoders' last fc2): exact gradient 0
        zero = [k for k, v in info["grad_norms"].items() if v <= 1e-4]
        assert len(zero) == 6 and all(named[k].grad.norm().item() <= 1e-5 for k in zero)
    for k in [f[len("grad__"):] for f in d.files if f.startswith("grad__")]:
        ref = d["grad__" + k]
        got = named[k].grad.float().cpu().numpy()
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        if fp32:
            assert rel <= (5e-2 if "DEM_ELEV" in k else 1e-2), f"{k}: relative grad error {rel}"
        else:
            cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
            assert cos >= 0.8, f"{k}: cosine {cos}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_swin_encoders_fused_match_the_reference_model(precision):
    """tests/golden/swin_two_mod.npz: the REFERENCE's FLAIR_HUB_Model (two Swin-T encoders, FusionHandler's
    placeholder-stage stripping, UPerNet task decoders + auxiliary decoder) on the same seeded weights and inputs"""
    import json
    import os
    import numpy as np
    from helpers import MOD, ROOT, TASK
    gold = os.path.join(ROOT, "tests", "golden")
    d = np.load(os.path.join(gold, "swin_two_mod.npz"))
    info = json.load(open(os.path.join(gold, "swin_two_mod.json")))
    task, cfg = _flair_swin(precision, {MOD: 96, "DEM_ELEV": 64})
    assert sorted(task.model.state_dict().keys()) == info["state_dict_keys"]
    assert list(task.model.encoders[MOD].seg_model.out_channels) == info["encoder_out_channels"]
    batch = {MOD: torch.from_numpy(d["x_aerial"]).cuda(), "DEM_ELEV": torch.from_numpy(d["x_dem"]).cuda(),
             TASK: torch.zeros(2, 19, 96, 96, device=DEV), "ALL_LABEL-LPIS": torch.zeros(2, 96, 96, dtype=torch.long, device=DEV)}
    with torch.no_grad():
        lt, la = task.model(batch)
    assert sorted(lt.keys()) == info["logit_keys"] and sorted(la.keys()) == info["aux_keys"]
    pairs = [(lt[TASK], d["logits_cosia"]), (lt["ALL_LABEL-LPIS"][:1], d["logits_lpis"]),
             (la["aux_AERIAL_RGBI_" + TASK][:1], d["logits_aux_cosia"])]
    for got, ref in pairs:
        got = got.float().cpu().numpy()
        assert got.shape == ref.shape
        if precision == "fp32":
            assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
        else:
            assert np.linalg.norm(got - ref) / np.linalg.norm(ref) <= 4e-2
            assert (got.argmax(1) == ref.argmax(1)).mean() >= 0.97


# ---------------------------------------------------------------------------------------------------- zonal loop

@pytest.mark.parametrize("output_type,precision", [("argmax", "fp32"), ("class_prob", "fp32"), ("argmax", "bf16")])
def test_zonal_run_with_swin_upernet_matches_the_oracle_loop(tmp_path, output_type, precision):
    """the fork's live configuration (configs/config_model_zonal_segmentation.yaml:26: a Swin + UPerNet checkpoint on
    RGB tiles) through run_inference, against the reference's tile loop restated around the CPU oracle network"""
    import os
    import numpy as np
    import yaml
    from helpers import MOD, ROOT, TASK, oracle_to_product_keys
    from flair_zonal_detection.inference import run_inference
    from flair_zonal_detection.raster import ArrayRaster
    from oracle.seeded_weights import fill_swin_state_dict
    from oracle.swin_upernet import SwinUPerNet as OracleNet
    from oracle.tile_bookkeeping import convert as o_convert, slice_tiles, write_window
    H, W, patch, margin, res = 300, 410, 128, 16, 0.2
    g = np.random.default_rng(3)
    img = g.integers(0, 255, (3, H, W)).astype(np.uint8)
    ras = ArrayRaster(img, 651992.36, 6860417.84, res)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "golden", "zonal_config.yaml")))
    means, stds = [105.66, 111.35, 102.18], [52.23, 45.62, 44.30]
    cfg.update({"output_path": str(tmp_path), "output_name": "z", "img_pixels_detection": patch, "margin": margin,
                "output_px_meters": res, "output_type": output_type, "batch_size": 4, "num_worker": 0,
                "monotemp_arch": "swin_tiny_patch4_window7_224-upernet", "hardware": {"precision": precision}})
    cfg["modalities"][MOD].update({"input_img_path": ras, "channels": [1, 2, 3],
                                   "normalization": {"type": "custom", "means": means, "stds": stds}})
    cfg["tasks"] = [{"name": TASK, "active": True, "class_names": {i: f"c{i}" for i in range(19)}}]
    oracle = OracleNet("swin_tiny_patch4_window7_224", 3, 19, patch).eval()
    oracle.load_state_dict(fill_swin_state_dict(oracle.state_dict(), seed=77))
    ckpt = {"state_dict": {"model." + k: v for k, v in oracle_to_product_keys(oracle.state_dict()).items()}}
    cfg["model_weights"] = str(tmp_path / "w.ckpt")
    torch.save(ckpt, cfg["model_weights"])
    got = run_inference(cfg)[TASK].data

    bounds = tuple(ras.bounds)
    canvas = np.zeros_like(got)
    for t in slice_tiles(bounds, bounds, patch, margin, res):
        x = ras.read_bounds([1, 2, 3], t["box"], patch).astype(np.float64)
        for c in range(3):
            x[c] = (x[c] - means[c]) / stds[c]
        with torch.no_grad():
            logits = oracle(torch.tensor(x[None], dtype=torch.float32))[0].numpy()
        p = o_convert(logits[:, margin:patch - margin, margin:patch - margin], output_type)
        col, row, w, h, skip = write_window(t["left"], t["top"], bounds, res, p.shape[-2], p.shape[-1])
        if skip:
            continue
        canvas[:, row:row + h, col:col + w] = p[:, :h, :w]
    if output_type == "argmax":
        assert (got == canvas).mean() >= (0.9995 if precision == "fp32" else 0.97)
    else:
        assert np.abs(got.astype(int) - canvas.astype(int)).max() <= 1
    assert got.any()


# ---------------------------------------------------------------------------------------------------- backward kernels

def test_linear_training_epilogues():
    """fc1 keeps its pre-activation, fc2's input gradient goes through gelu', DropPath scales rows per sample"""
    from flairhip import ops
    g = torch.Generator().manual_seed(21)
    M, K, N = 384, 128, 256
    x, w, b = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(N, K, generator=g) / math.sqrt(K)), torch.randn(N, generator=g)
    u_ref = _bf(F.linear(x.float(), w.float(), b))
    aux = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    h = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), act=ops.ACT_GELU, aux=aux)
    assert (aux.float().cpu() - u_ref.float()).abs().max().item() <= 3e-2
    assert (h.float().cpu() - F.gelu(aux.float().cpu())).abs().max().item() <= 2e-2
    # dgelu: out = (dy W2) * gelu'(u)
    dy = _bf(torch.randn(M, K, generator=g))
    w2t = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))  # [N_out = hidden, K = features of dy]
    u = aux.float().cpu().requires_grad_(True)
    F.gelu(u).backward(F.linear(dy.float(), w2t.float()))
    got = ops.linear(dy.to(DEV), w2t.to(DEV), None, act=ops.ACT_DGELU, aux=aux)
    assert (got.float().cpu() - u.grad).abs().max().item() <= 3e-2 * max(1.0, u.grad.abs().max().item())
    # row scale + residual
    rs = torch.tensor([0.0, 1.25, 1.0], device=DEV)
    r = _bf(torch.randn(M, N, generator=g))
    y = ops.linear(x.to(DEV), w.to(DEV), b.to(DEV), residual=r.to(DEV), row_scale=rs, rows_per_scale=128)
    ref = _bf(F.linear(x.float(), w.float(), b)).float() * rs.cpu().repeat_interleave(128)[:, None] + r.float()
    assert (y.float().cpu() - ref).abs().max().item() <= 3e-2 * max(1.0, ref.abs().max().item())
    assert torch.equal(y[:128].cpu(), r[:128])  # a dropped sample keeps its residual bit for bit


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [96, 256, 1024])
def test_layer_norm_backward(dtype, C):
    from flairhip import ops
    g = torch.Generator().manual_seed(C + 1)
    x = (torch.randn(3, 7, 9, C, generator=g) * 1.5 + 0.3).to(dtype)
    dy = torch.randn(3, 7, 9, C, generator=g).to(dtype)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    F.layer_norm(xr, (C,), wr, br, 1e-5).backward(dy.float())
    stats = torch.empty(3 * 7 * 9, 2, device=DEV)
    ops.layer_norm(x.to(DEV), w.to(DEV), b.to(DEV), stats=stats)
    dx, dg, db = ops.layer_norm_bwd(x.to(DEV), dy.to(DEV), w.to(DEV), stats)
    tol = 5e-5 if dtype == torch.float32 else 5e-2
    assert (dx.float().cpu() - xr.grad).abs().max().item() <= tol * max(1.0, xr.grad.abs().max().item())
    assert (dg.cpu() - wr.grad).abs().max().item() <= (1e-3 if dtype == torch.float32 else 0.3)
    assert (db.cpu() - br.grad).abs().max().item() <= 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_merge_norm_backward(dtype):
    from flairhip import ops
    g = torch.Generator().manual_seed(8)
    C = 96
    x = torch.randn(2, 8, 12, C, generator=g).to(dtype)
    dy = torch.randn(2, 4, 6, 4 * C, generator=g).to(dtype)
    w, b = torch.rand(4 * C, generator=g) + 0.5, torch.randn(4 * C, generator=g)
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    merged = xr.reshape(2, 4, 2, 6, 2, C).permute(0, 1, 3, 4, 2, 5).flatten(3)
    F.layer_norm(merged, (4 * C,), wr, br, 1e-5).backward(dy.float())
    stats = torch.empty(2 * 4 * 6, 2, device=DEV)
    ops.patch_merge_norm(x.to(DEV), w.to(DEV), b.to(DEV), stats=stats)
    dx, dg, db = ops.patch_merge_norm_bwd(x.to(DEV), dy.to(DEV), w.to(DEV), stats)
    tol = 5e-5 if dtype == torch.float32 else 5e-2
    assert (dx.float().cpu() - xr.grad).abs().max().item() <= tol * max(1.0, xr.grad.abs().max().item())
    assert (dg.cpu() - wr.grad).abs().max().item() <= (1e-3 if dtype == torch.float32 else 0.2)
    assert (db.cpu() - br.grad).abs().max().item() <= 1e-3


@pytest.mark.parametrize("H,W,heads,ws,shift", [
    (14, 14, 3, 7, 0), (16, 16, 3, 7, 3), (16, 20, 4, 7, 0), (4, 4, 3, 4, 0), (24, 24, 4, 12, 6), (32, 32, 4, 12, 6),
    (32, 32, 2, 12, 0), (12, 12, 8, 12, 0),
])
def test_window_attention_backward(H, W, heads, ws, shift):
    from flairhip import ops
    g = torch.Generator().manual_seed(H * 17 + ws + shift)
    C = heads * 32
    qkv = _bf(torch.randn(2, H, W, 3 * C, generator=g))
    dout = _bf(torch.randn(2, H, W, C, generator=g))
    bias = _bf(torch.randn(3 * C, generator=g)).float()
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    qr, br, tr = qkv.float().requires_grad_(True), bias.clone().requires_grad_(True), table.clone().requires_grad_(True)
    _attention_reference(qr, br, tr, heads, ws, shift).backward(dout.float())
    dq, dt, db = ops.window_attention_bwd(qkv.to(DEV), dout.to(DEV), bias.to(DEV), table.to(DEV), heads, ws, shift,
                                          32 ** -0.5)
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-12)).item()
    assert rel(dq.float().cpu(), qr.grad) <= 2e-2
    assert (dq.float().cpu() - qr.grad).abs().max().item() <= 5e-2 * max(1.0, qr.grad.abs().max().item())
    assert rel(dt.cpu(), tr.grad) <= 2e-2
    if br.grad.abs().max() > 0:  # the map is not a multiple of the window: padding tokens carry the bias
        assert rel(db.cpu(), br.grad) <= 2e-2
    else:
        assert db.abs().max().item() == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("hw_in,hw_out", [((4, 4), (16, 16)), ((6, 6), (16, 16)), ((16, 16), (16, 16)),
                                           ((32, 32), (16, 16)), ((1, 1), (16, 16)), ((8, 12), (32, 48)), ((32, 32), (128, 128))])
def test_bilinear_slice_backward(dtype, align, hw_in, hw_out):
    from flairhip import ops
    g = torch.Generator().manual_seed(13)
    C = 16
    dyw = torch.randn(2, *hw_out, 48, generator=g).to(dtype)
    xr = torch.zeros(2, C, *hw_in, requires_grad=True)
    F.interpolate(xr, size=hw_out, mode="bilinear", align_corners=align).backward(
        dyw[..., 16:32].float().permute(0, 3, 1, 2))
    dx = ops.bilinear_slice_bwd(dyw.to(DEV), hw_in, C, offset=16, align_corners=align).float().cpu()
    ref = xr.grad.permute(0, 2, 3, 1)
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert (dx - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adaptive_avg_pool_backward(dtype):
    from flairhip import ops
    g = torch.Generator().manual_seed(14)
    for s in (1, 2, 3, 6):
        dy = torch.randn(2, s, s, 32, generator=g).to(dtype)
        xr = torch.zeros(2, 32, 16, 16, requires_grad=True)
        F.adaptive_avg_pool2d(xr, s).backward(dy.float().permute(0, 3, 1, 2))
        dx = ops.adaptive_avg_pool_bwd(dy.to(DEV), (16, 16)).float().cpu()
        ref = xr.grad.permute(0, 2, 3, 1)
        assert (dx - ref).abs().max().item() <= (1e-5 if dtype == torch.float32 else 2e-2), s


# ---------------------------------------------------------------------------------------------------- training

def _grad_pair(name, ch, size, B=2, seed=7):
    from flairhip import ops
    from flairhip.swin import SwinUPerNet
    from oracle.swin_upernet import SwinUPerNet as OracleNet
    torch.manual_seed(1)
    oracle = OracleNet(name, ch, 19, size, drop_path_rate=0.0).train()
    _randomise(oracle, seed)
    model = SwinUPerNet(name, ch, 19, size, drop_path_rate=0.0)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.to(DEV).train()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, ch, size, size, generator=g)
    tgt = torch.randint(0, 19, (B, size, size), generator=g)
    return oracle, model, x, tgt


def test_swin_upernet_training_step_gradients_match_oracle_autograd():
    """bf16 training step (drop_path 0) against torch autograd through the fp32 CPU oracle: loss, every parameter
    gradient by cosine similarity / relative norm -- the bound is bf16 rounding, as for the U-Net's bf16 gradients"""
    from flairhip import nn as hnn
    from flairhip import ops
    name, ch, size = "swin_tiny_patch4_window7_224", 5, 128
    # batch 6: the PSP branch with pool size 1 runs a training-mode BatchNorm over B samples per channel; with 2 samples
    # its backward amplifies bf16 rounding of the last feature map to a 20 % error in every encoder gradient
    oracle, model, x, tgt = _grad_pair(name, ch, size, B=6)
    ref_logits = oracle(x)
    ref_loss = F.cross_entropy(ref_logits, tgt)
    ref_loss.backward()
    xn = ops.nchw_to_nhwc(x.to(DEV), torch.bfloat16, ops.pad_channels(ch))
    y = model(xn)
    logits = hnn.logits_view(y, 19)
    crit = hnn.HipCrossEntropyLoss(num_classes=19).to(DEV)
    loss = crit(logits, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) <= 2e-2 * abs(ref_loss.item())
    ref_grads = dict(oracle.named_parameters())
    bad, cos_all = [], []
    for k, p in model.named_parameters():
        if "fpn_stages.4" in k:  # the FPNBlock of the input image is never called (smp drops features[0])
            assert p.grad is None and ref_grads[k].grad is None
            continue
        assert p.grad is not None, k
        a, b = p.grad.float().cpu().flatten(), ref_grads[k].grad.flatten()
        if k.endswith("layers_3.blocks.1.mlp.fc2.bias"):
            # a per-channel constant added to the last feature map is removed by the training-mode BatchNorms that follow
            # every consumer (PSP convolutions): the exact gradient is 0, the oracle's value is f32 round-off
            assert b.abs().max().item() < 1e-6 and a.abs().max().item() < 5e-3
            continue
        cos = F.cosine_similarity(a, b, dim=0).item()
        cos_all.append(cos)
        ratio = (a.norm() / (b.norm() + 1e-20)).item()
        if cos < 0.95 or not (0.8 < ratio < 1.25):
            bad.append((k, round(cos, 4), round(ratio, 3)))
    assert not bad, bad[:12]
    assert sum(cos_all) / len(cos_all) > 0.99


def test_swin_training_reduces_the_loss_and_drop_path_is_stochastic():
    from flairhip import nn as hnn
    from flairhip import ops
    from flairhip.swin import SwinUPerNet
    torch.manual_seed(3)
    model = SwinUPerNet("swin_tiny_patch4_window7_224", 3, 19, 64, drop_path_rate=0.2).to(DEV).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
    tgt = torch.randint(0, 19, (4, 64, 64), generator=g).to(DEV)
    crit = hnn.HipCrossEntropyLoss(num_classes=19).to(DEV)
    xn = ops.nchw_to_nhwc(x, torch.bfloat16, ops.pad_channels(3))
    losses = []
    for _ in range(30):
        opt.zero_grad(set_to_none=True)
        loss = crit(hnn.logits_view(model(xn), 19), tgt)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert min(losses[-5:]) < 0.95 * losses[0], losses  # random per-pixel labels: memorising 4 tiles is slow
    with torch.no_grad():  # train mode without autograd: the evaluation kernels, no DropPath draw
        a = model(xn)
        b = model(xn)
    assert torch.equal(a, b)
    y1, y2 = model(xn), model(xn)  # two training forwards draw different DropPath masks
    assert not torch.equal(y1, y2)
    y32 = model(ops.nchw_to_nhwc(x, torch.float32, ops.pad_channels(3)))  # precision 32 trains too (f32 parity kernels)
    assert y32.dtype == torch.float32 and y32.requires_grad


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("hw", [(8, 8), (5, 12), (1, 3)])
def test_updown2x_slice_is_the_two_resizes_and_its_own_transpose(dtype, hw):
    """UPerNet's placeholder FPN stage: F.interpolate(x2) then F.interpolate(back) == one replicate-padded 3-tap filter"""
    from flairhip import ops
    g = torch.Generator().manual_seed(17)
    C = 16
    xw = torch.randn(2, *hw, 40, generator=g).to(dtype)
    xr = xw[..., 8:24].float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    up = F.interpolate(xr, size=(2 * hw[0], 2 * hw[1]), mode="bilinear", align_corners=False)
    ref = F.interpolate(up, size=hw, mode="bilinear", align_corners=False)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    wide = torch.full((2, *hw, 32), 3.0, dtype=dtype, device=DEV)
    ops.updown2x_slice(xw.to(DEV), C, x_offset=8, out=wide, offset=16)
    tol = 2e-6 if dtype == torch.float32 else 2e-2
    got = wide.float().cpu()
    assert (got[..., 16:] - ref.detach().permute(0, 2, 3, 1)).abs().max().item() <= tol
    assert torch.all(got[..., :16] == 3.0)
    dx = ops.updown2x_slice(dy.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV), C).float().cpu()
    assert (dx - xr.grad.permute(0, 2, 3, 1)).abs().max().item() <= (2e-6 if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("M,K,N", [(256, 96, 288), (1000, 96, 96), (4096, 384, 96), (2048, 768, 3072), (70, 128, 8),
                                    (8192, 256, 128), (333, 200, 136)])
def test_linear_wgrad_matches_torch(M, K, N):
    from flairhip import ops
    g = torch.Generator().manual_seed(M + K + N)
    x, dy = _bf(torch.randn(M, K, generator=g)), _bf(torch.randn(M, N, generator=g))
    ref = dy.float().t() @ x.float()
    dw = ops.linear_wgrad(x.to(DEV), dy.to(DEV))
    assert dw.shape == (N, K)
    assert (dw.cpu() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
    again = ops.linear_wgrad(x.to(DEV), dy.to(DEV))
    assert torch.equal(dw, again)  # fixed summation order
    acc = ops.linear_wgrad(x.to(DEV), dy.to(DEV), out=dw.clone(), accumulate=True)
    assert (acc.cpu() - 2 * ref).abs().max().item() <= 4e-3 * max(1.0, ref.abs().max().item())
    dw2, db = ops.linear_wgrad(x.to(DEV), dy.to(DEV), with_bias=True)  # nn.Linear's bias gradient from the same pass
    assert torch.equal(dw2, dw)
    ref_b = dy.float().sum(0)
    assert db.shape == (N,) and (db.cpu() - ref_b).abs().max().item() <= 1e-3 * max(1.0, ref_b.abs().max().item())


# ---------------------------------------------------------------------------------------------------- f32 parity mode of the training kernels

@pytest.mark.parametrize("M,K,N", [(256, 96, 288), (333, 200, 136), (70, 128, 8), (1000, 384, 96)])
def test_linear_f32_forward_epilogues_and_wgrad(M, K, N):
    from flairhip import ops
    g = torch.Generator().manual_seed(M + 3 * K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g)
    r = torch.randn(M, N, generator=g)
    tol = lambda ref: 2e-5 * max(1.0, ref.abs().max().item())
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    ref = F.linear(x, w, b)
    assert (ops.linear(xd, wd, bd).cpu() - ref).abs().max().item() <= tol(ref)
    aux = torch.empty(M, N, device=DEV)
    h = ops.linear(xd, wd, bd, act=ops.ACT_GELU, aux=aux)
    assert (aux.cpu() - ref).abs().max().item() <= tol(ref)
    assert (h.cpu() - F.gelu(ref)).abs().max().item() <= tol(ref)
    rps = (M + 2) // 3
    rs = torch.tensor([0.0, 1.25, 1.0], device=DEV)
    y = ops.linear(xd, wd, bd, residual=r.to(DEV), row_scale=rs, rows_per_scale=rps)
    ref2 = ref * rs.cpu().repeat_interleave(rps)[:M, None] + r
    assert (y.cpu() - ref2).abs().max().item() <= tol(ref2)
    # dgelu epilogue: (dy W^T-operand) * gelu'(aux)
    dy = torch.randn(M, K, generator=g)
    u = aux.cpu().clone().requires_grad_(True)
    F.gelu(u).backward(F.linear(dy, w))
    got = ops.linear(dy.to(DEV), wd, None, act=ops.ACT_DGELU, aux=aux)
    assert (got.cpu() - u.grad).abs().max().item() <= tol(u.grad)
    # weight / bias gradient
    dyo = torch.randn(M, N, generator=g)
    dw, db = ops.linear_wgrad(xd, dyo.to(DEV), with_bias=True)
    refw = dyo.t() @ x
    assert (dw.cpu() - refw).abs().max().item() <= 2e-5 * max(1.0, refw.abs().max().item()) * math.sqrt(M / 64)
    assert (db.cpu() - dyo.sum(0)).abs().max().item() <= 1e-4 * max(1.0, dyo.sum(0).abs().max().item())
    assert torch.equal(ops.linear_wgrad(xd, dyo.to(DEV)), dw)


@pytest.mark.parametrize("H,W,heads,ws,shift", [
    (14, 14, 3, 7, 0), (16, 16, 3, 7, 3), (16, 20, 4, 7, 0), (4, 4, 3, 4, 0), (24, 24, 4, 12, 6), (30, 26, 2, 12, 6),
])
def test_window_attention_backward_f32(H, W, heads, ws, shift):
    from flairhip import ops
    g = torch.Generator().manual_seed(H * 17 + ws + shift)
    C = heads * 32
    qkv = torch.randn(2, H, W, 3 * C, generator=g)
    dout = torch.randn(2, H, W, C, generator=g)
    bias = torch.randn(3 * C, generator=g)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g) * 0.5
    qr, br, tr = qkv.clone().requires_grad_(True), bias.clone().requires_grad_(True), table.clone().requires_grad_(True)
    _attention_reference(qr, br, tr, heads, ws, shift).backward(dout)
    dq, dt, db = ops.window_attention_bwd(qkv.to(DEV), dout.to(DEV), bias.to(DEV), table.to(DEV), heads, ws, shift,
                                          32 ** -0.5)
    assert (dq.cpu() - qr.grad).abs().max().item() <= 2e-5 * max(1.0, qr.grad.abs().max().item())
    assert (dt.cpu() - tr.grad).abs().max().item() <= 1e-4 * max(1.0, tr.grad.abs().max().item())
    assert (db.cpu() - br.grad).abs().max().item() <= 1e-4 * max(1.0, br.grad.abs().max().item())


@pytest.mark.parametrize("name,ch,size,B,bound", [
    ("swin_tiny_patch4_window7_224", 5, 128, 3, 2e-4),   # measured: worst 5.6e-5, median 2.8e-5
    # 24 blocks, 12 x 12 windows (shifted 4 x 4 / 2 x 2 grids, one-window and 6 x 6 stages): worst 1.5e-3, median 2e-4 --
    # f32 round-off through the PSP's BatchNorm over B samples, falling with B (4.7e-3 at B = 2, 1.9e-3 at 3)
    ("swin_base_patch4_window12_384", 3, 224, 4, 4e-3),
])
def test_swin_upernet_fp32_training_gradients_match_oracle_autograd(name, ch, size, B, bound):
    """precision 32: the same autograd nodes on the f32 kernels; every parameter gradient against torch autograd through
    the CPU oracle at f32 tolerance (the bf16 test above can only bound rounding noise)"""
    from flairhip import nn as hnn
    from flairhip import ops
    oracle, model, x, tgt = _grad_pair(name, ch, size, B=B)
    ref_loss = F.cross_entropy(oracle(x), tgt)
    ref_loss.backward()
    xn = ops.nchw_to_nhwc(x.to(DEV), torch.float32, ops.pad_channels(ch))
    logits = hnn.logits_view(model(xn), 19)
    loss = hnn.HipCrossEntropyLoss(num_classes=19).to(DEV)(logits, tgt.to(DEV))
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) <= 1e-5 * abs(ref_loss.item())
    ref_grads = dict(oracle.named_parameters())
    worst = []
    for k, p in model.named_parameters():
        if "fpn_stages.4" in k:
            assert p.grad is None and ref_grads[k].grad is None
            continue
        a, b = p.grad.float().cpu().flatten(), ref_grads[k].grad.flatten()
        if k.endswith("layers_3.blocks.1.mlp.fc2.bias"):  # exact gradient 0 (see the bf16 test)
            assert b.abs().max().item() < 1e-5 and a.abs().max().item() < 1e-5
            continue
        rel = ((a - b).norm() / (b.norm() + 1e-20)).item()
        worst.append((rel, k))
    worst.sort(reverse=True)
    assert worst[0][0] <= bound, worst[:8]
    assert worst[len(worst) // 2][0] <= bound / 4
