"""BASELINE.json configs[1] sizes on the GPU: 32 tiles of 512 x 512 x 5 per step, 19 classes.

The CPU oracle needs minutes for a batch of that size, so parity at full size is carried by
  (a) a direct comparison with the oracle on a two-tile sample AT THE FULL TILE SIZE, and
  (b) size-independent properties that tie the whole batch to that sample: batch-partition invariance of the
      eval-mode forward (bit-exact), batch statistics as a checksum of per-chunk checksums, the loss as the
      weight-normalised sum of per-tile losses, gradients of the logits summing to zero per pixel, argmax
      consistency, and run-to-run determinism of the full training step.
"""
import pytest
import torch
import torch.nn.functional as F

from helpers import MOD, TASK, make_pair

pytestmark = pytest.mark.gpu

B_FULL, TILE, CLASSES = 32, 512, 19
WEIGHTS = torch.tensor([1.0] * 15 + [0.0] * 4)


def _batch(n, seed=2025):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 5, TILE, TILE, generator=g)
    t = torch.randint(0, CLASSES, (n, TILE, TILE), generator=g)
    return x, t


def test_full_size_tiles_match_the_oracle_and_the_batch_is_partition_invariant(cuda):
    task, oracle, _ = make_pair(precision="fp32")
    x, t = _batch(B_FULL)
    oracle.eval()
    with torch.no_grad():
        ref = oracle(x[:2])  # two full-size tiles on the CPU: seconds
    task.eval()
    xd, td = x.to(cuda), t.to(cuda)
    with torch.no_grad():
        full = task.model({MOD: xd, TASK: td})[0][TASK]
        parts = torch.cat([task.model({MOD: xd[i:i + 8], TASK: td[i:i + 8]})[0][TASK] for i in range(0, B_FULL, 8)])
    assert full.shape == (B_FULL, CLASSES, TILE, TILE)
    got = full[:2].float().cpu()
    err = (got - ref).abs().max().item()
    assert err <= 1e-4 * max(1.0, ref.abs().max().item()), f"logit error {err}"
    assert (got.argmax(1) == ref.argmax(1)).float().mean().item() >= 0.9999
    # every tile is computed independently of its batch neighbours: the 32-tile launch (other grid, other tile ->
    # block mapping) gives the same bits as four 8-tile launches, so the sample above speaks for the whole batch
    assert torch.equal(full, parts)


def test_full_size_train_loss_matches_the_oracle_on_a_sample(cuda):
    task, oracle, _ = make_pair(precision="fp32")
    x, t = _batch(2, seed=7)
    oracle.train()
    ref = F.cross_entropy(oracle(x), t, weight=WEIGHTS)
    task.train()
    loss, preds, _ = task.step({MOD: x.to(cuda), TASK: t.to(cuda)}, training=True)
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    assert preds[TASK].shape == (2, TILE, TILE)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_batch_statistics_are_a_checksum_of_chunk_checksums(cuda, dtype):
    """stem-sized layer at full size (32 x 256 x 256 x 64 outputs): the statistics that come out of the conv epilogue
    equal the float64 combination of per-chunk sums of the stored output"""
    from flairhip import ops
    g = torch.Generator().manual_seed(11)
    x = ops.nchw_to_nhwc(torch.randn(B_FULL, 64, 256, 256, generator=g).to(cuda), dtype)
    w = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(cuda)
    pw = ops.pack_conv_weight(w, dtype, 1, 64)
    gamma, beta = torch.ones(64, device=cuda), torch.zeros(64, device=cuda)
    rm, rv = torch.zeros(64, device=cuda), torch.ones(64, device=cuda)
    y, scale, shift, mean, rstd = ops.conv2d_bn_stats(x, pw, 1, 64, gamma, beta, rm, rv, 0.1, 1e-5)
    n = B_FULL * 256 * 256
    s1 = torch.zeros(64, dtype=torch.float64, device=cuda)
    s2 = torch.zeros(64, dtype=torch.float64, device=cuda)
    for i in range(0, B_FULL, 4):  # chunk checksums, combined in float64
        c = y[i:i + 4].double().reshape(-1, 64)
        s1 += c.sum(0)
        s2 += (c * c).sum(0)
    m = s1 / n
    var = s2 / n - m * m
    assert torch.allclose(mean.double(), m, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rstd.double(), 1.0 / torch.sqrt(var + 1e-5), rtol=2e-5)
    assert torch.allclose(rm.double(), 0.1 * m, rtol=1e-5, atol=1e-7)  # momentum update from zero


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_loss_and_logit_gradient_properties_at_full_size(cuda, dtype):
    from flairhip import ops
    g = torch.Generator().manual_seed(13)
    logits = torch.zeros(B_FULL, TILE, TILE, 32, dtype=dtype, device=cuda)
    logits[..., :CLASSES] = (torch.randn(B_FULL, TILE, TILE, CLASSES, generator=g) * 3).to(cuda, dtype)
    t = torch.randint(0, CLASSES, (B_FULL, TILE, TILE), generator=g).to(torch.uint8).to(cuda)
    w = WEIGHTS.to(cuda)
    loss, wsum, dl, pred = ops.softmax_ce(logits, t, w, CLASSES, want_grad=True, want_pred=True)
    # the weighted mean over the batch is the weight-normalised sum of the per-tile losses
    num = den = 0.0
    for i in range(B_FULL):
        li, wi, _, _ = ops.softmax_ce(logits[i:i + 1], t[i:i + 1], w, CLASSES)
        num += li.double().item() * wi.double().item()
        den += wi.double().item()
    assert abs(wsum.item() - den) <= 1e-6 * den
    assert abs(loss.item() - num / den) <= 2e-6 * abs(num / den)
    assert wsum.item() == float((t < 15).sum().item())  # classes 15-18 carry weight 0
    # predictions are the first maximum of the real classes; pad channels never win
    assert torch.equal(pred, logits[..., :CLASSES].float().argmax(-1).to(torch.uint8))
    # d loss / d logits: zero on zero-weight pixels and in the pad channels, sums to zero over the classes of a pixel
    d = dl.float()
    assert not d[..., CLASSES:].any()
    assert not d[t >= 15].any()
    tol = 1e-9 if dtype == torch.float32 else 3e-8  # of order 1/den per pixel, rounded per element in bf16
    assert d.sum(-1).abs().max().item() <= tol
    # and against autograd on one tile
    z = logits[:1, ..., :CLASSES].float().permute(0, 3, 1, 2).requires_grad_(True)
    ref = F.cross_entropy(z, t[:1].long(), weight=w, reduction="sum")
    ref.backward()
    gref = z.grad.permute(0, 2, 3, 1) / wsum
    err = (d[:1, ..., :CLASSES] - gref).abs().max().item()
    assert err <= (1e-10 if dtype == torch.float32 else 2e-9 + 0.01 * gref.abs().max().item())


def test_full_size_training_step_is_deterministic(cuda):
    """the 32-tile bf16 training step twice from the same state: same loss bits, same bits in every gradient"""
    task, _, _ = make_pair(precision="bf16")
    x, t = _batch(B_FULL, seed=3)
    batch = {MOD: x.to(cuda), TASK: t.to(cuda)}
    task.train()
    state = {k: v.clone() for k, v in task.model.state_dict().items()}
    runs = []
    for _ in range(2):
        task.model.load_state_dict(state)
        task.model.zero_grad(set_to_none=True)
        loss, _, _ = task.step(batch, training=True)
        loss.backward()
        torch.cuda.synchronize()
        runs.append((loss.detach().clone(), {n: p.grad.clone() for n, p in task.model.named_parameters()
                                             if p.grad is not None}))
    assert torch.isfinite(runs[0][0]) and torch.equal(runs[0][0], runs[1][0])
    assert runs[0][1].keys() == runs[1][1].keys() and len(runs[0][1]) > 100
    for n, g0 in runs[0][1].items():
        assert torch.equal(g0, runs[1][1][n]), n


def _grad_errors(task, oracle):
    """relative L2 error of every parameter gradient of the product against the oracle's"""
    osd = dict(oracle.named_parameters())
    out = {}
    for name, p in task.model.named_parameters():
        if name.startswith("fusion_handler."):
            continue
        ok = ("encoder." if name.startswith("encoders.") else "") + name.split(".seg_model.", 1)[1]
        rg = osd[ok].grad
        out[name] = ((p.grad.float().cpu() - rg).norm() / (rg.norm() + 1e-12)).item()
    return out


def test_fp32_gradients_at_full_tile_size_match_the_oracle(cuda):
    """Every parameter gradient of a TRAINING step on two full-size tiles (fp32 mode) against the oracle's autograd
    evaluated in float64, next to the oracle's OWN float32 evaluation.

    Round-1 review: the 64 x 96 test needs a 1e-2 budget, explained by BatchNorm over 12 samples at the bottleneck;
    if that were the whole story the error would shrink at 512 x 512.  It does not -- tools/grad_parity.py shows it
    GROWS layer by layer from 4e-7 at the head to 1.2e-2 at the stem -- but the reference arithmetic itself (torch CPU
    float32, the oracle) is just as far from the float64 truth (1.1e-2 on the same parameters): the gradient of this
    random-initialised 50-layer BatchNorm network is ill-conditioned in float32, whoever evaluates it.  So the bar
    is: the product must be as close to the float64 gradients as the reference's own float32 arithmetic is (within
    2.5x, parameter by parameter).  The mechanism is not BatchNorm but ReLU: an activation within rounding distance of
    zero gets the other mask in float32 than in float64, a fraction f ~ 1e-6 of the elements of every layer, each
    contributing an O(1) error at its position -> sqrt(f) ~ 1e-3 relative per layer, accumulating over ~50 layers.
    With ReLU swapped for softplus in BOTH oracle evaluations the float32-vs-float64 error drops from 8e-3 to 1e-5
    (checked on the CPU oracle at 192 x 192)."""
    import copy
    task, oracle, _ = make_pair(precision="fp32")
    x, t = _batch(2, seed=17)
    o64 = copy.deepcopy(oracle).double().train()
    oracle.train()
    ref_loss = F.cross_entropy(oracle(x), t, weight=WEIGHTS)
    ref_loss.backward()
    F.cross_entropy(o64(x.double()), t, weight=WEIGHTS.double()).backward()
    task.train()
    loss, _, _ = task.step({MOD: x.to(cuda), TASK: t.to(cuda)}, training=True)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - ref_loss.item()) <= 2e-5 * max(1.0, abs(ref_loss.item()))
    g32, g64 = dict(oracle.named_parameters()), dict(o64.named_parameters())
    n, worst_ratio = 0, 0.0
    for name, p in task.model.named_parameters():
        if name.startswith("fusion_handler."):
            continue
        ok = ("encoder." if name.startswith("encoders.") else "") + name.split(".seg_model.", 1)[1]
        r64 = g64[ok].grad
        e_prod = ((p.grad.double().cpu() - r64).norm() / r64.norm()).item()
        e_ref = ((g32[ok].grad.double() - r64).norm() / r64.norm()).item()
        n += 1
        worst_ratio = max(worst_ratio, e_prod / (e_ref + 1e-6))
        assert e_prod <= 2.5 * e_ref + 1e-5, f"{name}: product {e_prod:.3e} vs float64, reference float32 {e_ref:.3e}"
        assert e_prod <= 3e-2, f"{name}: {e_prod:.3e}"
    print(f"fp32 512x512 gradients: {n} parameters, worst (product error) / (oracle-f32 error) = {worst_ratio:.2f}")
    assert n > 100


def test_bf16_at_full_tile_size_against_the_oracle(cuda):
    """The benchmarked mode (bf16 storage, f32 accumulation) on two full-size tiles against the fp32 oracle: eval
    logits / argmax, the training loss, and every parameter gradient.  Budgets are bf16 budgets: 8 mantissa bits per
    stored activation, ~50 layers deep."""
    task, oracle, _ = make_pair(precision="bf16")
    x, t = _batch(2, seed=19)
    batch = {MOD: x.to(cuda), TASK: t.to(cuda)}
    oracle.eval()
    with torch.no_grad():
        ref = oracle(x)
    task.eval()
    with torch.no_grad():
        got = task.model(batch)[0][TASK].float().cpu()
    rel = ((got - ref).norm() / ref.norm()).item()
    agree = (got.argmax(1) == ref.argmax(1)).float().mean().item()
    # the disagreeing pixels must be near-ties of the reference (top-2 margin within the bf16 error), not errors
    top2 = ref.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1])[got.argmax(1) != ref.argmax(1)]
    rms_err = (got - ref).pow(2).mean().sqrt().item()
    all_margin = top2[:, 0] - top2[:, 1]
    print(f"bf16 512x512 eval: relative logit error {rel:.3e} (rms {rms_err:.3e}, logit std {ref.std().item():.3e}), "
          f"argmax agreement {agree:.5f}; reference top-2 margin: median {all_margin.median().item():.3e}, among "
          f"disagreeing pixels median {margin.median().item() if margin.numel() else 0:.3e} / "
          f"max {margin.max().item() if margin.numel() else 0:.3e}")
    assert rel <= 0.05 and agree >= 0.95
    if margin.numel():
        # a pixel may change class only where the reference's decision margin is within the bf16 logit noise
        assert margin.max().item() <= 12.0 * rms_err and margin.median().item() <= 3.0 * rms_err
    oracle.train()
    ref_loss = F.cross_entropy(oracle(x), t, weight=WEIGHTS)
    ref_loss.backward()
    task.train()
    loss, _, _ = task.step(batch, training=True)
    loss.backward()
    torch.cuda.synchronize()
    lrel = abs(loss.item() - ref_loss.item()) / abs(ref_loss.item())
    # Parameter gradients: in bf16 the ReLU masks of the two evaluations differ on a fraction of ~1e-3 of the
    # activations per layer (see the fp32 test above for the mechanism), so element-wise agreement with the float32
    # oracle is out of reach for the deep layers by construction; what must hold is that the gradient is the same
    # DIRECTION: cosine similarity over all parameters, and near-exact agreement where no ReLU lies in between (head).
    osd = dict(oracle.named_parameters())
    dot = n1 = n2 = 0.0
    cos = {}
    for name, p in task.model.named_parameters():
        if name.startswith("fusion_handler."):
            continue
        ok = ("encoder." if name.startswith("encoders.") else "") + name.split(".seg_model.", 1)[1]
        a, b = p.grad.double().cpu().flatten(), osd[ok].grad.double().flatten()
        d, na, nb = float(a @ b), float(a @ a), float(b @ b)
        dot, n1, n2 = dot + d, n1 + na, n2 + nb
        cos[ok] = d / (na * nb) ** 0.5
    gcos = dot / (n1 * n2) ** 0.5
    head_err = _grad_errors(task, oracle)[f"main_decoders.{TASK}.seg_model.segmentation_head.0.weight"]
    print(f"bf16 512x512 train: loss {loss.item():.5f} vs {ref_loss.item():.5f} (rel {lrel:.2e}); gradient cosine over "
          f"all parameters {gcos:.4f}, norm ratio {(n1 / n2) ** 0.5:.4f}, worst per-parameter cosine "
          f"{min(cos.values()):.3f} at {min(cos, key=cos.get)}; head weight gradient relative error {head_err:.3e}")
    assert lrel <= 1e-3
    assert head_err <= 2e-2
    assert gcos >= 0.5 and 0.9 <= (n1 / n2) ** 0.5 <= 1.1  # measured: cosine 0.68, norm ratio 0.987


def test_bf16_parameter_gradients_are_as_close_to_fp32_as_the_oracles_own_bf16_storage_run(cuda):
    """Round-2 review, weak #2: the benchmarked mode had only a direction check (cosine >= 0.5) on whole-model gradients.
    The oracle has a bf16-STORAGE mode now (oracle/unet_resnet34.py: every tensor the product materialises in bf16 is
    rounded at the same point, forward and backward, f32 arithmetic in between).  Measured first: the product does NOT
    track that evaluation element by element either (per-parameter relative error 0.72, cosine 0.62 at 2 x 512 x 512) --
    a one-ulp difference of one stored bf16 activation (the f32 sums of two implementations differ in the last bits, 5e-4
    of the elements round the other way) moves ~600 outputs of the next convolution by ~1e-4 relative, of which ~5 %
    cross a bf16 rounding boundary in turn: the difference avalanches, and after a few layers two bf16-storage
    evaluations are as far from each other as each is from the float32 one.  What CAN be bounded parameter by parameter
    is the distance to the float32 gradient, with the reference arithmetic's own bf16-storage evaluation as the
    yardstick (the form of the fp32 test above, one precision level down; measured: yardstick error 0.92 median, product
    error / yardstick 0.999 median, 1.21 worst): a mis-scaled or partly wrong layer shows as a
    product error well above the yardstick for its parameters, and as a norm ratio away from 1."""
    task, oracle32, _ = make_pair(precision="bf16")
    from oracle.unet_resnet34 import UnetResNet34
    oracle16 = UnetResNet34(5, 19, storage_dtype=torch.bfloat16)
    oracle16.load_state_dict(oracle32.state_dict())
    x, t = _batch(2, seed=23)
    batch = {MOD: x.to(cuda), TASK: t.to(cuda)}
    grads = {}
    for tag, orc in (("f32", oracle32), ("b16", oracle16)):
        orc.train()
        loss_o = F.cross_entropy(orc(x), t, weight=WEIGHTS)
        loss_o.backward()
        grads[tag] = {k: p.grad.double().flatten() for k, p in orc.named_parameters()}
        grads[tag + "_loss"] = loss_o.item()
    task.train()
    loss, _, _ = task.step(batch, training=True)
    loss.backward()
    torch.cuda.synchronize()
    lrel = abs(loss.item() - grads["b16_loss"]) / abs(grads["b16_loss"])
    ratio, nrm, yard = {}, {}, {}
    for name, p in task.model.named_parameters():
        if name.startswith("fusion_handler."):
            continue
        ok = ("encoder." if name.startswith("encoders.") else "") + name.split(".seg_model.", 1)[1]
        a, r32, r16 = p.grad.double().cpu().flatten(), grads["f32"][ok], grads["b16"][ok]
        e_hip = float((a - r32).norm() / r32.norm())
        e_orc = float((r16 - r32).norm() / r32.norm())
        yard[ok], ratio[ok], nrm[ok] = e_orc, e_hip / max(e_orc, 1e-3), float(a.norm() / r16.norm())
    rv = sorted(ratio.values())
    worst = max(ratio, key=ratio.get)
    print(f"bf16 product vs fp32 oracle, yardstick = the oracle's own bf16-storage run, 2 x 512 x 512: loss rel {lrel:.2e} "
          f"(vs the bf16-storage oracle); yardstick error median {sorted(yard.values())[len(yard) // 2]:.3f}; "
          f"product error / yardstick: median {rv[len(rv) // 2]:.3f}, 90th percentile {rv[int(len(rv) * 0.9)]:.3f}, "
          f"worst {ratio[worst]:.3f} at {worst}; gradient norm product / bf16-storage oracle: "
          f"{min(nrm.values()):.3f} .. {max(nrm.values()):.3f}")
    assert lrel <= 2e-4
    assert ratio[worst] <= 1.5 and rv[len(rv) // 2] <= 1.15
    assert 0.7 <= min(nrm.values()) and max(nrm.values()) <= 1.4  # measured 0.83 .. 1.25
