"""hipGraph capture of the whole training step: replaying the graph must give what the eager step gives."""
import pytest
import torch

from helpers import MOD, TASK, make_pair

pytestmark = pytest.mark.gpu


def _run(graph: bool, steps: int = 4):
    from flairhip.graph import GraphedTrainStep, make_capturable
    task, _, cfg = make_pair(precision="bf16", seed=11)
    task.train()
    g = torch.Generator().manual_seed(1)
    batches = [{MOD: torch.randn(2, 5, 64, 64, generator=g).cuda(),
                TASK: torch.randint(0, 19, (2, 64, 64), generator=g).to(torch.uint8).cuda()} for _ in range(steps)]
    opt = torch.optim.AdamW(task.model.parameters(), lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=steps + 8, pct_start=0.2,
                                                cycle_momentum=False, div_factor=1000)
    losses = []
    if graph:
        # warm-up steps would move the weights: capture with zero warm-up influence by restoring the state
        state = {k: v.clone() for k, v in task.state_dict().items()}
        stepper = GraphedTrainStep(task, opt, batches[0], warmup_steps=2)
        task.load_state_dict(state)
        for st in opt.state.values():  # reset the moments the warm-up accumulated
            for k, v in st.items():
                if torch.is_tensor(v):
                    v.zero_()
        for b in batches:
            losses.append(stepper(b).item())
            sched.step()
    else:
        # the same optimizer arithmetic as under capture (device-resident fp32 lr and step counter): AdamW's
        # host-scalar path rounds lr differently, which Adam's normalised update turns into O(lr) weight noise
        make_capturable(opt)
        for b in batches:
            loss = task.training_step(b, 0)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            sched.step()
            losses.append(loss.item())
    torch.cuda.synchronize()
    w = task.model.state_dict()[f"encoders.{MOD}.seg_model.layer2.0.conv1.weight"].float().cpu()
    confmat = task.train_metrics[TASK].confmat.cpu()
    nbt = int(task.model.state_dict()[f"encoders.{MOD}.seg_model.bn1.num_batches_tracked"])
    return losses, w, confmat, nbt


def test_graph_replay_matches_eager(cuda):
    le, we, ce, ne = _run(False)
    lg, wg, cg, ng = _run(True)
    # every kernel on the path is deterministic (fixed-order reductions, no float atomics on the U-Net
    # path, whose bilinear resize is the identity), so the replayed trajectory is the eager one bit for bit
    assert le == lg, (le, lg)
    assert torch.equal(we, wg), ((we - wg).norm() / we.norm()).item()
    assert int(cg.sum()) == int(ce.sum()) + 2 * 2 * 64 * 64  # + the two warm-up batches
    # the warm-up batches were counted before task.load_state_dict(state) restored the pre-warm-up buffers: the
    # loaded num_batches_tracked is the truth (HipBatchNorm2d drops its pending count on load), so both runs end
    # with the four real batches
    assert ng == ne


def test_trainer_graph_mode_follows_the_eager_trainer(cuda):
    """HipTrainer(hip_graph=True): two eager steps, then the captured step; same loss trajectory as the eager trainer"""
    from flair_hub.tasks.trainers import HipTrainer

    def fit(hip_graph):
        task, _, cfg = make_pair(precision="bf16", seed=3)
        cfg["hyperparams"].update({"learning_rate": 1e-3, "total_steps": 6})
        g = torch.Generator().manual_seed(2)
        batches = [{MOD: torch.randn(2, 5, 64, 64, generator=g), TASK: torch.randint(0, 19, (2, 64, 64), generator=g)}
                   for _ in range(6)]
        losses = []
        orig = task.on_train_batch_end
        task.on_train_batch_end = lambda loss, batch, i: (losses.append(float(loss)), orig(loss, batch, i))[1]
        HipTrainer(max_epochs=1, hip_graph=hip_graph).fit(task, train_dataloaders=batches)
        return losses

    eager, graph = fit(False), fit(True)
    assert len(eager) == len(graph) == 6 and all(l == l for l in graph)
    assert eager[:3] == graph[:3]  # the first two steps are the same code; the third loss depends only on them
    assert all(abs(a - b) <= 5e-3 * abs(a) for a, b in zip(eager, graph))


def test_graph_plus_bucket_reduce_equals_the_eager_step(cuda):
    """data-parallel form of the graph step (graph A = forward + backward, gradients handed to GradSync.reduce_grads,
    graph B = the optimizer step): at world size 1 the reduction is the identity, so the trajectory is the eager one"""
    from flairhip.distributed import GradSync
    from flairhip.graph import GraphedTrainStep, make_capturable

    def run(graph):
        task, _, cfg = make_pair(precision="bf16", seed=13)
        task.train()
        g = torch.Generator().manual_seed(4)
        batches = [{MOD: torch.randn(2, 5, 64, 64, generator=g).cuda(),
                    TASK: torch.randint(0, 19, (2, 64, 64), generator=g).to(torch.uint8).cuda()} for _ in range(4)]
        opt = torch.optim.AdamW(task.model.parameters(), lr=1e-3, weight_decay=0.01, fused=True)
        losses = []
        if graph:
            state = {k: v.clone() for k, v in task.state_dict().items()}
            sync = GradSync(task.model, hooks=False)
            stepper = GraphedTrainStep(task, opt, batches[0], warmup_steps=2, grad_reduce=sync.reduce_grads)
            task.load_state_dict(state)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            for b in batches:
                losses.append(stepper(b).item())
        else:
            make_capturable(opt)  # the optimizer arithmetic of a captured step (device-resident lr and step counter)
            for b in batches:
                loss = task.training_step(b, 0)
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                losses.append(loss.item())
        w = task.model.state_dict()[f"encoders.{MOD}.seg_model.layer2.0.conv1.weight"].float().cpu()
        return losses, w

    le, we = run(False)
    lg, wg = run(True)
    assert le == lg and torch.equal(we, wg)


def test_eval_after_graph_replays_uses_the_trained_weights(cuda):
    """a replay moves weights and BatchNorm statistics on the device without touching any tensor version: caches
    keyed on versions (packed operands, eval-mode BatchNorm folds) must be rebuilt -- the eval forward after graph
    training equals the eval forward of a fresh model loaded with the trained state dict, and an eager training step
    after replays equals the same step on that fresh model"""
    from flairhip.graph import GraphedTrainStep
    from helpers import make_pair
    task, _, _ = make_pair(precision="bf16")
    g = torch.Generator().manual_seed(23)
    batch = {MOD: torch.randn(4, 5, 64, 64, generator=g).to(cuda),
             TASK: torch.randint(0, 19, (4, 64, 64), generator=g).to(torch.uint8).to(cuda)}
    task.eval()
    with torch.no_grad():
        before = task.model(batch)[0][TASK].clone()  # fills the eval-fold caches with the initial weights
    task.train()
    opt = torch.optim.AdamW(task.model.parameters(), lr=1e-2, fused=True)
    graphed = GraphedTrainStep(task, opt, batch, warmup_steps=2)
    for _ in range(5):
        graphed(batch)
    torch.cuda.synchronize()
    task.eval()
    with torch.no_grad():
        after = task.model(batch)[0][TASK].clone()
    fresh, _, _ = make_pair(precision="bf16", seed=99)  # other initial weights, then the trained state
    fresh.model.load_state_dict(task.model.state_dict())
    fresh.eval()
    with torch.no_grad():
        want = fresh.model(batch)[0][TASK]
    assert not torch.equal(after, before)
    assert torch.equal(after, want)
    # eager training step after replays: packed operands must be those of the current weights
    task.train()
    fresh.train()
    l1, _, _ = task.step(batch, training=True)
    l2, _, _ = fresh.step(batch, training=True)
    assert torch.equal(l1, l2)
