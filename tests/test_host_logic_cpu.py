"""Host-side bookkeeping of the product that needs no GPU (round-1 advisor findings)."""
import torch

from helpers import ROOT  # noqa: F401  (sets sys.path)


def test_batchnorm_pending_batch_count_does_not_survive_a_load():
    from flairhip.nn import HipBatchNorm2d
    bn = HipBatchNorm2d(8)
    bn.note_batch()
    bn.note_batch()  # two training batches counted lazily
    assert int(bn.state_dict()["num_batches_tracked"]) == 2
    bn.note_batch()
    sd = {k: v.clone() for k, v in bn.state_dict().items()}  # flushes: 3
    sd["num_batches_tracked"] = torch.tensor(40)
    bn.note_batch()  # pending again when the checkpoint arrives
    bn.load_state_dict(sd)
    assert int(bn.state_dict()["num_batches_tracked"]) == 40  # not 41: the loaded count is the truth
    bn.note_batch()
    assert int(bn.state_dict()["num_batches_tracked"]) == 41


def test_outgrown_workspaces_stay_alive_for_captured_graphs():
    from flairhip import ops
    dev = torch.device("cpu")
    a = ops.workspace(1 << 20, dev, "unit-test-slot")
    ptr = a.data_ptr()
    b = ops.workspace(4 << 20, dev, "unit-test-slot")
    assert b.numel() >= (4 << 20) and b.data_ptr() != ptr
    # the superseded buffer is still referenced (a replayed hipGraph may keep writing through its raw pointer)
    assert any(t.data_ptr() == ptr for t in ops._retired_workspaces)
    assert ops.workspace(1 << 20, dev, "unit-test-slot") is b  # grow-only
