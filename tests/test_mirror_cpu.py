"""Host-logic tests of the mirror on CPU plumbing (no GPU): the orchestration of CMMVAEModel.training_step --
phase order, optimiser routing, clipping, logged names, checkpoint keys -- replayed against the reference's golden
vectors.  The arithmetic on this path is plain torch (backend.cpu_plumbing), so this pins the HOST logic only; the
HIP kernels are pinned by the `-m gpu` tests."""
import pytest
import torch

from tests import helpers as H
from tests import mirror_utils as MU


@pytest.mark.parametrize("name", H.CASES + H.COND_CASES)
def test_training_steps_on_cpu_plumbing_match_reference(name):
    case, z, results = MU.replay_training(name, "cpu")
    MU.check_against_golden(case, z, results)


def test_cpu_tensors_are_refused_without_plumbing():
    import tempfile
    import pandas as pd
    from mmvae_amd import backend

    case, z = H.load_case("c1_small")
    with tempfile.TemporaryDirectory() as d, backend.cpu_plumbing():
        model = MU.build_mirror(case, "cpu", d)
    x = torch.rand(4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model.module(x, pd.DataFrame({"a": [0] * 4}), "human")


def test_state_dict_keys_match_reference_checkpoint():
    import tempfile
    from mmvae_amd import backend

    for name in H.CASES + H.COND_CASES:
        case, z = H.load_case(name)
        with tempfile.TemporaryDirectory() as d, backend.cpu_plumbing():
            model = MU.build_mirror(case, "cpu", d)
        ours = set(model.module.state_dict().keys())
        ref = {k[len("sd0/"):] for k in z.files if k.startswith("sd0/")}
        assert ours == ref, (sorted(ours - ref)[:5], sorted(ref - ours)[:5])
        assert all(k.startswith("module.") for k in model.state_dict().keys())


def test_optimizer_state_dict_round_trip_keeps_per_parameter_steps():
    """HipAdam on CPU plumbing: parameters without a gradient are skipped like torch.optim.Adam skips them and count
    their own steps; state_dict -> a fresh optimiser -> the same next update (the conditional layers' checkpoints)."""
    import torch

    from mmvae_amd import backend
    from mmvae_amd.optim import HipAdam

    def params():
        g = torch.Generator().manual_seed(0)
        return [torch.nn.Parameter(torch.randn(3, 5, generator=g)), torch.nn.Parameter(torch.randn(7, generator=g)),
                torch.nn.Parameter(torch.randn(2, 2, generator=g))]

    def grads(t):
        g = torch.Generator().manual_seed(100 + t)
        return [torch.randn(3, 5, generator=g), None if t % 2 else torch.randn(7, generator=g), torch.randn(2, 2, generator=g)]

    def step(opt, ps, t):
        for p, gr in zip(ps, grads(t)):
            p.grad = gr
        opt.step()

    with backend.cpu_plumbing(True):
        ref_p = params()
        ref = torch.optim.Adam(ref_p, lr=5e-3, weight_decay=1e-6)
        ps = params()
        opt = HipAdam(ps, lr=5e-3, weight_decay=1e-6)
        for t in range(3):
            step(ref, ref_p, t), step(opt, ps, t)
        sd = opt.state_dict()
        assert [int(sd["state"][i]["step"]) for i in range(3)] == [3, 2, 3]  # parameter 1 had no gradient at t = 1
        ps2 = params()
        for p, q in zip(ps2, ps):
            p.data.copy_(q.data)
        opt2 = HipAdam(ps2, lr=5e-3, weight_decay=1e-6)
        opt2.load_state_dict(sd)
        step(ref, ref_p, 3), step(opt, ps, 3), step(opt2, ps2, 3)
        for a, b, c in zip(ps, ps2, ref_p):
            assert torch.equal(a.data, b.data)
            assert torch.allclose(a.data, c.data, rtol=1e-6, atol=1e-7)
