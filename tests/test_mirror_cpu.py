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
