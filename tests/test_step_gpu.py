"""Whole-step parity on the GPU: CMMVAEModel.training_step (module path and graph-captured engine) through the HIP
library vs the golden vectors produced by the reference's own modules.  Tolerances (fp32, stated): scalar losses
rtol 2e-5; gradient norms rtol 5e-5; post-Adam parameters rel-L2 <= 1e-4; integer buffers exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests import helpers as H  # noqa: E402
from tests import mirror_utils as MU  # noqa: E402


@pytest.mark.parametrize("name", H.CASES)
def test_module_path_matches_reference(name):
    case, z, results = MU.replay_training(name, "cuda", use_engine=False)
    MU.check_against_golden(case, z, results)


@pytest.mark.parametrize("name", H.CASES)
def test_engine_path_matches_reference(name):
    case, z, results = MU.replay_training(name, "cuda", use_engine=True)
    MU.check_against_golden(case, z, results)


@pytest.mark.parametrize("name", H.CASES)
def test_engine_overlapped_exchange_program_matches_reference(name, monkeypatch):
    """The data-parallel program of the engine (early shared-VAE exchange on a second stream, the expert's exchange and
    clip + Adam deferred onto the communication stream, next use of the expert gated by an event), forced on one rank."""
    monkeypatch.setenv("MMVAE_DP_OVERLAP", "1")
    case, z, results = MU.replay_training(name, "cuda", use_engine=True)
    MU.check_against_golden(case, z, results)


def test_engine_overlapped_exchange_with_single_rank_rccl(monkeypatch):
    """Same program with a real process group (RCCL, one rank): every collective is issued, on both communicators,
    beside the graph replays."""
    import torch.distributed as td

    from mmvae_amd import dist as mdist

    monkeypatch.setenv("MMVAE_SINGLE_RANK_COLLECTIVES", "1")
    monkeypatch.setenv("MASTER_PORT", "29611")
    mdist.init_from_env()
    try:
        assert mdist.collectives_active()

        def prepare(model):
            model.optimizers()
            mdist.broadcast_parameters(model)
            red = mdist.attach(model)
            assert red.small_group is not red.group

        for name in ("two_mod_odd", "adversarial"):
            case, z, results = MU.replay_training(name, "cuda", use_engine=True, prepare=prepare)
            MU.check_against_golden(case, z, results)
    finally:
        torch.cuda.synchronize()
        td.destroy_process_group()
