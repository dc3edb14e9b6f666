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
