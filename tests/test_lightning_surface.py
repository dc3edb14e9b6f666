"""The Lightning-shaped plugin surface ("Trainer plugin surface"): `BaseModel` derives from
`lightning.pytorch.LightningModule` whenever that package is importable.  Lightning is not installed in the build image,
so the branch is driven through a stand-in package (tests/fake_lightning: wrapper optimisers, attachable trainer,
p.grad-clipping `clip_gradients`, counted calls) in a CHILD process, over golden cases of the reference on CPU plumbing:
the step must call `self.optimizers()` / `manual_backward` / `log_dict` of the base, must NOT leave the clip to the
base's p.grad clipping (HipAdam steps from its gathered arena), and its results must be the reference's.
Reference: /root/reference/src/cmmvae/models/base_model.py:51-123, models/cmmvae_model.py:138-217."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import torch
import mmvae_amd.models.base_model as bm
import lightning.pytorch as pl

assert bm.HAVE_LIGHTNING and issubclass(bm.BaseModel, pl.LightningModule), "the Lightning branch must be the one under test"
from mmvae_amd import backend
from tests import mirror_utils as MU

for name in sys.argv[1:]:
    holder = []
    with backend.cpu_plumbing():
        case, z, results = MU.replay_training(name, "cpu", use_engine=False, prepare=holder.append)
        MU.check_against_golden(case, z, results)
    model = holder[0]
    assert isinstance(model, pl.LightningModule) and model.automatic_optimization is False
    c = model.calls
    assert c["optimizers"] > 0 and c["manual_backward"] >= len(results) and c["log_dict"] > 0, dict(c)
    # the clip is the fused one: Lightning's p.grad clipping (too late for an optimiser that steps from its arena) never ran
    assert c["lightning_clip_gradients"] == 0, dict(c)
    opts = model.optimizers()
    assert isinstance(opts, list) and not any(isinstance(o, pl.LightningOptimizer) for o in opts)
    assert any("grad_norms/vae" == k for k in model.logged) and any(k.startswith("loss/training/") for k in model.logged)
    print("LIGHTNING_SURFACE_OK", name, dict(c), flush=True)
"""


def test_lightning_branch_runs_golden_cases_on_cpu_plumbing():
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "fake_lightning"), ROOT, env.get("PYTHONPATH", "")])
    cases = ["two_mod_odd", "adversarial", "clip_value"]
    r = subprocess.run([sys.executable, "-c", CHILD, *cases], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.stdout.count("LIGHTNING_SURFACE_OK") == len(cases), r.stdout[-1500:]


def test_stand_alone_branch_is_the_default_here():
    import mmvae_amd.models.base_model as bm

    assert bm.HAVE_LIGHTNING is False  # (no stand-in on this process's path: the suite runs the stand-alone surface)
