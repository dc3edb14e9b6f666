"""Two data-parallel ranks on ONE GPU (process group over gloo, which stages CUDA tensors through the host): a rehearsal
of the N > 1 step engine program with real cross-process collectives -- the early shared-VAE exchange on the second
communicator, the expert's exchange + clip + Adam deferred onto the communication stream beside the next step,
collectives issued between graph replays in the same order on both ranks.  Checks: no deadlock, parameters identical
on both ranks after every step, and the engine's data-parallel result equals the module path's (whose data-parallel
arithmetic is pinned against hand-averaged gradients in tests/test_dist_gloo.py)."""
import os
import socket
import tempfile

import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from tests import helpers as H  # noqa: E402
from tests import mirror_utils as MU  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, out_dir, use_engine, name="two_mod_odd"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MMVAE_DIST_TIMEOUT_S="120")  # a stuck collective fails this test after 2 minutes
    from mmvae_amd import dist as mdist

    assert mdist.init_from_env("gloo") == world
    torch.cuda.set_device(0)
    case, z = H.load_case(name)
    with tempfile.TemporaryDirectory() as d:
        model = MU.build_mirror(case, "cuda", d, use_engine=use_engine)
        MU.load_state(model, z, "sd0/")
        model.train()
        model.trainer.set_stage("training")
        model.optimizers()
        mdist.broadcast_parameters(model)
        red = mdist.attach(model)
        assert red.small_group is not red.group
        if use_engine:
            assert model._get_engine(torch.zeros(1, device="cuda")).overlap, "N > 1 must select the overlapped program"
        schedule = list(case["schedule"]) * 2  # 6 steps: every plan is built, captured and replayed
        for t, eid in enumerate(schedule):
            x, eps, masks, labels = H.step_inputs(z, t % len(case["schedule"]))
            B = x.shape[0] // world
            rows = slice(rank * B, (rank + 1) * B)  # each rank trains its own cells
            meta = {c: [f"{c}_{int(i)}" for i in idx[rows]] for c, idx in labels.items()} or {"dummy": [0] * B}
            model.module.vae.encoder.explicit_eps = eps[rows].cuda()
            enc = model.module.experts[eid].encoder
            enc.explicit_masks = {int(k.split(".")[4]): m[rows].cuda() for k, m in masks.items()
                                  if k.startswith(f"experts.{eid}.encoder.fc_layers.")}
            model.training_step((x[rows].cuda(), pd.DataFrame(meta), eid), t)
            model._flush_engine()
            torch.cuda.synchronize()
            flat = torch.cat([p.detach().flatten() for p in model.module.parameters()])
            gathered = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(gathered, flat)
            assert all(torch.equal(gathered[0], g) for g in gathered), f"ranks diverged at step {t}"
        if rank == 0:
            skip = H.bn_fed_biases(H.spec_from_case(case))
            torch.save({n: p.detach().cpu() for n, p in model.module.named_parameters() if n not in skip},
                       os.path.join(out_dir, f"{name}.engine{int(use_engine)}.pt"))
    torch.cuda.synchronize()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["two_mod_odd", "adversarial"])
def test_two_ranks_on_one_gpu_engine_equals_module_path(tmp_path, name):
    """`adversarial`: the discriminator / generator phases exchange their (small) arenas inline, between graph segments."""
    world = 2
    for use_engine in (True, False):
        mp.spawn(_run, args=(world, _free_port(), str(tmp_path), use_engine, name), nprocs=world, join=True)
    a = torch.load(os.path.join(tmp_path, f"{name}.engine1.pt"))
    b = torch.load(os.path.join(tmp_path, f"{name}.engine0.pt"))
    assert a.keys() == b.keys()
    for n in a:
        assert H.rel_l2(a[n], b[n]) < 1e-4, (n, H.rel_l2(a[n], b[n]))
