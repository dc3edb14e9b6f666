"""The program `bench.py` times -- BASELINE config 2 from its YAML, resident batches (pointer-keyed plans), captured and
REPLAYED hipGraphs with the library's Philox noise and the side branches on -- against the oracle, step by step
(oracle/program_check.py: pre-step state snapshotted, the noise the Philox fill left in the plan's buffers and the ReLU
slopes the step took fed to oracle.train_step).  Tolerances: losses 1e-4, gradient norms 5e-5, gradients 1e-4,
post-step parameters 1e-4 (cold Adam step 1e-3).  Reference: models/cmmvae_model.py:138-217."""
import argparse
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_program_matches_oracle():
    import bench
    from mmvae_amd import synthetic
    from oracle import program_check as PC

    a = argparse.Namespace(config="c2", genes="", no_engine=False)
    cfg = dict(synthetic.CONFIGS["c2"])
    device = torch.device("cuda", 0)
    model = bench.build_model(a, cfg, device).to(device)
    model.train()
    model.trainer.set_stage("training")
    model.optimizers()
    B = cfg["batch"]
    eids = list(cfg["experts"].keys())
    n_res = 2
    data = {eid: [(synthetic.synthetic_counts(B, G, seed=1234 + 97 * i + 13 * j, device=device),
                   synthetic.synthetic_metadata(B, seed=5 + j)) for j in range(n_res)]
            for i, (eid, G) in enumerate(cfg["experts"].items())}

    def batch(i):
        eid = eids[i % len(eids)]
        x, meta = data[eid][(i // len(eids)) % n_res]
        return x, meta, eid

    period = len(eids) * n_res
    step = 0
    # the very first step of every expert is a cold Adam step taken eagerly (plan build): checked too
    first = []
    for i in range(period):
        x, meta, eid = batch(step)
        first.append(PC.check_step(model, eid, x, meta, step))
        step += 1
    # bench.py's set-up: step every resident batch until its plan replays from the graph
    for i in range(3 * period):
        x, meta, eid = batch(step)
        model.training_step((x, meta, eid), step)
        step += 1
    rows = []
    for i in range(12):
        x, meta, eid = batch(step)
        r = PC.check_step(model, eid, x, meta, step)
        assert r["replayed"] and r["philox"], r  # the program under test: a replayed graph with device noise
        rows.append(r)
        step += 1
    assert any(r["forked"] for r in rows), "the bench program runs with its side branches"
    worst = {k: max(r[k] for r in rows) for k in ("loss", "recon_loss", "kl_loss", "grad_norm_vae", "grad_norm_expert",
                                                   "grad", "param")}
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "bench_program_parity.jsonl"), "w") as f:
        for r in first + rows:
            f.write(json.dumps(r) + "\n")
        f.write(json.dumps({"worst_of_12_replayed_steps": worst, "kinks": sum(r["kinks"] for r in rows)}) + "\n")
    model._engine.close()
