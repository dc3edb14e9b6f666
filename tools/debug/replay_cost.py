#!/usr/bin/env python3
"""CPU cost of one replay (hipGraphLaunch) of each captured step program: C2, C4, the conditional model (sequential and
parallel order).  Replays back to back without touching anything else; host ms per replay and the device's."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.bench_conditional as BC
from mmvae_amd import instantiate, synthetic
from mmvae_amd.modules import base

def measure(label, plans, n=100):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        plans[i % len(plans)].run()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    nodes = [sum(len(seg) for seg in p.segments if not isinstance(seg, tuple)) for p in plans]
    print(f"{label:40s} host {1e3 * (t1 - t0) / n:.3f} ms / replay, device {1e3 * (t2 - t0) / n:.3f}; launches per program {nodes}")

dev = torch.device("cuda", 0)
for cfg_name, yaml in (("c2", "c2_two_modality_20k.yaml"), ("c4", "c4_two_modality_20k_adversarial.yaml")):
    base.Adversarial.labels.clear()
    os.environ["MMVAE_LABELS_DIR"] = synthetic.write_label_dir(tempfile.mkdtemp(prefix="mmvae_labels_"))
    path = os.path.join(os.path.dirname(__file__), "..", "..", "configs", "model", yaml)
    if not os.path.exists(path):
        print("no", path); continue
    model = instantiate.load_yaml(path).to(dev)
    model.train(); model.trainer.set_stage("training"); model.optimizers()
    cfg = synthetic.CONFIGS[cfg_name]
    B = cfg["batch"]
    data = {eid: synthetic.synthetic_labelled_batch(B, G, seed=7 + i, device=dev) for i, (eid, G) in enumerate(cfg["experts"].items())}
    eids = list(data)
    for s in range(8):
        eid = eids[s % 2]
        x, m = data[eid]
        model.training_step((x, m, eid), s)
    torch.cuda.synchronize()
    measure(cfg_name, [p for p in model._engine._plans.values()][:2])
    model._engine.close()
for parallel in (False, True):
    with tempfile.TemporaryDirectory() as d:
        model = BC.build(d, 20000, use_engine=True, parallel=parallel)
        model.train(); model.trainer.set_stage("training")
        xs = {e: synthetic.synthetic_counts(512, 20000, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
        for i in range(8):
            eid = ("human", "mouse")[i % 2]
            model.training_step((xs[eid], BC.metadata(512, eid, i), eid), i)
        torch.cuda.synchronize()
        measure(f"conditional, {'parallel' if parallel else 'sequential'}", list(model._engine._plans.values()))
        model._engine.close()
