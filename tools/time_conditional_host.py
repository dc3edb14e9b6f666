#!/usr/bin/env python3
"""Host-side cost of one conditional-layer step through the captured engine (tools/bench_conditional.py's model):
wall time per step with and without the final synchronisation, and where the host spends it: CondProgram.load split into
metadata -> block indices ("li", of which the dictionary look-ups "lookup"), the per-position tables ("fill"), the optimiser job table ("jt"), waiting for a staging
slot ("take") and the upload call ("up": includes waiting for the previous graph launch to be handed to the device queue),
the replay call ("run") and the logging ("log").   usage: time_conditional_host.py [--parallel]"""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tools.bench_conditional as BC  # noqa: E402
from mmvae_amd import cond_tables, engine as E, engine_common, engine_cond, optim, synthetic  # noqa: E402

T = {}


def timed(owner, name, tag):
    orig = getattr(owner, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            T[tag] = T.get(tag, 0.0) + time.perf_counter() - t0

    setattr(owner, name, wrapper)


timed(engine_cond.CondProgram, "load", "load")
timed(engine_cond.CondProgram, "_local_indices", "li")
timed(cond_tables, "fill_all", "fill")
timed(cond_tables, "lookup_i32", "lookup")
timed(optim.HipAdam, "job_table", "jt")
timed(engine_common._PinnedRing, "take", "take")
timed(engine_common._PinnedRing, "upload", "up")
timed(E._Plan, "run", "run")
timed(E._Plan, "log", "log")
timed(E.StepEngine, "_check_signature", "sig")
timed(E.StepEngine, "_select_input", "sel")
timed(E.StepEngine, "training_step", "engine_step")

parallel = "--parallel" in sys.argv
with tempfile.TemporaryDirectory() as d:
    model = BC.build(d, 20000, use_engine=True, parallel=parallel)
    model.train()
    model.trainer.set_stage("training")
    B = 512
    xs = {e: synthetic.synthetic_counts(B, 20000, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
    mds = [BC.metadata(B, ("human", "mouse")[i % 2], i, categorical="--categorical" in sys.argv) for i in range(72)]
    for i in range(8):
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], mds[i], eid), i)
    torch.cuda.synchronize()
    T.clear()
    t0 = time.perf_counter()
    for i in range(8, 72):
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], mds[i], eid), i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    n = 64
    print(f"selection order {'parallel' if parallel else 'sequential'}: host ms/step {(t1 - t0) / n * 1e3:.3f}, "
          f"incl. the final sync {(t2 - t0) / n * 1e3:.3f}")
    print({k: round(v / n * 1e3, 3) for k, v in sorted(T.items(), key=lambda kv: -kv[1])})
