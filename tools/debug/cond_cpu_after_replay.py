#!/usr/bin/env python3
"""Is host code slower right behind a replay of the conditional program?  A fixed CPU-only workload (six dictionary
look-up passes over 512 strings + the native table builder) timed (a) with the device idle, (b) right behind a replay,
(c) behind a replay and a 0.3 ms sleep."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import tools.bench_conditional as BC
from mmvae_amd import synthetic, cond_tables as CT

with tempfile.TemporaryDirectory() as d:
    model = BC.build(d, 20000, use_engine=True, parallel=True)
    model.train(); model.trainer.set_stage("training")
    B = 512
    xs = {e: synthetic.synthetic_counts(B, 20000, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
    mds = [BC.metadata(B, ("human", "mouse")[i % 2], i) for i in range(16)]
    for i in range(8):
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], mds[i], eid), i)
    torch.cuda.synchronize()
    plans = list(model._engine._plans.values())
    rng = np.random.default_rng(0)
    keys = [f"donor_{i}" for i in range(4644)]
    table = {k: i for i, k in enumerate(keys)}
    cols = [[keys[i] for i in rng.integers(0, 4644, 512)] for _ in range(6)]
    out = np.zeros((6, 512), dtype=np.int32)
    seg = np.zeros(6 * CT.words(512), dtype=np.int32)
    base = np.zeros(6, dtype=np.int32)

    def work():
        t0 = time.perf_counter()
        for j in range(6):
            CT.lookup_i32(table, cols[j], out[j])
        CT.fill_all(seg, CT.words(512), out, base, 512)
        return time.perf_counter() - t0

    def stat(label, pre):
        ts = []
        for i in range(60):
            pre(i)
            ts.append(work())
            torch.cuda.synchronize()
        ts = sorted(ts[10:])
        print(f"{label:44s} median {1e6 * ts[len(ts) // 2]:7.1f} us   max {1e6 * ts[-1]:7.1f} us")

    stat("device idle", lambda i: None)
    stat("right behind a replay", lambda i: plans[i % 2].run())
    def pre(i):
        plans[i % 2].run(); time.sleep(0.0003)
    stat("behind a replay + 0.3 ms sleep", pre)
    def pre2(i):
        plans[i % 2].run(); plans[(i + 1) % 2].run()
    stat("behind two replays", pre2)
    print("cpu count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
