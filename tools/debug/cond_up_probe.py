#!/usr/bin/env python3
"""Inside the real training loop of the conditional program: what does the table upload wait for?  Per step: was the
previous replay already finished when the upload started (event query), how long the copy call and the event record took."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.bench_conditional as BC
from mmvae_amd import engine_common, engine as E, synthetic

S = {"copy": 0.0, "rec": 0.0, "idle": 0, "n": 0, "q": 0.0}
last = {"ev": None}
orig_run = E._Plan.run
def run(self):
    r = orig_run(self)
    ev = torch.cuda.Event(); ev.record(); last["ev"] = ev
    return r
if os.environ.get("PROBE_RECORD", "1") == "1":
    E._Plan.run = run
def upload(self, dst):
    t0 = time.perf_counter()
    if last["ev"] is not None:
        S["idle"] += int(last["ev"].query())
    t1 = time.perf_counter()
    dst.copy_(self.slots[self.i], non_blocking=True)
    t2 = time.perf_counter()
    ev = torch.cuda.Event(); ev.record(); self.events[self.i] = ev
    t3 = time.perf_counter()
    S["q"] += t1 - t0; S["copy"] += t2 - t1; S["rec"] += t3 - t2; S["n"] += 1
engine_common._PinnedRing.upload = upload

parallel = "--parallel" in sys.argv
with tempfile.TemporaryDirectory() as d:
    model = BC.build(d, 20000, use_engine=True, parallel=parallel)
    model.train(); model.trainer.set_stage("training")
    B = 512
    xs = {e: synthetic.synthetic_counts(B, 20000, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
    mds = [BC.metadata(B, ("human", "mouse")[i % 2], i) for i in range(72)]
    for i in range(8):
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], mds[i], eid), i)
    torch.cuda.synchronize()
    for k in S: S[k] = 0
    t0 = time.perf_counter()
    for i in range(8, 72):
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], mds[i], eid), i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    n = 64
    print(f"host ms/step {(t1 - t0) / n * 1e3:.3f}; previous replay finished at upload time in {S['idle']}/{S['n']} steps; "
          f"query {S['q'] / n * 1e3:.3f} copy call {S['copy'] / n * 1e3:.3f} ms, event record {S['rec'] / n * 1e3:.3f} ms")
