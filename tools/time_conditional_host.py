#!/usr/bin/env python3
"""Host-side cost of one conditional-layer step through the captured engine (tools/bench_conditional.py's model):
wall time per step with and without the final synchronisation, and the time inside _CondProgram.load split into
metadata -> block indices ("li"), the optimiser job table ("jt"), waiting for a staging slot ("take") and the upload
call ("up": includes waiting for the previous graph launch to be handed to the device queue)."""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.bench_conditional as BC
from mmvae_amd import synthetic, engine as E
T = {"load": 0.0, "copy": 0.0, "run": 0.0, "n": 0}
orig_load = E._CondProgram.load
def timed_load(self, md):
    t0 = time.perf_counter(); orig_load(self, md); T["load"] += time.perf_counter() - t0; T["n"] += 1
E._CondProgram.load = timed_load
import numpy as _np
_orig_li = E._CondProgram._local_indices
def _li(self, ent, key, md):
    t0 = time.perf_counter(); r = _orig_li(self, ent, key, md); T["li"] = T.get("li", 0) + time.perf_counter() - t0; return r
E._CondProgram._local_indices = _li
from mmvae_amd import optim as _O
_orig_jt = _O.HipAdam.job_table
def _jt(self, *a):
    t0 = time.perf_counter(); r = _orig_jt(self, *a); T["jt"] = T.get("jt", 0) + time.perf_counter() - t0; return r
_O.HipAdam.job_table = _jt
_orig_take = E._PinnedRing.take
def _take(self):
    t0 = time.perf_counter(); r = _orig_take(self); T["take"] = T.get("take", 0) + time.perf_counter() - t0; return r
E._PinnedRing.take = _take
_orig_up = E._PinnedRing.upload
def _up(self, d):
    t0 = time.perf_counter(); r = _orig_up(self, d); T["up"] = T.get("up", 0) + time.perf_counter() - t0; return r
E._PinnedRing.upload = _up
orig_run = E._Plan.run
def timed_run(self):
    t0 = time.perf_counter(); r = orig_run(self); T["run"] += time.perf_counter() - t0; return r
E._Plan.run = timed_run
with tempfile.TemporaryDirectory() as d:
    model = BC.build(d, 20000, use_engine=True); model.train(); model.trainer.set_stage("training")
    B=512
    xs = {e: synthetic.synthetic_counts(B, 20000, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
    mds = [BC.metadata(B, ("human", "mouse")[i % 2], i) for i in range(72)]
    for i in range(8):
        eid = ("human", "mouse")[i % 2]; model.training_step((xs[eid], mds[i], eid), i)
    torch.cuda.synchronize()
    for k in T: T[k] = 0
    t0=time.perf_counter()
    for i in range(8, 72):
        eid = ("human", "mouse")[i % 2]; model.training_step((xs[eid], mds[i], eid), i)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    n = 64
    print("host ms/step", (t1-t0)/n*1e3, "incl sync", (t2-t0)/n*1e3, "load", T["load"]/n*1e3, "run", T["run"]/n*1e3, {k: round(v/n*1e3, 3) for k, v in T.items() if k not in ("n",)})
