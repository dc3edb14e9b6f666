#!/usr/bin/env python3
"""What the host pays per replay of the conditional-layer program, piece by piece: the replay call alone (tables left as
they are), replay + one trailing runtime call (event record), the upload call alone behind a replay, the numpy part of
CondProgram.load alone.   usage: cond_launch_cost.py [--parallel]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import tools.bench_conditional as BC
from mmvae_amd import synthetic

parallel = "--parallel" in sys.argv
with tempfile.TemporaryDirectory() as d:
    model = BC.build(d, 20000, use_engine=True, parallel=parallel)
    model.train(); model.trainer.set_stage("training")
    B = 512
    xs = {e: synthetic.synthetic_counts(B, 20000, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
    mds = [BC.metadata(B, ("human", "mouse")[i % 2], i) for i in range(16)]
    for i in range(8):
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], mds[i], eid), i)
    torch.cuda.synchronize()
    eng = model._engine
    plans = list(eng._plans.values())
    n = 100

    def loop(body, label):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            body(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{label:58s} host {1e3 * (t1 - t0) / n:.3f} ms / iteration, with the final sync {1e3 * (t2 - t0) / n:.3f}")

    loop(lambda i: plans[i % 2].run(), "replay only")
    ev = torch.cuda.Event()
    def a(i):
        plans[i % 2].run(); ev.record()
    loop(a, "replay + event record")
    def b(i):
        p = plans[i % 2]; p.run(); p.cond.ring.take(); p.cond.ring.upload(p.cond.pack_dev)
    loop(b, "replay + table upload (same table)")
    def c(i):
        p = plans[i % 2]; p.cond.load(mds[i % 16]); p.run()
    loop(c, "load (numpy + upload) + replay")
    def e(i):
        p = plans[i % 2]
        t = time.perf_counter()
        while time.perf_counter() - t < 0.75e-3:
            pass
        p.cond.ring.take(); p.cond.ring.upload(p.cond.pack_dev); p.run()
    loop(e, "0.75 ms of host spinning + upload + replay")
    # (copies below go to a DUMMY destination: the program's tables are never touched out of order)
    side = torch.cuda.Stream()
    dummy = torch.zeros_like(plans[0].cond.pack_dev)
    pin = plans[0].cond.ring.slots[0]
    def f(i):
        plans[i % 2].run()
        dummy.copy_(pin, non_blocking=True)
    loop(f, "replay + H2D copy on the SAME stream (dummy target)")
    def g(i):
        plans[i % 2].run()
        with torch.cuda.stream(side):
            dummy.copy_(pin, non_blocking=True)
    loop(g, "replay + H2D copy on ANOTHER stream (dummy target)")
    evs = [torch.cuda.Event() for _ in range(8)]
    def h(i):
        plans[i % 2].run()
        e = evs[i % 8]
        e.record()
        with torch.cuda.stream(side):
            side.wait_event(e)
            dummy.copy_(pin, non_blocking=True)
    loop(h, "replay + record + other stream waits, then copies")
    def k(i):
        with torch.cuda.stream(side):
            dummy.copy_(pin, non_blocking=True)
            e = evs[i % 8]
            e.record()
        torch.cuda.current_stream().wait_event(e)
        plans[i % 2].run()
    loop(k, "copy on another stream, main waits for it, replay")
