"""What does a fork inside a captured graph cost per replay on this runtime?  Graphs of spin kernels (torch.cuda._sleep)
with the same critical path -- 10 kernels -- replayed back to back on one stream: single-stream; one branch forked and
joined in the middle; forked early / joined late; two branches; a branch whose join is the graph's last dependency.
Prints microseconds per replay and the excess over the single-stream graph.  usage: python tools/debug/graph_fork_gap.py"""
import time

import torch

dev = torch.device("cuda", 0)
CYC = 100_000  # ~50 us per kernel
side1, side2 = torch.cuda.Stream(), torch.cuda.Stream()


def k():
    torch.cuda._sleep(CYC)


def fork(s):
    s.wait_stream(torch.cuda.current_stream())


def join(s):
    torch.cuda.current_stream().wait_stream(s)


def branch(s, n):
    with torch.cuda.stream(s):
        for _ in range(n):
            k()


def single():
    for _ in range(10):
        k()


def fork_mid():
    for _ in range(4):
        k()
    fork(side1)
    branch(side1, 2)
    for _ in range(3):
        k()
    join(side1)
    for _ in range(3):
        k()


def fork_early_join_late():
    k()
    fork(side1)
    branch(side1, 6)
    for _ in range(8):
        k()
    join(side1)
    k()


def two_branches():
    for _ in range(3):
        k()
    fork(side1)
    branch(side1, 2)
    for _ in range(2):
        k()
    fork(side2)
    branch(side2, 2)
    for _ in range(3):
        k()
    join(side1)
    join(side2)
    for _ in range(2):
        k()


def join_is_last():
    for _ in range(7):
        k()
    fork(side1)
    branch(side1, 2)
    for _ in range(3):
        k()
    join(side1)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def per_replay(g, n=300):
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


base = None
for name, fn in (("single stream, 10 kernels", single), ("one branch, forked and joined mid-graph", fork_mid),
                 ("forked behind the first kernel, joined ahead of the last", fork_early_join_late),
                 ("two branches", two_branches), ("the join is the graph's last dependency", join_is_last)):
    us = per_replay(capture(fn))
    base = us if base is None else base
    print(f"{name:60s} {us:8.1f} us per replay   {us - base:+7.1f}")
