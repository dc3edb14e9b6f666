"""Do two captured graphs replayed on two streams run concurrently on this runtime?  Each graph holds spin kernels
(torch.cuda._sleep); prints the elapsed time of replaying both (A on s1, B on s2) against one alone and against the
same kernels launched eagerly on the two streams.  usage: python tools/debug/graph_two_streams.py"""
import time

import torch

dev = torch.device("cuda", 0)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
CYC = 2_000_000


def make():
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4):
            torch.cuda._sleep(CYC // 4)
    return g


ga, gb = make(), make()
torch.cuda.synchronize()


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def one():
    with torch.cuda.stream(s1):
        ga.replay()


def both_graphs():
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        gb.replay()


def both_eager():
    with torch.cuda.stream(s1):
        for _ in range(4):
            torch.cuda._sleep(CYC // 4)
    with torch.cuda.stream(s2):
        for _ in range(4):
            torch.cuda._sleep(CYC // 4)


def graph_and_eager():
    with torch.cuda.stream(s1):
        ga.replay()
    with torch.cuda.stream(s2):
        for _ in range(4):
            torch.cuda._sleep(CYC // 4)


print(f"one graph                      {timed(one):7.3f} ms")
print(f"two graphs on two streams      {timed(both_graphs):7.3f} ms")
print(f"eager kernels on two streams   {timed(both_eager):7.3f} ms")
print(f"graph on s1 + eager on s2      {timed(graph_and_eager):7.3f} ms")


# the engine's lane pattern: graph on s1, fork event, second graph on s1, a graph on s2 behind the event
ga2 = make()
torch.cuda.synchronize()


def lane_pattern():
    with torch.cuda.stream(s1):
        ga.replay()
        s2.wait_stream(s1)
        ga2.replay()
    with torch.cuda.stream(s2):
        gb.replay()


print(f"A on s1; s2 waits; A2 on s1, B on s2 (ideal 2x one graph)   {timed(lane_pattern):7.3f} ms")
