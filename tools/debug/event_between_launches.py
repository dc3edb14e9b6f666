"""When does a stream that waits on an event recorded BETWEEN two launches of another stream get going -- behind the
first launch (correct) or behind the second?  main: spin A (0.2 ms), event, spin B (0.2 ms); side: wait(event), short
spin; the side stream's finishing time is measured on the side stream itself.  Kernels and captured graphs, default and
pool main stream.  usage: python tools/debug/event_between_launches.py"""
import torch

dev = torch.device("cuda", 0)
CYC = 400_000


def graph_of(n):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        torch.cuda._sleep(n)
    return g


gA, gB, gS = graph_of(CYC), graph_of(CYC), graph_of(2000)
torch.cuda.synchronize()


def run(main, use_graphs, label):
    side = torch.cuda.Stream(device=dev)
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tm = torch.cuda.Event(enable_timing=True)
        ev = torch.cuda.Event()
        with torch.cuda.stream(side):
            t0.record()
        with torch.cuda.stream(main):
            gA.replay() if use_graphs else torch.cuda._sleep(CYC)
            ev.record()
            gB.replay() if use_graphs else torch.cuda._sleep(CYC)
            tm.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            gS.replay() if use_graphs else torch.cuda._sleep(2000)
            t1.record()
        torch.cuda.synchronize()
        res.append(t0.elapsed_time(t1) * 1e3)
    print(f"{label}: side stream done after " + " / ".join(f"{r:5.0f}" for r in res) + " us  (A alone ~200, A + B ~400)")


run(torch.cuda.default_stream(dev), False, "kernels, main = default stream")
run(torch.cuda.Stream(device=dev), False, "kernels, main = pool stream   ")
run(torch.cuda.default_stream(dev), True, "graphs,  main = default stream")
run(torch.cuda.Stream(device=dev), True, "graphs,  main = pool stream   ")
