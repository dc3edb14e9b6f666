"""How does the HIP graph executor schedule forked branches of a stream-captured graph?  Spin kernels of known length
(torch.cuda._sleep: one thread, no resource contention) in the shapes the engine emits; replay time tells which branches
overlapped.  usage: python tools/debug/graph_branch_probe.py"""
import sys
import time

import torch

dev = torch.device("cuda:0")
main = torch.cuda.Stream()
side = torch.cuda.Stream()
side2 = torch.cuda.Stream()

# calibrate: cycles per microsecond of _sleep
torch.cuda._sleep(1000)
torch.cuda.synchronize()
t0 = time.perf_counter()
torch.cuda._sleep(20_000_000)
torch.cuda.synchronize()
PER_US = 20_000_000 / ((time.perf_counter() - t0) * 1e6)


def k(us):
    torch.cuda._sleep(int(us * PER_US))


def run(name, body, ideal, serial):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        body()  # eager once
        main.synchronize()
        with torch.cuda.graph(g, stream=main):
            body()
        for _ in range(3):
            g.replay()
        main.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            g.replay()
        b.record()
        b.synchronize()
        print(f"{name:58s} {a.elapsed_time(b) * 100:7.1f} us   (ideal {ideal}, serial {serial})", flush=True)


def cur():
    return torch.cuda.current_stream()


def fork(s):
    s.wait_stream(cur())


def join(*ss):
    for s in ss:
        cur().wait_stream(s)


def on(s, *us):
    with torch.cuda.stream(s):
        for u in us:
            k(u)


def mark():
    e = torch.cuda.Event()
    e.record(cur())
    return e


def p_simple():  # A ; fork B ; C ; join ; D
    k(100); fork(side); on(side, 100); k(50); join(side); k(20)


def p_branch_first():  # chain ; fork side2 [V] ; K2 on main ; join
    k(50); fork(side2); on(side2, 30, 10); k(100); join(side2); k(20)


def p_main_first():  # chain ; mark ; K2 on main ; side2 waits mark [V] ; join
    k(50); e = mark(); k(100); side2.wait_event(e); on(side2, 30, 10); join(side2); k(20)


def p_step(early_late, late_first):
    """the engine's step: R ; K4b ; (early branch E on side2) ; fork side K4a ; chain ; K2 ‖ V on side2 ; join ; ADAM"""
    k(50)                      # recon
    e0 = mark()
    if not early_late:
        side2.wait_event(e0); on(side2, 10, 10)
    k(100)                     # K4b
    if early_late:
        side2.wait_event(e0); on(side2, 10, 10)
    fork(side); on(side, 150)  # K4a
    for _ in range(5):
        k(20)                  # chain
    e1 = mark()
    if late_first:
        side2.wait_event(e1); on(side2, 30, 10)
        k(100)                 # K2
    else:
        k(100)
        side2.wait_event(e1); on(side2, 30, 10)
    join(side, side2)
    k(50)


third = torch.cuda.Stream()


def p_step3():
    """as p_step but early and late branches on different streams"""
    k(50)
    e0 = mark()
    k(100)
    third.wait_event(e0); on(third, 10, 10)
    fork(side); on(side, 150)
    for _ in range(5):
        k(20)
    e1 = mark()
    k(100)
    side2.wait_event(e1); on(side2, 30, 10)
    join(side, side2, third)
    k(50)


print(f"spin calibration: {PER_US:.1f} cycles/us")
run("simple fork/join", p_simple, 220, 270)
run("branch emitted first, main kernel second", p_branch_first, 170, 210)
run("main kernel first, branch from a mark", p_main_first, 170, 210)
run("step: early branch before K4b, late branch before K2", lambda: p_step(False, True), 400, 460)
run("step: early before K4b, late after K2", lambda: p_step(False, False), 400, 460)
run("step: early after K4b, late after K2 (one side2 stream)", lambda: p_step(True, False), 400, 460)
run("step: early after K4b, late before K2", lambda: p_step(True, True), 400, 460)
run("step: early/late on different streams, both after", p_step3, 400, 460)


def p_v(early, late):
    """early: where the loss/bias kernels go -- 'main' (in line), 'side_after' (behind K4a on its stream), 'own' (own
    stream, emitted after K4b); late: where the VAE optimiser goes -- 'main', 'side' (behind K4a), 'side2' (own stream)"""
    k(50)
    e0 = mark()
    if early == "main":
        k(10); k(10)
    if early == "side_first":
        fork(side); on(side, 10, 10)
    if early == "own_first":
        fork(third); on(third, 10, 10)
    k(100)                     # K4b
    if early == "own":
        third.wait_event(e0); on(third, 10, 10)
    if early == "side_before":
        side.wait_event(e0); on(side, 10, 10)
    fork(side); on(side, 150)  # K4a
    if early == "side_after":
        on(side, 10, 10)
    for _ in range(5):
        k(20)                  # chain
    e1 = mark()
    if late == "main":
        k(30); k(10)
    k(100)                     # K2
    if late == "side":
        side.wait_event(e1); on(side, 30, 10)
    elif late == "side2":
        side2.wait_event(e1); on(side2, 30, 10)
    join(side)
    if late == "side2":
        join(side2)
    if early in ("own", "own_first"):
        join(third)
    k(50)


for early in ("side_after", "side_first", "own_first"):
    for late in ("main", "side", "side2"):
        ideal = 400 + (20 if early == "main" else 0) + (40 if late == "main" else 0)
        run(f"early={early} late={late}", lambda: p_v(early, late), ideal, 610)


def p_beside_adam():
    """VAE optimiser beside the expert's Adam: the expert waits for K4a only (an event on the side stream), the
    side stream's remaining work is joined at the end of the step"""
    k(50)
    k(100)                     # K4b
    fork(side); on(side, 150)  # K4a
    with torch.cuda.stream(side):
        evk = mark()
    on(side, 10, 10)           # loss words
    k(17)                      # bias column sums
    for _ in range(5):
        k(20)                  # chain
    e1 = mark()
    k(100)                     # K2
    side.wait_event(e1); on(side, 30, 10)   # VAE optimiser
    cur().wait_event(evk)
    k(50)                      # expert Adam
    join(side)
    k(5)


run("VAE optimiser beside the expert's Adam", p_beside_adam, 422, 632)


# ---- the same step as SEPARATE single-stream graphs joined by ordinary stream events (no forked capture)
def capture(body, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        body()
        stream.synchronize()
        with torch.cuda.graph(g, stream=stream):
            body()
    return g


def run_multi(name, launch, ideal):
    with torch.cuda.stream(main):
        for _ in range(3):
            launch()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            launch()
        b.record()
        b.synchronize()
        print(f"{name:58s} {a.elapsed_time(b) * 100:7.1f} us   (ideal {ideal})", flush=True)


g_a = capture(lambda: (k(50), k(100)), main)                        # recon, K4b
g_side1 = capture(lambda: (k(150), k(10), k(10)), side)             # K4a, loss words
g_chain = capture(lambda: [k(17)] + [k(20) for _ in range(5)], main)
g_k2 = capture(lambda: k(100), main)
g_v = capture(lambda: (k(30), k(10)), side)
g_tail = capture(lambda: (k(50), k(5)), main)
g_whole = capture(lambda: (k(50), k(100), k(17), [k(20) for _ in range(5)], k(100), k(50), k(5)), main)
e1, e2, e3 = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()


def launch_multi():
    g_a.replay(); e1.record(main)
    side.wait_event(e1)
    with torch.cuda.stream(side):
        g_side1.replay()
    g_chain.replay(); e2.record(main)
    g_k2.replay()
    side.wait_event(e2)
    with torch.cuda.stream(side):
        g_v.replay()
        e3.record(side)
    main.wait_event(e3)
    g_tail.replay()


def launch_cut_only():  # the main-stream work as four graphs back to back, nothing on the side
    g_a.replay(); g_chain.replay(); g_k2.replay(); g_tail.replay()


run_multi("main-stream kernels only, ONE graph", lambda: g_whole.replay(), 422)
run_multi("main-stream kernels only, four graphs back to back", launch_cut_only, 422)
run_multi("six single-stream graphs + stream events", launch_multi, 422)


def p_three(order):
    """VAE-owned work forks early (where the shared VAE's gradients are final), the expert's small work late:
    main: R K4b | side: K4a | chain1 | side2: E + Vvae | chain2 | K2 beside Vexp (side, behind K4a) | join | ADAM"""
    k(50); k(100)
    fork(side); on(side, 150)            # K4a
    for _ in range(5):
        k(20)                            # chain down to the VAE's last layer
    if order == "branch_first":
        fork(side2); on(side2, 10, 10, 30, 10)
        k(35); k(35)
    else:
        fork(side2)
        k(35); k(35)                     # the expert encoder's layers
        on(side2, 10, 10, 30, 10)
    fork(side)
    k(100)                               # K2
    on(side, 15, 5)                      # the expert's grouped GEMM + sums
    join(side, side2)
    k(50); k(5)


run("three streams, VAE branch emitted before chain2", lambda: p_three("branch_first"), 475, 700)
run("three streams, VAE branch emitted after chain2", lambda: p_three("main_first"), 475, 700)
