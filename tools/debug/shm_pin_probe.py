#!/usr/bin/env python3
"""Feasibility of a feed in a process of its own: a torch shared-memory tensor registered as page-locked in the parent
(hipHostRegister), written by a spawned child that never touches the GPU, copied to the device asynchronously by the parent."""
import sys, time
import torch
import torch.multiprocessing as mp


def child(buf, go, done):
    for i in range(5):
        go.get()
        buf.fill_(i + 1)
        done.put(i + 1)


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    words = 4 * 1024 * 1024  # 16 MB
    buf = torch.zeros(words, dtype=torch.int32).share_memory_()
    rc = torch.cuda.cudart().cudaHostRegister(buf.data_ptr(), buf.numel() * 4, 0)
    print("cudaHostRegister:", rc, "is_pinned:", buf.is_pinned())
    go, done = ctx.SimpleQueue(), ctx.SimpleQueue()
    p = ctx.Process(target=child, args=(buf, go, done), daemon=True)
    t0 = time.perf_counter(); p.start()
    dev = torch.empty(words, dtype=torch.int32, device="cuda")
    for i in range(5):
        go.put(1)
        v = done.get()
        if i == 0:
            print(f"child up after {time.perf_counter() - t0:.2f} s")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        dev.copy_(buf, non_blocking=True)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ok = bool((dev == v).all())
        print(f"round {i}: value {v} ok {ok}; copy call {1e3 * (t2 - t1):.3f} ms, done after {1e3 * (t3 - t1):.3f} ms")
    p.join(timeout=5)
