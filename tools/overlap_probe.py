#!/usr/bin/env python3
"""Can the expert's Adam pass (HBM-bound) hide beside the NEXT step's G-wide forward GEMM (matrix-core-bound) when the
chip is partitioned by compute units?  Adam confined to N units (mmvae_adam_set_workgroups) on one stream, the forward
GEMM of the C2 step capped to 256 - N workgroups (mmvae_gemm_set_workgroup_cap) on another; makespan of the pair against
the two run one after the other.  usage: overlap_probe.py [rounds]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvae_amd import _lib, ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B, G, H1 = 512, 20000, 1024
P = 42_034_208
dev = "cuda"
lib = _lib.load()
gen = torch.Generator(device=dev).manual_seed(0)
x = torch.zeros(B + 32, G, device=dev)[:B]
x.copy_(torch.randn(B, G, device=dev, generator=gen))
W1 = torch.randn(H1, G, device=dev, generator=gen) * 0.01
param, grad = torch.randn(P, device=dev, generator=gen), torch.randn(P, device=dev, generator=gen) * 1e-3
m, v = torch.zeros(P, device=dev), torch.zeros(P, device=dev)
state = torch.tensor([0, 0, 1, 1, 1, 0, 0, 0], dtype=torch.float32, device=dev)
main, side = torch.cuda.Stream(), torch.cuda.Stream()


def gemm():
    return ops.gemm_planes(0, x, W1, raw_slabs=True)


def adam():
    rc = lib.mmvae_adam_step(P, param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(), 5e-3,
                             0.9, 0.999, 1e-8, 0.0, 1.0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0


def timed(fn_main, fn_side, reps=6):
    """Median makespan (us) and the two durations of fn_main on `main` beside fn_side on `side` (either may be None)."""
    span, dm, ds = [], [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m1, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(main):
            t0.record()
        side.wait_event(t0)
        if fn_side is not None:
            with torch.cuda.stream(side):
                fn_side()
                s1.record()
        with torch.cuda.stream(main):
            if fn_main is not None:
                fn_main()
            m1.record()
            if fn_side is not None:
                main.wait_event(s1)
            t1.record()
        torch.cuda.synchronize()
        span.append(t0.elapsed_time(t1) * 1e3)
        dm.append(t0.elapsed_time(m1) * 1e3)
        ds.append(t0.elapsed_time(s1) * 1e3 if fn_side is not None else 0.0)
    return statistics.median(span), statistics.median(dm), statistics.median(ds)


def burst(fn, n=4):
    def run():
        for _ in range(n):
            fn()
    return run


for _ in range(3):
    gemm(), adam()
torch.cuda.synchronize()
lib.mmvae_gemm_set_workgroup_cap(0)
lib.mmvae_adam_set_workgroups(0)
g_alone = timed(burst(gemm), None)[0] / 4
a_alone = timed(burst(adam), None)[0] / 4
print(f"alone, whole chip (bursts of 4): forward GEMM {g_alone:7.1f} us   Adam {a_alone:7.1f} us   one after the other "
      f"{g_alone + a_alone:7.1f} us")
serial = statistics.median(timed(lambda: (gemm(), adam()), None)[0] for _ in range(rounds))
serial2 = statistics.median(timed(lambda: (gemm(), gemm(), adam()), None)[0] for _ in range(rounds))
print(f"one launch each on ONE stream: GEMM + Adam {serial:7.1f} us   2 GEMMs + Adam {serial2:7.1f} us")
for n in (32, 48, 64, 80, 96, 128):
    lib.mmvae_adam_set_workgroups(n)
    a_n = timed(burst(adam), None)[0] / 4
    lib.mmvae_gemm_set_workgroup_cap(256 - n)
    g_n = timed(burst(gemm), None)[0] / 4
    res = [timed(gemm, adam) for _ in range(rounds)]
    span = statistics.median(r[0] for r in res)
    dg = statistics.median(r[1] for r in res)
    da = statistics.median(r[2] for r in res)
    # two GEMMs on the main stream (a forward GEMM, then a stand-in for what follows it) beside one Adam pass
    res2 = [timed(burst(gemm, 2), adam) for _ in range(rounds)]
    span2 = statistics.median(r[0] for r in res2)
    print(f"Adam on {n:3d} CUs alone {a_n:6.1f} us ({P * 28 / a_n / 1e6:4.2f} TB/s) | GEMM on {256 - n:3d} alone {g_n:6.1f} us | "
          f"together: span {span:6.1f} (GEMM done {dg:6.1f}, Adam done {da:6.1f}) | 2 GEMMs beside it: span {span2:6.1f}")
lib.mmvae_gemm_set_workgroup_cap(0)
lib.mmvae_adam_set_workgroups(0)
