#!/usr/bin/env python3
"""SURVEY 8 f1 / VERDICT r3 item 8, answered by measurement: the weight gradient of the expert encoder's first layer,
dW = dY^T x, from the sparse batch (gene-major ELL + mmvae_dw_sparse_ell_f32) against the dense bf16x3 GEMM over
pre-split planes (what the step runs), at BASELINE's and the reference's gene counts, 5 / 10 / 19 % of the entries
stored (19 %: the bench's synthetic batches).  Run on the GPU box; prints one table."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvae_amd import _lib, ops


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    lib = _lib.load()
    B, M = 512, 1024
    dev = "cuda"
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    print(f"{'genes':>6} {'stored':>7} | {'ELL build':>9} {'sparse dW':>10} {'= sum':>8} | {'dense dW (planes)':>18} {'x split':>8} | rel diff")
    for G in (20000, 52437, 60530):
        g = torch.Generator(device=dev).manual_seed(G)
        dY = torch.randn(B, M, device=dev, generator=g)
        for density in (0.05, 0.10, 0.19):
            m = torch.rand(B, G, device=dev, generator=g) < density
            x = torch.where(m, torch.rand(B, G, device=dev, generator=g) * 9.0, torch.zeros((), device=dev))
            Gp = (G + 7) // 8 * 8
            xpad = torch.zeros(B + 32, Gp, device=dev)
            xpad[:B, :G] = x
            rows = torch.zeros(G * B, dtype=torch.int32, device=dev)
            vals = torch.zeros(G * B, dtype=torch.float32, device=dev)
            cnt = torch.zeros(G, dtype=torch.int32, device=dev)
            dW = torch.empty(M, G, device=dev)

            def ell():
                _lib.check(lib.mmvae_ell_from_dense_f32(B, G, x.data_ptr(), G, B, rows.data_ptr(), vals.data_ptr(),
                                                        cnt.data_ptr(), st()), "ell")

            def sparse():
                _lib.check(lib.mmvae_dw_sparse_ell_f32(B, G, M, dY.data_ptr(), M, rows.data_ptr(), vals.data_ptr(),
                                                       cnt.data_ptr(), B, dW.data_ptr(), G, st()), "dw")

            t_ell, t_sp = timeit(ell), timeit(sparse)
            ref = (dY.double().t() @ x.double())
            err = float((dW.double() - ref).norm() / ref.norm())
            t_dense = t_split = float("nan")
            if Gp == G:
                xp, dYp = ops.split_planes(xpad[:B]), ops.split_planes(dY)
                t_split = timeit(lambda: ops.split_planes(xpad[:B], out=xp))
                t_dense = timeit(lambda: ops.gemm_planes(ops.GEMM_TN, None, None, a_planes=dYp, b_planes=xp, want_sq=True))
            else:
                dYs = torch.zeros(B + 32, M, device=dev)
                dYs[:B] = dY
                t_dense = timeit(lambda: ops.gemm(ops.GEMM_TN, dYs[:B], xpad[:B, :G]))
            print(f"{G:6d} {density * 100:6.0f}% | {t_ell:9.1f} {t_sp:10.1f} {t_ell + t_sp:8.1f} | {t_dense:18.1f} {t_split:8.1f} | {err:.1e}")
    print("us per 512-cell batch; dense: the TN bf16x3 GEMM the step runs (pre-split planes where the gene count allows), uncapped")


if __name__ == "__main__":
    main()
