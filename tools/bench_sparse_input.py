#!/usr/bin/env python3
"""SURVEY 8 f1, answered by measurement: the expert encoder's first layer fed from the CSR batch directly (gather SpMM
over the transposed weight, mmvae_csr_spmm_wt_i32_f32) against the product path (densify the CSR batch with
mmvae_csr_to_dense, then the dense bf16x3 GEMM), at BASELINE and at the reference's real gene counts, 5 % and 10 % of
the entries stored.  Also priced: keeping a transposed copy of the weight (one pass per step) and the sparse-aware
reconstruction term.  Run on the GPU box; prints one table."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvae_amd import ops


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
    B, N = 512, 1024
    dev = "cuda"
    print(f"{'genes':>6} {'stored':>7} | {'densify':>8} {'dense GEMM':>11} {'= product':>10} | {'CSR SpMM':>9} {'+ W^T copy':>11} | gathered MB")
    for G in (20000, 52437, 60530):
        g = torch.Generator(device=dev).manual_seed(G)
        W = torch.randn(N, G, device=dev, generator=g) * 0.02
        Wt = W.t().contiguous()
        for density in (0.05, 0.10):
            m = torch.rand(B, G, device=dev, generator=g) < density
            d = torch.where(m, torch.rand(B, G, device=dev, generator=g) * 9.0, torch.zeros((), device=dev))
            xc = d.to_sparse_csr()
            x = torch.sparse_csr_tensor(xc.crow_indices().int(), xc.col_indices().int(), xc.values(), size=(B, G))
            out = torch.empty(B, G, device=dev)
            t_dens = timeit(lambda: ops.csr_to_dense(x, out=out))
            t_gemm = timeit(lambda: ops.gemm(ops.GEMM_NT, out, W))
            t_spmm = timeit(lambda: ops.csr_spmm_wt(x, Wt))
            t_tr = timeit(lambda: Wt.copy_(W.t()))  # a transposing pass per step (an upper bound: torch's copy kernel)
            y1, y2 = ops.gemm(ops.GEMM_NT, out, W), ops.csr_spmm_wt(x, Wt)
            err = float((y1 - y2).norm() / y1.norm())
            nnz = int(x.values().numel())
            print(f"{G:6d} {density * 100:6.0f}% | {t_dens:8.1f} {t_gemm:11.1f} {t_dens + t_gemm:10.1f} | {t_spmm:9.1f} {t_spmm + t_tr:11.1f} | "
                  f"{nnz * 4 * N / 1e6:8.0f}   (rel diff {err:.1e})")
    print("times in us per 512-cell batch, forward product of the first layer only; the weight gradient of that layer "
          "(dW = dz^T x) has the same gather / scatter shape")


if __name__ == "__main__":
    main()
