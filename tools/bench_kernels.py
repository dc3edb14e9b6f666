#!/usr/bin/env python3
"""Micro-benchmark of the individual HIP kernels at the BASELINE config-2 shapes (B=512, G=20000).
Prints one line per kernel: time, TFLOP/s (fp32 MFMA peak 157.3) or GB/s (HBM peak 8000).  Run on the GPU box."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mmvae_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--G", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    B, G, H1, H2 = a.B, a.G, 1024, 512
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    x, W1, dY1 = r(B, G), r(H1, G) * 0.01, r(B, H1)
    h, W4, dP = r(B, H1), r(G, H1) * 0.01, r(B, G)
    b4 = r(G)
    rows = []

    def rec(name, t, flops=None, bytes_=None):
        if flops:
            rows.append(f"{name:34s} {t*1e6:9.1f} us  {flops/t/1e12:7.1f} TFLOP/s  ({flops/t/157.3e12*100:5.1f}% of fp32 MFMA peak)")
        else:
            rows.append(f"{name:34s} {t*1e6:9.1f} us  {bytes_/t/1e9:7.0f} GB/s     ({bytes_/t/8e12*100:5.1f}% of HBM peak)")

    fl = 2.0 * B * G * H1
    rec(f"k1 enc-L1 fwd NT [{B}x{H1}x{G}] slabs", timeit(lambda: ops.gemm_slabs(0, x, W1), a.iters), fl)
    rec(f"k1 enc-L1 fwd NT +reduce", timeit(lambda: ops.gemm(0, x, W1), a.iters), fl)
    dW1 = torch.empty(H1, G, device=dev)
    rec(f"k2 enc-L1 dW TN [{H1}x{G}x{B}]", timeit(lambda: ops.gemm(2, dY1, x, out=dW1), a.iters), fl)
    xh, dPo, sep = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev), torch.empty(ops.recon_tiles(G), B, device=dev)
    rec(f"k3 dec-L2 fwd+recon NT [{B}x{G}x{H1}]", timeit(lambda: ops.decoder_recon(h, W4, b4, x, xhat=None, want_xhat=False, dP=dPo, se_part=sep), a.iters), fl)
    rec(f"k3 plain NT gemm same shape", timeit(lambda: ops.gemm(0, h, W4, out=xh), a.iters), fl)
    dW4 = torch.empty(G, H1, device=dev)
    rec(f"k4a dec-L2 dW TN [{G}x{H1}x{B}]", timeit(lambda: ops.gemm(2, dP, h, out=dW4), a.iters), fl)
    rec(f"k4b dec-L2 dX NN [{B}x{H1}x{G}] slabs", timeit(lambda: ops.gemm_slabs(1, dP, W4), a.iters), fl)
    for (M, N, K) in [(B, 512, 1024), (B, 256, 512), (B, 128, 256), (B, 1024, 512)]:
        A_, B_ = r(M, K), r(N, K)
        rec(f"k5 small NT [{M}x{N}x{K}]", timeit(lambda: ops.gemm(0, A_, B_), a.iters), 2.0 * M * N * K)
        At, Bt = r(M, N), r(M, K)  # dW = dy^T x : [N,K]
        rec(f"k5 small TN dW [{N}x{K}x{M}]", timeit(lambda: ops.gemm(2, At, Bt), a.iters), 2.0 * M * N * K)
    n = H1 * G * 2 + 2 * H1 * H2
    p, gr, m, v = r(n), r(n), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    st, part = torch.zeros(8, device=dev), torch.empty(ops.sqnorm_partials(n), device=dev)
    rec("k12 grad sqnorm", timeit(lambda: ops.clip_adam_step(p, gr, m, v, st, part, max_norm=10.0, do_step=False), a.iters), bytes_=4.0 * n)
    rec("k13 adam step", timeit(lambda: ops.clip_adam_step(p, gr, m, v, st, part, do_norm=False), a.iters), bytes_=28.0 * n)
    sl = r(16, B, H1)
    bn = dict(gamma=torch.ones(H1, device=dev), beta=torch.zeros(H1, device=dev), running_mean=torch.zeros(H1, device=dev),
              running_var=torch.ones(H1, device=dev), num_batches_tracked=None)
    mask = (torch.rand(B, H1, device=dev) > 0.1).to(torch.uint8)
    rec("k6 bn+relu+dropout fwd (16 slabs)", timeit(lambda: ops.fc_epilogue_fwd(sl, None, bn=bn, relu=True, keep_mask=mask, dropout_p=0.1), a.iters), bytes_=4.0 * B * H1 * 19)
    rec("k9 mse fwd+bwd", timeit(lambda: ops.mse_sum_fwd_bwd(xh, x), a.iters), bytes_=12.0 * B * G)
    rec("colsum dP (db4)", timeit(lambda: ops.fc_epilogue_bwd(dP, want_dz=False), a.iters), bytes_=4.0 * B * G)
    print("\n".join(rows))


if __name__ == "__main__":
    main()
