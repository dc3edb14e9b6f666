#!/usr/bin/env python3
"""A/B of two builds of libmmvae_hip.so on the five G-wide GEMMs of the C2 step, interleaved in ONE process
(rule: perf deltas come from interleaved rounds in one process).  usage: ab_gemm.py old.so new.so [rounds]"""
import ctypes as C
import statistics
import sys

import torch

libs = [C.CDLL(p) for p in sys.argv[1:3]]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
B, G, H1 = 512, 20000, 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
x, W1, dY1 = r(B, G), r(H1, G) * 0.01, r(B, H1)
h, W4, dP, b4 = r(B, H1), r(G, H1) * 0.01, r(B, G), r(G)
ws = torch.empty(64 * B * H1, device=dev)
out_big = torch.empty(H1 * G, device=dev)
sep = torch.empty(256 * B, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
f, i64, i32, u32, z = C.c_float, C.c_int64, C.c_int, C.c_uint, C.c_size_t
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(lib, layout, M, N, K, A, lda, Bm, ldb, Cm, ldc, flags, sk):
    rc = lib.mmvae_gemm_f32(i32(layout), i32(M), i32(N), i32(K), f(1.0), p(A), i64(lda), p(Bm), i64(ldb), p(Cm), i64(ldc),
                            None, u32(flags), i32(sk), p(ws), z(ws.numel() * 4), S())
    assert rc == 0, rc


cases = {
    "k1 enc-L1 fwd NT slabs x16": lambda lib: gemm(lib, 0, B, H1, G, x, G, W1, G, ws, H1, 4, 16),
    "k2 enc-L1 dW  TN 1024x20000x512": lambda lib: gemm(lib, 2, H1, G, B, dY1, H1, x, G, out_big, G, 0, 1),
    "k3 dec-L2 fwd+recon NT": lambda lib: lib.mmvae_decoder_recon_rows_f32(
        i32(B), i32(B), i32(G), i32(H1), p(h), i64(H1), p(W4), i64(H1), p(b4), p(x), i64(G), None, i64(0), p(out_big),
        i64(G), p(sep), S()),
    "k4a dec-L2 dW TN 20000x1024x512": lambda lib: gemm(lib, 2, G, H1, B, dP, G, h, H1, out_big, H1, 0, 1),
    "k4b dec-L2 dX NN slabs x16": lambda lib: gemm(lib, 1, B, H1, G, dP, G, W4, H1, ws, H1, 4, 16),
}


def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, fn in cases.items():
    for lib in libs:
        for _ in range(3):
            fn(lib)
    torch.cuda.synchronize()
    t = [[], []]
    for _ in range(rounds):
        for k, lib in enumerate(libs):
            t[k].append(timeit(lambda: fn(lib)))
    a, b = statistics.median(t[0]), statistics.median(t[1])
    print(f"{name:34s} A {a:7.1f} us (min {min(t[0]):6.1f})   B {b:7.1f} us (min {min(t[1]):6.1f})   B/A {b / a:5.3f}")

# bitwise agreement of the two builds (same products, same order -> identical results expected)
print("bitwise agreement of A and B:")
outs = []
for lib in libs:
    res = {}
    ws.zero_(); out_big.zero_(); sep.zero_()
    gemm(lib, 0, B, H1, G, x, G, W1, G, ws, H1, 4, 16); res["k1"] = ws[:16 * B * H1].clone()
    gemm(lib, 2, H1, G, B, dY1, H1, x, G, out_big, G, 0, 1); res["k2"] = out_big.clone()
    cases["k3 dec-L2 fwd+recon NT"](lib); res["k3_dP"] = out_big[:B * G].clone(); res["k3_se"] = sep.clone()
    gemm(lib, 2, G, H1, B, dP, G, h, H1, out_big, H1, 0, 1); res["k4a"] = out_big.clone()
    gemm(lib, 1, B, H1, G, dP, G, W4, H1, ws, H1, 4, 16); res["k4b"] = ws[:16 * B * H1].clone()
    torch.cuda.synchronize()
    outs.append(res)
for k in outs[0]:
    a, b = outs[0][k], outs[1][k]
    print(f"  {k}: equal={bool(torch.equal(a, b))} max|diff|={(a - b).abs().max().item():.3e} nan={bool(torch.isnan(b).any())}")
