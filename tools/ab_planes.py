#!/usr/bin/env python3
"""Pre-split operands (include/mmvae_hip.h, "Pre-split operands") against the in-kernel split on the G-wide GEMMs of the
C2 step: bitwise agreement and interleaved timings in ONE process.  usage: ab_planes.py [rounds] [B] [G]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvae_amd import ops

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
G = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
H1 = 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
Bp = (B + 31) // 32 * 32


def padded(rows, cols):  # engine-style buffer: 32 zero rows of slack behind the matrix
    full = torch.zeros(rows + 32, cols, device=dev)
    full[:rows] = r(rows, cols)
    return full[:rows]


x, dY1, h, dP = padded(B, G), padded(B, H1), padded(B, H1), padded(B, G)
x.mul_((torch.rand(B, G, device=dev, generator=g) < 0.1).float())  # ~90 % zeros like the normalised counts
W1, W4 = r(H1, G) * 0.01, r(G, H1) * 0.01
xp, dYp, hp, dPp = (ops.split_planes(t) for t in (x, dY1, h, dP))
for t, p in ((x, xp), (dY1, dYp), (h, hp), (dP, dPp)):
    assert torch.equal(p.to_float(), t), "planes do not sum to the matrix"
    assert int(p.data[:, t.shape[0]:].abs().max()) == 0
print("split: planes sum exactly to the fp32 matrices; slack rows zero")

TN, NT, NN = 2, 0, 1
cases = {
    "k2 enc-L1 dW TN 1024xGxB": (lambda: ops.gemm_planes(TN, dY1, x, K=Bp, want_sq=True),
                                  lambda: ops.gemm_planes(TN, None, None, a_planes=dYp, b_planes=xp, K=Bp, want_sq=True)),
    "k4a dec-L2 dW TN Gx1024xB": (lambda: ops.gemm_planes(TN, dP, h, K=Bp, want_sq=True),
                                   lambda: ops.gemm_planes(TN, None, None, a_planes=dPp, b_planes=hp, K=Bp, want_sq=True)),
    "k1 enc-L1 fwd NT slabs": (lambda: ops.gemm_planes(NT, x, W1, raw_slabs=True),
                               lambda: ops.gemm_planes(NT, None, W1, a_planes=xp, raw_slabs=True)),
    "k4b dec-L2 dX NN slabs": (lambda: ops.gemm_planes(NN, dP, W4, raw_slabs=True),
                               lambda: ops.gemm_planes(NN, None, W4, a_planes=dPp, raw_slabs=True)),
}
b4 = r(G)
hrelu = torch.zeros(B + 32, H1, device=dev)
hrelu[:B] = torch.relu(r(B, H1))  # the decoder's hidden activations are post-ReLU
hrelu = hrelu[:B]
hrp = ops.split_planes(hrelu)
nrt = ops.recon_row_tiles(B)


def recon(planes):
    cp = torch.zeros(nrt, G, device=dev)
    _, dPo, se = ops.decoder_recon(hrelu, W4, b4, x, want_xhat=False, col_part=cp, h_planes=hrp if planes else None)
    return dPo, se, cp


cases["k3 dec-L2 fwd+recon NT"] = (lambda: recon(False), lambda: recon(True))
# (r5) the weights pre-split as well: the reconstruction kernel whose stagers only move planes, and dX with W4 planes
W4p = ops.split_planes(W4)


def recon_w():
    cp = torch.zeros(nrt, G, device=dev)
    _, dPo, se = ops.decoder_recon(hrelu, W4, b4, x, want_xhat=False, col_part=cp, h_planes=hrp, W_planes=W4p)
    return dPo, se, cp


cases["k3w recon NT h+W planes"] = (lambda: recon(False), recon_w)
cases["k4c dec-L2 dX NN W planes"] = (lambda: ops.gemm_planes(NN, dP, W4, raw_slabs=True),
                                      lambda: ops.gemm_planes(NN, dP, None, b_planes=W4p, raw_slabs=True))
splits = {"split x [BxG]": lambda: ops.split_planes(x, xp), "split dY [Bx1024]": lambda: ops.split_planes(dY1, dYp),
          "split W4 [Gx1024]": lambda: ops.split_planes(W4, W4p)}


def flat(o):
    return torch.cat([t.reshape(-1) for t in o]) if isinstance(o, tuple) else o.reshape(-1)


ok = True
for name, (f32, pl) in cases.items():
    if name.startswith("k3"):
        (dA, sA, cA), (dB, sB, cB) = f32(), pl()
        torch.cuda.synchronize()
        same = bool(torch.equal(dA, dB)) and bool(torch.equal(sA, sB))
        cerr = ((cA.sum(0) - cB.sum(0)).norm() / cA.sum(0).norm()).item()
        ok &= same and cerr < 1e-6
        print(f"{name:30s} dP, se_part bitwise equal: {same}   bias-gradient partial sums rel-L2 {cerr:.2e}")
        continue
    a, b = flat(f32()), flat(pl())
    torch.cuda.synchronize()
    same = bool(torch.equal(a, b))
    ok &= same
    print(f"{name:30s} bitwise equal: {same}   max|diff| {(a - b).abs().max().item():.3e}   nan {bool(torch.isnan(b).any())}")


def timeit(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


fl = 2.0 * B * G * H1
for name, (f32, pl) in cases.items():
    for f in (f32, pl):
        for _ in range(3):
            f()
    t = [[], []]
    for _ in range(rounds):
        for k, f in enumerate((f32, pl)):
            t[k].append(timeit(f))
    a, b = statistics.median(t[0]), statistics.median(t[1])
    print(f"{name:30s} fp32 operands {a:7.1f} us ({fl / a / 1e6:5.1f} TF)   planes {b:7.1f} us ({fl / b / 1e6:5.1f} TF, "
          f"{6 * fl / b / 1e6 / 2500:.3f} of the bf16 peak)   ratio {b / a:5.3f}")
for name, f in splits.items():
    for _ in range(3):
        f()
    print(f"{name:30s} {statistics.median(timeit(f) for _ in range(rounds)):7.1f} us")
print("ALL BITWISE EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
