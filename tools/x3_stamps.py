"""Diagnostic: per-phase cycle sums of the bf16x3 loop (library built with EXTRA=-DMMVAE_X3_STAMPS=1)."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from mmvae_amd import _lib, ops

lib = _lib.load()
fn = lib.mmvae_debug_x3_stamps
B, G, H = 512, 20000, 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
cases = {
    "k2 TN dW [1024x20000x512]": (ops.GEMM_TN, torch.randn(B, H, device=dev, generator=g), torch.randn(B, G, device=dev, generator=g), (H, G)),
    "k3 NT [512x20000x1024]": (ops.GEMM_NT, torch.randn(B, H, device=dev, generator=g), torch.randn(G, H, device=dev, generator=g), (B, G)),
    "k4b NN [512x1024x20000]": (ops.GEMM_NN, torch.randn(B, G, device=dev, generator=g), torch.randn(G, H, device=dev, generator=g), (B, H)),
}
names = ["frag0 wait", "mfma0+split", "gload+frag1 wait", "mfma1 issue", "barrier1", "lds write", "barrier2"]
for name, (lay, a, b, shp) in cases.items():
    out = torch.empty(shp, device=dev)
    for _ in range(3):
        ops.gemm(lay, a, b, out=out, splitk=1)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    assert fn(buf) == 0
    print(name)
    for w in range(4):
        n = buf[w * 8 + 7]
        tot = sum(buf[w * 8 + i] for i in range(7))
        print(f"  wave {w}: tiles {n}  total/tile {tot / max(n, 1):7.0f} cyc   " +
              "  ".join(f"{names[i]} {buf[w * 8 + i] / max(n, 1):6.0f}" for i in range(7)))
