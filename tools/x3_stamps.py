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
    "k1 NT split [512x1024x20000]": (ops.GEMM_NT, torch.randn(B, G, device=dev, generator=g), torch.randn(H, G, device=dev, generator=g), (B, H)),
    "k4b NN [512x1024x20000]": (ops.GEMM_NN, torch.randn(B, G, device=dev, generator=g), torch.randn(G, H, device=dev, generator=g), (B, H)),
}
names = ["frag0 wait", "mfma0+split", "gload+frag1 wait", "mfma1 issue", "barrier1", "lds write", "barrier2"]
for name, (lay, a, b, shp) in cases.items():
    out = torch.empty(shp, device=dev)
    for _ in range(3):
        ops.gemm(lay, a, b, out=out, splitk=(0 if 'split' in name or 'k4b' in name else 1))
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    assert fn(buf) == 0
    nb = 2048
    tr = (ctypes.c_longlong * (4 * nb))()
    nb_used = {"k2": 1000, "k3": 500, "k1": 512, "k4b": 512, "k4a": 1000}[name.split()[0]]
    assert lib.mmvae_debug_x3_trace(tr, nb) == 0
    import numpy as np
    T = np.array(tr[:], dtype=np.int64).reshape(nb, 4)
    T = T[:nb_used]
    t0 = T[:, 0].min()
    T = (T - t0) / 100.0  # us
    loop_us = T[:, 2] - T[:, 1]
    xcd = np.arange(len(T)) % 8
    print("   main loop by XCD (bid % 8):", " ".join(f"{loop_us[xcd == x].mean():.0f}/{loop_us[xcd == x].max():.0f}" for x in range(8)), "(mean/max us)")
    slot = np.arange(len(T)) // 8
    print("   main loop by bid//8 quartile:", " ".join(f"{loop_us[(slot * 4 // (slot.max() + 1)) == k].mean():.0f}" for k in range(4)))
    order = np.argsort(T[:, 0])
    T = T[order]
    print(name, f": {len(T)} workgroups; kernel span {T[:, 3].max():.1f} us")
    print(f"   entry: first wave of workgroups (<5us) {int((T[:, 0] < 5).sum())}, later starts at "
          f"{np.percentile(T[T[:, 0] >= 5, 0], [0, 50, 100]) if (T[:, 0] >= 5).any() else '-'} us")
    print(f"   prologue (entry->loop) median {np.median(T[:, 1] - T[:, 0]):.1f} us, max {np.max(T[:, 1] - T[:, 0]):.1f}")
    print(f"   main loop median {np.median(T[:, 2] - T[:, 1]):.1f} us, min {np.min(T[:, 2] - T[:, 1]):.1f}, max {np.max(T[:, 2] - T[:, 1]):.1f}")
    print(f"   epilogue median {np.median(T[:, 3] - T[:, 2]):.1f} us, max {np.max(T[:, 3] - T[:, 2]):.1f}")
    first = T[T[:, 0] < 5]
    print(f"   first-round exits: {np.percentile(first[:, 3], [0, 50, 100])} us")
    for w in range(4):
        n = buf[w * 8 + 7]
        tot = sum(buf[w * 8 + i] for i in range(7))
        print(f"  wave {w}: tiles {n}  total/tile {tot / max(n, 1):7.0f} cyc   " +
              "  ".join(f"{names[i]} {buf[w * 8 + i] / max(n, 1):6.0f}" for i in range(7)))
