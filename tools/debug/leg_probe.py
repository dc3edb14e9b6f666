"""Roofline-leg variants: duration of the G-wide GEMMs launched back to back, alone, vs inside the step (profiles)."""
import sys
import torch
sys.path.insert(0, ".")
from mmvae_amd import ops

B, G, H = 512, 20000, 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(B, G, device=dev, generator=g)
W = torch.randn(H, G, device=dev, generator=g)
dY = torch.randn(B, H, device=dev, generator=g)
dW = torch.empty(H, G, device=dev)
cases = {"k1 NT slabs": lambda: ops.gemm_slabs(ops.GEMM_NT, X, W),
         "k2 TN dW": lambda: ops.gemm(ops.GEMM_TN, dY, X, out=dW, splitk=1)}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for iters in (1, 2, 4, 10, 20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{name}: {iters:2d} back to back: {e0.elapsed_time(e1) / iters * 1e3:.1f} us per launch")
