"""Launches the roofline kernel of bench.py (the forward GEMM of the G-wide expert encoder layer; `dw`: the TN
weight-gradient GEMM) a few times, for
rocprofv3 --pmc passes:   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python3 tools/roofline_kernel.py"""
import sys

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmvae_amd import ops, synthetic  # noqa: E402

cfg = synthetic.CONFIGS["c2"]
B, G, H1 = cfg["batch"], max(cfg["experts"].values()), 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn(B, G, device=dev, generator=g)
W = torch.randn(H1, G, device=dev, generator=g)
dY = torch.randn(B, H1, device=dev, generator=g)
dW = torch.empty(H1, G, device=dev)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
for _ in range(8):
    if which == "fwd":   # bench.py's roofline kernel: Y[B, 1024] = X . W^T as 16 raw split-K slabs
        ops.gemm_slabs(ops.GEMM_NT, X, W)
    else:                # the weight gradient dW[1024, G] = dY^T . X (round 1's roofline kernel)
        ops.gemm(ops.GEMM_TN, dY, X, out=dW, splitk=1)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", 4 * (B * H1 + B * G + H1 * G))
