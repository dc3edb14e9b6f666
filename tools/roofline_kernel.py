"""Launches the roofline kernel of bench.py (the TN weight-gradient GEMM of a G-wide layer) a few times, for
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
dY = torch.randn(B, H1, device=dev, generator=g)
X = torch.randn(B, G, device=dev, generator=g)
dW = torch.empty(H1, G, device=dev)
for _ in range(8):
    ops.gemm(ops.GEMM_TN, dY, X, out=dW, splitk=1)
torch.cuda.synchronize()
print("algorithmic bytes per launch:", 4 * (B * H1 + B * G + H1 * G))
