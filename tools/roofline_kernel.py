"""Launches the five G-wide GEMMs of the C2 step in the operand forms the engine uses (the expert encoder's first
layer: forward, weight gradient with dY pre-split; the decoder's last layer: fused forward + reconstruction, input
gradient, weight gradient with h pre-split) a few times each, for rocprofv3 --pmc passes (one counter group per run):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -o p -- python3 tools/roofline_kernel.py
usage: roofline_kernel.py [family | fwd | dw | dw_planes | fwd_half]
(fwd_half: the forward GEMM over half the genes -- half the MFMAs: shows that SQ_VALU_MFMA_BUSY_CYCLES counts, it is
equal for the five kernels of the family because they do equal work)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmvae_amd import ops, synthetic  # noqa: E402

cfg = synthetic.CONFIGS["c2"]
B, G, H1 = cfg["batch"], max(cfg["experts"].values()), 1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s: torch.randn(*s, device=dev, generator=g)


def padded(t):  # engine-style buffer: 32 zero slack rows behind the matrix
    full = torch.zeros(t.shape[0] + 32, t.shape[1], device=dev)
    full[: t.shape[0]] = t
    return full[: t.shape[0]]


X = padded(synthetic.synthetic_counts(B, G, seed=1234, device=dev))  # the bench's input distribution (~90 % zeros)
W1, W4, b4 = r(H1, G) * 0.03, r(G, H1) * 0.03, r(G) * 0.1
dY, h = padded(r(B, H1)), padded(torch.relu(r(B, H1)))
dP = padded(r(B, G) * (torch.rand(B, G, device=dev, generator=g) < 0.5))
dW1, dW4 = torch.empty(H1, G, device=dev), torch.empty(G, H1, device=dev)
sep = torch.empty(ops.recon_tiles(G), B, device=dev)
Xp, dYp, hp = ops.split_planes(X), ops.split_planes(dY), ops.split_planes(h)
which = sys.argv[1] if len(sys.argv) > 1 else "family"
if os.environ.get("MMVAE_RK_CAP"):  # the persistent kernels' grid capped as the engine caps it beside a branch
    from mmvae_amd import _lib

    _lib.load().mmvae_gemm_set_workgroup_cap(int(os.environ["MMVAE_RK_CAP"]))
for _ in range(8):
    if which in ("family", "fwd"):  # Y[B, 1024] = X . W1^T as 16 raw split-K slabs
        ops.gemm_slabs(ops.GEMM_NT, X, W1)
    if which == "family":
        ops.decoder_recon(h, W4, b4, X, want_xhat=False, dP=dP.clone(), se_part=sep)  # fused last layer + recon
        ops.gemm_slabs(ops.GEMM_NN, dP, W4)  # dX[B, 1024] = dP . W4
        ops.gemm_planes(ops.GEMM_TN, dP, None, b_planes=hp, out=dW4, want_sq=True)  # dW4[G, 1024] = dP^T . h  (h pre-split)
    if which == "family":  # dW1[1024, G] = dY^T . X with dY pre-split and the batch read as fp32 (the engine's form since late r5)
        ops.gemm_planes(ops.GEMM_TN, None, X, a_planes=dYp, out=dW1, want_sq=True)
    if which == "dw_planes":  # ... from pre-split planes of both (LDS-DMA stagers; gene counts off a multiple of 4)
        ops.gemm_planes(ops.GEMM_TN, None, None, a_planes=dYp, b_planes=Xp, out=dW1, want_sq=True)
    if which == "fwd_half":
        ops.gemm_slabs(ops.GEMM_NT, X[:, : G // 2], W1[:, : G // 2])
    if which == "dw":  # the same product with the in-kernel split (round 2's form)
        ops.gemm_planes(ops.GEMM_TN, dY, X, out=dW1, want_sq=True)
torch.cuda.synchronize()
print("algorithmic bytes per launch (forward GEMM):", 4 * (B * H1 + B * G + H1 * G))
