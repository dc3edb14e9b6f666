"""Diagnostic: where the wave-specialised bf16x3 kernel over PRE-SPLIT operands spends its cycles (library built with
-DMMVAE_X3_STAMPS=1 for gemm_planes.hip; MMVAE_LIB points at it)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from mmvae_amd import _lib, ops

lib = _lib.load()
B, G, H = 512, 20000, 1024
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
x, dY, h, dP, W1, W4 = r(B, G), r(B, H), r(B, H), r(B, G), r(H, G), r(G, H)
xp, dYp, hp, dPp = (ops.split_planes(t) for t in (x, dY, h, dP))
TN, NT, NN = 2, 0, 1
cases = {
    "k2 TN dW [1024x20000x512] planes/planes": lambda: ops.gemm_planes(TN, None, None, a_planes=dYp, b_planes=xp, want_sq=True),
    "k4a TN dW [20000x1024x512] planes/planes": lambda: ops.gemm_planes(TN, None, None, a_planes=dPp, b_planes=hp, want_sq=True),
    "k1 NT slabs [512x1024x20000] planes/fp32": lambda: ops.gemm_planes(NT, None, W1, a_planes=xp, raw_slabs=True),
    "k4b NN slabs [512x1024x20000] planes/fp32": lambda: ops.gemm_planes(NN, None, W4, a_planes=dPp, raw_slabs=True),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    assert lib.mmvae_debug_x3p_stamps(buf) == 0
    nb = 256
    tr = (ctypes.c_longlong * (4 * 2048))()
    assert lib.mmvae_debug_x3p_trace(tr, 2048) == 0
    T = np.array(tr[:], dtype=np.int64).reshape(2048, 4)[:nb]
    T = (T - T[:, 0].min()) / 100.0
    print(f"{name}: kernel span {T[:, 3].max():.1f} us; last item's loop end {np.median(T[:, 2]):.1f} us (median), epilogue+exit {np.median(T[:, 3] - T[:, 2]):.1f} us")
    m = np.array(buf[:16], dtype=np.float64).reshape(4, 4)  # [what][wave]
    n = m[3]
    print("   multipliers per k-tile:  before barrier %s   in barrier %s   behind it %s   (k-tiles %s)" % (
        (m[0] / n).round(0), (m[1] / n).round(0), (m[2] / n).round(0), n))
    sg = np.array(buf[16:32], dtype=np.float64).reshape(4, 4)
    ns = sg[3]
    print("   stagers per k-tile:      DMA issue %s   split+write / landing wait %s   barrier wait %s" % (
        (sg[1] / ns).round(0), (sg[0] / ns).round(0), (sg[2] / ns).round(0)))
