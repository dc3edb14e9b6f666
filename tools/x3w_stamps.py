"""Diagnostic: where the wave-specialised bf16x3 kernel spends its cycles (library built with -DMMVAE_X3_STAMPS=1)."""
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
cases = {
    "k2 TN dW [1024x20000x512] 256x160": (lambda: ops.gemm(ops.GEMM_TN, r(B, H), r(B, G), splitk=1)),
    "k4a TN dW [20000x1024x512] 160x256": (lambda: ops.gemm(ops.GEMM_TN, r(B, G), r(B, H), splitk=1)),
    "k3 NT [512x20000x1024] 256x160": (lambda: ops.gemm(ops.GEMM_NT, r(B, H), r(G, H), splitk=1)),
    "k1 NT slabs [512x1024x20000] 256x128": (lambda: ops.gemm_slabs(ops.GEMM_NT, r(B, G), r(H, G))),
    "k4b NN slabs [512x1024x20000] 256x128": (lambda: ops.gemm_slabs(ops.GEMM_NN, r(B, G), r(G, H))),
    "k5 NT + reconstruction epilogue [512x20000x1024] 256x160": (
        lambda: ops.decoder_recon(torch.relu(r(B, H)), r(G, H) * 0.03, r(G) * 0.1, torch.relu(r(B, G)), want_xhat=False,
                                  dP=torch.empty(B, G, device=dev),
                                  se_part=torch.empty(ops.recon_tiles(G), B, device=dev))),
    "k5b the same over 5120 rows (K-sample programs: 10 tiles per CU)": (
        lambda: ops.decoder_recon(torch.relu(r(10 * B, H)), r(G, H) * 0.03, r(G) * 0.1, torch.relu(r(B, G)), want_xhat=False,
                                  dP=torch.empty(10 * B, G, device=dev),
                                  se_part=torch.empty(ops.recon_tiles(G), 10 * B, device=dev))),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    assert lib.mmvae_debug_x3_stamps(buf) == 0
    nb = 256
    tr = (ctypes.c_longlong * (4 * 2048))()
    assert lib.mmvae_debug_x3_trace(tr, 2048) == 0
    T = np.array(tr[:], dtype=np.int64).reshape(2048, 4)[:nb]
    T = (T - T[:, 0].min()) / 100.0
    print(f"{name}: kernel span {T[:, 3].max():.1f} us; last item's loop end {np.median(T[:, 2]):.1f} us (median), epilogue+exit {np.median(T[:, 3] - T[:, 2]):.1f} us")
    m = np.array(buf[:16], dtype=np.float64).reshape(4, 4)  # [what][wave]
    n = m[3]
    print("   multipliers per k-tile:  before barrier %s   in barrier %s   behind it %s   (k-tiles %s)" % (
        (m[0] / n).round(0), (m[1] / n).round(0), (m[2] / n).round(0), n))
    sg = np.array(buf[16:32], dtype=np.float64).reshape(4, 4)
    ns = sg[3]
    print("   stagers per k-tile:      split+write %s   offsets+loads %s   barrier wait %s" % (
        (sg[0] / ns).round(0), (sg[1] / ns).round(0), (sg[2] / ns).round(0)))
