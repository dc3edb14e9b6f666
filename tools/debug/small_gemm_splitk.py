"""Split-K sweep of the core layers' 64x64-tile GEMMs (raw slabs, as the engine launches them): us per launch inside a
captured graph of 20 launches (launch gaps excluded the way the step sees them)."""
import sys
import torch
sys.path.insert(0, ".")
from mmvae_amd import ops

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
B = 512
cases = [("fwd NT 1024->512", ops.GEMM_NT, (B, 1024), (512, 1024)),
         ("fwd NT 512->1024", ops.GEMM_NT, (B, 512), (1024, 512)),
         ("fwd NT 512->256", ops.GEMM_NT, (B, 512), (256, 512)),
         ("fwd NT 256->512", ops.GEMM_NT, (B, 256), (512, 256)),
         ("dX NN 1024<-512 (K=1024)", ops.GEMM_NN, (B, 1024), (1024, 512)),
         ("dX NN (K=512) ->1024", ops.GEMM_NN, (B, 512), (512, 1024)),
         ("dX NN (K=256) ->512", ops.GEMM_NN, (B, 256), (256, 512))]
for name, layout, sa, sb in cases:
    a, b = r(*sa), r(*sb)
    M, K = sa
    N = sb[0] if layout == ops.GEMM_NT else sb[1]
    tile, sk0 = ops.gemm_plan(layout, M, N, K)
    line = f"{name:28s} plan tile {tile} splitk {sk0}: "
    for sk in (1, 2, 4, 8, 16):
        if K // sk < 32:
            continue
        for _ in range(2):
            ops.gemm_slabs(layout, a, b, splitk=sk)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                ops.gemm_slabs(layout, a, b, splitk=sk)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        line += f" S={sk}: {e0.elapsed_time(e1) / 20 * 1e3:5.1f}"
    print(line, flush=True)
