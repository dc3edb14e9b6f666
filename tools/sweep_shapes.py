#!/usr/bin/env python3
"""Step time of the captured engine over a grid of batch sizes, gene counts and sample counts K (2 modalities): looks for
shape cliffs -- sizes at which a kernel falls off its fast path.  Prints ms / step and fp32-equivalent TFLOP/s.
usage: sweep_shapes.py [B,B,...] [G,G,...] [K,K,...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import pandas as pd  # noqa: E402
import torch  # noqa: E402

from mmvae_amd import synthetic  # noqa: E402


def run(B, G, K=1, steps=12, warm=6):
    model = synthetic.build_model({"human": G, "mouse": G}, n_samples=K, seed=0).cuda()
    model.train()
    model.trainer.set_stage("training")
    xs = {e: synthetic.synthetic_counts(B, G, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
    md = pd.DataFrame({"dummy": [0] * B})
    for i in range(warm + steps):
        if i == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        eid = ("human", "mouse")[i % 2]
        model.training_step((xs[eid], md, eid), i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    tf = synthetic.flops_per_cell(G, K) * B / ms / 1e9
    del model, xs
    torch.cuda.empty_cache()
    return ms, tf


if __name__ == "__main__":
    Bs = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "256,500,512,1000,1024,2048".split(","))]
    Gs = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "5000,20000,33333,60530".split(","))]
    Ks = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1"])]
    for K in Ks:
        print(f"K = {K}   B \\ G  " + "".join(f"{g:>22d}" for g in Gs))
        for B in Bs:
            cells = []
            for G in Gs:
                ms, tf = run(B, G, K)
                cells.append(f"{ms:8.3f} ms {tf:6.1f} TF")
            print(f"{B:6d}  " + "".join(f"{c:>22s}" for c in cells), flush=True)
