#!/usr/bin/env python3
"""Headline benchmark: cells/sec through one MMVAE training step (BASELINE.json metric), on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = zero_grad, forward, ELBO, [adversarial D/G phases], backward, global-norm clip, Adam for the shared VAE and
the active expert, on a synthetic batch that is already resident in HBM.  Modalities alternate round-robin on a fixed,
rank-synchronous schedule.  N > 1: data parallel over cells (weak scaling: per-GPU batch fixed), gradients averaged
with RCCL all-reduce over the flat gradient arenas.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", help="BASELINE config: c1..c5 (c2 = the configuration the metric is quoted on)")
    ap.add_argument("--no-engine", action="store_true", help="module (autograd) path instead of the captured engine")
    ap.add_argument("--mode", default="train", choices=["train", "validate", "predict"],
                    help="train = the headline metric; validate / predict = the forward-only programs (SURVEY 8 f3)")
    ap.add_argument("--input", default="dense", choices=["dense", "csr", "npz"],
                    help="csr: batches arrive as torch.sparse_csr (resident in HBM) and are densified per step (8 f1); "
                         "npz: batches stream from npz-CSR / pkl chunk files on disk through mmvae_amd.data (8 f4: host "
                         "feed and PCIe transfer inside the timed region)")
    ap.add_argument("--genes", default="", help="diagnostics: comma-separated gene counts replacing the config's (one per "
                                                "modality), e.g. 60530,52437 = the reference's human / mouse widths")
    ap.add_argument("--sim-comm", default="", help="diagnostics: CUS,LDS_KB,MICROS -- a stand-in for a collective beside the "
                                                   "step: that many workgroups holding that much LDS each spin on a side "
                                                   "stream for that long, started with every step (DESIGN.md section 7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-baseline sample")
    return ap.parse_args()


# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks (no sparsity)
BF16_DENSE_TFLOPS = 2500.0
F32_MFMA_TFLOPS = 157.3
# HBM bytes per launch of the roofline kernel, from the separate rocprofv3 --pmc passes summarised in
# profiles/r1d_pmc_roofline_kernel.csv (r1b: 48174): 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B) + WRITE_SIZE, in KiB.
# Algorithmic bytes of the launch: 4 (B*1024 + B*G + 1024*G) = 125.0 MB (X is re-read by the 8 row tiles: 2x fetch).
HBM_TRAFFIC_PMC_BYTES = (2 * 47450 + 80000) * 1024


def time_dominant_kernel(cfg, device, iters=10):
    """Roofline leg: the dominant kernel of the step (2 launches per step -- enc-L1 dW and, with the tile transposed,
    dec-L2 dW -- ~23 % of the GPU time in profiles/r1_bench_kernel_stats.csv) -- the weight-gradient GEMM of a G-wide layer, dW[1024, G] = dY^T[1024, B] .
    X[B, G] (TN layout, fp32 in / fp32 out, computed as 6 bf16 MFMAs per product = gemm_x3_kernel<TN>) -- launched on
    the current stream with HIP events around each launch.  Returns (avg seconds per launch, algorithmic FLOPs per
    launch = 2 M N K)."""
    from mmvae_amd import ops

    B, G, H1 = cfg["batch"], max(cfg["experts"].values()), 1024
    g = torch.Generator(device=device).manual_seed(1)
    dY = torch.randn(B, H1, device=device, generator=g)
    X = torch.randn(B, G, device=device, generator=g)
    dW = torch.empty(H1, G, device=device)
    for _ in range(3):
        ops.gemm(ops.GEMM_TN, dY, X, out=dW, splitk=1)
    torch.cuda.synchronize()
    # one event pair around `iters` back-to-back launches on the launch stream: the average is the kernel's duration
    # (an event pair per launch would add the ~10 us host launch latency to every sample).  Kept short: after ~1.7 ms
    # of nothing but this GEMM the chip lowers its clock and the same launch goes from 132 to 170 us
    # (profiles/r1d: kernel trace of the leg) -- a state the training step, which interleaves lighter kernels, is never in
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(ops.GEMM_TN, dY, X, out=dW, splitk=1)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / iters * 1e-3
    return t, 2.0 * H1 * G * B


def cpu_baseline(cfg, seconds):
    """The oracle (CPU restatement of the reference step, checked against the reference's golden vectors) timed on the
    host cores of this box on a bounded sample of the same workload."""
    from oracle import mmvae_oracle as O

    # cores actually available to this process (cgroup / affinity), not the machine's core count
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(avail, 64)))
    B, K = cfg["batch"], cfg["K"]
    spec = O.ModelSpec(
        experts={eid: (O.FCSpec.make([G, 1024, 512], dropout_rate=0.1, use_batch_norm=True, relu=True),
                       O.FCSpec.make([512, 1024, G], relu=True)) for eid, G in cfg["experts"].items()},
        vae_encoder=O.FCSpec.make([512, 256], use_batch_norm=True, relu=True, return_hidden=True),
        vae_decoder=O.FCSpec.make([128, 256, 512], relu=True), latent_dim=128)
    hp = O.HParams()
    sd = O.init_state(spec, seed=0)
    opt_state = {}
    eids = list(cfg["experts"].keys())
    xs = {eid: O.synthetic_counts(B, G, seed=1234 + i) for i, (eid, G) in enumerate(cfg["experts"].items())}
    g = torch.Generator().manual_seed(7)

    def one(i):
        nonlocal sd
        eid = eids[i % len(eids)]
        eps = torch.randn((K, B, 128) if K > 1 else (B, 128), generator=g)
        masks = {f"experts.{eid}.encoder.fc_layers.0.dr": (torch.rand(B, 1024, generator=g) >= 0.1),
                 f"experts.{eid}.encoder.fc_layers.1.dr": (torch.rand(B, 512, generator=g) >= 0.1)}
        _, sd = O.train_step(spec, sd, opt_state, xs[eid], eid, eps, masks, None, 1.0, hp)

    one(0)
    one(1)
    n, t0 = 0, time.perf_counter()
    while True:
        one(n)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 200:
            break
    return {"value": B * n / el, "unit": "cells/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} training steps of the same workload (B={B}, K={K}, modalities alternating) in {el:.1f} s, "
                      f"torch {torch.__version__} CPU fp32"}


def main():
    a = parse()
    from mmvae_amd import dist as mdist, synthetic

    world = mdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    rank = mdist.rank()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MMVAE_SINGLE_DEVICE", "0") != "0":  # rehearsal: every rank on GPU 0 (with MMVAE_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    cfg = dict(synthetic.CONFIGS[a.config])
    if a.genes:
        widths = [int(v) for v in a.genes.split(",")]
        cfg["experts"] = {eid: widths[i % len(widths)] for i, eid in enumerate(cfg["experts"])}
    B, K = cfg["batch"], cfg["K"]

    model = synthetic.build_model(cfg["experts"], adversarial=cfg["adversarial"], n_samples=K,
                                  use_engine=not a.no_engine, seed=0).to(device)
    model.train()
    model.trainer.set_stage("training")
    model.optimizers()
    mdist.broadcast_parameters(model)
    mdist.attach(model)

    eids = list(cfg["experts"].keys())
    n_res = 2  # resident batches per modality
    data = {}
    for i, (eid, G) in enumerate(cfg["experts"].items()):
        data[eid] = [(synthetic.synthetic_counts(B, G, seed=1234 + 97 * i + 13 * j + 1000 * rank, device=device),
                      synthetic.synthetic_metadata(B, seed=5 + j + 1000 * rank)) for j in range(n_res)]

    if a.input == "csr":
        data = {eid: [(x.to_sparse_csr(), m) for x, m in v] for eid, v in data.items()}
    feed = None
    if a.input == "npz":  # synthetic chunks on disk -> the reference's on-disk format -> CSR batches staged to the GPU
        import tempfile

        import pandas as pd
        import scipy.sparse as sp

        from mmvae_amd import data as mdata
        from mmvae_amd.trainer import MultiModalBatches

        tmp = tempfile.mkdtemp(prefix="mmvae_bench_")
        feeds = {}
        for i, (eid, G) in enumerate(cfg["experts"].items()):
            rows = torch.cat([synthetic.synthetic_counts(B, G, seed=77 + 31 * i + j, device="cpu") for j in range(8)])
            meta = pd.concat([synthetic.synthetic_metadata(B, seed=9 + j) for j in range(8)], ignore_index=True)
            mdata.write_chunks(os.path.join(tmp, eid), eid, sp.csr_matrix(rows.numpy()), meta, chunk_rows=4 * B,
                               compressed=False)  # stored members: the feed memory-maps them
            feeds[eid] = mdata.SpeciesChunks(os.path.join(tmp, eid), f"{eid}_train_counts_*.npz",
                                             f"{eid}_train_metadata_*.pkl", B, eid, seed=i, device=device,
                                             rank=0, world=1)

        def endless():
            while True:
                yield from MultiModalBatches(feeds, seed=0, round_robin=True)

        # the whole feed (gather, metadata slices, H2D copies on a stream of its own) runs ahead of the step loop
        feed = iter(mdata.Prefetcher(endless(), depth=3, device=device))
    if a.mode != "train":
        model.eval()
        model.trainer.set_stage("validation" if a.mode == "validate" else "predict")

    sim = None
    if a.sim_comm:
        from mmvae_amd import _lib as _l

        cus, lds_kb, micros = (int(v) for v in a.sim_comm.split(","))
        sim = (_l.load(), torch.cuda.Stream(device=device), cus, lds_kb * 1024, micros)

    def step(i):
        if sim is not None:
            lib_, side_, cus_, lds_, us_ = sim
            side_.wait_stream(torch.cuda.current_stream())
            _l.check(lib_.mmvae_debug_occupy(cus_, lds_, us_, None, side_.cuda_stream), "mmvae_debug_occupy")
        if feed is not None:
            x, meta, eid = next(feed)
        else:
            eid = eids[i % len(eids)]  # rank-synchronous round-robin schedule
            x, meta = data[eid][(i // len(eids)) % n_res]
        if a.mode == "train":
            model.training_step((x, meta, eid), i)
        elif a.mode == "validate":
            model.validation_step((x, meta, eid))
        else:
            model.predict_step((x, meta, eid), i)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # Engine set-up (untimed, before the W warm-up steps): a step plan is built on its first run and captured into a
    # hipGraph on its second; a resident batch is recognised by its pointer on its second sight.  Step every resident
    # batch until its plan replays, so that neither the warm-up nor the timed steps contain plan builds.
    period = len(eids) * n_res
    n_setup = 4 * period
    for i in range(n_setup):
        step(i)
    sync()
    for i in range(a.warmup):
        step(n_setup + i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(n_setup + a.warmup + i)
    sync()
    el = time.perf_counter() - t0
    if feed is not None:
        feed.close()  # stops the feed's background threads before the interpreter shuts down
    if world > 1:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t)
    loss = {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in model.logged.items()
            if k.startswith(("loss/", "recon_loss/", "kl_loss/"))}

    if rank == 0:
        G = max(cfg["experts"].values())
        cells_per_s = B * world * a.steps / el
        out = {
            "metric": {"train": "cells/sec per MMVAE train step", "validate": "cells/sec per MMVAE validation step",
                       "predict": "cells/sec per MMVAE predict step (latent embeddings)"}[a.mode], "value": cells_per_s, "unit": "cells/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if a.input != "npz" else "synthetic, streamed from npz-CSR / pkl chunk files",
            "config": {"workload": f"{a.config}: {len(eids)}-modality MMVAE train step, {G} genes each, latent 128, "
                                   f"K={K}, batch {B}/GPU, adversarial={cfg['adversarial']}"
                                   + (", CSR input densified per step" if a.input == "csr" else "")
                                   + (", npz-CSR chunks streamed from disk" if a.input == "npz" else ""),
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "path": "module" if (a.no_engine or not model._engine) else "engine(hipGraph)"},
            "step_flops_per_cell": synthetic.flops_per_cell(G, K),
            "step_tflops": synthetic.flops_per_cell(G, K) * cells_per_s / world / 1e12,
            "last_losses": loss, "setup_steps": n_setup,
        }
        from mmvae_amd import _lib

        tk, fl = time_dominant_kernel(cfg, device)
        x3 = _lib.load().mmvae_gemm_get_precision() == _lib.GEMM_PRECISION_BF16X3
        # bf16x3: every fp32 product costs 6 bf16 MFMA products, so the matrix-core ceiling for ALGORITHMIC fp32
        # flops is the dense bf16 peak / 6; the exact-f32 mode is bounded by the fp32 MFMA peak.
        peak = BF16_DENSE_TFLOPS / 6.0 if x3 else F32_MFMA_TFLOPS
        out["roofline"] = {"bound": "mfma",
                           "kernel": ("gemm_x3_kernel<TN,128x160> (bf16x3 MFMA)" if x3 else "gemm_f32_kernel<TN> (f32 MFMA)")
                                     + ": dW of a G-wide layer",
                           "achieved": fl / tk / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl / tk / (peak * 1e12),
                           "traffic": HBM_TRAFFIC_PMC_BYTES, "us_per_launch": tk * 1e6, "flops_per_launch": fl,
                           "peak_note": "dense bf16 MFMA peak 2500 / 6 MFMA products per fp32 product" if x3
                                        else "dense fp32 MFMA peak",
                           "frac_of_f32_mfma_peak": fl / tk / (F32_MFMA_TFLOPS * 1e12)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_seconds)
        print(json.dumps(out))
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
