#!/usr/bin/env python3
"""Headline benchmark: cells/sec through one MMVAE training step (BASELINE.json metric), on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = zero_grad, forward, ELBO, [adversarial D/G phases], backward, global-norm clip, Adam for the shared VAE and
the active expert, on a synthetic batch that is already resident in HBM.  Modalities alternate round-robin on a fixed,
rank-synchronous schedule.  N > 1: data parallel over cells (weak scaling: per-GPU batch fixed), gradients averaged
with RCCL all-reduce over the flat gradient arenas.  Prints ONE JSON line (rank 0).

Launched plainly with --gpus N > 1 (no torchrun environment) the script starts the N rank processes itself -- fresh
children, before this process has touched a GPU -- passes rank 0's line through and exits non-zero if any rank fails.
The model is instantiated from configs/model/<config>.yaml (the reference's LightningCLI schema).
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
    ap.add_argument("--hidden", type=int, default=0, help="diagnostics: width of the experts' hidden layer next to the "
                    "genes (1024 in every BASELINE config; e.g. 1000 = not a multiple of the 32-wide k-tile)")
    ap.add_argument("--sim-comm", default="", help="diagnostics: CUS,LDS_KB,MICROS -- a stand-in for a collective beside the "
                                                   "step: that many workgroups holding that much LDS each spin on a side "
                                                   "stream for that long, started with every step (DESIGN.md section 7)")
    ap.add_argument("--sim-world", type=int, default=0, help="diagnostics (timing only): with --gpus 1 and "
                    "MMVAE_SINGLE_RANK_COLLECTIVES=1, give the sharded expert update the slice of a world of N ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-lookahead", action="store_true", help="do not tell the model which batch comes next (the step "
                    "engine then computes every step's first forward product inside that step: no pipelining across steps)")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity block (one step of the timed program "
                                                             "against the oracle, ~2 s, outside the timed region)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget of the CPU-baseline sample (all thread counts)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                    help="cpu: rehearsal of the launch / exchange / timing logic on the gloo backend with CPU plumbing "
                         "(module path, plain torch ops; tests/test_bench_entry.py) -- never a measurement")
    return ap.parse_args()


CONFIG_FILES = {"c1": "c1_core_vae.yaml", "c2": "c2_two_modality_20k.yaml", "c3": "c3_two_modality_20k_k10.yaml",
                "c4": "c4_two_modality_20k_adversarial.yaml", "c5": "c5_three_modality_30k_k5.yaml"}


def spawn_ranks(argv, n: int) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as fresh child processes (this
    process has not touched a GPU), one per GPU, rendezvous on 127.0.0.1.  Rank 0 prints the JSON line on the inherited
    stdout.  Returns the exit code: non-zero if any rank failed (the others are then terminated)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if "cpu" in [a_ for i_, a_ in enumerate(argv) if i_ > 0 and argv[i_ - 1] == "--device"]:
            # a CPU rehearsal never opens a GPU, also on a box that has one (such boxes bound the number of processes per
            # GPU: eight rehearsal ranks that merely asked torch whether a device exists were killed by the pool's guard)
            env.update(HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    code = 0
    try:
        while procs:
            for p in list(procs):
                rc = p.poll()
                if rc is None:
                    continue
                procs.remove(p)
                if rc != 0:
                    code = code or rc
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    return code


# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks (no sparsity), HBM3E peak
BF16_DENSE_TFLOPS = 2500.0
F32_MFMA_TFLOPS = 157.3
HBM_PEAK_TBS = 8.0
# The dominant kernel family of the step: its five G-wide GEMMs (2 B G 1024 FLOP each, ~half of the critical path), by
# the engine's probe tag; the substring identifies the kernel's row in profiles/*_pmc_family.csv (rocprofv3 --pmc passes
# of tools/roofline_kernel.py, one counter group per run, summarised by tools/pmc_summary.py).
FAMILY = [
    ("enc_l1_fwd", "forward GEMM of the G-wide expert encoder layer (NT, split-K 16 raw slabs)", "gemm_x3w_kernel<0, 0, 256, 128"),
    ("dec_l2_recon", "expert decoder's last layer + reconstruction loss epilogue (NT)", "gemm_x3w_kernel<0, 0, 256, 160, 4, 1, 1"),
    ("dec_l2_dx", "input gradient of the decoder's last layer (NN, split-K 16 raw slabs)", "gemm_x3w_kernel<0, 1, 256, 128"),
    ("dec_l2_dw", "weight gradient of the decoder's last layer (TN)", "gemm_x3w_kernel<1, 1, 160, 256"),
    ("enc_l1_dw", "weight gradient of the encoder's first layer (TN)", "gemm_x3w_kernel<1, 1, 256, 160"),
]
def _latest_pmc_family():
    """profiles/r<N>_pmc_family.csv of the latest round (tools/collect_profiles.sh writes it ahead of the bench line)."""
    import glob
    import re

    found = [(int(re.search(r"r(\d+)_pmc_family", f).group(1)), f)
             for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_family.csv")) if re.search(r"r(\d+)_pmc_family", f)]
    return max(found)[1] if found else os.path.join(ROOT, "profiles", "r3_pmc_family.csv")


PMC_FAMILY_FILE = _latest_pmc_family()


def pmc_traffic():
    """HBM bytes per launch of the family's kernels from the tracked PMC summary: 2 x FETCH_SIZE (gfx950 tallies 128-B
    read requests at 64 B) + WRITE_SIZE, KiB -> bytes.  {kernel substring: bytes} ({} when the file is absent)."""
    import csv

    if not os.path.exists(PMC_FAMILY_FILE):
        return {}
    per = {}
    with open(PMC_FAMILY_FILE) as f:
        for row in csv.DictReader(f):
            for _, _, sub in FAMILY:
                if sub in row["kernel"] and row["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
                    per.setdefault(sub, {})[row["counter"]] = float(row["value_per_launch"])
    return {sub: int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) for sub, v in per.items() if len(v) == 2}


def time_step_kernels(model, step, first, steps=8):
    """Roofline leg: every G-wide GEMM of the step and the expert's Adam pass, measured INSIDE the step: `steps` more
    training steps are run eagerly (same launches in the same order, not replayed from the graph) with a HIP event pair
    around each probed launch on the stream it is launched on (engine probe hook, _Plan._probed; the pair includes one
    event marker, 2-8 us on a busy stream -- reported as measured, so a figure never over-states its kernel).  A GEMM
    whose grid the engine caps to share the chip with a branch (`cus`) is timed as it runs there.  Launched alone the
    same kernels are slower (cold clocks and caches, tools/debug/leg_probe.py).  Returns {tag: (median seconds,
    work per launch, meta)}."""
    eng = model._engine
    eng.eager_only = True
    probe = {}
    try:
        for p in eng._plans.values():
            p.probe = probe
        for i in range(steps):
            step(first + i)
        torch.cuda.synchronize()
    finally:
        for p in eng._plans.values():
            p.probe = None
        eng.eager_only = False
    meta = {}
    for p in eng._plans.values():
        meta.update(p.probe_meta)
    out = {}
    for tag, pairs in probe.items():
        pairs = pairs[2:]  # (event, event, work, event) per launch; the first two steps warm up
        if not pairs:
            continue
        ts = sorted(e0.elapsed_time(e1) for e0, e1, _, e2 in pairs)
        out[tag] = (ts[len(ts) // 2] * 1e-3, pairs[0][2], meta.get(tag, {}))
    if "enc_l1_fwd" not in out:
        raise RuntimeError("roofline leg: the engine did not run the probed GEMMs")
    return out


def parity_block(model, step_args, next_batch=None, settled=False):
    """One more step of the timed program (replayed graph, Philox noise, branches) against the oracle -- the checker,
    outside the timed region (oracle/program_check.py; tests/test_bench_program_gpu.py holds 12 such steps to the same
    tolerances)."""
    from oracle import program_check as PC

    x, meta, eid, i = step_args
    t0 = time.perf_counter()
    r = PC.check_step(model, eid, x, meta, i, strict=False, next_batch=next_batch)
    tol = dict(PC.TOL)
    if settled:
        # the adversarial configuration is checked 2 000 steps in, near a stationary point of the reconstruction term: its
        # gradients are sums over the cells that nearly cancel, and two fp32 evaluations in different summation orders
        # differ by the GEMMs' 2e-6 times that cancellation (tests/test_bench_program_gpu.py: 4.5e-5 .. 1.05e-4 measured)
        tol.update(grad=3e-4, grad_norm=1e-4, note="settled regime: gradient bounds 3x / 2x those of the untrained C2 program")
    adv_loss = [v for k, v in r.items() if k.startswith("adversarial_loss_")]
    adv_norm = [v for k, v in r.items() if k.startswith(("grad_norm_discriminator", "grad_norm_generator"))]
    ok = (max([r["loss"], r["recon_loss"], r["kl_loss"]] + adv_loss) <= tol["loss"]
          and max([r["grad_norm_vae"], r["grad_norm_expert"]] + adv_norm) <= tol["grad_norm"] and r["grad"] <= tol["grad"]
          and r["param"] <= (tol["param_cold"] if r["cold"] else tol["param"]))
    return {"checked": "one step of the timed program (replayed hipGraph, device Philox noise, side branches) against "
                       "oracle.train_step from the snapshotted pre-step state, at the noise and ReLU slopes the step took",
            "pass": bool(ok), "tolerance": tol,
            "rel_err": {k: v for k, v in r.items() if isinstance(v, float)},
            "worst_tensors": {"grad": r.get("worst_grad"), "param": r.get("worst_param")},
            "relu_kinks": r["kinks"], "replayed_graph": r["replayed"], "forked_branches": r["forked"],
            "first_product_from_previous_step": r["first_product_from_previous_step"],
            "computes_next_first_product": r["computes_next_first_product"],
            "seconds": round(time.perf_counter() - t0, 2)}


def cpu_baseline(cfg, seconds):
    """The oracle (CPU restatement of the reference step, pinned against the reference's golden vectors) timed on the
    host cores of this box on a bounded sample of the same workload: parameters and optimiser state updated in place
    (oracle.InPlaceStepper: leaf tensors, torch.optim.Adam, backward(), clip_grad_norm_ -- a trainer's bookkeeping, no
    per-step clone of the 84 M parameters), thread counts {8, 16, 32, min(all available, 64)} swept, the best one reported."""
    from oracle import mmvae_oracle as O

    try:  # cores actually available to this process (cgroup / affinity), not the machine's core count
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    B, K = cfg["batch"], cfg["K"]
    spec = O.ModelSpec(
        experts={eid: (O.FCSpec.make([G, 1024, 512], dropout_rate=0.1, use_batch_norm=True, relu=True),
                       O.FCSpec.make([512, 1024, G], relu=True)) for eid, G in cfg["experts"].items()},
        vae_encoder=O.FCSpec.make([512, 256], use_batch_norm=True, relu=True, return_hidden=True),
        vae_decoder=O.FCSpec.make([128, 256, 512], relu=True), latent_dim=128)
    stepper = O.InPlaceStepper(spec, O.init_state(spec, seed=0), O.HParams())
    eids = list(cfg["experts"].keys())
    xs = {eid: O.synthetic_counts(B, G, seed=1234 + i) for i, (eid, G) in enumerate(cfg["experts"].items())}
    g = torch.Generator().manual_seed(7)

    def one(i):
        eid = eids[i % len(eids)]
        eps = torch.randn((K, B, 128) if K > 1 else (B, 128), generator=g)
        masks = {f"experts.{eid}.encoder.fc_layers.0.dr": (torch.rand(B, 1024, generator=g) >= 0.1),
                 f"experts.{eid}.encoder.fc_layers.1.dr": (torch.rand(B, 512, generator=g) >= 0.1)}
        stepper.step(xs[eid], eid, eps, masks, None, 1.0)

    # (torch's CPU GEMMs stop scaling long before a GPU host's 100-200 hardware threads: 256 threads measured 15 cells/s
    # against 1 986 at 16 -- the sweep stops at 64)
    counts = sorted({t for t in (8, 16, 32, min(avail, 64)) if t <= avail} or {avail})
    sweep = {}
    for threads in counts:
        torch.set_num_threads(threads)
        for i in range(len(eids)):
            one(i)  # untimed: first touch of this thread count
        n, t0 = 0, time.perf_counter()
        while True:
            one(n)
            n += 1
            el = time.perf_counter() - t0
            if (el >= seconds / len(counts) and n >= len(eids)) or n >= 200:
                break
        sweep[threads] = (B * n / el, n, el)
    best = max(sweep, key=lambda t: sweep[t][0])
    rate, n, el = sweep[best]
    return {"value": rate, "unit": "cells/s", "cores": best, "kind": "port",
            "sample": f"{n} training steps of the same workload (B={B}, K={K}, modalities alternating) in {el:.1f} s with "
                      f"{best} threads of {avail} available, parameters / Adam state in place, torch {torch.__version__} "
                      f"CPU fp32; sweep " + ", ".join(f"{t} threads: {sweep[t][0]:.0f} cells/s" for t in counts)}


def build_model(a, cfg, device):
    """The benchmark's model: configs/model/<config>.yaml through the class_path / init_args instantiator (the drop-in
    surface), or -- `--genes` diagnostics -- the same architecture from mmvae_amd.synthetic with other gene counts."""
    from mmvae_amd import instantiate, synthetic

    torch.manual_seed(0)
    if a.genes or getattr(a, "hidden", 0):
        return synthetic.build_model(cfg["experts"], adversarial=cfg["adversarial"], n_samples=cfg["K"],
                                     use_engine=not a.no_engine, seed=0, **({"h1": a.hidden} if getattr(a, "hidden", 0) else {}))
    if cfg["adversarial"]:  # unique_expression_<condition>.csv files with the reference's class counts
        import tempfile

        from mmvae_amd.modules import base

        base.Adversarial.labels.clear()
        os.environ["MMVAE_LABELS_DIR"] = synthetic.write_label_dir(tempfile.mkdtemp(prefix="mmvae_labels_"))
    model = instantiate.load_yaml(os.path.join(ROOT, "configs", "model", CONFIG_FILES[a.config]))
    model.use_engine = not a.no_engine
    got = {e.id: e.encoder.config.layers[0] for e in model.module.experts.values()}
    assert got == cfg["experts"] and model.module.vae.encoder.n_samples == cfg["K"], (got, cfg)
    return model


def main():
    a = parse()
    if a.sim_world:
        os.environ["MMVAE_DP_SIM_WORLD"] = str(a.sim_world)
    if a.gpus > 1 and "RANK" not in os.environ:  # plain launch: this process becomes the launcher (no GPU call so far)
        sys.exit(spawn_ranks(sys.argv[1:], a.gpus))
    from mmvae_amd import backend, dist as mdist, synthetic

    # The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner when its
    # communicator comes up): file descriptor 1 points at stderr until the line is written.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    on_gpu = a.device == "cuda"
    world = mdist.init_from_env(None if on_gpu else "gloo")
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    rank = mdist.rank()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MMVAE_SINGLE_DEVICE", "0") != "0":  # rehearsal: every rank on GPU 0 (with MMVAE_DIST_BACKEND=gloo)
        local = 0
    if on_gpu:
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:  # rehearsal of the launch / exchange / timing logic: CPU plumbing, module path
        device = torch.device("cpu")
        backend.set_cpu_plumbing(True)
        a.no_engine = True
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(world, 1)))
    cfg = dict(synthetic.CONFIGS[a.config])
    if a.genes:
        widths = [int(v) for v in a.genes.split(",")]
        cfg["experts"] = {eid: widths[i % len(widths)] for i, eid in enumerate(cfg["experts"])}
    B, K = cfg["batch"], cfg["K"]

    model = build_model(a, cfg, device).to(device)
    model.train()
    model.trainer.set_stage("training")
    model.optimizers()
    mdist.broadcast_parameters(model)
    mdist.attach(model)

    eids = list(cfg["experts"].keys())
    n_res = 2  # resident batches per modality
    data = {}
    for i, (eid, G) in enumerate(cfg["experts"].items()):
        if cfg["adversarial"]:  # labels (and counts) are functions of the cell: the discriminators have signal, as on real data
            data[eid] = [synthetic.synthetic_labelled_batch(B, G, seed=1234 + 97 * i + 13 * j + 1000 * rank, device=device)
                         for j in range(n_res)]
        else:
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

    def batch_of(i):
        eid = eids[i % len(eids)]  # rank-synchronous round-robin schedule
        x, meta = data[eid][(i // len(eids)) % n_res]
        return x, meta, eid

    pending = []

    def step(i):
        if sim is not None:
            lib_, side_, cus_, lds_, us_ = sim
            side_.wait_stream(torch.cuda.current_stream())
            _l.check(lib_.mmvae_debug_occupy(cus_, lds_, us_, None, side_.cuda_stream), "mmvae_debug_occupy")
        if feed is not None:  # streamed batches: one batch of look-ahead, like mmvae_amd.trainer.Lookahead
            if not pending:
                pending.append(next(feed))
            x, meta, eid = pending.pop()
            pending.append(next(feed))
            if a.mode == "train" and not a.no_lookahead:
                model.hint_next_batch(pending[0])
        else:
            x, meta, eid = batch_of(i)
        if a.mode == "train":
            if feed is None and not a.no_lookahead:
                # the loop knows its next batch (as mmvae_amd.trainer's does, one batch ahead): the engine may compute
                # that step's first forward product beside this step's forward chain.  Every timed step still runs
                # exactly one such product -- the next step's instead of its own.
                model.hint_next_batch(batch_of(i + 1))
            model.training_step((x, meta, eid), i)
        elif a.mode == "validate":
            model.validation_step((x, meta, eid))
        else:
            model.predict_step((x, meta, eid), i)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    # Engine set-up (untimed, before the W warm-up steps): a step plan is built on its first run and captured into a
    # hipGraph on its second; a resident batch is recognised by its pointer on its second sight.  Step every resident
    # batch until its plan replays, so that neither the warm-up nor the timed steps contain plan builds.
    period = len(eids) * n_res
    n_setup = 4 * period if on_gpu else 0
    if cfg["adversarial"] and on_gpu and a.mode == "train":
        # the adversarial game's first ~1 500 steps are violent (gradient reversal at adv_weight 25 against cold Adam
        # steps: KL of 1e6 .. 1e8, spikes of 1e12 in the loss; profiles/r5_c4_stability.txt) before it settles: the warm-up,
        # the timed steps, the roofline leg and the parity-checked step all run in the settled regime (2 s, untimed)
        n_setup = max(n_setup, 2000 // period * period)
    for i in range(n_setup):
        step(i)
    # data parallelism: the engine times its two GEMM kernel families on its first replayed steps and keeps the faster
    # (engine._dp_autotune); those steps belong to the set-up
    eng = getattr(model, "_engine", None) if a.mode == "train" else None
    while eng is not None and getattr(eng, "_tune", None) is not None and n_setup < 400:
        for _ in range(period):
            step(n_setup)
            n_setup += 1
    # ... and until a whole period of steps has REPLAYED its program (a plan is built on its first run, captured on its
    # second; with the look-ahead a resident batch is staged on its first two announcements and read in place from the third,
    # which changes the program's key once more): no plan build may fall into the warm-up or the timed steps
    eng_any = getattr(model, "_engine", None)
    replayed = 0
    while eng_any and replayed < period and n_setup < 400:
        step(n_setup)
        n_setup += 1
        plan = getattr(eng_any, "last_plan", None) if a.mode == "train" else None
        replayed = replayed + 1 if (plan is None or (plan._graphs is not None and plan._runs >= 3)) else 0
    sync()
    for i in range(a.warmup):
        step(n_setup + i)
    sync()
    # one event behind every timed step (recorded on the launch stream, read after the run): per-step durations for the
    # median beside the mean -- the events cost ~1 us each and no synchronisation
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if on_gpu else []
    prof = None
    if os.environ.get("MMVAE_HOST_PROFILE") == "1":  # diagnostics: where the host spends a step (not a measurement run)
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    if marks:
        marks[0].record()
    for i in range(a.steps):
        step(n_setup + a.warmup + i)
        if marks:
            marks[i + 1].record()
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(28)
    if a.mode == "train" and hasattr(model, "_flush_engine"):
        model._flush_engine()  # the last step's deferred expert update belongs to the timed work
    host_el = time.perf_counter() - t0  # the host's share: enqueueing the K steps (a step time close to it is host-bound)
    sync()
    el = time.perf_counter() - t0
    if os.environ.get("MMVAE_STAMPS") == "1" and getattr(model, "_engine", None):
        # diagnostics: milestones of the LAST replayed step, from marker launches inside the captured program (no tracer
        # attached; the markers cost ~2 us each: this run's ms/step is not a result)
        plan = model._engine.last_plan
        st = model._engine.buf("debug.stamps", (256,), torch.int64).cpu().tolist()[:len(plan.stamp_names)]
        t_first = min(st)
        for t, name in sorted(zip(st, plan.stamp_names)):
            print(f"{(t - t_first) / 100.0:9.1f} us  {name}", file=sys.stderr)
    if feed is not None:
        feed.close()  # stops the feed's background threads before the interpreter shuts down
    if world > 1:
        t = torch.tensor([el], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t)
    leg = None
    if on_gpu and a.mode == "train" and feed is None and getattr(model, "_engine", None) and not a.no_engine:
        leg = time_step_kernels(model, step, n_setup + a.warmup + a.steps)  # every rank: the steps exchange gradients
    # N > 1: the same loop once more with the gradient all-reduces skipped (mmvae_amd.dist.DRY_RUN) -- what the step
    # costs without the transfers; `ms_per_step` minus this is the exchange the overlapped program did NOT hide
    dp_diag = None
    if world > 1 and a.mode == "train" and feed is None:
        n_dry = max(min(a.steps, 20), 2 * len(eids))
        base = n_setup + a.warmup + a.steps + 16
        mdist.DRY_RUN = True
        try:
            for i in range(2 * len(eids)):
                step(base + i)
            sync()
            t1 = time.perf_counter()
            for i in range(n_dry):
                step(base + 2 * len(eids) + i)
            sync()
            dry = time.perf_counter() - t1
        finally:
            mdist.DRY_RUN = False
        t = torch.tensor([dry], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dp_diag = {"ms_per_step_without_transfers": float(t) / n_dry * 1e3, "steps": n_dry,
                   "note": "the same data-parallel program with the all-reduce calls skipped (max over ranks), run behind "
                           "the timed region; ms_per_step minus this = exchange time the program did not hide"}
    if getattr(model, "_engine", None):
        model._flush_engine()
    loss = {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in model.logged.items()
            if k.startswith(("loss/", "recon_loss/", "kl_loss/"))}
    import math

    finite = {"last_losses": all(math.isfinite(v) for v in loss.values())}
    if on_gpu:
        opts = list(model.optimizers())
        finite["parameters"] = all(bool(torch.isfinite(o.arena.data).all()) for o in opts)
        finite["adam_moments"] = all(bool(torch.isfinite(o.arena.exp_avg).all()) and bool(torch.isfinite(o.arena.exp_avg_sq).all())
                                     for o in opts)
    parity = None
    if (leg is not None and world == 1 and rank == 0 and a.config in ("c2", "c4") and a.input == "dense" and not a.genes
            and not a.hidden and not a.no_parity and not a.sim_world):
        i_par = n_setup + a.warmup + a.steps + 8
        x_par, m_par, eid_par = batch_of(i_par)
        parity = parity_block(model, (x_par, m_par, eid_par, i_par), None if a.no_lookahead else batch_of(i_par + 1),
                              settled=bool(cfg["adversarial"]))

    if rank == 0:
        G = max(cfg["experts"].values())
        cells_per_s = B * world * a.steps / el
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)) if marks else []
        backend_name = torch.distributed.get_backend() if world > 1 or mdist.collectives_active() else "none"
        one_expert = len(eids) == 1
        out = {
            "metric": {"train": "cells/sec per MMVAE train step", "validate": "cells/sec per MMVAE validation step",
                       "predict": "cells/sec per MMVAE predict step (latent embeddings)"}[a.mode], "value": cells_per_s, "unit": "cells/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
            "ms_per_step_median": per_step[len(per_step) // 2] if per_step else None,
            "host_ms_per_step": host_el / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic" if a.input != "npz" else "synthetic, streamed from npz-CSR / pkl chunk files",
            "config": {"workload": f"{a.config}: {len(eids)}-modality MMVAE train step, {G} genes each, latent 128, "
                                   f"K={K}, batch {B}/GPU, adversarial={cfg['adversarial']}"
                                   + (", CSR input densified per step" if a.input == "csr" else "")
                                   + (", npz-CSR chunks streamed from disk" if a.input == "npz" else ""),
                       "model_file": None if a.genes else f"configs/model/{CONFIG_FILES[a.config]}",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "exchange": ("none" if world == 1 and not mdist.collectives_active() else
                                    "in-order all-reduce (one expert: nothing to overlap the expert's exchange with)"
                                    if one_expert else "expert all-reduce + update overlapped with the next modality's step"),
                       "path": "module" if (a.no_engine or not model._engine) else "engine(hipGraph)"},
            "rccl_ranks": torch.distributed.get_world_size() if backend_name == "nccl" else 0,
            "dist_backend": backend_name,
            # (algorithmic FLOPs of the program that was timed: the forward-only modes do a third / a sixth of a training step)
            "step_flops_per_cell": synthetic.flops_per_cell(G, K, mode=a.mode, **({"h1": a.hidden} if a.hidden else {})),
            "step_tflops": synthetic.flops_per_cell(G, K, mode=a.mode, **({"h1": a.hidden} if a.hidden else {})) * cells_per_s / world / 1e12,
            "pipelined_first_product": (dict(eng.prefetch_stats,
                                             cap=eng.settings.prefetch_adv if cfg["adversarial"] else eng.settings.prefetch,
                                             note="software pipelining across steps: every step computes the NEXT step's first "
                                                  "forward GEMM (the loop's look-ahead) beside its own forward chain and starts "
                                                  "from the slabs the previous step left; one such GEMM per timed step")
                                        if (eng and eng.prefetch_stats["issued"]) else None),
            "last_losses": loss, "setup_steps": n_setup,
            # every number above was measured on finite arithmetic: the last logged losses, the parameters and both Adam
            # moments of every optimiser after the last step (a diverged run times degenerate operands: VERDICT r4)
            "finite": finite,
        }
        if not all(finite.values()):
            out["invalid"] = "non-finite values after the timed steps: " + ", ".join(k for k, v in finite.items() if not v)
            print("bench.py: " + out["invalid"], file=sys.stderr)
        if leg is not None:
            from mmvae_amd import _lib

            x3 = _lib.load().mmvae_gemm_get_precision() == _lib.GEMM_PRECISION_BF16X3
            # bf16x3: every fp32 product costs 6 bf16 MFMA products, so the matrix-core ceiling for ALGORITHMIC fp32
            # flops is the dense bf16 peak / 6; the exact-f32 mode is bounded by the fp32 MFMA peak.
            peak = BF16_DENSE_TFLOPS / 6.0 if x3 else F32_MFMA_TFLOPS
            # (the tracked counters are C2's: default gene counts and hidden width, dense input)
            profiled_shape = a.config == "c2" and not a.genes and not a.hidden and a.input == "dense"
            traffic = pmc_traffic() if profiled_shape else {}
            kernels, fl_sum, t_sum, tr_sum, tr_n, t_held = [], 0.0, 0.0, 0, 0, 0.0
            for tag, what, sub in FAMILY:
                if tag not in leg:
                    continue
                tk, fl, meta = leg[tag]
                fl_sum += fl
                t_sum += tk
                t_held += tk * (meta.get("cus") or 256) / 256.0
                tr = traffic.get(sub)
                if tr is not None:
                    tr_sum += tr
                    tr_n += 1
                kernels.append({"name": tag, "what": what, "shape": meta.get("shape"), "us": tk * 1e6,
                                "tflops": fl / tk / 1e12, "frac": fl / tk / (peak * 1e12),
                                # a launch whose grid the engine caps runs BESIDE a latency-bound lane of the program on
                                # `cus` of the 256 CUs: its rate against the ceiling of the CUs it held
                                "frac_of_held_cus": fl / tk / (peak * 1e12) * 256.0 / (meta.get("cus") or 256),
                                "cus": meta.get("cus") or 256, "operands": {"": "fp32, split in the kernel", "A+B": "pre-split bf16 planes (LDS-DMA)",
                                             "A": "A pre-split bf16 planes (LDS-DMA), B fp32 split in the kernel",
                                             "B": "A fp32 split in the kernel, B pre-split bf16 planes (LDS-DMA)"
                                             }.get(meta.get("planes") or "", "fp32, split in the kernel"),
                                "traffic": tr})
            if "adam_expert" in leg:
                tk, by, meta = leg["adam_expert"]
                kernels.append({"name": "adam_expert", "what": "fused clip + Adam over the active expert's flat arenas",
                                "shape": meta.get("shape"), "us": tk * 1e6, "bound": "hbm", "tb_per_s": by / tk / 1e12,
                                "frac": by / tk / (HBM_PEAK_TBS * 1e12), "cus": 256})
            step_tf = synthetic.flops_per_cell(G, K, **({"h1": a.hidden} if a.hidden else {})) * cells_per_s / world / 1e12
            out["roofline"] = {
                "bound": "mfma",
                "kernel": ("the five G-wide GEMMs of the step, time-weighted (bf16x3 MFMA: 6 bf16 products per fp32 product)"
                           if x3 else "the five G-wide GEMMs of the step, time-weighted (f32 MFMA)"),
                "achieved": fl_sum / t_sum / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl_sum / t_sum / (peak * 1e12),
                # three of the five launches run capped beside a latency-bound lane (the next step's first product beside the
                # forward chain, the two weight gradients beside the backward chain / the VAE's optimiser), and the
                # reconstruction launch starts while the first still holds CUs: `frac` prices every launch against the WHOLE
                # chip for as long as it lasted; this one prices it against the CUs it was given
                "frac_of_held_cus": fl_sum / t_held / (peak * 1e12),
                "traffic": int(tr_sum / tr_n) if tr_n else None,
                "traffic_note": (f"mean HBM bytes per launch over {tr_n} of the family's kernels, 2 x FETCH_SIZE + WRITE_SIZE "
                                 f"from {os.path.relpath(PMC_FAMILY_FILE, ROOT)} (separate rocprofv3 --pmc passes)"
                                 if tr_n else "no tracked PMC summary found"),
                "algorithmic_bytes": 4 * (cfg["batch"] * 1024 + (cfg["batch"] + 1024) * max(cfg["experts"].values())),
                "us_per_launch": t_sum / max(len([k for k in kernels if k["name"] != "adam_expert"]), 1) * 1e6,
                "flops_per_launch": leg["enc_l1_fwd"][1],
                "kernels": kernels,
                "whole_step": {"tflops": step_tf, "frac": step_tf / peak,
                               "note": "algorithmic FLOPs of the whole step / step time against the same ceiling"},
                "measured": "per kernel: median over 6 eagerly launched training steps behind the timed region of the HIP "
                            "event pair around its launch on its launch stream (one event marker included); `cus` < 256: "
                            "the grid is capped to share the chip with a branch of the program, timed as it runs there",
                "peak_note": "dense bf16 MFMA peak 2500 / 6 MFMA products per fp32 product" if x3 else "dense fp32 MFMA peak",
                "frac_of_f32_mfma_peak": fl_sum / t_sum / (F32_MFMA_TFLOPS * 1e12)}
            if parity is not None:
                out["parity"] = parity
        elif not on_gpu:
            out["rehearsal"] = "CPU plumbing over gloo: launch / exchange / timing logic only, not a measurement"
        if eng is not None and getattr(eng, "dp_tuned", None):
            out["dp_kernels"] = dict(eng.dp_tuned, note="ms per step of the two GEMM kernel families under the exchange, "
                                     "timed on this run's first replayed steps (max over ranks); the faster one ran the timed region")
        if eng is not None and getattr(eng, "streams_probe", None):
            # engine.concurrent_streams: how many of the exchange program's three streams (bulk exchange, small exchanges,
            # the adversaries' lane) were OBSERVED to run beside the main stream on this device
            out["dp_streams"] = eng.streams_probe
        if dp_diag is not None:
            dp_diag["ms_exposed_exchange"] = el / a.steps * 1e3 - dp_diag["ms_per_step_without_transfers"]
            out["data_parallel"] = dp_diag
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.cpu_seconds)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        if on_gpu and torch.distributed.get_backend() == "nccl":
            # RCCL: leave without tearing the communicators down.  destroy_process_group() has aborted the process once in
            # a while on this pool (`Fatal Python error: Aborted` inside it, behind a finished run: tests/helpers.py
            # run_in_child) -- and a rank that dies there turns a measured run into a failed launch.  Every rank is past
            # its last collective (barrier), the device is idle, the line is written.
            if getattr(model, "_engine", None):
                model._engine.close()
            torch.distributed.barrier()
            torch.cuda.synchronize()
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
