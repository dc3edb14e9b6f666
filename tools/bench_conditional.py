#!/usr/bin/env python3
"""Conditional-layer model at the reference's scale: Z = 128, one Linear(128, 128) + LayerNorm per
condition, conditionals assay (8), sex (2), dataset_id (273), donor_id (4644) + the per-species block, B = 512, two
20 000-gene modalities.  Times CMMVAEModel.training_step through the captured engine (conditional layers inside the
program), on the module path with the grouped HIP kernels, and on the module path with the per-condition loop of the
reference (MMVAE_COND_GROUPED=0).  --engine-only: the first of the three."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

SIZES = {"assay": 8, "sex": 2, "dataset_id": 273, "donor_id": 4644}


FULL_GENES = {"human": 60530, "mouse": 52437}  # configs/model/compare/adversarial-conditional.yaml:85,101


def build(root, G=20000, Z=128, use_engine=False, full=False, parallel=False):
    """full: the reference's adversarial-conditional model at its real size -- 60 530 / 52 437 genes, conditional layers
    assay / dataset_id / donor_id / tissue (per species) / species, two adversaries (on h1 [256,128,64] and on z [128,64], human_only.yaml:
    103-157) with heads for the four conditions whose class counts ship with the reference."""
    from mmvae_amd.config import AutogradConfig, GradientClipConfig
    from mmvae_amd.models import CMMVAEModel
    from mmvae_amd.modules import CLVAE, CMMVAE, base

    os.makedirs(os.path.join(root, "shared"), exist_ok=True)
    for k, n in SIZES.items():
        pd.Series([f"{k}_{i}" for i in range(n)]).to_csv(os.path.join(root, "shared", f"unique_expression_{k}.csv"),
                                                        header=False, index=False)
    for sp in ("human", "mouse"):  # a species-specific key so that the species block exists
        os.makedirs(os.path.join(root, sp), exist_ok=True)
        pd.Series([f"t_{sp}_{i}" for i in range(4)]).to_csv(os.path.join(root, sp, "unique_expression_tissue.csv"),
                                                            header=False, index=False)

    def cfg(layers, dropout=0.0, bn=False, relu=True, hidden=False, ln=False):
        return base.FCBlockConfig(layers=list(layers), dropout_rate=dropout, use_batch_norm=bn, use_layer_norm=ln,
                                  activation_fn=nn.ReLU if relu else None, return_hidden=hidden)

    genes = FULL_GENES if full else {"human": G, "mouse": G}
    experts = [base.Expert(e, cfg([g, 1024, 512], dropout=0.1, bn=True), cfg([512, 1024, g])) for e, g in genes.items()]
    # (a species-specific key is what makes the per-species block exist: components.py:420-464 omits species without files)
    keys = ["assay", "dataset_id", "donor_id", "tissue", "species"] if full else list(SIZES) + ["tissue", "species"]
    advs = None
    if full:
        os.makedirs(os.path.join(root, "labels", "human"), exist_ok=True)
        for k, n in SIZES.items():
            pd.Series([f"{k}_{i}" for i in range(n)]).to_csv(
                os.path.join(root, "labels", "human", f"unique_expression_{k}.csv"), header=False, index=False)
        base.Adversarial.labels.clear()
        advs = [base.Adversarial(encoder=cfg(enc), heads=cfg([enc[-1]], relu=False), conditions=list(SIZES),
                                 labels_dir=os.path.join(root, "labels")) for enc in ([256, 128, 64], [Z, 64])]
    # parallel: the reference's default (human_only.yaml:78-79) -- every layer reads z, the outputs are concatenated and a
    # Linear(n_keys * Z, Z) + ReLU in front of the decoder takes them in (clvae.py:55-79)
    extra = dict(concat_config=base.ConcatBlockConfig(dropout_rate=0.0, use_batch_norm=False, use_layer_norm=False,
                                                      activation_fn=nn.ReLU)) if parallel else {}
    vae = CLVAE(latent_dim=Z, encoder_config=cfg([512, 256], bn=True, hidden=True), decoder_config=cfg([Z, 256, 512]),
                conditional_config=cfg([Z], relu=False, ln=True), conditionals_directory=root, conditionals=list(keys),
                selection_order=["parallel"] if parallel else list(keys), hidden_z=full, **extra)
    clip = lambda: GradientClipConfig(val=10, algorithm="norm")
    torch.manual_seed(0)
    return CMMVAEModel(CMMVAE(vae, base.Experts(experts), advs), adv_weight=25 if full else None,
                       autograd_config=AutogradConfig(clip(), clip(), clip()), use_engine=use_engine).cuda()


_CATS = {}


def metadata(B, eid, seed, categorical=False):
    """categorical: `category` columns whose categories are all labels of the key and are SHARED by every frame -- what
    row slices of one chunk's obs frame look like (census obs frames hold categorical columns); default: str columns."""
    rng = np.random.default_rng(seed)
    md = {k: [f"{k}_{i}" for i in rng.integers(0, n, B)] for k, n in SIZES.items()}
    md["tissue"] = [f"t_{eid}_{i}" for i in rng.integers(0, 4, B)]
    md = pd.DataFrame(md)
    if categorical:
        for k in md.columns:
            if k not in _CATS:
                labels = [f"{k}_{i}" for i in range(SIZES[k])] if k in SIZES else [f"t_{e}_{i}" for e in ("human", "mouse") for i in range(4)]
                _CATS[k] = pd.CategoricalDtype(labels)
            md[k] = md[k].astype(_CATS[k])
    return md


def run_engine(steps=60, warm=12, B=512, G=20000, full=False, parallel=False):
    """The same model through the captured engine (conditional layers inside the program); no host read-back inside the
    timed region, metadata frames prepared beforehand (the feed's job)."""
    from mmvae_amd import synthetic

    with tempfile.TemporaryDirectory() as d:
        model = build(d, G, use_engine=True, full=full, parallel=parallel)
        model.train()
        model.trainer.set_stage("training")
        if full:
            # random synthetic labels + the reference's lr 5e-3 drive this model's KL term to overflow within ~15 steps
            # (on the autograd module path just the same); the step's work does not depend on the learning rate, so the
            # timing run uses a small one and stays finite
            for opt in model.optimizers():
                for g in opt.param_groups:
                    g["lr"] = 1e-5
        genes = FULL_GENES if full else {"human": G, "mouse": G}
        xs = {e: synthetic.synthetic_counts(B, g, seed=3 + i, device="cuda") for i, (e, g) in enumerate(genes.items())}
        n_md = min(steps, 72) // 2 * 2  # (long runs cycle through 72 frames: --steps N)
        mds = [metadata(B, ("human", "mouse")[i % 2], i, categorical="--categorical" in sys.argv) for i in range(n_md)]
        first = []
        for i in range(steps):
            if i == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            eid = ("human", "mouse")[i % 2]
            if i + 1 < steps and os.environ.get("MMVAE_BENCH_LOOKAHEAD", "1") != "0":  # the loop's look-ahead (trainer.Lookahead)
                nxt = ("human", "mouse")[(i + 1) % 2]
                model.hint_next_batch((xs[nxt], mds[(i + 1) % n_md], nxt))
            model.training_step((xs[eid], mds[i % n_md], eid), i)
            if i < 3:
                first.append(float(model.logged[f"loss/training/{eid}"]))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (steps - warm) * 1e3
        assert model._engine and all(p.cond is not None for p in model._engine._plans.values())
        last = float(model.logged[f"loss/training/{eid}"])
    return ms, first, last


def run(grouped: bool, steps=12, B=512, G=20000):
    from mmvae_amd import synthetic

    os.environ["MMVAE_COND_GROUPED"] = "1" if grouped else "0"
    with tempfile.TemporaryDirectory() as d:
        model = build(d, G)
        model.train()
        model.trainer.set_stage("training")
        xs = {e: synthetic.synthetic_counts(B, G, seed=3 + i, device="cuda") for i, e in enumerate(("human", "mouse"))}
        losses, t0 = [], None
        for i in range(steps):
            if i == 4:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            eid = ("human", "mouse")[i % 2]
            torch.manual_seed(100 + i)  # dropout masks / rsample noise of the module path come from torch's generator
            model.training_step((xs[eid], metadata(B, eid, i), eid), i)
            losses.append(float(model.logged[f"loss/training/{eid}"]))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / (steps - 4) * 1e3
        n_params = sum(p.numel() for p in model.module.vae.conditionals.parameters())
    return ms, losses, n_params


if __name__ == "__main__":
    STEPS = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 60
    if "--full" in sys.argv:
        ms_f, first_f, last_f = run_engine(steps=STEPS, full=True)
        print(f"reference's adversarial-conditional model, 60530 / 52437 genes, B = 512, captured engine: "
              f"{ms_f:8.2f} ms / step = {512 / ms_f * 1e3:,.0f} cells/s   losses {first_f} ... {last_f:.1f}")
        sys.exit(0)
    if "--parallel" in sys.argv:  # the reference's default selection order; MMVAE_COND_BATCHED=0: one launch per position
        import random

        random.seed(0)
        ms_p, first_p, last_p = run_engine(steps=STEPS, parallel=True)
        print(f"captured engine, selection order 'parallel' (MMVAE_COND_BATCHED={os.environ.get('MMVAE_COND_BATCHED', '1')}"
              f"{', categorical metadata' if '--categorical' in sys.argv else ''}): "
              f"{ms_p:8.2f} ms / step   losses {first_p} ... {last_p:.1f}")
        sys.exit(0)
    ms_e, first_e, last_e = run_engine(steps=STEPS)
    print(f"captured engine     : {ms_e:8.2f} ms / step   losses {first_e} ... {last_e:.1f} (device Philox noise)")
    if "--engine-only" in sys.argv:
        sys.exit(0)
    ms_g, loss_g, n = run(True)
    ms_l, loss_l, _ = run(False)
    print(f"conditional parameters: {n / 1e6:.1f} M in {sum(SIZES.values()) + 8 + 2} blocks")
    print(f"grouped HIP kernels : {ms_g:8.1f} ms / step   losses {loss_g[:3]} ... {loss_g[-1]:.1f}")
    print(f"per-condition loop  : {ms_l:8.1f} ms / step   losses {loss_l[:3]} ... {loss_l[-1]:.1f}")
    rel = max(abs(a - b) / abs(b) for a, b in zip(loss_g, loss_l))
    print(f"max relative loss difference over {len(loss_g)} steps: {rel:.2e}")
