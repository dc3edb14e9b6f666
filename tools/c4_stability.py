#!/usr/bin/env python3
"""Loss trajectory of the C4 (adversarial) program on synthetic data: does the training stay finite under the reference's
hyper-parameters (adv_weight 25, Adam 5e-3 / 1e-6, clip 10 by norm; cmmvae_model.py:59-136,182-184)?
usage: c4_stability.py STEPS MODE [N_RES]   MODE: uniform (labels independent of the cells) | cells (labels and counts
both functions of a latent cell type: synthetic.synthetic_labelled_batch)"""
import math
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mmvae_amd import instantiate, synthetic
from mmvae_amd.modules import base

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "uniform"
n_res = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda", 0)
torch.manual_seed(0)
base.Adversarial.labels.clear()
os.environ["MMVAE_LABELS_DIR"] = synthetic.write_label_dir(tempfile.mkdtemp(prefix="mmvae_labels_"))
model = instantiate.load_yaml(os.path.join(os.path.dirname(__file__), "..", "configs", "model",
                                           "c4_two_modality_20k_adversarial.yaml")).to(dev)
model.train()
model.trainer.set_stage("training")
model.optimizers()
cfg = synthetic.CONFIGS["c4"]
B = cfg["batch"]
data = {}
for i, (eid, G) in enumerate(cfg["experts"].items()):
    if mode == "cells":
        data[eid] = [synthetic.synthetic_labelled_batch(B, G, seed=1234 + 97 * i + 13 * j, device=dev) for j in range(n_res)]
    else:
        data[eid] = [(synthetic.synthetic_counts(B, G, seed=1234 + 97 * i + 13 * j, device=dev),
                      synthetic.synthetic_metadata(B, seed=5 + j)) for j in range(n_res)]
eids = list(data)
worst = 0.0
for s in range(steps):
    eid = eids[s % len(eids)]
    x, m = data[eid][(s // len(eids)) % n_res]
    model.training_step((x, m, eid), s)
    if s % max(1, steps // 25) in (0, 1) or s >= steps - 2:
        if getattr(model, "_engine", None):
            model._flush_engine()
        torch.cuda.synchronize()
        L = {k: float(v) for k, v in model.logged.items()}
        tot = L.get(f"loss/training/{eid}")
        adv = {k.split("/")[0]: round(v, 1) for k, v in L.items() if k.endswith("summed") and eid in k}
        norms = {k.split("/")[-1]: round(v, 1) for k, v in L.items() if k.startswith("grad_norms/")}
        worst = max(worst, abs(tot)) if math.isfinite(tot) else float("inf")
        print(f"step {s:6d} {eid:6s} loss {tot:.4g} recon {L.get(f'recon_loss/training/{eid}'):.4g} "
              f"kl {L.get(f'kl_loss/training/{eid}'):.4g} adv {adv} norms {norms}", flush=True)
print(f"worst |loss| seen {worst:.4g}; finite {math.isfinite(worst)}")
