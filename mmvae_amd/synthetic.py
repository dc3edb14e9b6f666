"""Synthetic single-cell batches and canonical model builders for the BASELINE configs.

Inputs mimic the reference's on-disk value distribution: counts c ~ Poisson(lambda_g), lambda_g = 0.15 LogNormal(0,1),
then x = log1p(1e4 c / rowsum(c))  (scripts/data-preprocessing/data_processing_functions.py:12-31), dense fp32 [B, G].
Model widths are the reference defaults with G substituted (configs/model/compare/adversarial-conditional.yaml:34-108).
"""
from __future__ import annotations

import os
import tempfile
import warnings
from typing import Dict, List, Optional

import pandas as pd
import torch
import torch.nn as nn

ADV_CLASSES = {"assay": 8, "sex": 2, "dataset_id": 273, "donor_id": 4644}  # sizes of data/conditional_layers/*.csv

CONFIGS = {
    # name: experts {id: G}, per-GPU batch, K samples, adversaries
    "c1": dict(experts={"human": 2000}, batch=128, K=1, adversarial=False),
    "c2": dict(experts={"human": 20000, "mouse": 20000}, batch=512, K=1, adversarial=False),
    "c3": dict(experts={"human": 20000, "mouse": 20000}, batch=512, K=10, adversarial=False),
    "c4": dict(experts={"human": 20000, "mouse": 20000}, batch=512, K=1, adversarial=True),
    "c5": dict(experts={"human": 30000, "mouse": 30000, "macaque": 30000}, batch=1024, K=5, adversarial=False),
}


def synthetic_counts(B: int, G: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    g = torch.Generator().manual_seed(seed)
    lam = 0.15 * torch.exp(torch.randn(G, generator=g))
    c = torch.poisson(lam.expand(B, G), generator=g)
    x = torch.log1p(1e4 * c / c.sum(1, keepdim=True).clamp_min(1.0))
    return x.to(device)


def synthetic_metadata(B: int, seed: int = 1234, classes: Optional[Dict[str, int]] = None) -> pd.DataFrame:
    classes = classes or ADV_CLASSES
    g = torch.Generator().manual_seed(seed)
    return pd.DataFrame({c: [f"{c}_{int(i)}" for i in torch.randint(0, n, (B,), generator=g)]
                         for c, n in classes.items()})


def synthetic_labelled_batch(B: int, G: int, seed: int = 1234, device="cpu", classes: Optional[Dict[str, int]] = None,
                             n_patterns: int = 64, effect: float = 0.75):
    """(x [B, G], metadata) whose labels are a FUNCTION OF THE CELL (r5; the adversarial BASELINE config): every cell has a
    donor d, the donor's expression pattern (one of `n_patterns` per-gene log-fold-change vectors, d % n_patterns) scales
    its Poisson rates, and donor_id = d, dataset_id / assay / sex = d modulo the class counts.  The discriminators of
    `gradient_reversal_domain_classifier` (cmmvae_model.py:103-136) then have signal to learn -- as on real data -- and
    the game against the gradient-reversed encoder stays bounded; with labels independent of the cells
    (synthetic_metadata) the discriminator can only memorise, the generator phase drives its loss up without limit and
    the step runs on inf / NaN within tens of steps (profiles/r5_c4_stability.txt).  Same value distribution as
    synthetic_counts (counts -> log1p(1e4 c / rowsum))."""
    classes = classes or ADV_CLASSES
    order = list(classes)
    n_top = max(classes.values())
    g = torch.Generator().manual_seed(seed)
    gp = torch.Generator().manual_seed(4242 + G)  # the gene rates and the donors' patterns belong to the modality
    lam = 0.15 * torch.exp(torch.randn(G, generator=gp))
    patterns = torch.randn(n_patterns, G, generator=gp)
    donors = torch.randint(0, n_top, (B,), generator=g)
    rates = lam[None, :] * torch.exp(effect * patterns[donors % n_patterns] - 0.5 * effect * effect)
    c = torch.poisson(rates, generator=g)
    x = torch.log1p(1e4 * c / c.sum(1, keepdim=True).clamp_min(1.0))
    meta = pd.DataFrame({cond: [f"{cond}_{int(d) % classes[cond]}" for d in donors] for cond in order})
    return x.to(device), meta


def write_label_dir(root: str, classes: Optional[Dict[str, int]] = None) -> str:
    """`<root>/human/unique_expression_<cond>.csv` files as Adversarial expects (components.py:656)."""
    classes = classes or ADV_CLASSES
    os.makedirs(os.path.join(root, "human"), exist_ok=True)
    for cond, n in classes.items():
        pd.Series([f"{cond}_{i}" for i in range(n)]).to_csv(
            os.path.join(root, "human", f"unique_expression_{cond}.csv"), header=False, index=False)
    return root


def build_model(experts: Dict[str, int], *, latent_dim: int = 128, h1: int = 1024, h2: int = 512, hv: int = 256,
                dropout: float = 0.1, adversarial: bool = False, adv_weight: Optional[float] = None,
                labels_dir: Optional[str] = None, n_samples: int = 1, use_engine: bool = True, seed: int = 0):
    """Canonical MMVAE of the BASELINE configs behind the mirror's public classes."""
    from .config import AutogradConfig, GradientClipConfig
    from .models import CMMVAEModel
    from .modules import CMMVAE, CLVAE, base

    def cfg(layers, **kw):
        return base.FCBlockConfig(layers=list(layers), activation_fn=kw.pop("act", nn.ReLU), **kw)

    torch.manual_seed(seed)
    exps = [base.Expert(eid, cfg([G, h1, h2], dropout_rate=dropout, use_batch_norm=True), cfg([h2, h1, G]))
            for eid, G in experts.items()]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = CLVAE(latent_dim=latent_dim, encoder_config=cfg([h2, hv], use_batch_norm=True, return_hidden=True),
                    decoder_config=cfg([latent_dim, hv, h2]), hidden_z=adversarial)
    vae.encoder.n_samples = n_samples
    advs = None
    if adversarial:
        base.Adversarial.labels.clear()
        labels_dir = labels_dir or write_label_dir(tempfile.mkdtemp(prefix="mmvae_labels_"))
        conds = list(ADV_CLASSES.keys())
        advs = [base.Adversarial(cfg([hv, 128, 64]), cfg([64], act=None), conds, labels_dir),
                base.Adversarial(cfg([latent_dim, 64]), cfg([64], act=None), conds, labels_dir)]
        adv_weight = 25 if adv_weight is None else adv_weight
    clip = lambda: GradientClipConfig(val=10, algorithm="norm")  # noqa: E731
    return CMMVAEModel(CMMVAE(vae, base.Experts(exps), advs), adv_weight=adv_weight,
                       autograd_config=AutogradConfig(clip(), clip(), clip()), use_engine=use_engine)


def flops_per_cell(G: int, K: int = 1, h1: int = 1024, h2: int = 512, hv: int = 256, Z: int = 128,
                   mode: str = "train") -> float:
    """Algorithmic FLOPs per cell of one step.  train: fwd + dW everywhere + dX except the input layer;
    validate: the forward pass alone (encoder + decoder, ONE sample: the forward-only programs draw one rsample whatever
    K is); predict: the encoder alone (the program ends at z)."""
    E = G * h1 + h1 * h2 + h2 * hv + 2 * hv * Z
    D = Z * hv + hv * h2 + h2 * h1 + h1 * G
    if mode == "validate":
        return 2.0 * (E + D)
    if mode == "predict":
        return 2.0 * E
    return 6.0 * E - 2.0 * G * h1 + 6.0 * K * D
