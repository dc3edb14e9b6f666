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


def flops_per_cell(G: int, K: int = 1, h1: int = 1024, h2: int = 512, hv: int = 256, Z: int = 128) -> float:
    """Algorithmic FLOPs per cell of one training step: fwd + dW everywhere + dX except the input layer."""
    E = G * h1 + h1 * h2 + h2 * hv + 2 * hv * Z
    D = Z * hv + hv * h2 + h2 * h1 + h1 * G
    return 6.0 * E - 2.0 * G * h1 + 6.0 * K * D
