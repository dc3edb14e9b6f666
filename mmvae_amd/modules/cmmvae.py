"""CMMVAE: expert encoder -> shared CLVAE -> expert decoder, plus optional adversarial discriminators.
Mirror of `cmmvae/modules/cmmvae.py:11-142` (call signature and return tuple are part of the drop-in boundary)."""
from __future__ import annotations

import warnings
from typing import Optional

import pandas as pd
import torch
from torch import nn

from ..constants import REGISTRY_KEYS as RK
from .base import Adversarial, Experts
from .clvae import CLVAE


class CMMVAE(nn.Module):
    def __init__(self, vae: CLVAE, experts: Experts, adversarials: Optional[list] = None):
        super().__init__()
        self.vae = vae
        self.experts = experts
        # The reference leaves `adversarials` undefined when None is passed, which breaks configure_optimizers
        # (SURVEY 8b "stale-config caveat"); here None / [] simply mean "no adversaries".
        self.adversarials = nn.ModuleList([adv for adv in (adversarials or []) if adv])

    def forward(self, x: torch.Tensor, metadata: pd.DataFrame, expert_id: str, cross_generate: bool = False):
        """-> (qz, pz, z, {expert_id: xhat}, hidden)   (cmmvae.py:51-113)"""
        shared_x = self.experts[expert_id].encode(x)
        qz, pz, z, shared_xhat, hidden = self.vae(shared_x, metadata, species=expert_id)
        xhats = {}
        if cross_generate:
            if self.training:
                warnings.warn("CMMVAE is cross-generating during training: gradients accumulate for every expert")
            for eid in self.experts:
                xhats[eid] = self.experts[eid].decode(shared_xhat)
        else:
            xhats[expert_id] = self.experts[expert_id].decode(shared_xhat)
        return qz, pz, z, xhats, hidden

    @torch.no_grad()
    def get_latent_embeddings(self, x: torch.Tensor, metadata: pd.DataFrame, expert_id: str) -> dict:
        """{"z": (z, metadata)} with metadata["species"] = expert_id written in place (cmmvae.py:115-142)."""
        x = self.experts[expert_id].encode(x)
        _, z, _ = self.vae.encode(x)
        metadata["species"] = expert_id
        return {RK.Z: (z, metadata)}
