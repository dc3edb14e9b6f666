"""Shared-core VAE of the mirror: encode -> reparameterise -> [conditionals] -> decode, and the ELBO.

Mirrors `cmmvae/modules/vae.py` (BaseVAE :12-176, VAE :178-205).  On device tensors the ELBO terms come from the
HIP kernels: the Gaussian KL row sums were already produced by the fused reparameterisation kernel
(Encoder.forward) and the reconstruction term is the sum-of-squares kernel (functional.MseSumFn).
"""
from __future__ import annotations

import pandas as pd
import torch
import torch.nn.functional as F
from torch import nn
from torch.distributions import Distribution, Normal, kl_divergence

from .. import backend
from .. import functional as HF
from ..constants import REGISTRY_KEYS as RK
from . import base


class BaseVAE(nn.Module):
    def __init__(self, encoder: base.Encoder, decoder: nn.Module):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def encode(self, x: torch.Tensor, **kwargs):
        qz, z, hidden = self.encoder(x)
        return qz, z, hidden

    def decode(self, z: torch.Tensor, **kwargs) -> torch.Tensor:
        return self.decoder(z)

    def after_reparameterize(self, z: torch.Tensor, metadata: pd.DataFrame, **kwargs) -> torch.Tensor:
        return z

    def forward(self, x: torch.Tensor, metadata: pd.DataFrame, **kwargs):
        """-> (qz, pz, z, xhat, hidden)  (vae.py:80-102).  With the K-sample extension z is [K,B,Z] and the decoder
        runs on the K*B stacked samples."""
        qz, z, hidden = self.encode(x, **kwargs)
        pz = Normal(torch.zeros_like(qz.loc), torch.ones_like(qz.loc), validate_args=False)
        z = self.after_reparameterize(z, metadata, **kwargs)
        xhat = self.decode(z.reshape(-1, z.shape[-1]) if z.dim() == 3 else z, **kwargs)
        return qz, pz, z, xhat, hidden

    def elbo(self, qz: Distribution, pz: Distribution, x: torch.Tensor, xhat: torch.Tensor, kl_weight: float,
             **kwargs) -> dict:
        """loss = recon + kl_weight * KL, KL = sum over latent, MEAN over cells; recon = SUM of squared errors over
        cells x genes (vae.py:136-152).  xhat with K*B rows selects the K-sample log-mean-exp extension."""
        if x.layout == torch.sparse_csr:
            x = backend.to_dense(x)
        if backend.on_hip(xhat):
            cache = getattr(qz, "_mmvae", None)
            if cache is not None:
                z_kl_div = cache["kl_sum"] / cache["batch"]
            else:  # foreign distribution objects: generic (non-fused) KL
                z_kl_div = kl_divergence(qz, pz).sum(dim=-1).mean()
            K = xhat.shape[0] // x.shape[0]
            if K == 1:
                recon_loss = HF.MseSumFn.apply(xhat.contiguous(), x.contiguous())
            else:
                recon_loss = HF.KSampleReconFn.apply(xhat.contiguous(), x.contiguous(), K)
        else:
            z_kl_div = kl_divergence(qz, pz).sum(dim=-1).mean()
            recon_loss = F.mse_loss(xhat, x, reduction="sum")
        loss = recon_loss + kl_weight * z_kl_div
        return {RK.LOSS: loss, RK.RECON_LOSS: recon_loss, RK.KL_LOSS: z_kl_div, RK.KL_WEIGHT: kl_weight}

    @torch.no_grad()
    def get_latent_embeddings(self, x: torch.Tensor, metadata: pd.DataFrame, **kwargs) -> dict:
        _, z, _ = self.encode(x)
        return {RK.Z: z, f"{RK.Z}_{RK.METADATA}": metadata}


class VAE(BaseVAE):
    """BaseVAE built from two FCBlockConfigs; extra kwargs go to the Encoder (vae.py:178-205)."""

    def __init__(self, encoder_config: base.FCBlockConfig, decoder_config: base.FCBlockConfig, **encoder_kwargs):
        super().__init__(
            encoder=base.Encoder(fc_block_config=encoder_config, return_dist=True, **encoder_kwargs),
            decoder=base.FCBlock(decoder_config),
        )
