"""Conditional-latent VAE (mirror of `cmmvae/modules/clvae.py:10-111`): a VAE whose latent sample optionally passes
through metadata-conditioned layers before decoding; "parallel" mode widens the decoder input."""
from __future__ import annotations

import warnings
from typing import Optional

import pandas as pd
import torch

from .base import ConcatBlockConfig, ConditionalLayers, FCBlockConfig
from .vae import VAE


class CLVAE(VAE):
    def __init__(self, encoder_config: FCBlockConfig, decoder_config: FCBlockConfig,
                 conditional_config: Optional[FCBlockConfig] = None, conditionals_directory: Optional[str] = None,
                 conditionals: Optional[list] = None, selection_order: Optional[list] = None,
                 concat_config: Optional[ConcatBlockConfig] = None, **encoder_kwargs):
        conditionals_module = None
        if conditional_config and conditionals and conditionals_directory:
            conditionals_module = ConditionalLayers(directory=conditionals_directory, conditionals=conditionals,
                                                    fc_block_config=conditional_config,
                                                    selection_order=selection_order)
        else:
            warnings.warn("No conditionals found for vae")
        if selection_order and selection_order[0] == "parallel":
            if not concat_config:
                raise RuntimeError("Please define concat_config when selection_order = parallel")
            if conditionals_module is None:
                raise RuntimeError("selection_order = parallel needs conditional layers")
            concat_dim = len(conditionals_module.selection_order) * conditional_config.layers[-1]
            # one extra decoder layer concat_dim -> old input width, described by concat_config (clvae.py:55-79)
            decoder_config.layers = [concat_dim] + decoder_config.layers
            for name in ("activation_fn", "dropout_rate", "return_hidden", "use_layer_norm", "use_batch_norm"):
                setattr(decoder_config, name, [getattr(concat_config, name)] + list(getattr(decoder_config, name)))
        super().__init__(encoder_config=encoder_config, decoder_config=decoder_config, **encoder_kwargs)
        self.conditionals = conditionals_module

    def after_reparameterize(self, z: torch.Tensor, metadata: pd.DataFrame, **kwargs) -> torch.Tensor:
        if self.conditionals:
            if z.dim() != 2:
                # the K-sample ELBO is this build's extension, conditional layers are the reference's: the reference
                # routes one latent row per cell through the cell's condition blocks and defines nothing for K > 1
                raise ValueError("conditional layers are defined for one latent sample per cell (encoder n_samples = 1)")
            return self.conditionals(z, metadata, **kwargs)
        return z
