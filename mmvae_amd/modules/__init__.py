"""nn.Modules of the mirror (same public names as `cmmvae.modules`)."""
from . import base
from .vae import VAE
from .clvae import CLVAE
from .cmmvae import CMMVAE

__all__ = ["base", "CLVAE", "CMMVAE", "VAE"]
