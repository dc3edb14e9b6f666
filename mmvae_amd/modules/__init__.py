"""nn.Modules of the mirror (the public names of `cmmvae.modules`): `base` building blocks, the plain and the
conditional-latent VAE, and the multi-expert model that routes a batch through one expert and the shared VAE."""
from . import base  # noqa: F401
from .clvae import CLVAE  # noqa: F401
from .cmmvae import CMMVAE  # noqa: F401
from .vae import VAE  # noqa: F401

__all__ = ("base", "VAE", "CLVAE", "CMMVAE")
