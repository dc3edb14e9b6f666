"""Trainer plugin surface of the mirror (the public names of `cmmvae.models`): the LightningModule-shaped base class
with its logging helpers, and the MMVAE training / validation / prediction steps on top of it."""
from .base_model import BaseModel, tag_log_dict  # noqa: F401
from .cmmvae_model import CMMVAEModel  # noqa: F401

__all__ = ("CMMVAEModel", "BaseModel", "tag_log_dict")
