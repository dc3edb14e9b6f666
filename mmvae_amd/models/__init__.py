"""Trainer plugin surface of the mirror (same public names as `cmmvae.models`)."""
from .base_model import BaseModel, tag_log_dict
from .cmmvae_model import CMMVAEModel

__all__ = ["BaseModel", "CMMVAEModel", "tag_log_dict"]
