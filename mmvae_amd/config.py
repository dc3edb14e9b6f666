"""Gradient-clipping configuration objects of the trainer plugin surface.

Mirrors reference `cmmvae/config.py:4-26` (class names, argument names and iteration behaviour are part of the
YAML schema: configs/model/human_only.yaml:10-27)."""
from typing import Literal, Optional, Union


class GradientClipConfig:
    """(val, algorithm) pair; iterable so it can be splatted into clip_gradients(optimizer, *cfg)."""

    def __init__(self, val: Optional[Union[int, float]] = None,
                 algorithm: Optional[Literal["norm", "value"]] = None):
        self.val = val
        self.algorithm = algorithm

    def __iter__(self):
        yield self.val
        yield self.algorithm

    def __repr__(self):
        return f"GradientClipConfig(val={self.val!r}, algorithm={self.algorithm!r})"


class AutogradConfig:
    """Per-optimiser-family clipping: adversaries, the shared VAE, the active expert."""

    def __init__(self, adversarial_gradient_clip: Optional[GradientClipConfig] = None,
                 vae_gradient_clip: Optional[GradientClipConfig] = None,
                 expert_gradient_clip: Optional[GradientClipConfig] = None):
        self.adversarial_gradient_clip = adversarial_gradient_clip
        self.vae_gradient_clip = vae_gradient_clip
        self.expert_gradient_clip = expert_gradient_clip
