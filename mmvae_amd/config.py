"""Gradient-clipping configuration objects of the trainer plugin surface.

Mirrors reference `cmmvae/config.py:4-26` (class names, argument names and iteration behaviour are part of the
YAML schema: configs/model/human_only.yaml:10-27)."""
from dataclasses import dataclass, fields
from typing import Iterator, Optional, Union

Number = Union[int, float]


@dataclass
class GradientClipConfig:
    """How one optimiser's gradients are clipped.  Iterating yields (val, algorithm), so that the object can be
    splatted into `clip_gradients(optimizer, *cfg)` the way the reference's training step does."""

    val: Optional[Number] = None
    algorithm: Optional[str] = None  # "norm" | "value"

    def __iter__(self) -> Iterator:
        return (getattr(self, f.name) for f in fields(self))


@dataclass
class AutogradConfig:
    """One clipping rule per optimiser family; None = no clipping for that family."""

    adversarial_gradient_clip: Optional[GradientClipConfig] = None  # every adversary's optimiser
    vae_gradient_clip: Optional[GradientClipConfig] = None          # the shared VAE
    expert_gradient_clip: Optional[GradientClipConfig] = None       # the expert that is active in the step
