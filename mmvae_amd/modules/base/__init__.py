"""Building blocks of the mirror (same public names as `cmmvae.modules.base`)."""
from .components import (
    Adversarial,
    ConcatBlockConfig,
    ConditionalLayer,
    ConditionalLayers,
    Encoder,
    Expert,
    Experts,
    FCBlock,
    FCBlockConfig,
    GradientReversalFunction,
)
from .annealing_fn import KLAnnealingFn, LinearKLAnnealingFn

__all__ = [
    "Adversarial", "ConditionalLayer", "ConditionalLayers", "ConcatBlockConfig", "Encoder", "Expert", "Experts",
    "FCBlock", "FCBlockConfig", "GradientReversalFunction", "KLAnnealingFn", "LinearKLAnnealingFn",
]
