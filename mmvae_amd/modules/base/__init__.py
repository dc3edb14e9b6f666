"""Building blocks of the mirror: the public names of `cmmvae.modules.base`."""
from .annealing_fn import KLAnnealingFn, LinearKLAnnealingFn
from .components import (Adversarial, ConcatBlockConfig, ConditionalLayer, ConditionalLayers, Encoder, Expert, Experts,
                         FCBlock, FCBlockConfig, GradientReversalFunction)

__all__ = ["Adversarial", "ConcatBlockConfig", "ConditionalLayer", "ConditionalLayers", "Encoder", "Expert", "Experts",
           "FCBlock", "FCBlockConfig", "GradientReversalFunction", "KLAnnealingFn", "LinearKLAnnealingFn"]
