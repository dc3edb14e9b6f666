"""Building blocks of the mirror: the public names of `cmmvae.modules.base`, re-exported from where they live here."""
from . import annealing_fn as _schedules
from . import components as _blocks

_EXPORTS = {
    _blocks: ("FCBlockConfig", "ConcatBlockConfig", "FCBlock", "Encoder", "Expert", "Experts", "ConditionalLayer",
              "ConditionalLayers", "Adversarial", "GradientReversalFunction"),
    _schedules: ("KLAnnealingFn", "LinearKLAnnealingFn"),
}
for _module, _names in _EXPORTS.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)
__all__ = sorted(name for names in _EXPORTS.values() for name in names)
del _module, _names, _name
