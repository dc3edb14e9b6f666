"""KL-weight schedules (host-side scalars).  Mirrors reference `cmmvae/modules/base/annealing_fn.py:1-42`."""


class KLAnnealingFn:
    """Constant KL weight; `step()` is the per-training-step hook subclasses override."""

    def __init__(self, kl_weight: float):
        self._kl_weight = kl_weight

    @property
    def kl_weight(self):
        return self._kl_weight

    @kl_weight.setter
    def kl_weight(self, weight):
        self._kl_weight = weight

    def step(self) -> None:
        pass


class LinearKLAnnealingFn(KLAnnealingFn):
    """min for `warmup_steps` steps, then a linear ramp of slope (max - min) / climax_steps, clamped to [min, max].
    The step counter starts at -warmup_steps; the weight only changes once the counter is >= 0
    (annealing_fn.py:34-42)."""

    def __init__(self, min_kl_weight: float = 1e-7, max_kl_weight: float = 1e-5, warmup_steps: float = 1e3,
                 climax_steps: float = 1e4):
        super().__init__(min_kl_weight)
        self._min = min_kl_weight
        self._max = max_kl_weight
        self._warmup_steps = warmup_steps
        self._climax_steps = climax_steps
        self.m = (max_kl_weight - min_kl_weight) / climax_steps
        self.b = min_kl_weight
        self.x = -warmup_steps

    def step(self) -> None:
        self.x += 1
        if self.x >= 0:
            self.kl_weight = min(max(self.m * self.x + self.b, self._min), self._max)
