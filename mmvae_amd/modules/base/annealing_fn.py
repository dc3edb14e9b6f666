"""KL-weight schedules (host-side scalars).  Mirrors reference `cmmvae/modules/base/annealing_fn.py:1-42`."""


class KLAnnealingFn:
    """A KL weight that stays what it was set to; `step()` is the once-per-training-step hook of the schedules."""

    def __init__(self, kl_weight: float):
        self._kl_weight = kl_weight

    @property
    def kl_weight(self) -> float:
        """Current weight of the KL term (read by the step, settable by the user)."""
        return self._kl_weight

    @kl_weight.setter
    def kl_weight(self, value: float) -> None:
        self._kl_weight = value

    def step(self) -> None:
        """Constant schedule: nothing to advance."""


class LinearKLAnnealingFn(KLAnnealingFn):
    """`min_kl_weight` during the first `warmup_steps` calls of `step()`, then a straight ramp that reaches
    `max_kl_weight` after `climax_steps` more calls and stays there (annealing_fn.py:17-42: a counter that starts at
    -warmup_steps; the weight is only recomputed once the counter is non-negative).  `m`, `b`, `x` (slope, intercept,
    counter) are the reference's public attribute names and are kept: resuming a run restores `x`."""

    def __init__(self, min_kl_weight: float = 1e-7, max_kl_weight: float = 1e-5, warmup_steps: float = 1e3,
                 climax_steps: float = 1e4):
        super().__init__(min_kl_weight)
        self.low, self.high = min_kl_weight, max_kl_weight
        self.m = (max_kl_weight - min_kl_weight) / climax_steps
        self.b = min_kl_weight
        self.x = -warmup_steps

    def step(self) -> None:
        self.x += 1
        if self.x >= 0:
            self.kl_weight = min(max(self.m * self.x + self.b, self.low), self.high)
