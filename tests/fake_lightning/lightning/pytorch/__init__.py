"""TEST SCAFFOLDING -- the part of `lightning.pytorch.LightningModule` that the reference's manual-optimisation step
touches (models/base_model.py:51-123, models/cmmvae_model.py:138-217 of the reference: `self.optimizers()`,
`self.manual_backward`, `self.clip_gradients`, `self.log` / `self.log_dict`, `self.trainer` stage flags), restated in
the shape Lightning gives it: optimisers handed out as LightningOptimizer-style WRAPPERS that carry a copy of the
optimiser's attribute dictionary and forward step() to the wrapped object, `clip_gradients` clipping `p.grad` in place
the way the precision plugin does, a `trainer` that must be attached.  Every call is counted (`calls`) so the test can
tell which implementation ran.  Not Lightning: none of its loops, strategies or loggers."""
import collections

import torch
import torch.nn as nn


class _Trainer:
    def __init__(self):
        self.training = True
        self.validating = self.sanity_checking = self.predicting = self.testing = False
        self.global_step = 0
        self.gradient_clip_val = None

    def set_stage(self, stage):  # (what a Trainer's loops flip; the mirror's tests call it directly)
        self.training, self.validating = stage == "training", stage == "validation"
        self.testing, self.predicting = stage == "test", stage == "prediction"
        self.sanity_checking = stage == "sanity_checking"


class LightningOptimizer:
    """Like lightning.pytorch.core.optimizer.LightningOptimizer: same class lineage as the wrapped optimiser, a COPY of its
    __dict__, step() forwarded to the wrapped object and counted by the trainer."""

    def __init__(self, optimizer, trainer):
        self.__dict__ = {k: v for k, v in optimizer.__dict__.items() if k not in ("step", "__del__")}
        self.__class__ = type("Lightning" + optimizer.__class__.__name__, (self.__class__, optimizer.__class__), {})
        self._optimizer, self._fl_trainer = optimizer, trainer

    @property
    def optimizer(self):
        return self._optimizer

    def step(self, closure=None, **kwargs):
        self._fl_trainer.global_step += 1
        return self._optimizer.step()


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self.automatic_optimization = True
        self._trainer = _Trainer()
        self._fabric = None
        self._fl_opts = None
        self.calls = collections.Counter()

    @property
    def trainer(self):
        if self._trainer is None:
            raise RuntimeError("LightningModule is not attached to a Trainer")
        return self._trainer

    @trainer.setter
    def trainer(self, t):
        self._trainer = t

    def optimizers(self, use_pl_optimizer=True):
        self.calls["optimizers"] += 1
        if self._fl_opts is None:
            self._fl_opts = list(self.configure_optimizers())
        if use_pl_optimizer:
            return [LightningOptimizer(o, self._trainer) for o in self._fl_opts]
        return list(self._fl_opts)

    def manual_backward(self, loss, *args, **kwargs):
        self.calls["manual_backward"] += 1
        loss.backward(*args, **kwargs)

    def clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        self.calls["lightning_clip_gradients"] += 1  # (the precision plugin's clip: on p.grad, in place)
        params = [p for g in optimizer.param_groups for p in g["params"]]
        if gradient_clip_algorithm == "value":
            torch.nn.utils.clip_grad_value_(params, gradient_clip_val)
        else:
            torch.nn.utils.clip_grad_norm_(params, gradient_clip_val)

    def log(self, name, value, **kwargs):
        self.calls["log"] += 1

    def log_dict(self, dictionary, **kwargs):
        self.calls["log_dict"] += 1
