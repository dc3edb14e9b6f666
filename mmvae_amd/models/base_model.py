"""Trainer-facing base class of the mirror.

The reference's `BaseModel` is a `lightning.pytorch.LightningModule` (cmmvae/models/base_model.py:52-355).  Lightning
is an optional dependency here: when importable, this class derives from it and plugs into a stock Lightning
Trainer; otherwise it provides the small part of the LightningModule surface the MMVAE step uses (log / log_dict /
optimizers / manual_backward / clip_gradients / trainer stage flags) so mmvae_amd.trainer.Trainer can drive it.

Logged scalar names are part of the drop-in boundary (`{key}/{stage}/{expert}`, base_model.py:265-297).
"""
from __future__ import annotations

from typing import Iterable, Literal, Optional, Union

import torch
import torch.nn as nn

from ..modules.base import KLAnnealingFn
from ..modules.base import init as _init  # noqa: F401  (module import keeps `init` reachable like the reference)
from ..modules.base.init import he_init_weights

try:  # Lightning is not installed in the build image: tests/test_lightning_surface.py drives this branch through a
    # stand-in `lightning.pytorch` (tests/fake_lightning) -- the calls the step makes, not Lightning's own machinery
    import lightning.pytorch as pl

    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    _Base = nn.Module
    HAVE_LIGHTNING = False


def tag_log_dict(log_dict: dict, tags: Iterable[str] = (), sep: str = "/",
                 key_pos: Union[Literal["first"], Literal["last"]] = "first") -> dict:
    """{key: v} -> {"key/tag1/tag2": v} (key_pos "first") or {"tag1/tag2/key": v} ("last"); base_model.py:14-49."""
    if key_pos not in ("first", "last"):
        raise ValueError(f"Key position {key_pos} is not supported!")
    tags_str = sep.join(tags)
    if not tags_str:
        return dict(log_dict)
    if key_pos == "first":
        return {f"{k}{sep}{tags_str}": v for k, v in log_dict.items()}
    return {f"{tags_str}{sep}{k}": v for k, v in log_dict.items()}


class _TrainerStub:
    """Stage flags a LightningModule reads from `self.trainer` (base_model.py:158-176)."""

    def __init__(self):
        self.training = True
        self.validating = self.sanity_checking = self.predicting = self.testing = False
        self.global_step = 0

    def set_stage(self, stage: str) -> None:
        self.training = stage == "training"
        self.validating = stage == "validation"
        self.testing = stage == "test"
        self.predicting = stage == "prediction"
        self.sanity_checking = stage == "sanity_checking"


class BaseModel(_Base):
    def __init__(self, record_gradients: bool = False, save_gradients_interval: int = 25,
                 gradient_record_cap: int = 20, kl_annealing_fn: Optional[KLAnnealingFn] = None,
                 predict_dir: str = "", predict_save_interval: int = 600, initial_save_index: int = -1,
                 use_he_init_weights: bool = True):
        super().__init__()
        self.record_gradients = record_gradients
        self.save_gradients_interval = save_gradients_interval
        self.gradient_record_cap = gradient_record_cap
        self.predict_dir = predict_dir
        self.predict_save_interval = predict_save_interval
        self._curr_save_idx = initial_save_index
        self._running_predictions = []
        self.kl_annealing_fn = kl_annealing_fn or KLAnnealingFn(1.0)
        self._use_he_init_weights = use_he_init_weights
        if not HAVE_LIGHTNING:
            self._trainer = _TrainerStub()
            self._optimizers = None
            self.automatic_optimization = True
            self.logged: dict = {}  # name -> latest value (device scalars are kept as tensors: no host sync)

    # ------------------------------------------------------------------ LightningModule surface (stand-alone mode)
    if not HAVE_LIGHTNING:

        @property
        def trainer(self):
            return self._trainer

        @trainer.setter
        def trainer(self, t):
            self._trainer = t

        def log(self, name, value, **kwargs):
            self.logged[name] = value

        def log_dict(self, d, **kwargs):
            self.logged.update(d)

        def optimizers(self):
            if self._optimizers is None:
                self._optimizers = self.configure_optimizers()
            return self._optimizers

        def manual_backward(self, loss, *args, **kwargs):
            loss.backward(*args, **kwargs)

        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    else:  # ------------------------------------------------------------- under a real LightningModule base

        def optimizers(self, use_pl_optimizer: bool = False):
            """The RAW optimisers, always as a list.  Lightning's default hands out LightningOptimizer wrappers that carry
            a COPY of the optimiser's attribute dictionary and forward step() to the wrapped object: state this build sets
            on the wrapper (the fused clip's max_grad_norm, the cached gradient norm) would never reach the HipAdam that
            steps.  The step is manual optimisation (automatic_optimization = False), which needs nothing of the wrapper."""
            opts = super().optimizers(use_pl_optimizer=False)
            return list(opts) if isinstance(opts, (list, tuple)) else [opts]

        @property
        def logged(self) -> dict:
            """name -> latest logged value, as in stand-alone mode (tests and mmvae_amd.trainer read it; Lightning's own
            logger connector receives the same values through log / log_dict)."""
            return self.__dict__.setdefault("_logged", {})

        def log(self, name, value, **kwargs):
            self.logged[name] = value
            if getattr(self, "_trainer", None) is not None or getattr(self, "_fabric", None) is not None:
                try:
                    super().log(name, value, **kwargs)
                except Exception:  # noqa: BLE001  (outside a Trainer loop Lightning refuses to log: the dict still has it)
                    pass

        def log_dict(self, d, **kwargs):
            self.logged.update(d)
            if getattr(self, "_trainer", None) is not None or getattr(self, "_fabric", None) is not None:
                try:
                    super().log_dict(d, **kwargs)
                except Exception:  # noqa: BLE001
                    pass

    def clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
        """`LightningModule.clip_gradients(optimizer, val, algorithm)` as the reference calls it (cmmvae_model.py:126-129,
        203-209), for BOTH bases.  A HipAdam folds the clip into its fused update (set_clip / set_clip_value: the gradient
        arena was gathered when its norm was logged, and step() reuses it -- clipping `p.grad` afterwards, as Lightning's
        precision plugin would, never reaches the arena); any other optimiser gets torch's clip_grad_norm_ / _value_,
        which is what Lightning does at precision 32."""
        if gradient_clip_val is None:
            return
        algorithm = gradient_clip_algorithm or "norm"
        if algorithm not in ("norm", "value"):
            raise ValueError(f"gradient_clip_algorithm {algorithm!r}: 'norm' or 'value' (config.py:8)")
        if hasattr(optimizer, "set_clip"):  # fused into HipAdam.step()
            if algorithm == "norm":
                optimizer.set_clip(float(gradient_clip_val))
            else:
                optimizer.set_clip_value(float(gradient_clip_val))
        else:
            params = [p for g in optimizer.param_groups for p in g["params"]]
            if algorithm == "norm":
                torch.nn.utils.clip_grad_norm_(params, float(gradient_clip_val))
            else:
                torch.nn.utils.clip_grad_value_(params, float(gradient_clip_val))

    # ------------------------------------------------------------------ shared helpers (same names as the reference)
    def init_weights(self):
        if self._use_he_init_weights:
            he_init_weights(self)

    @property
    def stage_name(self) -> str:
        t = self.trainer
        for flag, name in (("training", "training"), ("validating", "validation"),
                           ("sanity_checking", "sanity_checking"), ("predicting", "prediction"),
                           ("testing", "test")):
            if getattr(t, flag, False):
                return name
        return ""

    def log_gradient_norms(self, optimizer_dict, tag_prefix="grad_norms"):
        """Global L2 norm of each optimiser's gradients (base_model.py:111-123).  The reference syncs once per
        parameter (`.item()`); HipAdam computes the norm in its fused pass and it is logged as a device scalar."""
        for name, optimizer in optimizer_dict.items():
            if isinstance(optimizer, dict):
                self.log_gradient_norms(optimizer, f"{tag_prefix}/{name}")
            elif hasattr(optimizer, "compute_grad_norm"):
                self.log(f"{tag_prefix}/{name}", optimizer.compute_grad_norm())
            else:
                sq = 0.0
                for group in optimizer.param_groups:
                    for p in group["params"]:
                        if p.grad is not None:
                            sq += float(p.grad.data.norm(2)) ** 2
                self.log(f"{tag_prefix}/{name}", sq ** 0.5)

    def auto_log(self, log_dict: dict, tags: Iterable[str] = (), sep: str = "/",
                 key_pos: Literal["first", "last"] = "first", log_sanity_checking: bool = False):
        t = self.trainer
        if t is not None and getattr(t, "sanity_checking", False) and not log_sanity_checking:
            return
        self.log_dict(tag_log_dict(log_dict, tags, sep, key_pos), on_step=bool(getattr(t, "training", True)),
                      on_epoch=True, logger=True)
