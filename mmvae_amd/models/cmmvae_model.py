"""CMMVAEModel: the trainer plugin of the mirror (reference `cmmvae/models/cmmvae_model.py:15-351`).

Same constructor arguments, step methods, optimiser layout (`configure_optimizers` -> flat list + `optimizer_map`)
and logged scalar names as the reference.  `training_step` has two executions of the same algorithm:

  * engine path (default on device, mmvae_amd.engine.StepEngine): the whole step -- forward, ELBO, adversarial D/G
    phases, backward, clip, Adam -- is a fixed sequence of libmmvae_hip.so launches over pre-allocated buffers,
    captured once per expert in a hipGraph and replayed; nothing is traced, nothing is allocated per step, and no
    value is read back to the host unless logging asks for it.
  * module path (any FCBlock configuration, CPU plumbing): torch autograd sequences the same kernels through
    mmvae_amd.functional.  This is the code below, written as the sections of the engine's program; behaviour pinned
    against the reference's training_step :138-217, grf :59-101, gradient_reversal_domain_classifier :103-136 by the
    golden vectors (tests/test_step_gpu.py, tests/test_mirror_cpu.py).
"""
from __future__ import annotations

from typing import Optional

import pandas as pd
import torch
import torch.nn as nn

from .. import backend
from .. import functional as HF
from ..config import AutogradConfig
from ..constants import REGISTRY_KEYS as RK
from ..modules import CMMVAE
from ..modules.base.components import Adversarial, GradientReversalFunction
from ..optim import HipAdam
from .base_model import BaseModel


def _ce_sum(logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    if logits.is_cuda:
        return HF.CrossEntropySumFn.apply(logits.contiguous(), labels)
    return nn.functional.cross_entropy(logits, labels, reduction="sum")  # CPU plumbing


class CMMVAEModel(BaseModel):
    def __init__(self, module: CMMVAE, adv_weight: Optional[float] = None,
                 autograd_config: Optional[AutogradConfig] = None, *args, use_engine: bool = True, **kwargs):
        super().__init__(*args, **kwargs)
        self.module = module
        self.automatic_optimization = False  # manual optimisation (cmmvae_model.py:50-52)
        self.adversarial_criterion = nn.CrossEntropyLoss(reduction="sum")
        self.init_weights()
        self.adv_weight = adv_weight if adv_weight else 1.0  # NB: 0 becomes 1.0, as in the reference (:56)
        self.autograd_config = autograd_config or AutogradConfig()
        self.use_engine = use_engine
        self._engine = None
        self.optimizer_map = None

    # ------------------------------------------------------------------------------------------ adversarial phases
    # The module path states the step the way DESIGN.md section 2 and the engine's program do: sections
    # (forward + ELBO) -> (discriminator round: learn on detached features) -> (generator round: same nets behind the
    # gradient-reversal layer, added to the ELBO) -> backward -> per-optimiser (norm, clip, Adam).  The three public
    # methods of the reference's surface (grf, gradient_reversal_domain_classifier, training_step) are entry points
    # into these sections.
    _ROUNDS = {True: "discriminator", False: "generator"}

    def _adversary_round(self, features, labels: dict, expert_id: str, learn: bool) -> list:
        """One loss per (feature, adversary) pair: the sum over the adversary's heads of CE(sum) against the cells'
        class indices.  learn=True feeds detached features (the adversary's own update); learn=False feeds them
        through GradientReversalFunction(alpha = 1), so the loss pushes the encoder the other way.  Logged as
        `{round}_{n}/{stage}/{expert}/adversarial_loss/{condition|summed}` (cmmvae_model.py:87-98)."""
        totals = []
        for number, (feature, adversary) in enumerate(zip(features, self.module.adversarials), start=1):
            source = feature.detach() if learn else GradientReversalFunction.apply(feature, 1)
            code = adversary.encoder(source)
            per_head = {condition: _ce_sum(adversary.heads[condition](code), classes)
                        for condition, classes in labels.items()}
            per_head["summed"] = torch.stack(tuple(per_head.values())).sum()
            self.auto_log(per_head, tags=[f"{self._ROUNDS[learn]}_{number}", self.stage_name, expert_id, RK.ADV_LOSS],
                          key_pos="last")
            totals.append(per_head["summed"])
        return totals

    def grf(self, hidden_representations, labels: dict, expert_id: str, detach: bool = False):
        """Reference entry point (cmmvae_model.py:59-101): the round's per-adversary losses."""
        return self._adversary_round(hidden_representations, labels, expert_id, learn=detach)

    @staticmethod
    def adversarial_labels(metadata: pd.DataFrame, device) -> dict:
        """Metadata columns -> int64 class-index tensors through the class-level Adversarial.labels tables
        (cmmvae_model.py:111-115), one vectorised lookup per condition."""
        out = {}
        for condition, table in Adversarial.labels.items():
            classes = metadata[condition].map(table)
            if classes.isna().any():
                raise KeyError(f"{condition}: {metadata[condition][classes.isna()].iloc[0]!r} is not a known class")
            out[condition] = torch.as_tensor(classes.to_numpy(dtype="int64"), device=device)
        return out

    def _norm_clip(self, name: str, optimizer, rule) -> None:
        """Log the optimiser's gradient norm as grad_norms/{name}, then apply its clipping rule (HipAdam folds both into
        its fused update)."""
        self.log_gradient_norms({name: optimizer}, tag_prefix="grad_norms")
        if rule:
            self.clip_gradients(optimizer, *rule)

    def gradient_reversal_domain_classifier(self, hidden_representations, metadata: pd.DataFrame, expert_id: str,
                                            adversarial_optimizers: dict):
        """Discriminator round (each adversary: backward, norm, clip, Adam, gradients cleared) followed by the generator
        round on the just-updated adversaries; returns the generator losses (cmmvae_model.py:103-136)."""
        assert len(self.module.adversarials) > 0
        labels = self.adversarial_labels(metadata, hidden_representations[0].device)
        learned = self._adversary_round(hidden_representations, labels, expert_id, learn=True)
        for number, (loss, optimizer) in enumerate(zip(learned, adversarial_optimizers.values()), start=1):
            self.manual_backward(loss)
            self._norm_clip(f"discriminator_{number}", optimizer, self.autograd_config.adversarial_gradient_clip)
            optimizer.step()
            optimizer.zero_grad()
        return self._adversary_round(hidden_representations, labels, expert_id, learn=False)

    # ------------------------------------------------------------------------------------------------ step methods
    def training_step(self, batch, batch_idx: int) -> None:
        """One training step on one expert's batch (cmmvae_model.py:138-217).  Device batches run the captured engine
        program; what follows is the autograd statement of the same program."""
        x, metadata, expert_id = batch
        metadata["species"] = expert_id
        engine = self._get_engine(x)
        hint, self._next_hint = getattr(self, "_next_hint", None), None
        if engine is not None:
            return engine.training_step(x, metadata, expert_id, next_batch=hint)
        self._flush_engine()  # (a step on the module path behind engine steps: their deferred updates land first)
        if getattr(self.module.vae.encoder, "elbo_mode", "analytic") != "analytic":
            raise NotImplementedError("elbo_mode='iwae' (the opt-in full-IWAE objective) runs in the captured engine only")

        live = self.get_optimizers()
        adversary_optims = live.get("adversarials") or {}
        stepped = {"vae": (live["vae"], self.autograd_config.vae_gradient_clip),
                   f"expert_{expert_id}": (live["experts"][expert_id], self.autograd_config.expert_gradient_clip)}
        for optimizer in [o for o, _ in stepped.values()] + list(adversary_optims.values()):
            optimizer.zero_grad()

        # forward + ELBO
        qz, pz, z, xhats, features = self.module(x=x, metadata=metadata, expert_id=expert_id)
        dense_x = backend.to_dense(x) if x.layout == torch.sparse_csr else x
        terms = self.module.vae.elbo(qz, pz, dense_x, xhats[expert_id], self.kl_annealing_fn.kl_weight)
        terms["Mean"], terms["Variance"] = self._posterior_stats(qz)
        objective = terms[RK.LOSS]
        # adversarial rounds: the generator losses join the objective with weight adv_weight
        if len(self.module.adversarials) > 0:
            for loss in self.gradient_reversal_domain_classifier(features, metadata, expert_id, adversary_optims):
                objective = objective + self.adv_weight * loss
        terms[RK.LOSS] = objective

        self.manual_backward(objective)
        for name, (optimizer, rule) in stepped.items():  # norms of both are logged before either is clipped
            self.log_gradient_norms({name: optimizer}, tag_prefix="grad_norms")
        for key, optimizer in adversary_optims.items():  # gradients the main backward left on the adversaries: logged,
            self.log_gradient_norms({f"generator_{key}": optimizer}, tag_prefix="grad_norms")  # never stepped
        for optimizer, rule in stepped.values():
            if rule:
                self.clip_gradients(optimizer, *rule)
        for optimizer, _ in stepped.values():
            optimizer.step()
        self.kl_annealing_fn.step()
        self.auto_log(terms, tags=[self.stage_name, expert_id])

    @staticmethod
    def _posterior_stats(qz):
        """qz.mean.mean(), qz.variance.mean() (:170-171); from the fused kernel's row sums when available."""
        cache = getattr(qz, "_mmvae", None)
        if cache is not None:
            from .. import ops

            stat = cache["stat_row"]
            n = float(qz.loc.numel())
            return ops.sum_f32(stat[0]).reshape(()) / n, ops.sum_f32(stat[1]).reshape(()) / n
        return qz.mean.mean(), qz.variance.mean()

    def _flush_engine(self):
        if self._engine:
            self._engine.flush()

    def hint_next_batch(self, batch) -> None:
        """Optional: tell the model which batch the NEXT training_step will receive -- `(x, metadata, expert_id)` as the
        loader yields it, or None.  The step engine then computes that step's first forward product as filler work beside
        this step's latency-bound forward chain (software pipelining across steps, mmvae_amd.engine: same kernels on the
        same operands, bit-identical results; a hint that turns out wrong costs one wasted product).  The hinted tensor
        must stay unmodified until its step.  mmvae_amd.trainer.Trainer and mmvae_amd.data.Lookahead look one batch ahead;
        the reference's loop (no look-ahead) simply never calls this."""
        if batch is None:
            self._next_hint = None
            return
        x, _, expert_id = batch
        self._next_hint = (x, expert_id)

    def gather_optimizer_state(self) -> None:
        """Under data parallelism the engine updates each expert arena sharded (this rank's slice of the Adam moments
        only): gather them.  COLLECTIVE -- call it on EVERY rank before a checkpoint that only one rank writes
        (`if rank == 0: torch.save(opt.state_dict())`); HipAdam.state_dict() itself is collective too once an
        optimiser has taken sharded steps.  No-op on one rank."""
        self._flush_engine()
        for opt in self.optimizers():
            opt.sync_sharded_state()

    def state_dict(self, *args, **kwargs):
        self._flush_engine()  # deferred expert updates must have landed before parameters are read
        return super().state_dict(*args, **kwargs)

    def validation_step(self, batch, batch_idx: int = 0):
        """Eval-mode forward + ELBO, logged under the current stage (:219-248)."""
        self._flush_engine()
        x, metadata, expert_id = batch
        engine = None if self.module.training else self._get_engine(x)
        if engine is not None:  # forward-only captured program (eval-mode BatchNorm, no dropout, one rsample)
            loss_dict = engine.validation_step(x, metadata, expert_id)
        else:
            qz, pz, z, xhats, hidden_representations = self.module(x, metadata, expert_id)
            if x.layout == torch.sparse_csr:
                x = backend.to_dense(x)
            loss_dict = self.module.vae.elbo(qz, pz, x, xhats[expert_id], self.kl_annealing_fn.kl_weight)
        self.auto_log(loss_dict, tags=[self.stage_name, expert_id])
        if getattr(self.trainer, "validating", False):
            self.log("val_loss", loss_dict[RK.LOSS], logger=False, on_epoch=True)
        return loss_dict

    test_step = validation_step

    def predict_step(self, batch, batch_idx: int = 0):
        self._flush_engine()
        x, metadata, species = batch
        engine = None if self.module.training else self._get_engine(x)
        if engine is not None:
            z = engine.latent_embeddings(x, metadata, species)
            metadata["species"] = species
            return {RK.Z: (z, metadata)}
        return self.module.get_latent_embeddings(x, metadata, species)

    # -------------------------------------------------------------------------------------------------- optimisers
    def get_optimizers(self, zero_all: bool = False):
        """The optimisers in the shape of `optimizer_map`: {"experts": {id: opt}, "vae": opt, "adversarials": {n: opt}}
        (cmmvae_model.py:267-297)."""
        flat = self.optimizers()
        if zero_all:
            for optimizer in flat:
                optimizer.zero_grad()
        return _map_leaves(self.optimizer_map, lambda index: flat[index])

    def configure_optimizers(self, optim_cls="Adam"):
        """One Adam(lr=5e-3, weight_decay=1e-6) per expert, one for the VAE, one per adversary, as a flat list plus
        `optimizer_map` (:299-351).  "Adam" builds the fused flat-arena HipAdam; torch.optim.AdamW for "AdamW"."""
        def make(params, pack=None):
            if optim_cls == "Adam":
                return HipAdam(params, lr=5e-3, weight_decay=1e-6, pack=pack)
            return torch.optim.AdamW(params, lr=5e-3, weight_decay=1e-6)

        def head_packs(adv):
            """Single-Linear heads: their weights back to back, then their biases (one GEMM over all heads)."""
            lins = []
            for head in adv.heads.values():
                if len(head.fc_layers) != 1 or [n for n, _ in head.fc_layers[0].named_children()] != ["lin"]:
                    return None
                lins.append(head.fc_layers[0].lin)
            if len(lins) < 2 or any(l.bias is None for l in lins):
                return None
            n_e = lins[0].in_features
            if any(l.in_features != n_e for l in lins):
                return None
            # class counts padded to multiples of 4 rows: the packed matrix keeps 16-byte-regular shapes (2, 273 classes)
            return [([l.weight for l in lins], 4 * n_e), ([l.bias for l in lins], 4)]

        optim_dict = {"experts": {eid: make(m.parameters()) for eid, m in self.module.experts.items()},
                      "vae": make(self.module.vae.parameters())}
        if getattr(self.module.vae, "conditionals", None) is not None and optim_cls == "Adam":
            # condition blocks absent from a rank's batch have no gradient there: under data parallelism the set of
            # parameters that step is the union over the ranks (HipAdam._allreduce)
            optim_dict["vae"].sparse_presence = True
        if len(self.module.adversarials) > 0:
            optim_dict["adversarials"] = {i: make(m.parameters(), head_packs(m) if optim_cls == "Adam" else None)
                                          for i, m in enumerate(self.module.adversarials, start=1)}
        optimizers: list = []
        self.optimizer_map = convert_to_flat_list_and_map(optim_dict, optimizers)
        return optimizers

    # ------------------------------------------------------------------------------------------------------ engine
    def _get_engine(self, x):
        if not self.use_engine or not x.is_cuda:
            return None
        if self._engine is None:
            from ..engine import StepEngine

            self._engine = StepEngine.try_build(self) or False
        return self._engine or None


def _map_leaves(tree, fn):
    """Same-shaped copy of a nested dict with fn applied to every leaf."""
    return {k: _map_leaves(v, fn) if isinstance(v, dict) else fn(v) for k, v in tree.items()}


def convert_to_flat_list_and_map(d: dict, flat_list: Optional[list] = None) -> dict:
    """Nested dict of optimisers -> same-shaped dict of positions in `flat_list`, which receives the optimisers in
    traversal order (the list Lightning's `configure_optimizers` wants; cmmvae_model.py:326-351)."""
    flat_list = [] if flat_list is None else flat_list

    def place(optimizer):
        flat_list.append(optimizer)
        return len(flat_list) - 1

    return _map_leaves(d, place)
