"""CMMVAEModel: the trainer plugin of the mirror (reference `cmmvae/models/cmmvae_model.py:15-351`).

Same constructor arguments, step methods, optimiser layout (`configure_optimizers` -> flat list + `optimizer_map`)
and logged scalar names as the reference.  `training_step` has two executions of the same algorithm:

  * engine path (default on device, mmvae_amd.engine.StepEngine): the whole step -- forward, ELBO, adversarial D/G
    phases, backward, clip, Adam -- is a fixed sequence of libmmvae_hip.so launches over pre-allocated buffers,
    captured once per expert in a hipGraph and replayed; nothing is traced, nothing is allocated per step, and no
    value is read back to the host unless logging asks for it.
  * module path (any FCBlock configuration, CPU plumbing): torch autograd sequences the same kernels through
    mmvae_amd.functional.  This is the code below; it follows the reference line by line in behaviour:
    training_step :138-217, grf :59-101, gradient_reversal_domain_classifier :103-136.
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
    def grf(self, hidden_representations, labels: dict, expert_id: str, detach: bool = False):
        """Per adversary: CE(sum) of every head on (detached | gradient-reversed) hidden features, summed (:59-101)."""
        adv_losses = []
        for i, (hidden_rep, adversary) in enumerate(zip(hidden_representations, self.module.adversarials), start=1):
            if detach:
                hidden_rep = hidden_rep.detach()
                loss_tag = f"discriminator_{i}"
            else:
                hidden_rep = GradientReversalFunction.apply(hidden_rep, 1)
                loss_tag = f"generator_{i}"
            encoded = adversary.encoder(hidden_rep)
            head_losses = []
            for condition, label in labels.items():
                disc_loss = _ce_sum(adversary.heads[condition](encoded), label)
                head_losses.append(disc_loss)
                self.auto_log({condition: disc_loss}, tags=[loss_tag, self.stage_name, expert_id, RK.ADV_LOSS],
                              key_pos="last")
            summed = torch.sum(torch.stack(head_losses))
            self.auto_log({"summed": summed}, tags=[loss_tag, self.stage_name, expert_id, RK.ADV_LOSS], key_pos="last")
            adv_losses.append(summed)
        return adv_losses

    @staticmethod
    def adversarial_labels(metadata: pd.DataFrame, device) -> dict:
        """metadata columns -> int64 class-index tensors via the class-level Adversarial.labels maps (:111-115)."""
        return {cond: torch.tensor([mp[v] for v in metadata[cond].values], dtype=torch.int64, device=device)
                for cond, mp in Adversarial.labels.items()}

    def gradient_reversal_domain_classifier(self, hidden_representations, metadata: pd.DataFrame, expert_id: str,
                                            adversarial_optimizers: dict):
        assert len(self.module.adversarials) > 0
        labels = self.adversarial_labels(metadata, hidden_representations[0].device)
        # D phase: every adversary learns on detached features, one backward/clip/step each (:118-131)
        adv_losses = self.grf(hidden_representations, labels, expert_id, detach=True)
        for i, (adv_loss, adv_optimizer) in enumerate(zip(adv_losses, adversarial_optimizers.values()), start=1):
            self.manual_backward(adv_loss)
            self.log_gradient_norms({f"discriminator_{i}": adv_optimizer}, tag_prefix="grad_norms")
            if self.autograd_config.adversarial_gradient_clip:
                self.clip_gradients(adv_optimizer, *self.autograd_config.adversarial_gradient_clip)
            adv_optimizer.step()
            adv_optimizer.zero_grad()
        # G phase: same nets (just updated) behind a gradient-reversal layer (:134-136)
        return self.grf(hidden_representations, labels, expert_id, detach=False)

    # ------------------------------------------------------------------------------------------------ step methods
    def training_step(self, batch, batch_idx: int) -> None:
        x, metadata, expert_id = batch
        metadata["species"] = expert_id
        engine = self._get_engine(x)
        if engine is not None:
            return engine.training_step(x, metadata, expert_id)
        if getattr(self.module.vae.encoder, "elbo_mode", "analytic") != "analytic":
            raise NotImplementedError("elbo_mode='iwae' (the opt-in full-IWAE objective) runs in the captured engine only")

        optims = self.get_optimizers()
        expert_optimizer = optims["experts"][expert_id]
        vae_optimizer = optims["vae"]
        adversarial_optimizers = optims.get("adversarials")
        vae_optimizer.zero_grad()
        expert_optimizer.zero_grad()
        if adversarial_optimizers:
            for optim in adversarial_optimizers.values():
                optim.zero_grad()

        qz, pz, z, xhats, hidden_representations = self.module(x=x, metadata=metadata, expert_id=expert_id)
        if x.layout == torch.sparse_csr:
            x = backend.to_dense(x)
        main_loss_dict = self.module.vae.elbo(qz, pz, x, xhats[expert_id], self.kl_annealing_fn.kl_weight)
        main_loss_dict["Mean"], main_loss_dict["Variance"] = self._posterior_stats(qz)
        total_loss = main_loss_dict[RK.LOSS]

        adv_losses = None
        if len(self.module.adversarials) > 0:
            adv_losses = self.gradient_reversal_domain_classifier(hidden_representations, metadata, expert_id,
                                                                  adversarial_optimizers)
        if adv_losses:
            for adv_loss in adv_losses:
                total_loss = total_loss + adv_loss * self.adv_weight

        self.manual_backward(total_loss)
        main_loss_dict[RK.LOSS] = total_loss
        self.log_gradient_norms({"vae": vae_optimizer, f"expert_{expert_id}": expert_optimizer},
                                tag_prefix="grad_norms")
        if adversarial_optimizers:
            for key, optim in adversarial_optimizers.items():
                self.log_gradient_norms({f"generator_{key}": optim}, tag_prefix="grad_norms")
        if self.autograd_config.vae_gradient_clip:
            self.clip_gradients(vae_optimizer, *self.autograd_config.vae_gradient_clip)
        if self.autograd_config.expert_gradient_clip:
            self.clip_gradients(expert_optimizer, *self.autograd_config.expert_gradient_clip)
        vae_optimizer.step()
        expert_optimizer.step()
        self.kl_annealing_fn.step()
        self.auto_log(main_loss_dict, tags=[self.stage_name, expert_id])

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
        optimizers = self.optimizers()
        if zero_all:
            for optim in optimizers:
                optim.zero_grad()

        def resolve(mapping):
            if isinstance(mapping, dict):
                return {k: resolve(v) for k, v in mapping.items()}
            return optimizers[mapping]

        return resolve(self.optimizer_map)

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


def convert_to_flat_list_and_map(d: dict, flat_list: Optional[list] = None) -> dict:
    """Nested dict of optimisers -> same-shaped dict of indices into `flat_list` (appended in traversal order)."""
    if flat_list is None:
        flat_list = []
    mapping = {}
    for key, value in d.items():
        if isinstance(value, dict):
            mapping[key] = convert_to_flat_list_and_map(value, flat_list)
        else:
            flat_list.append(value)
            mapping[key] = len(flat_list) - 1
    return mapping
