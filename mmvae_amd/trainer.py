"""Minimal fit / validate / predict loop for CMMVAEModel when Lightning is not installed.

Stands in for the part of `lightning.pytorch.Trainer` the reference's trainer YAML exercises (configs/trainer/config.yaml:
max_epochs / max_steps / limit_*_batches / check_val_every_n_epoch / num_sanity_val_steps) -- manual optimisation, so
the loop only sets the stage flags and calls the step methods.  Batches are `(x, metadata, expert_id)` tuples as
produced by the reference's datamodules (data/local/cellxgene_manager.py:76-88); `MultiModalBatches` below mirrors
`MultiModalDataLoader` (data/local/multi_modal_loader.py:36-68) with a seeded, rank-synchronous modality choice.
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, Iterator, List, Optional

import torch
import yaml


class MultiModalBatches:
    """Interleaves per-modality batch iterables: at every step one modality that still has batches is drawn (seeded
    `random.Random`, identical on every rank so that data-parallel ranks train the same expert)."""

    def __init__(self, loaders: Dict[str, Iterable], seed: int = 0, round_robin: bool = False):
        self.loaders = loaders
        self.seed = seed
        self.round_robin = round_robin
        self.epoch = 0

    def __iter__(self) -> Iterator:
        rng = random.Random(self.seed + self.epoch)
        self.epoch += 1
        its = {k: iter(v) for k, v in self.loaders.items()}
        order: List[str] = list(its.keys())
        i = 0
        while its:
            key = order[i % len(order)] if self.round_robin else rng.choice(sorted(its.keys()))
            i += 1
            if key not in its:
                continue
            try:
                yield next(its[key])
            except StopIteration:
                del its[key]
                order = [k for k in order if k in its]


class Lookahead:
    """Wraps a batch iterable and tells the model, before each batch is handed out, which batch comes AFTER it
    (`CMMVAEModel.hint_next_batch`): the step engine then computes the next step's first forward product beside the
    current step's forward chain (software pipelining across steps).  Works around any loop that pulls batches one at a
    time -- this module's Trainer, or a Lightning Trainer given `Lookahead(dataloader, model)` as its train dataloader."""

    def __init__(self, batches: Iterable, model):
        self.batches, self.model = batches, model

    def __len__(self):
        return len(self.batches)

    def __iter__(self) -> Iterator:
        it = iter(self.batches)
        try:
            cur = next(it)
        except StopIteration:
            return
        for nxt in it:
            self.model.hint_next_batch(nxt)
            yield cur
            cur = nxt
        self.model.hint_next_batch(None)
        yield cur


class Trainer:
    def __init__(self, max_epochs: int = 10, max_steps: int = -1, limit_train_batches: Optional[int] = None,
                 limit_val_batches: Optional[int] = None, check_val_every_n_epoch: int = 1,
                 num_sanity_val_steps: int = 0, **unused):
        self.max_epochs = max_epochs
        self.max_steps = max_steps
        self.limit_train_batches = limit_train_batches
        self.limit_val_batches = limit_val_batches
        self.check_val_every_n_epoch = check_val_every_n_epoch
        self.num_sanity_val_steps = num_sanity_val_steps
        self.global_step = 0
        self.history: List[dict] = []

    @classmethod
    def from_yaml(cls, path: str) -> "Trainer":
        with open(path) as f:
            cfg = yaml.safe_load(f) or {}
        keys = ("max_epochs", "max_steps", "limit_train_batches", "limit_val_batches", "check_val_every_n_epoch",
                "num_sanity_val_steps")
        return cls(**{k: cfg[k] for k in keys if cfg.get(k) is not None})

    def _snapshot(self, model) -> dict:
        return {k: (float(v.detach()) if torch.is_tensor(v) else v) for k, v in model.logged.items()}

    def fit(self, model, train_batches: Iterable, val_batches: Optional[Iterable] = None):
        model.optimizers()
        stub = model.trainer
        for epoch in range(self.max_epochs):
            model.train()
            stub.set_stage("training")
            for i, batch in enumerate(Lookahead(train_batches, model) if hasattr(model, "hint_next_batch") else train_batches):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                model.training_step(batch, i)
                self.global_step += 1
                stub.global_step = self.global_step
                if 0 < self.max_steps <= self.global_step:
                    break
            self.history.append({"epoch": epoch, "stage": "training", **self._snapshot(model)})
            if val_batches is not None and (epoch + 1) % self.check_val_every_n_epoch == 0:
                self.validate(model, val_batches)
            if 0 < self.max_steps <= self.global_step:
                break
        return self.history

    @torch.no_grad()
    def validate(self, model, batches: Iterable):
        model.eval()
        model.trainer.set_stage("validation")
        for i, batch in enumerate(batches):
            if self.limit_val_batches is not None and i >= self.limit_val_batches:
                break
            model.validation_step(batch, i)
        self.history.append({"stage": "validation", **self._snapshot(model)})

    @torch.no_grad()
    def predict(self, model, batches: Iterable, writer=None) -> list:
        """`predict_step` over the batches; with a `mmvae_amd.predictions.PredictionWriter` every batch is appended
        to `predictions.h5` as it is produced (the reference's write_interval="batch" callback) and nothing is kept."""
        model.eval()
        model.trainer.set_stage("prediction")
        if writer is None:
            return [model.predict_step(batch, i) for i, batch in enumerate(batches)]
        writer.on_predict_start(self, model)
        for i, batch in enumerate(batches):
            writer.write_on_batch_end(self, model, model.predict_step(batch, i), None, batch, i, 0)
        writer.on_predict_epoch_end(self, model)
        return []
