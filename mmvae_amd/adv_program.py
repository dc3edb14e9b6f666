"""The adversaries of a training step as two launches per phase (include/mmvae_hip.h, "Row-owner adversary passes").

Reference: `CMMVAEModel.gradient_reversal_domain_classifier` / `grf` (models/cmmvae_model.py:59-136) over
`Adversarial` (modules/base/components.py:638-674) and `GradientReversalFunction` (:879-899).  The discriminator phase
of EVERY adversary reads detached features and steps its own optimiser, the generator phase of every adversary then
runs on the updated weights -- the adversaries are independent of each other inside a phase, so one launch serves all:

    pass (discriminator)  ->  weight gradients + norm / clip / step count  ->  Adam (all adversaries)
    pass (generator, gradient reversal)  ->  weight gradients + norm (logged only, cmmvae_model.py:196-200)

This module only builds the device job tables and enqueues the launches; torch is memory and streams.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch

from . import _lib


def _p(t):
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


@dataclass
class AdvLayer:
    """One Linear (+ ReLU) (+ Dropout) layer of an adversary's encoder."""
    W: torch.Tensor          # [n_out, n_in]
    b: Optional[torch.Tensor]
    gW: torch.Tensor         # gradient-arena views
    gb: Optional[torch.Tensor]
    relu: bool
    p_drop: float = 0.0
    masks: dict = field(default_factory=dict)  # phase -> uint8 [B, n_out] keep mask (p_drop > 0)


@dataclass
class AdvNet:
    """One adversary: the hidden representation it reads, its encoder layers, its stacked heads and its optimiser."""
    x: torch.Tensor          # [>= B, width[0]] rows 0..B-1 are read
    ldx: int
    layers: List[AdvLayer]
    Wh: torch.Tensor         # [Ct, n_e]: the heads' weights stacked (rows of head h start at col[h])
    bh: Optional[torch.Tensor]
    gWh: torch.Tensor
    gbh: Optional[torch.Tensor]
    col: List[int]
    classes: List[int]
    opt: object = None       # HipAdam (arena, state_dev, param_groups) or None


def supported(lib, net: AdvNet, B: int, splits: int = 1, tiles: int = 0):
    """(head tile count, LDS bytes, partial floats) of mmvae_adv_pass_plan -- at `tiles` head tiles when the launch is
    shared with a wider adversary -- or None when the shape is outside the kernels."""
    if not (1 <= len(net.layers) <= _lib.ADV_MAX_LAYERS and 1 <= len(net.classes) <= _lib.ADV_MAX_HEADS):
        return None
    job = _lib.AdvJob()
    job.n_layers, job.H, job.Ct, job.B = len(net.layers), len(net.classes), net.Wh.shape[0], B
    job.width[0] = net.layers[0].W.shape[1]
    for l, lay in enumerate(net.layers):
        job.width[l + 1] = lay.W.shape[0]
    for h, (c, n) in enumerate(zip(net.col, net.classes)):
        job.col[h], job.classes[h] = c, n
    nt, lds, pf = C.c_int(tiles), C.c_size_t(0), C.c_int64(0)
    if lib.mmvae_adv_pass_plan(C.addressof(job), splits, C.byref(nt), C.byref(lds), C.byref(pf), None) != _lib.OK:
        return None
    return nt.value, lds.value, pf.value


class AdvProgram:
    """Device tables + launches of the adversarial phases of one plan.

    `alloc(name, shape, dtype)` hands out zero-initialised device buffers (the engine's pool); `phases`: for each phase
    name a dict(gscale, reverse (write -d loss / d x), loss_each[i] / loss_total[i] (device addresses per adversary),
    total_loss (address or None), total_scale, opt flags / max_norm / norm_out[i]).
    """

    def __init__(self, lib, alloc: Callable, nets: List[AdvNet], B: int, labels: torch.Tensor, device, tag: str = "adv",
                 splits: Optional[int] = None):
        self.lib, self.nets, self.B, self.labels, self.device = lib, nets, B, labels, device
        self.H = len(nets[0].classes)
        n_rt = (B + 15) // 16
        tiles = min(sum((((c + 3) // 4 * 4) + 15) // 16 for c in n.classes) for n in nets)
        # class splits per cell tile: enough workgroups to cover the chip (results depend on the split count only
        # through the order in which the partial sums are merged)
        self.splits = max(1, min(8, tiles, 256 // max(1, n_rt * len(nets)))) if splits is None else int(splits)
        plans = [supported(lib, n, B, self.splits) for n in nets]
        if any(p is None for p in plans):
            raise _lib.HipLibraryError("adv_program: unsupported adversary shape (call supported() first)")
        self.net = max(p[0] for p in plans)  # one launch: every job runs at the widest job's head tile count
        plans = [supported(lib, n, B, self.splits, self.net) for n in nets]
        if any(p is None for p in plans):
            raise _lib.HipLibraryError("adv_program: unsupported adversary shape at the launch's tile count")
        self.lds = max(p[1] for p in plans)
        self.keep: list = []
        self.bufs = []
        for i, (n, plan) in enumerate(zip(nets, plans), start=1):
            widths = [n.layers[0].W.shape[1]] + [l.W.shape[0] for l in n.layers]
            b = dict(
                act=[alloc(f"{tag}{i}.act{l}", (B, w), torch.float32) for l, w in enumerate(widths[1:])],
                dz=[alloc(f"{tag}{i}.dz{l}", (B, w), torch.float32) for l, w in enumerate(widths[1:])],
                logits=alloc(f"{tag}{i}.logits_all", (B, n.Wh.shape[0]), torch.float32),
                lse=alloc(f"{tag}{i}.lse", (self.H, B), torch.float32),
                loss_rows=alloc(f"{tag}{i}.ce_rows", (self.H, B), torch.float32),
                gx=alloc(f"{tag}{i}.gh", (B, widths[0]), torch.float32),
                partials=alloc(f"{tag}{i}.partials.{self.splits}.{self.net}", (plan[2],), torch.float32),
            )
            self.bufs.append(b)
        self.launch_tickets = alloc(f"{tag}.launch_tickets", (4,), torch.int32)
        self.phase_tables: dict = {}

    def _host_job(self, n: AdvNet, b: Optional[dict], phase: Optional[str] = None, cfg: Optional[dict] = None,
                  i: int = 0) -> _lib.AdvJob:
        job = _lib.AdvJob()
        job.x, job.ldx = _p(n.x), n.ldx
        job.n_layers, job.H, job.Ct, job.B = len(n.layers), len(n.classes), n.Wh.shape[0], self.B
        job.width[0] = n.layers[0].W.shape[1]
        for l, lay in enumerate(n.layers):
            job.width[l + 1] = lay.W.shape[0]
            job.W[l], job.b[l] = _p(lay.W), _p(lay.b)
            job.relu[l] = int(lay.relu)
            job.p_drop[l] = float(lay.p_drop)
            if phase is not None:
                job.mask[l] = _p(lay.masks.get(phase))
        for h, (c, k) in enumerate(zip(n.col, n.classes)):
            job.col[h], job.classes[h] = c, k
        job.Wh, job.bh, job.labels = _p(n.Wh), _p(n.bh), _p(self.labels)
        if b is not None:
            for l in range(len(n.layers)):
                job.act[l], job.dz[l] = _p(b["act"][l]), _p(b["dz"][l])
            job.logits, job.lse, job.loss_rows = _p(b["logits"]), _p(b["lse"]), _p(b["loss_rows"])
            job.partials = _p(b["partials"])
            job.gx = _p(b["gx"]) if cfg["reverse"] else None
            job.loss_each, job.loss_total = cfg["loss_each"][i], cfg["loss_total"][i]
            job.total_loss, job.total_scale = cfg.get("total_loss"), float(cfg.get("total_scale", 0.0))
            job.gscale = float(cfg["gscale"])
        return job

    def _to_device(self, arr) -> torch.Tensor:
        t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.keep.append(t)
        return t

    def build_phase(self, phase: str, cfg: dict) -> None:
        """Device tables of one phase.  cfg: see the class docstring; cfg["opts"]: None (no norm bookkeeping in the
        weight-gradient launch: a gradient exchange comes first) or per adversary dict(flags, max_norm, norm_out)."""
        lib, nets = self.lib, self.nets
        jobs = (_lib.AdvJob * len(nets))(*[self._host_job(n, b, phase, cfg, i)
                                           for i, (n, b) in enumerate(zip(nets, self.bufs))])
        fast = 1
        for j in jobs:  # the 16-byte loaders need every job's rows aligned (mmvae_adv_pass_plan decides)
            f, nt = C.c_int(0), C.c_int(self.net)
            _lib.check(lib.mmvae_adv_pass_plan(C.addressof(j), self.splits, C.byref(nt), None, None, C.byref(f)),
                       "mmvae_adv_pass_plan")
            fast &= f.value
        dw = []
        for i, (n, b) in enumerate(zip(nets, self.bufs)):
            inputs = [(n.x, n.ldx)] + [(a, a.shape[1]) for a in b["act"]]
            for l, lay in enumerate(n.layers):
                j = _lib.AdvDwJob()
                j.dz, j.ld_dz = _p(b["dz"][l]), lay.W.shape[0]
                j.inp, j.ld_inp = _p(inputs[l][0]), inputs[l][1]
                j.gW, j.gb = _p(lay.gW), _p(lay.gb)
                j.M, j.N, j.B, j.H, j.opt = lay.W.shape[0], lay.W.shape[1], self.B, 0, i
                j.gscale = 1.0
                dw.append(j)
            j = _lib.AdvDwJob()
            j.dz, j.ld_dz = _p(b["logits"]), n.Wh.shape[0]
            j.inp, j.ld_inp = _p(inputs[-1][0]), inputs[-1][1]
            j.gW, j.gb = _p(n.gWh), _p(n.gbh)
            j.lse, j.labels, j.gscale = _p(b["lse"]), _p(self.labels), float(cfg["gscale"])
            j.M, j.N, j.B, j.H, j.opt = n.Wh.shape[0], n.Wh.shape[1], self.B, len(n.classes), i
            for h, (c, k) in enumerate(zip(n.col, n.classes)):
                j.col[h], j.classes[h] = c, k
            dw.append(j)
        dw_arr = (_lib.AdvDwJob * len(dw))(*dw)
        total, dw_fast = C.c_int(0), C.c_int(0)
        _lib.check(lib.mmvae_adv_dw_prepare(len(dw), C.addressof(dw_arr), C.byref(total), C.byref(dw_fast)),
                   "mmvae_adv_dw_prepare")
        opts_dev, n_opts = None, 0
        if cfg.get("opts") is not None:
            arr = []
            for n, o in zip(nets, cfg["opts"]):
                g = n.opt.param_groups[0]
                a = _lib.AdvOpt()
                a.state, a.norm_out = _p(n.opt.state_dev), o.get("norm_out")
                a.max_norm, a.grad_scale = float(o["max_norm"]), float(cfg.get("grad_scale", 1.0))
                a.beta1, a.beta2 = g["betas"]
                a.flags = int(o["flags"])
                arr.append(a)
            opts_dev, n_opts = self._to_device((_lib.AdvOpt * len(arr))(*arr)), len(arr)
        self.phase_tables[phase] = dict(
            jobs=self._to_device(jobs), fast=fast, dw=self._to_device(dw_arr), dw_fast=dw_fast.value, n_dw=len(dw), blocks=total.value, opts=opts_dev,
            n_opts=n_opts)
        self.dw_partials = getattr(self, "dw_partials", None)
        if self.dw_partials is None or self.dw_partials.numel() < total.value:
            self.dw_partials = torch.zeros(total.value, dtype=torch.float32, device=self.device)

    def build_adam(self, grad_scale: float = 1.0) -> None:
        arr = []
        for n in self.nets:
            a, g = n.opt.arena, n.opt.param_groups[0]
            e = _lib.AdamArena()
            e.p, e.g, e.m, e.v, e.state, e.n = _p(a.data), _p(a.grad), _p(a.exp_avg), _p(a.exp_avg_sq), _p(n.opt.state_dev), a.numel
            e.lr, (e.beta1, e.beta2), e.eps, e.weight_decay = g["lr"], g["betas"], g["eps"], g["weight_decay"]
            e.grad_scale = float(grad_scale)
            arr.append(e)
        self.adam_table = self._to_device((_lib.AdamArena * len(arr))(*arr))
        self.adam_max_n = max(n.opt.arena.numel for n in self.nets)

    # ------------------------------------------------------------------------------------------------ launches
    def launch_pass(self, phase: str) -> None:
        t = self.phase_tables[phase]
        _lib.check(self.lib.mmvae_adv_pass_f32(len(self.nets), t["jobs"].data_ptr(), self.B, self.splits, self.net, t["fast"],
                                               self.lds, _stream()), "mmvae_adv_pass_f32")

    def launch_dw(self, phase: str) -> None:
        t = self.phase_tables[phase]
        _lib.check(self.lib.mmvae_adv_dw_f32(t["n_dw"], t["dw"].data_ptr(), t["blocks"], t["n_opts"], _p(t["opts"]),
                                             self.dw_partials.data_ptr(), self.launch_tickets.data_ptr(), len(self.nets),
                                             t["jobs"].data_ptr(), t["dw_fast"], _stream()), "mmvae_adv_dw_f32")

    def launch_adam(self) -> None:
        _lib.check(self.lib.mmvae_adam_step_multi(len(self.nets), self.adam_table.data_ptr(), self.adam_max_n, _stream()),
                   "mmvae_adam_step_multi")
