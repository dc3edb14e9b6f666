"""Data parallelism over the batch axis: one process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm, over
xGMI).  The reference has no distributed code (`strategy: auto`, never exercised); this is the build's own.

Cells are independent, so the only exchange step is the gradient all-reduce.  Because every optimiser owns ONE
contiguous gradient arena (mmvae_amd.optim.ParamArena), the all-reduce is a few large contiguous buckets instead of
one call per tensor.  Semantics = Lightning DDP: gradients are averaged over ranks (sum all-reduce, then the 1/world
factor is folded into the fused clip+Adam kernel as `grad_scale`, so no extra pass over the gradients exists).
BatchNorm statistics stay per-rank (reference `sync_batchnorm: false`, configs/trainer/config.yaml:77).
All ranks must train the same expert on a step (rank-synchronous schedule) so the same arenas are live everywhere.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist

BUCKET_BYTES = 64 << 20  # xGMI ring all-reduce is per-link bound: few, large messages
# Diagnostics (bench.py --gpus N > 1, behind its timed region): skip the gradient all-reduces while everything else of
# the data-parallel program (streams, events, the norm of the "reduced" arena, the deferred update) runs unchanged --
# the step time without the transfers; the difference to the timed figure is the exchange that was NOT hidden.  The
# replicas drift apart from there on: only ever set at the end of a measurement process.
DRY_RUN = False


def init_from_env(backend: Optional[str] = None) -> int:
    """Initialise the default process group from torchrun's env (RANK / WORLD_SIZE / MASTER_*).  Returns world size."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if (world > 1 or _single_rank_collectives()) and not dist.is_initialized():
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:  # MMVAE_DIST_BACKEND=gloo: rehearse several ranks on one GPU (gloo stages CUDA tensors)
            backend = os.environ.get("MMVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        kwargs = {}
        if os.environ.get("MMVAE_DIST_TIMEOUT_S"):  # tests: a lost peer fails the collective instead of hanging for 30 min
            import datetime

            kwargs["timeout"] = datetime.timedelta(seconds=float(os.environ["MMVAE_DIST_TIMEOUT_S"]))
        dist.init_process_group(backend=backend, **kwargs)
    return world


def _single_rank_collectives() -> bool:
    """Diagnostics (MMVAE_SINGLE_RANK_COLLECTIVES=1): with one rank, still create the process group, issue every
    collective and run the step engine's overlapped data-parallel program -- rehearses the N > 1 code path (streams,
    events, RCCL launches beside graph replays) on a one-GPU box."""
    return os.environ.get("MMVAE_SINGLE_RANK_COLLECTIVES", "0") != "0"


def collectives_active() -> bool:
    return world_size() > 1 or (_single_rank_collectives() and dist.is_available() and dist.is_initialized())


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def bucket_ranges(numel: int, bucket_bytes: int = BUCKET_BYTES) -> List[range]:
    per = max(bucket_bytes // 4, 1)
    return [range(s, min(s + per, numel)) for s in range(0, numel, per)]


def all_reduce_flat(flat: torch.Tensor, group=None, bucket_bytes: int = BUCKET_BYTES) -> None:
    """Sum-all-reduce a flat tensor in place, in contiguous buckets, ordered after the CURRENT stream (the process
    group's communication stream waits for the current stream and the current stream waits for the collective, both
    as stream dependencies: the host never blocks)."""
    if not collectives_active() or DRY_RUN:
        return
    for r in bucket_ranges(flat.numel(), bucket_bytes):
        dist.all_reduce(flat[r.start:r.stop], op=dist.ReduceOp.SUM, group=group)


class GradAllReducer:
    """Sum-all-reduces a flat gradient arena in contiguous buckets, optionally on a side stream so that it overlaps
    with whatever the compute stream still has to do (remaining backward GEMMs / the other optimiser's update).

    Two communicators: `group` carries the bulk exchange (the active expert's ~170 MB gradient arena, which the step
    engine overlaps with the NEXT step), `small_group` the latency-bound ones (shared-VAE and adversary arenas, a few
    MB) so that they never queue behind a bulk transfer on the same communicator.  Every rank issues the collectives
    of both in the same program order."""

    def __init__(self, group=None, bucket_bytes: int = BUCKET_BYTES, side_stream: bool = True, small_group=None):
        self.group = group
        self.small_group = small_group if small_group is not None else group
        self.bucket_bytes = bucket_bytes
        self.stream = torch.cuda.Stream() if (side_stream and torch.cuda.is_available()) else None
        self._pending: list = []

    def reduce_here(self, flat_grad: torch.Tensor, small: bool = False) -> None:
        """Bucketed all-reduce ordered after (and completing on) the current stream."""
        all_reduce_flat(flat_grad, self.small_group if small else self.group, self.bucket_bytes)

    def launch(self, flat_grad: torch.Tensor) -> None:
        """Enqueue the all-reduce of `flat_grad` (in place).  Returns immediately; call wait() before reading it."""
        if not collectives_active() or DRY_RUN:
            return
        if self.stream is not None and flat_grad.is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for r in bucket_ranges(flat_grad.numel(), self.bucket_bytes):
                    dist.all_reduce(flat_grad[r.start:r.stop], op=dist.ReduceOp.SUM, group=self.group)
            self._pending.append(flat_grad)
        else:
            for r in bucket_ranges(flat_grad.numel(), self.bucket_bytes):
                dist.all_reduce(flat_grad[r.start:r.stop], op=dist.ReduceOp.SUM, group=self.group)

    def wait(self) -> None:
        if self._pending and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self._pending.clear()


_HOST_GROUP = None


def host_group():
    """A process group for small HOST-side exchanges (metadata-derived flags): gloo, created at first use -- a collective
    call, every rank must reach it at the same point of its program.  With a gloo default group it is that group."""
    global _HOST_GROUP
    if _HOST_GROUP is None:
        _HOST_GROUP = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _HOST_GROUP


def host_all_reduce_max(flags):
    """MAX all-reduce of a small numpy int32 array over the host group, in place; no device work, no stream sync."""
    t = torch.from_numpy(flags)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_group())
    return flags


def attach(model, reducer: Optional[GradAllReducer] = None) -> GradAllReducer:
    """Make every HipAdam of `model` average its gradients over the default process group.  Collective call: every
    rank must call it (a second communicator for the small arenas is created here)."""
    w = world_size()
    if reducer is None:
        small = dist.new_group() if (collectives_active() and os.environ.get("MMVAE_DP_SMALL_GROUP", "1") != "0") else None
        reducer = GradAllReducer(small_group=small)
    for opt in model.optimizers():
        opt.reducer = reducer if collectives_active() else None
        opt.grad_scale = 1.0 / w
    return reducer


def broadcast_parameters(model, src: int = 0) -> None:
    """Replicate rank `src`'s parameters and buffers (all ranks start from identical state)."""
    if not collectives_active():
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)
