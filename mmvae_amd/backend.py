"""Execution-path selection.  The product path is HIP: device tensors -> libmmvae_hip.so, always.

CPU tensors are refused unless the caller has explicitly switched on *CPU plumbing* (BASELINE config C1:
"CPU reference path via configs/trainer (plumbing, no GPU)"): a plain-torch path used to exercise the host logic
(config validation, YAML instantiation, trainer loop, checkpoints) in containers without a GPU.  It is selected by
the caller, never as a fallback: a device tensor never takes it, and a missing .so never falls back to it.
"""
from contextlib import contextmanager

import torch

_CPU_PLUMBING = False


def cpu_plumbing_enabled() -> bool:
    return _CPU_PLUMBING


def set_cpu_plumbing(enabled: bool) -> None:
    global _CPU_PLUMBING
    _CPU_PLUMBING = bool(enabled)


@contextmanager
def cpu_plumbing(enabled: bool = True):
    global _CPU_PLUMBING
    prev, _CPU_PLUMBING = _CPU_PLUMBING, bool(enabled)
    try:
        yield
    finally:
        _CPU_PLUMBING = prev


def on_hip(t: torch.Tensor) -> bool:
    """True -> run the HIP kernels.  False -> caller-enabled CPU plumbing.  Anything else raises."""
    if t.is_cuda:
        return True
    if _CPU_PLUMBING:
        return False
    raise RuntimeError(
        f"mmvae_amd got a {t.device} tensor: the HIP path needs device tensors and there is no CPU fallback. "
        "Move the model and batch to the GPU, or (host-logic tests / config C1 only) wrap the call in "
        "mmvae_amd.backend.cpu_plumbing()."
    )


def to_dense(x: torch.Tensor) -> torch.Tensor:
    """Dense rows of a `torch.sparse_csr` batch: the HIP pass (ops.csr_to_dense) on the device; torch's own on CPU
    plumbing.  Dense tensors pass through."""
    if x.layout != torch.sparse_csr:
        return x
    if on_hip(x.values()):
        from . import ops

        return ops.csr_to_dense(x)
    return x.to_dense()
