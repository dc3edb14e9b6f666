"""Shared pieces of the step engine's program builder (mmvae_amd.engine): layout constants, pointer helpers, a page-locked
staging ring, layer records and bf16-plane buffers."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .modules.base.components import FCBlock

NT, NN, TN = _lib.GEMM_NT, _lib.GEMM_NN, _lib.GEMM_TN
RAW = _lib.GEMM_RAW_SLABS
ACC = _lib.GEMM_ACCUMULATE
RELU = _lib.GEMM_RELU
SLACK = _lib.GEMM_OPERAND_SLACK  # every operand the engine hands to a GEMM has 16 readable bytes behind it


# Norm-partial slots the GEMM epilogues of one optimiser may fill (mmvae_gemm_f32_sq: one per output tile).  The planning
# call counts the tiles of the element-guarded 128 x 128 kernel when an operand's rows are not 16-byte groups -- 3 784 for
# one weight gradient at the reference's 60 530 genes, two per expert: 4 096 slots sent the second one (and the whole
# gradient arena: a 93 us norm pass) back to the separate norm launch.
SQ_FUSED_SLOTS = 8192


def _p(t):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


class _PinnedRing:
    """Page-locked staging slots for per-step host tables.  Pinning a fresh tensor per step costs 0.2-0.8 ms on this
    runtime (measured; the copy itself is ~4 us to enqueue), so the slots are allocated once and reused round-robin;
    a slot is rewritten only after the copy that last read it has completed (event)."""

    def __init__(self, numel: int, dtype=torch.int32, slots: int = 4):
        self.slots = [torch.zeros(numel, dtype=dtype).pin_memory() for _ in range(slots)]
        self.views = [t.numpy() for t in self.slots]
        self.events = [None] * slots
        self.i = 0

    def take(self):
        """The next slot as a numpy array (its previous upload has completed)."""
        self.i = (self.i + 1) % len(self.slots)
        ev = self.events[self.i]
        if ev is not None:
            ev.synchronize()
        return self.views[self.i]

    def upload(self, dst: torch.Tensor) -> None:
        """The current slot into `dst`, on the current stream, by a kernel that reads the page-locked slot in place
        (mmvae_upload_words): `dst.copy_(slot, non_blocking=True)` is a hipMemcpyAsync, which this runtime hands to the
        SDMA engine -- and an SDMA copy behind a captured program blocked the HOST for ~0.5 ms per step of the conditional
        programs (tools/debug/cond_up_env.sh: with HSA_ENABLE_SDMA=0 the call took 9 us and the loop became device-bound)."""
        src = self.slots[self.i]
        n_bytes = src.numel() * src.element_size()
        if dst.numel() * dst.element_size() != n_bytes or n_bytes % 4:
            raise _lib.HipLibraryError("pinned ring: slot and destination differ in size")
        _lib.check(_lib.load().mmvae_upload_words(n_bytes // 4, src.data_ptr(), dst.data_ptr(), _s()), "mmvae_upload_words")
        ev = self.events[self.i]
        if ev is None:
            ev = self.events[self.i] = torch.cuda.Event()
        ev.record()


class _LayerRef:
    """One FCBlock layer bound to its parameter / gradient-arena tensors."""

    def __init__(self, seq: nn.Sequential, grad_of, return_hidden: bool, block=None, index: int = 0):
        self.block, self.index = block, index  # the FCBlock it belongs to: explicit keep masks are looked up there
        lin = seq.lin
        self.n_in, self.n_out = lin.in_features, lin.out_features
        self.W, self.b = lin.weight, lin.bias
        self.gW, self.gb = grad_of(lin.weight), grad_of(lin.bias)
        bn = getattr(seq, "bn", None)
        self.bn = bn
        if bn is not None:
            self.ggamma, self.gbeta = grad_of(bn.weight), grad_of(bn.bias)
        self.relu = isinstance(getattr(seq, "af", None), nn.ReLU)
        dr = getattr(seq, "dr", None)
        self.p = float(dr.p) if dr is not None else 0.0
        self.return_hidden = return_hidden and hasattr(seq, "af")


def _supported_block(block: FCBlock) -> bool:
    for seq in block.fc_layers:
        if hasattr(seq, "ln"):
            return False
        af = getattr(seq, "af", None)
        if af is not None and not isinstance(af, nn.ReLU):
            return False
    return True


class _PlaneBuf:
    """Three bf16 planes of an engine buffer [rows, cols] (int16 [3, rows + 32, ld], ld = cols rounded up to 8); the 32
    slack rows of every plane stay zero: a weight-gradient GEMM runs its K over them (kpad).  The columns between cols
    and ld are zeros too (written by the split kernels): a rows-contiguous planes operand is fetched in 16-byte groups."""

    def __init__(self, eng, name: str, rows: int, cols: int):
        self.rows, self.cols, self.ld = rows, cols, (cols + 7) // 8 * 8
        self.data = eng.buf(name, (3, rows + 32, self.ld), torch.int16)
        self.pstride = (rows + 32) * self.ld

    def ptr(self) -> int:
        return self.data.data_ptr()

    def args(self):
        """(planes pointer, leading dimension, plane stride) as the C-ABI takes them"""
        return self.data.data_ptr(), self.ld, self.pstride


_NOPL = (None, 0, 0)


def _planes_desc(planes) -> str:
    """Which operands of a launch are pre-split planes ("", "A", "B", "A+B"): probe metadata."""
    if not planes:
        return ""
    return "+".join(n for n, p in zip("AB", planes) if p is not None)
