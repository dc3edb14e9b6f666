"""Functional wrappers over the C-ABI (one Python function per entry point of include/mmvae_hip.h).

PyTorch is plumbing here: it owns device memory and the HIP stream.  Every function takes device tensors, passes
raw pointers / leading dimensions / the current stream to libmmvae_hip.so and returns tensors.  Nothing in this
module computes with torch ops, and nothing here runs on CPU tensors (a CPU tensor raises).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import GEMM_NT, GEMM_NN, GEMM_TN, GEMM_RELU, GEMM_ACCUMULATE, GEMM_RAW_SLABS  # noqa: F401

_workspaces: dict = {}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: Optional[torch.Tensor], name: str, dtype=torch.float32) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.HipLibraryError(
            f"{name}: mmvae_amd HIP ops need a device tensor (got {t.device}); there is no CPU fallback. "
            "For CPU plumbing (BASELINE config C1) enable mmvae_amd.backend.cpu_plumbing()."
        )
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _mat(t: torch.Tensor, name: str) -> Tuple[int, int, int]:
    """(rows, cols, ld) of a 2-D row-major view whose inner stride is 1."""
    if t.dim() != 2:
        raise ValueError(f"{name}: expected 2-D, got {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        raise ValueError(f"{name}: inner stride must be 1 (got strides {t.stride()})")
    ld = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0))
    if ld < t.shape[1]:
        raise ValueError(f"{name}: leading dimension {ld} < cols {t.shape[1]}")
    return t.shape[0], t.shape[1], ld


def workspace(nbytes: int, device, kind: str = "gemm") -> torch.Tensor:
    """Grow-only per-device scratch arena (split-K slabs / column partials).  Stream-ordered use only."""
    key = (kind, torch.device(device).index, _stream())
    ws = _workspaces.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(max(nbytes // 4 + 1, 1 << 20), dtype=torch.float32, device=device)
        _workspaces[key] = ws
    return ws


def gemm_plan(layout: int, M: int, N: int, K: int) -> Tuple[int, int]:
    lib = _lib.load()
    tile, sk = C.c_int(0), C.c_int(0)
    _lib.check(lib.mmvae_gemm_plan(layout, M, N, K, C.byref(tile), C.byref(sk)), "mmvae_gemm_plan")
    return tile.value, sk.value


def gemm(
    layout: int,
    a: torch.Tensor,
    b: torch.Tensor,
    *,
    out: Optional[torch.Tensor] = None,
    bias: Optional[torch.Tensor] = None,
    alpha: float = 1.0,
    relu: bool = False,
    accumulate: bool = False,
    splitk: int = 0,
) -> torch.Tensor:
    """C = alpha * op(A) op(B) (+bias)(relu)(+C).  NT: a[M,K], b[N,K]; NN: a[M,K], b[K,N]; TN: a[K,M], b[K,N]."""
    lib = _lib.load()
    _chk(a, "a"), _chk(b, "b"), _chk(bias, "bias")
    ar, ac, lda = _mat(a, "a")
    br, bc, ldb = _mat(b, "b")
    if layout == GEMM_NT:
        M, K, N = ar, ac, br
        if bc != K:
            raise ValueError(f"NT gemm: a {tuple(a.shape)} vs b {tuple(b.shape)}")
    elif layout == GEMM_NN:
        M, K, N = ar, ac, bc
        if br != K:
            raise ValueError(f"NN gemm: a {tuple(a.shape)} vs b {tuple(b.shape)}")
    elif layout == GEMM_TN:
        K, M, N = ar, ac, bc
        if br != K:
            raise ValueError(f"TN gemm: a {tuple(a.shape)} vs b {tuple(b.shape)}")
    else:
        raise ValueError("layout")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs out")
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _chk(out, "out")
    orr, oc, ldc = _mat(out, "out")
    if (orr, oc) != (M, N):
        raise ValueError(f"out {tuple(out.shape)} != {(M, N)}")
    if bias is not None and (bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("bias must be contiguous [N]")
    if splitk == 0:
        _, splitk = gemm_plan(layout, M, N, K)
    nbytes = lib.mmvae_gemm_workspace_bytes(layout, M, N, K, splitk)
    ws = workspace(nbytes, a.device) if nbytes else None
    flags = (GEMM_RELU if relu else 0) | (GEMM_ACCUMULATE if accumulate else 0)
    rc = lib.mmvae_gemm_f32(
        layout, M, N, K, alpha, _ptr(a), lda, _ptr(b), ldb, _ptr(out), ldc, _ptr(bias), flags, splitk, _ptr(ws),
        nbytes, _stream(),
    )
    _lib.check(rc, "mmvae_gemm_f32")
    return out


def gemm_sq(layout: int, a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None,
            bias: Optional[torch.Tensor] = None, alpha: float = 1.0, accumulate: bool = False):
    """Unsplit GEMM (as gemm()) that also returns the per-tile partial sums of squares of what it stored
    (mmvae_gemm_f32_sq): (out, partials) with partials.sum() == (out ** 2).sum() up to fp32 rounding."""
    lib = _lib.load()
    _chk(a, "a"), _chk(b, "b"), _chk(bias, "bias")
    ar, ac, lda = _mat(a, "a")
    br, bc, ldb = _mat(b, "b")
    M, K, N = (ar, ac, br) if layout == GEMM_NT else (ar, ac, bc) if layout == GEMM_NN else (ac, ar, bc)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _chk(out, "out")
    n = lib.mmvae_gemm_sq_partials(layout, M, N, K, 0)
    if n <= 0:
        raise ValueError("gemm_sq: this shape is planned as a split-K launch")
    partials = torch.empty(n, dtype=torch.float32, device=a.device)
    flags = GEMM_ACCUMULATE if accumulate else 0
    _lib.check(lib.mmvae_gemm_f32_sq(layout, M, N, K, alpha, _ptr(a), lda, _ptr(b), ldb, _ptr(out), _mat(out, "out")[2],
                                     _ptr(bias), flags, _ptr(partials), n, _stream()), "mmvae_gemm_f32_sq")
    return out, partials


class Planes:
    """The exact three-way bf16 split of an fp32 matrix [rows, cols] (include/mmvae_hip.h, "Pre-split operands"):
    `data` is int16 [3, rows + slack_rows, ld]; the slack rows stay zero (a weight-gradient GEMM pads its K to 32)."""

    def __init__(self, rows: int, cols: int, device, slack_rows: int = 32):
        # (a column count off a multiple of 8: the leading dimension is rounded up, the columns in between stay zero)
        self.rows, self.cols, self.ld = rows, cols, (cols + 7) // 8 * 8
        self.data = torch.zeros((3, rows + slack_rows, self.ld), dtype=torch.int16, device=device)
        self.plane_stride = (rows + slack_rows) * self.ld

    def ptr(self):
        return self.data.data_ptr()

    def to_float(self) -> torch.Tensor:
        """p0 + p1 + p2 as fp32 (exactly the matrix that was split; tests)."""
        planes = (self.data[:, : self.rows, : self.cols].to(torch.int32) << 16).view(torch.float32)
        return (planes[0] + planes[1]) + planes[2]


def split_planes(src: torch.Tensor, out: Optional[Planes] = None) -> Planes:
    """mmvae_split_planes_f32: write the three bf16 planes of `src` [rows, cols]."""
    lib = _lib.load()
    _chk(src, "src")
    rows, cols, ld = _mat(src, "src")
    if out is None:
        out = Planes(rows, cols, src.device)
    if (out.rows, out.cols) != (rows, cols):
        raise ValueError("split_planes: shape mismatch")
    _lib.check(lib.mmvae_split_planes_f32(rows, cols, _ptr(src), ld, out.ptr(), out.ld, out.plane_stride, _stream()),
               "mmvae_split_planes_f32")
    return out


def gemm_planes(layout: int, a: Optional[torch.Tensor], b: Optional[torch.Tensor], *, a_planes: Optional[Planes] = None,
                b_planes: Optional[Planes] = None, K: Optional[int] = None, out: Optional[torch.Tensor] = None,
                accumulate: bool = False, splitk: int = 0, raw_slabs: bool = False, want_sq: bool = False,
                operand_slack: bool = False):
    """mmvae_gemm_planes_f32: gemm() / gemm_slabs() / gemm_sq() with optional pre-split operands (a / b may be None
    when the planes are given).  K overrides the reduction length of a TN product (padded over the zero slack rows).
    operand_slack: the caller vouches for 16 readable bytes behind every fp32 operand (MMVAE_GEMM_OPERAND_SLACK)."""
    lib = _lib.load()

    def dims(t, pl):
        return (t.shape[0], t.shape[1]) if t is not None else (pl.rows, pl.cols)

    (ar, ac), (br, bc) = dims(a, a_planes), dims(b, b_planes)
    if layout == GEMM_NT:
        M, Kk, N = ar, ac, br
    elif layout == GEMM_NN:
        M, Kk, N = ar, ac, bc
    else:
        Kk, M, N = ar, ac, bc
    if K is not None:
        Kk = K
    dev = (a if a is not None else a_planes.data).device
    if splitk == 0 and not want_sq:
        _, splitk = gemm_plan(layout, M, N, Kk)
    flags = ((GEMM_ACCUMULATE if accumulate else 0) | (GEMM_RAW_SLABS if raw_slabs else 0)
             | (_lib.GEMM_OPERAND_SLACK if operand_slack else 0))
    if raw_slabs:
        out = torch.empty((splitk, M, N), dtype=torch.float32, device=dev)
        ldc, ws, nbytes = N, None, 0
    else:
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=dev)
        ldc = _mat(out, "out")[2]
        nbytes = 0 if want_sq else lib.mmvae_gemm_workspace_bytes(layout, M, N, Kk, splitk)
        ws = workspace(nbytes, dev) if nbytes else None
    partials, n = None, 0
    if want_sq:
        n = lib.mmvae_gemm_sq_partials(layout, M, N, Kk, 0)
        if n <= 0:
            raise ValueError("gemm_planes: this shape is planned as a split-K launch")
        partials = torch.empty(n, dtype=torch.float32, device=dev)
    rc = lib.mmvae_gemm_planes_f32(
        layout, M, N, Kk, 1.0,
        _ptr(a), a.stride(0) if a is not None else 0, a_planes.ptr() if a_planes else None,
        a_planes.ld if a_planes else 0, a_planes.plane_stride if a_planes else 0,
        _ptr(b), b.stride(0) if b is not None else 0, b_planes.ptr() if b_planes else None,
        b_planes.ld if b_planes else 0, b_planes.plane_stride if b_planes else 0,
        _ptr(out), ldc, None, flags, splitk, _ptr(ws), nbytes, _ptr(partials), n, _stream())
    _lib.check(rc, "mmvae_gemm_planes_f32")
    return (out, partials) if want_sq else out


def gemm_slabs(layout: int, a: torch.Tensor, b: torch.Tensor, splitk: int = 0) -> torch.Tensor:
    """Raw split-K partial products [S, M, N] (no epilogue), to be summed by fc_epilogue_fwd / _bwd."""
    lib = _lib.load()
    _chk(a, "a"), _chk(b, "b")
    ar, ac, lda = _mat(a, "a")
    br, bc, ldb = _mat(b, "b")
    if layout == GEMM_NT:
        M, K, N = ar, ac, br
    elif layout == GEMM_NN:
        M, K, N = ar, ac, bc
    else:
        K, M, N = ar, ac, bc
    if splitk == 0:
        _, splitk = gemm_plan(layout, M, N, K)
    out = torch.empty((splitk, M, N), dtype=torch.float32, device=a.device)
    rc = lib.mmvae_gemm_f32(
        layout, M, N, K, 1.0, _ptr(a), lda, _ptr(b), ldb, _ptr(out), N, None, GEMM_RAW_SLABS, splitk, None, 0, _stream()
    )
    _lib.check(rc, "mmvae_gemm_f32(raw slabs)")
    return out


def recon_tiles(G: int) -> int:
    return _lib.load().mmvae_recon_tiles(G)


def recon_row_tiles(rows: int) -> int:
    return _lib.load().mmvae_recon_row_tiles(rows)


def decoder_recon(
    h: torch.Tensor,
    W: torch.Tensor,
    bias: Optional[torch.Tensor],
    x: torch.Tensor,
    *,
    want_xhat: bool = True,
    want_dP: bool = True,
    xhat: Optional[torch.Tensor] = None,
    dP: Optional[torch.Tensor] = None,
    se_part: Optional[torch.Tensor] = None,
    col_part: Optional[torch.Tensor] = None,
    h_planes: Optional["Planes"] = None,
    dP_planes: Optional["Planes"] = None,
    h_kpad: bool = False,
    W_planes: Optional["Planes"] = None,
):
    """Fused last decoder layer + squared error.  h [R,H] (R = K*B rows), W [G,H], x [B,G].
    h_kpad: h is a view of a buffer whose columns H .. round_up(H, 32) - 1 are ZERO and W's rows may be read that far
    (mmvae_recon_set_h_kpad): a hidden width that is not a multiple of 32 then takes the pipelined kernels.
    h_planes: the pre-split h (mmvae_decoder_recon_planes_f32; h stays the fallback).
    Returns (xhat [R,G] | None, dP [R,G] | None, se_part [tiles, R]).  col_part [recon_row_tiles(R), G] (optional):
    receives the column sums of dP per row tile (their sum over the tiles is the bias gradient when K = 1)."""
    lib = _lib.load()
    _chk(h, "h"), _chk(W, "W"), _chk(bias, "bias"), _chk(x, "x")
    R, H, ldh = _mat(h, "h")
    G, H2, ldw = _mat(W, "W")
    B, G2, ldx = _mat(x, "x")
    if H2 != H or G2 != G or R % B != 0:
        raise ValueError(f"decoder_recon shapes: h {tuple(h.shape)} W {tuple(W.shape)} x {tuple(x.shape)}")
    T = lib.mmvae_recon_tiles(G)
    if xhat is None and want_xhat:
        xhat = torch.empty((R, G), dtype=torch.float32, device=h.device)
    if dP is None and want_dP:
        dP = torch.empty((R, G), dtype=torch.float32, device=h.device)
    if se_part is None:
        se_part = torch.empty((T, R), dtype=torch.float32, device=h.device)
    ldxh = _mat(xhat, "xhat")[2] if xhat is not None else 0
    lddp = _mat(dP, "dP")[2] if dP is not None else 0
    if col_part is not None and (tuple(col_part.shape) != (lib.mmvae_recon_row_tiles(R), G) or not col_part.is_contiguous()):
        raise ValueError(f"decoder_recon: col_part must be a contiguous [{lib.mmvae_recon_row_tiles(R)}, {G}] tensor")
    if W_planes is not None:  # weights pre-split as well (mmvae_decoder_recon_wplanes_f32; taken together with h_planes)
        if dP_planes is not None:
            raise ValueError("decoder_recon: W_planes and dP_planes have no common entry point")
        hp, wp = h_planes, W_planes
        rc = lib.mmvae_decoder_recon_wplanes_f32(
            R, B, G, H, _ptr(h), ldh, hp.ptr() if hp else None, hp.ld if hp else 0, hp.plane_stride if hp else 0,
            _ptr(W), ldw, wp.ptr(), wp.ld, wp.plane_stride, _ptr(bias), _ptr(x), ldx, _ptr(xhat), ldxh, _ptr(dP), lddp,
            _ptr(se_part), _ptr(col_part), _stream(),
        )
        _lib.check(rc, "mmvae_decoder_recon_wplanes_f32")
        return xhat, dP, se_part
    if h_planes is not None or dP_planes is not None:
        hp, dpp = h_planes, dP_planes
        rc = lib.mmvae_decoder_recon_planes_f32(
            R, B, G, H, _ptr(h), ldh, hp.ptr() if hp else None, hp.ld if hp else 0, hp.plane_stride if hp else 0,
            _ptr(W), ldw, _ptr(bias), _ptr(x), ldx, _ptr(xhat), ldxh, _ptr(dP), lddp,
            dpp.ptr() if dpp else None, dpp.ld if dpp else 0, dpp.plane_stride if dpp else 0,
            _ptr(se_part), _ptr(col_part), _stream(),
        )
        _lib.check(rc, "mmvae_decoder_recon_planes_f32")
        return xhat, dP, se_part
    if h_kpad and ldh < (H + 31) // 32 * 32:
        raise ValueError("decoder_recon: h_kpad needs h's leading dimension to cover the padded width")
    lib.mmvae_recon_set_h_kpad(1 if h_kpad else 0)
    try:
        rc = lib.mmvae_decoder_recon_rows_colsum_f32(
            R, B, G, H, _ptr(h), ldh, _ptr(W), ldw, _ptr(bias), _ptr(x), ldx, _ptr(xhat), ldxh, _ptr(dP), lddp,
            _ptr(se_part), _ptr(col_part), _stream(),
        )
    finally:
        lib.mmvae_recon_set_h_kpad(0)
    _lib.check(rc, "mmvae_decoder_recon_rows_colsum_f32")
    return xhat, dP, se_part


def fc_epilogue_fwd(
    inp: torch.Tensor,
    bias: Optional[torch.Tensor],
    *,
    bn: Optional[dict] = None,
    training: bool = True,
    relu: bool = False,
    keep_mask: Optional[torch.Tensor] = None,
    dropout_p: float = 0.0,
    want_a: bool = True,
):
    """inp: [B,N] or slabs [S,B,N].  bn: dict(gamma, beta, running_mean, running_var, num_batches_tracked,
    momentum, eps) or None.  Returns dict(z, a, d, mean, invstd) (entries may be None)."""
    lib = _lib.load()
    _chk(inp, "inp"), _chk(bias, "bias"), _chk(keep_mask, "keep_mask", torch.uint8)
    if inp.dim() == 2:
        inp = inp.unsqueeze(0)
    if not inp.is_contiguous():
        raise ValueError("inp must be contiguous")
    S, B, N = inp.shape
    dev = inp.device
    has_drop = keep_mask is not None
    z = torch.empty((B, N), dtype=torch.float32, device=dev) if bn is not None else None
    d = torch.empty((B, N), dtype=torch.float32, device=dev)
    a = torch.empty((B, N), dtype=torch.float32, device=dev) if (has_drop and want_a) else None
    mean = invstd = None
    bnp = None
    if bn is not None:
        if training:
            mean = torch.empty(N, dtype=torch.float32, device=dev)
            invstd = torch.empty(N, dtype=torch.float32, device=dev)
        nbt = bn.get("num_batches_tracked")
        bnp = _lib.BnParams(
            _ptr(bn.get("gamma")), _ptr(bn.get("beta")), _ptr(bn.get("running_mean")), _ptr(bn.get("running_var")),
            _ptr(nbt) if nbt is not None else None, float(bn.get("momentum", 0.01)), float(bn.get("eps", 1e-3)),
        )
    if has_drop and (tuple(keep_mask.shape) != (B, N) or not keep_mask.is_contiguous()):
        raise ValueError("keep_mask must be contiguous [B,N] uint8")
    nws = lib.mmvae_fc_workspace_bytes(B, N)
    ws = workspace(nws, dev, "fc")
    rc = lib.mmvae_fc_epilogue_fwd(
        B, N, _ptr(inp), N, S, _ptr(bias), C.byref(bnp) if bnp is not None else None, int(training), int(relu),
        _ptr(keep_mask), float(dropout_p), _ptr(z), _ptr(a), _ptr(d), N, _ptr(mean), _ptr(invstd), _ptr(ws), nws,
        _stream(),
    )
    _lib.check(rc, "mmvae_fc_epilogue_fwd")
    return {"z": z, "a": a if a is not None else d, "d": d, "mean": mean, "invstd": invstd}


def fc_epilogue_bwd(
    din: torch.Tensor,
    *,
    addend: Optional[torch.Tensor] = None,
    addend_a: Optional[torch.Tensor] = None,
    row_scale: Optional[torch.Tensor] = None,
    keep_mask: Optional[torch.Tensor] = None,
    dropout_p: float = 0.0,
    relu: bool = False,
    a: Optional[torch.Tensor] = None,
    z: Optional[torch.Tensor] = None,
    gamma: Optional[torch.Tensor] = None,
    mean: Optional[torch.Tensor] = None,
    invstd: Optional[torch.Tensor] = None,
    has_bn: bool = False,
    want_dz: bool = True,
    want_dbias: bool = True,
    dbias_out: Optional[torch.Tensor] = None,
    dgamma_out: Optional[torch.Tensor] = None,
    dbeta_out: Optional[torch.Tensor] = None,
):
    """Backward of the layer tail.  din: [B,N] or slabs [S,B,N]; `addend`: further gradient on the layer output d;
    `addend_a`: gradient on the pre-dropout activation (bypasses the keep mask).  Returns (dz, dbias, dgamma, dbeta)."""
    lib = _lib.load()
    _chk(din, "din")
    if din.dim() == 2:
        din = din.unsqueeze(0)
    if not din.is_contiguous():
        raise ValueError("din must be contiguous")
    S, B, N = din.shape
    dev = din.device
    for name, t in (("addend", addend), ("addend_a", addend_a), ("a", a), ("z", z)):
        _chk(t, name)
        if t is not None and (tuple(t.shape) != (B, N) or not t.is_contiguous()):
            raise ValueError(f"{name} must be contiguous [B,N]")
    _chk(row_scale, "row_scale"), _chk(keep_mask, "keep_mask", torch.uint8)
    dz = torch.empty((B, N), dtype=torch.float32, device=dev) if (want_dz or has_bn) else None
    dbias = dbias_out if dbias_out is not None else (torch.empty(N, dtype=torch.float32, device=dev) if want_dbias else None)
    dgamma = dbeta = None
    if has_bn:
        dgamma = dgamma_out if dgamma_out is not None else torch.empty(N, dtype=torch.float32, device=dev)
        dbeta = dbeta_out if dbeta_out is not None else torch.empty(N, dtype=torch.float32, device=dev)
    nws = lib.mmvae_fc_workspace_bytes(B, N)
    ws = workspace(nws, dev, "fc")
    rc = lib.mmvae_fc_epilogue_bwd(
        B, N, _ptr(din), N, S, _ptr(addend), _ptr(addend_a), _ptr(row_scale), _ptr(keep_mask), float(dropout_p), int(relu), _ptr(a),
        _ptr(z), _ptr(gamma), _ptr(mean), _ptr(invstd), int(has_bn), _ptr(dz), N, _ptr(dbias), _ptr(dgamma),
        _ptr(dbeta), _ptr(ws), nws, _stream(),
    )
    _lib.check(rc, "mmvae_fc_epilogue_bwd")
    return dz, dbias, dgamma, dbeta


def layernorm_fwd(x: torch.Tensor, eps: float = 1e-5):
    lib = _lib.load()
    _chk(x, "x")
    B, N, ldx = _mat(x, "x")
    y = torch.empty((B, N), dtype=torch.float32, device=x.device)
    invstd = torch.empty(B, dtype=torch.float32, device=x.device)
    _lib.check(lib.mmvae_layernorm_fwd(B, N, _ptr(x), ldx, eps, _ptr(y), N, None, _ptr(invstd), _stream()), "layernorm_fwd")
    return y, invstd


def layernorm_bwd(dy: torch.Tensor, y: torch.Tensor, invstd: torch.Tensor):
    lib = _lib.load()
    _chk(dy, "dy"), _chk(y, "y"), _chk(invstd, "invstd")
    B, N, lddy = _mat(dy, "dy")
    _, _, ldy = _mat(y, "y")
    dx = torch.empty((B, N), dtype=torch.float32, device=dy.device)
    _lib.check(lib.mmvae_layernorm_bwd(B, N, _ptr(dy), lddy, _ptr(y), ldy, _ptr(invstd), _ptr(dx), N, _stream()), "layernorm_bwd")
    return dx


def reparam_kl_fwd(mu: torch.Tensor, a_raw: torch.Tensor, eps: Optional[torch.Tensor], var_eps: float = 1e-4,
                   want_stats: bool = True):
    """mu, a_raw [B,Z]; eps [K,B,Z] or [B,Z] or None.  Returns (std [B,Z], z [K,B,Z]|None, kl_row [B], stat_row [2,B]|None)."""
    lib = _lib.load()
    _chk(mu, "mu"), _chk(a_raw, "a_raw"), _chk(eps, "eps")
    if not (mu.is_contiguous() and a_raw.is_contiguous()):
        raise ValueError("mu / a_raw must be contiguous")
    B, Z = mu.shape
    K = 1
    z = None
    if eps is not None:
        if not eps.is_contiguous():
            raise ValueError("eps must be contiguous")
        K = eps.shape[0] if eps.dim() == 3 else 1
        if eps.numel() != K * B * Z:
            raise ValueError("eps shape")
        z = torch.empty((K, B, Z) if eps.dim() == 3 else (B, Z), dtype=torch.float32, device=mu.device)
    std = torch.empty_like(mu)
    kl_row = torch.empty(B, dtype=torch.float32, device=mu.device)
    stat = torch.empty((2, B), dtype=torch.float32, device=mu.device) if want_stats else None
    rc = lib.mmvae_reparam_kl_fwd(B, Z, K, _ptr(mu), _ptr(a_raw), _ptr(eps), var_eps, _ptr(std), _ptr(z), _ptr(kl_row),
                                  _ptr(stat), _stream())
    _lib.check(rc, "mmvae_reparam_kl_fwd")
    return std, z, kl_row, stat


def reparam_kl_bwd(mu, std, eps, dz, *, dmu_extra=None, dstd_extra=None, dkl_row=None, kl_scale_dev=None,
                   kl_scale: float = 1.0, var_eps: float = 1e-4):
    lib = _lib.load()
    for n, t in (("mu", mu), ("std", std), ("eps", eps), ("dz", dz), ("dmu_extra", dmu_extra),
                 ("dstd_extra", dstd_extra), ("dkl_row", dkl_row), ("kl_scale_dev", kl_scale_dev)):
        _chk(t, n)
        if t is not None and not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous")
    B, Z = mu.shape
    K = 1
    if dz is not None:
        K = dz.numel() // (B * Z)
    dmu = torch.empty_like(mu)
    da = torch.empty_like(mu)
    rc = lib.mmvae_reparam_kl_bwd(B, Z, K, _ptr(mu), _ptr(std), _ptr(eps), _ptr(dz), _ptr(dmu_extra), _ptr(dstd_extra),
                                  _ptr(dkl_row), _ptr(kl_scale_dev), float(kl_scale), var_eps, _ptr(dmu), _ptr(da),
                                  _stream())
    _lib.check(rc, "mmvae_reparam_kl_bwd")
    return dmu, da


def mse_sum_fwd_bwd(xhat: torch.Tensor, x: torch.Tensor, *, want_grad: bool = True, gscale_dev=None,
                    gscale: float = 1.0):
    """Returns (se_row [B], dxhat [B,G] | None) with dxhat = gscale * 2 (xhat - x)."""
    lib = _lib.load()
    _chk(xhat, "xhat"), _chk(x, "x"), _chk(gscale_dev, "gscale_dev")
    B, G, ldxh = _mat(xhat, "xhat")
    B2, G2, ldx = _mat(x, "x")
    if (B, G) != (B2, G2):
        raise ValueError("mse shapes")
    se = torch.empty(B, dtype=torch.float32, device=x.device)
    dx = torch.empty((B, G), dtype=torch.float32, device=x.device) if want_grad else None
    rc = lib.mmvae_mse_sum_fwd_bwd(B, G, _ptr(xhat), ldxh, _ptr(x), ldx, _ptr(se), _ptr(dx), G, _ptr(gscale_dev),
                                   float(gscale), _stream())
    _lib.check(rc, "mmvae_mse_sum_fwd_bwd")
    return se, dx


def elbo_finalize(se_part: torch.Tensor, kl_row, stat_row, *, B: int, K: int = 1, Z: int = 0, kl_weight_dev=None,
                  kl_weight: float = 1.0, want_w: bool = False):
    """se_part [T, K*B].  Returns (out6 [6] = loss, recon, kl, kl_weight, mean(mu), mean(var); w [K*B] | None)."""
    lib = _lib.load()
    _chk(se_part, "se_part"), _chk(kl_row, "kl_row"), _chk(stat_row, "stat_row"), _chk(kl_weight_dev, "klw")
    if se_part.dim() == 1:
        se_part = se_part.unsqueeze(0)
    if not se_part.is_contiguous() or se_part.shape[1] != K * B:
        raise ValueError("se_part must be contiguous [T, K*B]")
    T = se_part.shape[0]
    out = torch.empty(6, dtype=torch.float32, device=se_part.device)
    w = torch.empty(K * B, dtype=torch.float32, device=se_part.device) if want_w else None
    recon_row = torch.empty(B, dtype=torch.float32, device=se_part.device)
    rc = lib.mmvae_elbo_finalize(B, K, T, _ptr(se_part), _ptr(kl_row), _ptr(stat_row), Z, _ptr(kl_weight_dev),
                                 float(kl_weight), _ptr(out), _ptr(w), _ptr(recon_row), _stream())
    _lib.check(rc, "mmvae_elbo_finalize")
    return out, w


def cross_entropy_sum(logits: torch.Tensor, labels: torch.Tensor, *, want_grad: bool = True, gscale: float = 1.0,
                      gscale_dev: Optional[torch.Tensor] = None):
    """Returns (loss_rows [B], dlogits [B,C] | None)."""
    lib = _lib.load()
    _chk(logits, "logits"), _chk(labels, "labels", torch.int64)
    B, Cn, ld = _mat(logits, "logits")
    if labels.numel() != B or not labels.is_contiguous():
        raise ValueError("labels must be contiguous [B] int64")
    rows = torch.empty(B, dtype=torch.float32, device=logits.device)
    dl = torch.empty((B, Cn), dtype=torch.float32, device=logits.device) if want_grad else None
    _chk(gscale_dev, "gscale_dev")
    rc = lib.mmvae_cross_entropy_sum(B, Cn, _ptr(logits), ld, _ptr(labels), _ptr(rows), _ptr(dl), Cn,
                                     _ptr(gscale_dev), float(gscale), _stream())
    _lib.check(rc, "mmvae_cross_entropy_sum")
    return rows, dl


def sum_f32(v: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    lib = _lib.load()
    _chk(v, "v")
    if not v.is_contiguous():
        raise ValueError("v must be contiguous")
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=v.device)
    _lib.check(lib.mmvae_sum_f32(v.numel(), _ptr(v), _ptr(out), int(accumulate), _stream()), "mmvae_sum_f32")
    return out


def sqnorm_partials(n: int) -> int:
    return _lib.load().mmvae_sqnorm_partials(n)


def clip_adam_step(param, grad, exp_avg, exp_avg_sq, state, partials, *, lr=5e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                   weight_decay=1e-6, max_norm=0.0, grad_scale=1.0, do_norm=True, do_step=True, advance=True):
    """Global-norm clip + Adam over one flat arena.  `state`: float32[8] device tensor (step, norm, clip, bc1, bc2).
    do_norm: recompute the gradient norm; advance: increment the step counter; do_step: apply the update."""
    lib = _lib.load()
    for n_, t in (("param", param), ("grad", grad), ("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq),
                  ("state", state), ("partials", partials)):
        _chk(t, n_)
        if not t.is_contiguous():
            raise ValueError(f"{n_} must be contiguous")
    n = param.numel()
    if grad.numel() != n or exp_avg.numel() != n or exp_avg_sq.numel() != n or state.numel() < 8:
        raise ValueError("arena sizes")
    s = _stream()
    npart = lib.mmvae_sqnorm_partials(n)
    if do_norm:
        if partials.numel() < npart:
            raise ValueError("partials too small")
        _lib.check(lib.mmvae_grad_sqnorm(n, _ptr(grad), _ptr(partials), s), "mmvae_grad_sqnorm")
    flags = (_lib.PREPARE_NORM if do_norm else 0) | (_lib.PREPARE_ADVANCE if (advance and do_step) else 0)
    _lib.check(lib.mmvae_adam_prepare(npart, _ptr(partials), float(max_norm), float(grad_scale), beta1, beta2,
                                      _ptr(state), flags, s), "mmvae_adam_prepare")
    if do_step:
        _lib.check(lib.mmvae_adam_step(n, _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(state), lr,
                                       beta1, beta2, eps, weight_decay, float(grad_scale), s), "mmvae_adam_step")


def philox_keep_mask(shape, p_drop: float, rng_state: torch.Tensor, stream_id: int = 0, advance: bool = True,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _chk(rng_state, "rng_state", torch.int64)
    if out is None:
        out = torch.empty(shape, dtype=torch.uint8, device=rng_state.device)
    _lib.check(lib.mmvae_philox_keep_mask(out.numel(), float(p_drop), _ptr(out), _ptr(rng_state), stream_id,
                                          int(advance), _stream()), "mmvae_philox_keep_mask")
    return out


def philox_normal(shape, rng_state: torch.Tensor, stream_id: int = 1, advance: bool = True,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _chk(rng_state, "rng_state", torch.int64)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=rng_state.device)
    _lib.check(lib.mmvae_philox_normal(out.numel(), _ptr(out), _ptr(rng_state), stream_id, int(advance), _stream()),
               "mmvae_philox_normal")
    return out


def axpby(alpha: float, x: torch.Tensor, beta: float, y: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    _chk(x, "x"), _chk(y, "y")
    if not (x.is_contiguous() and y.is_contiguous()) or x.numel() != y.numel():
        raise ValueError("axpby needs contiguous same-size tensors")
    _lib.check(lib.mmvae_axpby(x.numel(), float(alpha), _ptr(x), float(beta), _ptr(y), _stream()), "mmvae_axpby")
    return y


def sum_parts_batch(jobs) -> None:
    """jobs: list of (src [n_parts, rows, cols] contiguous tensor, dst [rows, cols] tensor view, alpha, accumulate).
    One launch; every dst = (dst +) alpha * sum over parts in part order."""
    lib = _lib.load()
    if not jobs:
        return
    arr = (_lib.SumJob * len(jobs))()
    dev = jobs[0][0].device
    for k, (src, dst, alpha, accumulate) in enumerate(jobs):
        _chk(src, "src"), _chk(dst, "dst")
        if src.dim() != 3 or not src.is_contiguous() or dst.dim() != 2 or tuple(src.shape[1:]) != tuple(dst.shape):
            raise ValueError("sum_parts_batch: src must be contiguous [parts, rows, cols] and dst [rows, cols]")
        if dst.shape[1] > 1 and dst.stride(1) != 1:
            raise ValueError("sum_parts_batch: dst rows must be contiguous")
        P, R, Cn = src.shape
        arr[k] = _lib.SumJob(_ptr(src), _ptr(dst), R * Cn, Cn, dst.stride(0) if R > 1 else Cn, P, R, Cn, float(alpha),
                             _lib.GEMM_ACCUMULATE if accumulate else 0, 0)
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    _lib.check(lib.mmvae_sum_parts_batch(len(jobs), _ptr(jobs_dev), max(int(j.rows) * int(j.cols) for j in arr), _stream()),
               "mmvae_sum_parts_batch")
    torch.cuda.current_stream().synchronize()  # jobs_dev must outlive the launch


def csr_to_dense(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense fp32 [B, G] rows of a `torch.sparse_csr` batch (device), written by one HIP pass."""
    lib = _lib.load()
    if x.layout != torch.sparse_csr or x.dim() != 2:
        raise ValueError("csr_to_dense: expected a 2-D torch.sparse_csr tensor")
    crow, col, val = x.crow_indices(), x.col_indices(), x.values()
    if not val.is_cuda:
        raise _lib.HipLibraryError("csr_to_dense: CPU tensors have no HIP path")
    if crow.dtype != col.dtype or crow.dtype not in (torch.int32, torch.int64):
        crow, col = crow.long(), col.long()
    fn = lib.mmvae_csr_to_dense_i32_f32 if crow.dtype == torch.int32 else lib.mmvae_csr_to_dense_f32
    if val.dtype != torch.float32:
        val = val.float()
    B, G = x.shape
    if out is None:
        out = torch.empty((B, G), dtype=torch.float32, device=val.device)
    _chk(out, "out")
    if tuple(out.shape) != (B, G):
        raise ValueError(f"out {tuple(out.shape)} != {(B, G)}")
    nnz = int(val.numel())
    crow, col, val = crow.contiguous(), col.contiguous(), val.contiguous()
    _lib.check(fn(B, G, nnz, crow.data_ptr(), col.data_ptr() if nnz else None, val.data_ptr() if nnz else None, _ptr(out),
                  _mat(out, "out")[2], _stream()), "mmvae_csr_to_dense")
    return out


def csr_spmm_wt(x: torch.Tensor, wt: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """f1 measurement: y[B, N] = x_csr[B, G] . wt[G, N] (+ bias) straight from a `torch.sparse_csr` batch with int32
    indices; wt is the layer's weight TRANSPOSED.  Not on the product path (the engine densifies)."""
    lib = _lib.load()
    if x.layout != torch.sparse_csr or x.dim() != 2:
        raise ValueError("csr_spmm_wt: expected a 2-D torch.sparse_csr tensor")
    crow, col, val = x.crow_indices().int().contiguous(), x.col_indices().int().contiguous(), x.values().float().contiguous()
    _chk(wt, "wt"), _chk(bias, "bias")
    B, G = x.shape
    if wt.shape[0] != G or wt.stride(1) != 1 or wt.shape[1] % 4:
        raise ValueError("wt must be [G, N] with N contiguous and N % 4 == 0")
    N = wt.shape[1]
    y = torch.empty((B, N), dtype=torch.float32, device=val.device)
    nnz = int(val.numel())
    _lib.check(lib.mmvae_csr_spmm_wt_i32_f32(B, N, G, nnz, crow.data_ptr(), col.data_ptr() if nnz else None,
                                             val.data_ptr() if nnz else None, _ptr(wt), wt.stride(0), _ptr(bias), _ptr(y), N,
                                             _stream()), "mmvae_csr_spmm_wt_i32_f32")
    return y


def cond_linear_fwd(x: torch.Tensor, params: torch.Tensor, w_off: torch.Tensor, b_off: torch.Tensor,
                    cond: torch.Tensor, n_out: int, rows: torch.Tensor = None) -> torch.Tensor:
    """y[b] = W[cond[b]] x[b] + bias[cond[b]]; blocks addressed by element offsets into the flat `params` arena.
    `rows` (int32 [B], the cells sorted by condition) selects the kernel that reads a shared block once per 8 cells."""
    lib = _lib.load()
    _chk(x, "x"), _chk(params, "params")
    B, n_in, ldx = _mat(x, "x")
    y = torch.empty((B, n_out), dtype=torch.float32, device=x.device)
    _lib.check(lib.mmvae_cond_linear_fwd(B, n_in, n_out, _ptr(x), ldx, _ptr(params), _ptr(w_off), _ptr(b_off),
                                         _ptr(cond), _ptr(rows), _ptr(y), n_out, _stream()), "mmvae_cond_linear_fwd")
    return y


_COND_PARTIALS: dict = {}


def cond_tables_to_device(t: dict, device) -> dict:
    """The index tables of `cond_tables.group_tables` as ONE device upload; returns int32 device views by name."""
    import numpy as np

    names = ("cond", "rows", "chunk_dst", "chunk_beg", "chunk_end", "red_cond", "red_slot", "red_n")
    packed = torch.from_numpy(np.concatenate([t[n] for n in names])).to(device, non_blocking=True)
    out, off = {}, 0
    for n in names:
        out[n] = packed[off:off + len(t[n])]
        off += len(t[n])
    return out


def cond_linear_bwd(dy: torch.Tensor, x: torch.Tensor, params: torch.Tensor, grads: torch.Tensor, w_off: torch.Tensor,
                    b_off: torch.Tensor, tables: dict, dx: torch.Tensor = None, accumulate: bool = False,
                    sorted_dx: bool = True) -> torch.Tensor:
    """dx (into `dx`, added to it when `accumulate`), and dW / db of the PRESENT conditions written into `grads` (same
    layout as `params`).  `tables`: device int32 arrays cond, rows, chunk_*, red_* (cond_tables.group_tables)."""
    from . import cond_tables

    lib = _lib.load()
    _chk(dy, "dy"), _chk(x, "x")
    B, n_out, lddy = _mat(dy, "dy")
    _, n_in, ldx = _mat(x, "x")
    if dx is None:
        dx = torch.empty((B, n_in), dtype=torch.float32, device=x.device)
    s = _stream()
    _lib.check(lib.mmvae_cond_linear_bwd_dx(B, n_in, n_out, _ptr(dy), lddy, _ptr(params), _ptr(w_off), _ptr(tables["cond"]),
                                            _ptr(tables["rows"]) if sorted_dx else None, _ptr(dx), _mat(dx, "dx")[2],
                                            int(accumulate), s), "mmvae_cond_linear_bwd_dx")
    n_red = int(tables["red_cond"].numel())
    partials = None
    if n_red:
        need = cond_tables.partial_slots(B) * (n_in * n_out + n_out)
        key = (x.device, torch.cuda.current_stream().cuda_stream)
        partials = _COND_PARTIALS.get(key)
        if partials is None or partials.numel() < need:
            partials = _COND_PARTIALS[key] = torch.empty(need, dtype=torch.float32, device=x.device)
    _lib.check(lib.mmvae_cond_linear_bwd_dw(int(tables["chunk_dst"].numel()), _ptr(tables["chunk_dst"]),
                                            _ptr(tables["chunk_beg"]), _ptr(tables["chunk_end"]), _ptr(tables["rows"]),
                                            n_in, n_out, _ptr(dy), lddy, _ptr(x), ldx, _ptr(grads), _ptr(w_off), _ptr(b_off),
                                            n_red, _ptr(tables["red_cond"]) if n_red else None,
                                            _ptr(tables["red_slot"]) if n_red else None,
                                            _ptr(tables["red_n"]) if n_red else None, _ptr(partials), s),
               "mmvae_cond_linear_bwd_dw")
    return dx


def gemm_batch(jobs) -> None:
    """Grouped launch of small GEMMs.  jobs: list of dicts {layout, a, b, out, bias=None, alpha=1.0, relu=False,
    accumulate=False} with the operand conventions of gemm().  One launch (mmvae_gemm_batch_f32); raises when a job
    does not meet the alignment requirements of the grouped kernel."""
    lib = _lib.load()
    if not jobs:
        return
    arr = (_lib.GemmJob * len(jobs))()
    for k, j in enumerate(jobs):
        a, b, out, bias, layout = j["a"], j["b"], j["out"], j.get("bias"), j["layout"]
        _chk(a, "a"), _chk(b, "b"), _chk(out, "out"), _chk(bias, "bias")
        ar, ac, lda = _mat(a, "a")
        br, bc, ldb = _mat(b, "b")
        M, N, ldc = _mat(out, "out")
        if layout == GEMM_NT:
            K, ok = ac, (ar, bc, br) == (M, ac, N)
        elif layout == GEMM_NN:
            K, ok = ac, (ar, br, bc) == (M, ac, N)
        elif layout == GEMM_TN:
            K, ok = ar, (ac, br, bc) == (M, ar, N)
        else:
            raise ValueError("layout")
        if not ok or (bias is not None and (bias.numel() != N or not bias.is_contiguous())):
            raise ValueError(f"gemm_batch job {k}: shapes a {tuple(a.shape)} b {tuple(b.shape)} out {tuple(out.shape)}")
        flags = (GEMM_RELU if j.get("relu") else 0) | (GEMM_ACCUMULATE if j.get("accumulate") else 0)
        arr[k] = _lib.GemmJob(_ptr(a), _ptr(b), _ptr(out), _ptr(bias), lda, ldb, ldc, layout, M, N, K,
                              float(j.get("alpha", 1.0)), flags, 0, 0)
    total = C.c_int(0)
    _lib.check(lib.mmvae_gemm_batch_prepare(len(jobs), C.addressof(arr), C.byref(total)), "mmvae_gemm_batch_prepare")
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(jobs[0]["a"].device)
    _lib.check(lib.mmvae_gemm_batch_f32(len(jobs), _ptr(jobs_dev), total.value, _stream()), "mmvae_gemm_batch_f32")
    torch.cuda.current_stream().synchronize()  # jobs_dev must outlive the launch


def weighted_colsum(x: torch.Tensor, row_weight: torch.Tensor) -> torch.Tensor:
    """partials [chunks of 256 rows, N]: sum over a chunk's rows of row_weight[r] * x[r, :] (mmvae_weighted_colsum_f32)."""
    lib = _lib.load()
    _chk(x, "x"), _chk(row_weight, "row_weight")
    B, N, ldx = _mat(x, "x")
    parts = torch.empty((lib.mmvae_weighted_colsum_chunks(B), N), dtype=torch.float32, device=x.device)
    _lib.check(lib.mmvae_weighted_colsum_f32(B, N, _ptr(x), ldx, _ptr(row_weight), _ptr(parts), _stream()),
               "mmvae_weighted_colsum_f32")
    return parts


def scale_rows(x: torch.Tensor, row_scale: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    _chk(x, "x"), _chk(row_scale, "row_scale")
    B, N, ldx = _mat(x, "x")
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _lib.check(lib.mmvae_scale_rows(B, N, _ptr(x), ldx, _ptr(row_scale), _ptr(out), _mat(out, "out")[2], _stream()),
               "mmvae_scale_rows")
    return out
