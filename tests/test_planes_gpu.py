"""Pre-split operands (include/mmvae_hip.h, "Pre-split operands of the bf16x3 GEMMs"): the split itself, the GEMMs whose
stagers move the planes by LDS-DMA (ds_read_b64_tr_b16 fragments for the rows-contiguous forms), the producers that
write planes on the way out (reconstruction epilogue, layer tails) -- each against the fp32-operand entry point it
replaces (bit-identical: same products, same order) and against an fp64 product.  Calls go through the C-ABI."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import rel_l2  # noqa: E402

NT, NN, TN = 0, 1, 2


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a device"
    from mmvae_amd import ops as _ops, _lib

    assert _lib.load().mmvae_abi_version() >= 5
    return _ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def padded(t, slack=32):
    """engine-style buffer: zero slack rows behind the matrix (a weight-gradient GEMM pads its K to 32 over them)"""
    full = torch.zeros(t.shape[0] + slack, t.shape[1], device=t.device)
    full[: t.shape[0]] = t
    return full[: t.shape[0]]


@pytest.mark.parametrize("rows,cols", [(8, 8), (33, 64), (512, 1024), (100, 20000), (33, 60530), (64, 52437), (100, 10001),
                                       (5, 12)])
def test_split_is_exact(ops, rows, cols):
    x = rnd(rows, cols, seed=rows + cols) * torch.logspace(-6, 6, cols, device="cuda")  # wide dynamic range
    # (the split is exact while the residuals stay normal numbers: |a| >= 2^-110; below that the pieces' bit patterns no
    # longer hold them -- absolute error < 2^-126, the same in the GEMMs' own split)
    x[0, :4] = torch.tensor([0.0, -0.0, 1e-30, 3.0e38], device="cuda")
    p = ops.split_planes(x)
    assert torch.equal(p.to_float(), x)
    assert int(p.data[:, rows:].abs().max()) == 0  # slack rows untouched
    assert p.ld % 8 == 0 and p.ld - cols < 8
    if p.ld > cols:  # a column count off a multiple of 8: the columns up to the leading dimension hold zeros
        p.data[:, :, cols:] = 77
        ops.split_planes(x, out=p)
        assert int(p.data[:, :rows, cols:].abs().max()) == 0
    planes = (p.data[:, :rows, :cols].to(torch.int32) << 16).view(torch.float32)
    assert torch.equal(planes[0].view(torch.int32), x.view(torch.int32) & -65536)  # plane 0 = the top 16 bits


CASES = [
    # (layout, M, N, K, pre-split operands, raw split-K slabs)
    # (the planes kernels are the wave-specialised ones: a shape needs >= 160 work items to be taken)
    (TN, 2048, 5120, 512, "ab", False),   # 256x160 tiles
    (TN, 5120, 2048, 512, "ab", False),   # 160x256 tiles
    (TN, 4096, 4096, 256, "ab", False),   # 256x128 tiles
    (TN, 1024, 20000, 512, "ab", False),  # dW of the expert encoder's first layer at C2
    (TN, 20000, 1024, 512, "ab", False),  # dW of the expert decoder's last layer
    (TN, 2000, 5008, 480, "ab", False),   # ragged tiles (extents % 8 == 0 only), K padded to 512 over the slack rows
    (TN, 20000, 1024, 512, "b", False),   # the same product with only h pre-split (A = dP split in the kernel)
    (TN, 2000, 5008, 480, "b", False),
    (TN, 1024, 20000, 512, "a", False),   # (late r5) dW of the first layer with only dY pre-split: the batch x stays fp32
    (TN, 1024, 30000, 1024, "a", False),  # ... at C5's width and batch
    (TN, 2000, 5008, 480, "a", False),
    # gene counts off a multiple of 8 (r5; human_only.yaml:90, adversarial-conditional.yaml:107): planes with a padded ld
    (TN, 1024, 60530, 512, "ab", False),  # dW of the first layer at the reference's human width
    (TN, 1024, 52437, 512, "ab", False),  # ... and the mouse width (odd)
    (TN, 1024, 10001, 500, "ab", False),  # golden case mid_odd
    (TN, 10001, 1024, 512, "ab", False),  # the rows-contiguous A operand with an odd extent
    (TN, 60530, 1024, 512, "b", False),   # dW of the last layer: dP fp32 (odd rows, slack behind), h pre-split
    (TN, 2000, 5003, 480, "b", False),
    (NT, 512, 1024, 20000, "a", True),    # forward of the first layer: x pre-split, W fp32
    (NN, 512, 1024, 20000, "a", True),    # dX of the last layer: dP pre-split, W fp32
    (NT, 500, 1024, 8192, "a", True),
    (NN, 768, 1280, 8192, "a", True),
]


@pytest.mark.parametrize("layout,M,N,K,pre,slabs", CASES)
def test_gemm_planes_bitwise_and_fp64(ops, layout, M, N, K, pre, slabs):
    Kp = (K + 31) // 32 * 32
    if layout == TN:
        a, b = padded(rnd(K, M, seed=1)), padded(rnd(K, N, seed=2))
    elif layout == NT:
        a, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.02)
    else:
        a, b = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.02)
    a[:, ::3] = 0  # count-like sparsity
    ref = (a.double().t() @ b.double() if layout == TN else a.double() @ b.double().t() if layout == NT
           else a.double() @ b.double())
    ap = ops.split_planes(a) if "a" in pre else None
    bp = ops.split_planes(b) if "b" in pre else None
    # (fp32 operands whose rows are not 16-byte groups reach the same kernel only when the caller vouches for the slack
    # behind them: padded() provides it)
    kw = dict(K=Kp if layout == TN else None, raw_slabs=slabs, want_sq=(layout == TN), operand_slack=(layout == TN))
    lib = __import__("mmvae_amd._lib", fromlist=["load"]).load()
    assert lib.mmvae_gemm_planes_supported(layout, M, N, Kp if layout == TN else K, 0 if slabs else 1, int("a" in pre),
                                           int("b" in pre)) == 1
    f32 = ops.gemm_planes(layout, a, b, **kw)
    pl = ops.gemm_planes(layout, a if ap is None else None, b if bp is None else None, a_planes=ap, b_planes=bp, **kw)
    if layout == TN:
        (f32, sq0), (pl, sq1) = f32, pl
        assert torch.equal(sq0, sq1)
        assert abs(float(sq1.double().sum()) / float((pl.double() ** 2).sum()) - 1) < 1e-5
    assert torch.equal(f32, pl), f"planes path differs from the fp32-operand path: {(f32 - pl).abs().max()}"
    out = pl.sum(0) if slabs else pl
    assert rel_l2(out.double(), ref) <= 2e-6


def test_gemm_planes_falls_back_or_fails_loudly(ops):
    """A shape off the wave-specialised kernel: fp32 fallback when given, MMVAE_ERR_ARG otherwise."""
    a, b = padded(rnd(64, 96, seed=3)), padded(rnd(64, 72, seed=4))
    ap, bp = ops.split_planes(a), ops.split_planes(b)
    ref = ops.gemm(TN, a, b)
    assert torch.equal(ops.gemm_planes(TN, a, b, a_planes=ap, b_planes=bp), ref)
    with pytest.raises(Exception):
        ops.gemm_planes(TN, None, None, a_planes=ap, b_planes=bp)


@pytest.mark.parametrize("B,G,H", [(512, 20000, 1024), (256, 2560, 512)])
def test_recon_epilogue_writes_dp_planes(ops, B, G, H):
    h, W, bias = torch.relu(rnd(B, H, seed=5)), rnd(G, H, seed=6, scale=0.03), rnd(G, seed=7, scale=0.1)
    x = torch.relu(rnd(B, G, seed=8)) * (torch.rand(B, G, device="cuda") < 0.1)
    nrt = ops.recon_row_tiles(B)
    cp0, cp1 = torch.zeros(nrt, G, device="cuda"), torch.zeros(nrt, G, device="cuda")
    _, dP0, se0 = ops.decoder_recon(h, W, bias, x, want_xhat=False, col_part=cp0)
    dpp = ops.Planes(B, G, "cuda")
    _, dP1, se1 = ops.decoder_recon(h, W, bias, x, want_xhat=False, want_dP=False, col_part=cp1, dP_planes=dpp)
    assert dP1 is None
    assert torch.equal(dpp.to_float(), dP0) and torch.equal(se0, se1) and torch.equal(cp0, cp1)
    assert int(dpp.data[:, B:].abs().max()) == 0
    # with h pre-split too (wave-specialised kernel, 256-row tiles): same dP and per-cell errors, bias partials re-ordered
    if B % 256 == 0 and G >= 20000:
        dpp2, cp2 = ops.Planes(B, G, "cuda"), torch.zeros(nrt, G, device="cuda")
        _, dP2, se2 = ops.decoder_recon(h, W, bias, x, want_xhat=False, col_part=cp2, h_planes=ops.split_planes(h),
                                        dP_planes=dpp2)
        assert torch.equal(dP2, dP0) and torch.equal(dpp2.to_float(), dP0) and torch.equal(se2, se0)
        assert rel_l2(cp2.sum(0).double(), cp0.sum(0).double()) < 1e-6


def _bn(N):
    return dict(gamma=torch.rand(N, device="cuda") + 0.5, beta=rnd(N, seed=11), running_mean=torch.zeros(N, device="cuda"),
                running_var=torch.ones(N, device="cuda"), num_batches_tracked=None)


@pytest.mark.parametrize("B,N,S", [(512, 1024, 4), (100, 256, 1), (33, 64, 2)])
def test_layer_tails_write_planes(ops, B, N, S):
    """mmvae_fc_epilogue_fwd_planes / _bwd_planes: the planes are exactly the fp32 outputs of the plain entry points."""
    import ctypes as C

    from mmvae_amd import _lib

    lib = _lib.load()
    slabs = rnd(S, B, N, seed=12)
    bias = rnd(N, seed=13)
    ref = ops.fc_epilogue_fwd(slabs, bias, relu=True)
    d = torch.empty(B, N, device="cuda")
    pl = ops.Planes(B, N, "cuda")
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.mmvae_fc_epilogue_fwd_planes(B, N, slabs.data_ptr(), N, S, bias.data_ptr(), None, 1, 1, None, 0.0, None, None,
                                          d.data_ptr(), N, None, None, None, 0, pl.ptr(), pl.ld, pl.plane_stride, st)
    assert rc == 0
    assert torch.equal(d, ref["d"]) and torch.equal(pl.to_float(), d)
    # backward through a BatchNorm layer: the final dz and its planes
    z = rnd(B, N, seed=14)
    mean, invstd = z.mean(0), 1.0 / torch.sqrt(z.var(0, unbiased=False) + 1e-3)
    gamma = torch.rand(N, device="cuda") + 0.5
    a_act = torch.relu((z - mean) * invstd * gamma)
    ws = torch.empty(lib.mmvae_fc_workspace_bytes(B, N) // 4, device="cuda")
    outs = []
    for planes in (None, ops.Planes(B, N, "cuda")):
        dz = torch.empty(B, N, device="cuda")
        args = (B, N, slabs.data_ptr(), N, S, None, None, None, None, 0.0, 1, a_act.data_ptr(), z.data_ptr(),
                gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 1, dz.data_ptr(), N, None, None, None,
                ws.data_ptr(), ws.numel() * 4)
        if planes is None:
            rc = lib.mmvae_fc_epilogue_bwd(*args, st)
        else:
            rc = lib.mmvae_fc_epilogue_bwd_planes(*args, planes.ptr(), planes.ld, planes.plane_stride, st)
        assert rc == 0
        outs.append((dz, planes))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[1][1].to_float(), outs[1][0])


def test_layer_tail_with_piggybacked_split(ops):
    """mmvae_fc_epilogue_fwd_split: the tail's own outputs are unchanged and the extra workgroups split a row range of an
    unrelated matrix (the engine's input batch, a third of the rows per tail)."""
    from mmvae_amd import _lib

    lib = _lib.load()
    B, N, S, G = 512, 1024, 16, 20000
    slabs, bias, bn = rnd(S, B, N, seed=20), rnd(N, seed=21), _bn(N)
    mask = (torch.rand(B, N, device="cuda") > 0.1).to(torch.uint8)
    ref = ops.fc_epilogue_fwd(slabs, bias, bn=dict(bn, running_mean=bn["running_mean"].clone(),
                                                     running_var=bn["running_var"].clone()),
                              relu=True, keep_mask=mask, dropout_p=0.1)
    x = rnd(B, G, seed=22)
    xp = ops.Planes(B, G, "cuda")
    z, a, d = (torch.empty(B, N, device="cuda") for _ in range(3))
    mean, invstd = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    ws = torch.empty(lib.mmvae_fc_workspace_bytes(B, N) // 4, device="cuda")
    bnp = _lib.BnParams(bn["gamma"].data_ptr(), bn["beta"].data_ptr(), bn["running_mean"].data_ptr(),
                        bn["running_var"].data_ptr(), None, 0.01, 1e-3)
    import ctypes as C

    r0, nrows = 171, 171  # the middle third
    rc = lib.mmvae_fc_epilogue_fwd_split(B, N, slabs.data_ptr(), N, S, bias.data_ptr(), C.byref(bnp), 1, 1, mask.data_ptr(),
                                         0.1, z.data_ptr(), a.data_ptr(), d.data_ptr(), N, mean.data_ptr(),
                                         invstd.data_ptr(), ws.data_ptr(), ws.numel() * 4, nrows, G,
                                         x.data_ptr() + 4 * r0 * G, G, xp.ptr() + 2 * r0 * xp.ld, xp.ld, xp.plane_stride,
                                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    for k, t in (("z", z), ("a", a), ("d", d), ("mean", mean), ("invstd", invstd)):
        assert torch.equal(t, ref[k]), k
    got = xp.to_float()
    assert torch.equal(got[r0:r0 + nrows], x[r0:r0 + nrows])
    assert float(got[:r0].abs().max()) == 0 and float(got[r0 + nrows:].abs().max()) == 0
    # values in the fp32 denormal range (below ~2^-110 the third piece of the split keeps low bits): the
    # piggy-backed split must write the same planes as mmvae_split_planes_f32 -- not OR a piece's low half into its neighbour
    tiny = torch.zeros(B, G, device="cuda")
    e = torch.randint(-149, -110, (B, G), device="cuda").float()
    tiny[:] = torch.exp2(e) * (1 + torch.rand(B, G, device="cuda")) * torch.where(torch.rand(B, G, device="cuda") < 0.5, -1.0, 1.0)
    tiny[::3, ::5] = rnd(B, G, seed=23)[::3, ::5]  # ordinary neighbours beside the tiny ones
    want = ops.split_planes(tiny)
    xp2 = ops.Planes(B, G, "cuda")
    rc = lib.mmvae_fc_epilogue_fwd_split(B, N, slabs.data_ptr(), N, S, bias.data_ptr(), C.byref(bnp), 1, 1, mask.data_ptr(),
                                         0.1, z.data_ptr(), a.data_ptr(), d.data_ptr(), N, mean.data_ptr(),
                                         invstd.data_ptr(), ws.data_ptr(), ws.numel() * 4, B, G, tiny.data_ptr(), G,
                                         xp2.ptr(), xp2.ld, xp2.plane_stride, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    assert torch.equal(xp2.data[:, :B], want.data[:, :B])


@pytest.mark.parametrize("layout", [NT, NN, TN])
@pytest.mark.parametrize("M,N,K,splitk", [(512, 512, 1024, 0), (512, 256, 512, 1), (96, 257, 300, 1), (130, 70, 1000, 5),
                                          (33, 47, 19, 1), (64, 48, 33, 0)])
def test_small_tile_bf16x3_kernel(ops, monkeypatch, layout, M, N, K, splitk):
    """MMVAE_X3S=1: the 64x64-tile GEMMs of the core layers on the bf16 matrix cores (opt-in: measured level with the
    exact-f32 MFMA kernel inside the step).  fp32-GEMM accuracy, exact on integer-valued operands."""
    monkeypatch.setenv("MMVAE_X3S", "1")
    a, b = rnd(M, K, seed=31), rnd(K, N, seed=32)
    ref = a.double() @ b.double()
    A = a if layout != TN else a.t().contiguous()
    Bm = b.t().contiguous() if layout == NT else b
    out = ops.gemm(layout, A, Bm, splitk=splitk)
    assert rel_l2(out.double(), ref) <= 2e-6
    ai, bi = torch.round(a * 4), torch.round(b * 4) + 1
    Ai = ai if layout != TN else ai.t().contiguous()
    Bi = bi.t().contiguous() if layout == NT else bi
    assert torch.equal(ops.gemm(layout, Ai, Bi, splitk=splitk).double(), ai.double() @ bi.double())
