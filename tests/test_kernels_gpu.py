"""Per-kernel parity tests: every C-ABI entry point of libmmvae_hip.so against a CPU computation of the same math
(fp64 matmuls / torch autograd on CPU).  These call THROUGH the C-ABI (mmvae_amd.ops -> ctypes -> .so).
Tolerances: GEMM outputs rel-L2 <= 2e-6 vs fp64 (bf16x3 and exact-f32 MFMA paths alike); elementwise 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import rel_l2  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a device"
    from mmvae_amd import ops as _ops, _lib

    lib = _lib.load()
    assert lib.mmvae_abi_version() >= 3
    return _ops


def dev(t):
    return t.cuda()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# ----------------------------------------------------------------------------------------------------------- GEMM
def _asym(m, n):
    """Asymmetric integer-valued matrix: catches row/col swaps and k-permutation mismatches exactly."""
    i = torch.arange(m, dtype=torch.float32).unsqueeze(1)
    j = torch.arange(n, dtype=torch.float32).unsqueeze(0)
    return ((3 * i + 5 * j) % 7) - 3.0 + ((i * j) % 3)


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (64, 64, 64), (33, 47, 19), (200, 136, 100), (512, 256, 96)])
def test_gemm_exact_integers(ops, layout, M, N, K):
    a = _asym(M, K)
    b = _asym(K, N) + 1.0
    ref = a.double() @ b.double()
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    out = ops.gemm(layout, dev(A), dev(Bm))
    assert torch.equal(out.cpu().double(), ref), f"max err {(out.cpu().double() - ref).abs().max()}"


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize(
    "M,N,K,splitk",
    [(512, 1024, 2000, 0), (512, 512, 1024, 0), (96, 257, 300, 1), (130, 70, 1000, 5), (512, 2000, 512, 0),
     (1024, 1200, 512, 1), (64, 48, 33, 0), (8, 24, 64, 0),
     # K tail slab (K % 32 != 0 over 16-byte-regular operands: whole k-tiles on the pipelined kernel + one tail slab)
     (512, 1024, 3000, 0), (512, 1024, 30000, 0), (256, 512, 2012, 4), (128, 128, 100, 3), (128, 128, 68, 2)],
)
def test_gemm_random(ops, layout, M, N, K, splitk):
    a, b = rnd(M, K, seed=1), rnd(K, N, seed=2)
    bias = rnd(N, seed=3)
    ref = a.double() @ b.double() + bias.double()
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    out = ops.gemm(layout, dev(A), dev(Bm), bias=dev(bias), splitk=splitk)
    assert rel_l2(out, ref) < 2e-6
    out = ops.gemm(layout, dev(A), dev(Bm), bias=dev(bias), relu=True, splitk=splitk)
    assert rel_l2(out, ref.clamp_min(0)) < 2e-6
    c0 = rnd(M, N, seed=4)
    out = ops.gemm(layout, dev(A), dev(Bm), out=dev(c0), alpha=-0.5, accumulate=True, splitk=splitk)
    assert rel_l2(out, -0.5 * (a.double() @ b.double()) + c0.double()) < 2e-6


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(508, 19996, 72), (19996, 508, 72), (1024, 20000, 64), (20000, 1024, 40)])
def test_gemm_wide_and_tall_tiles(ops, layout, M, N, K):
    """Chip-filling outputs where the planner picks the 128x160 / 160x128 bf16x3 tiles (fewer rounds of resident
    workgroups): ragged last tiles, a K tail (K % 32 != 0) and the DPP-transposed rows 128..159 of RC operands."""
    a, b = _asym(M, K), _asym(K, N) + 1.0
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    out = ops.gemm(layout, dev(A), dev(Bm), splitk=1)
    ref = dev(a).double() @ dev(b).double()
    assert torch.equal(out.double(), ref), f"max err {(out.double() - ref).abs().max()}"
    a, b = rnd(M, K, seed=11), rnd(K, N, seed=12)
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    out = ops.gemm(layout, dev(A), dev(Bm), splitk=1)
    ref = dev(a).double() @ dev(b).double()
    assert rel_l2(out.cpu(), ref.cpu()) < 2e-6


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(1024, 20000, 512), (20000, 1024, 512), (512, 20000, 1024), (1000, 19996, 96),
                                   (2048, 5000, 64), (5120, 8000, 32), (300, 30000, 128)])
def test_gemm_wave_specialised_kernel(ops, layout, M, N, K, monkeypatch):
    """The wave-specialised bf16x3 kernel (one 512-thread workgroup per CU: 4 multiplier + 4 stager wavefronts, double
    buffered LDS, persistent over its work items): shapes that select its 256x160 / 160x256 / 256x128 tiles, with
    whole and ragged edge tiles, one and several items per workgroup, against fp64 -- and bit for bit against the
    2 x 4-wave kernel's sum order is NOT required (different accumulation split), only fp32-GEMM accuracy."""
    import ctypes

    from mmvae_amd import _lib

    monkeypatch.setenv("MMVAE_X3W", "1")  # (off by default: level with the 2 x 4-wave kernel inside the step)
    t, sk = ctypes.c_int(), ctypes.c_int()
    a, b = rnd(M, K, seed=M + 7 * layout), rnd(K, N, seed=N + 3)
    ref = a.double() @ b.double()
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    out = ops.gemm(layout, dev(A), dev(Bm), splitk=1)
    assert rel_l2(out, ref) < 2e-6
    # exact on integer-valued operands (every bf16 piece and every partial sum is exact)
    ai, bi = _asym(M, K), _asym(K, N) + 1.0
    Ai = ai if layout != 2 else ai.t().contiguous()
    Bi = bi.t().contiguous() if layout == 0 else bi
    assert torch.equal(ops.gemm(layout, dev(Ai), dev(Bi), splitk=1).cpu().double(), ai.double() @ bi.double())


@pytest.mark.parametrize("layout", [0, 1])
def test_gemm_wave_specialised_split_slabs(ops, layout, monkeypatch):
    """Split-K slabs of the K = G reductions (enc-L1 forward NT, dec-L2 input gradient NN): 16 slices x 16 tiles of
    256x128 = one workgroup per CU; the slabs must add up to the product."""
    monkeypatch.setenv("MMVAE_X3W", "1")
    M, N, K = 512, 1024, 20000
    a, b = rnd(M, K, seed=1), rnd(K, N, seed=2, scale=0.05)
    Bm = b.t().contiguous() if layout == 0 else b
    slabs = ops.gemm_slabs(layout, dev(a), dev(Bm))
    assert slabs.shape[0] >= 8
    assert rel_l2(slabs.double().sum(0), a.double() @ b.double()) < 2e-6


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_unaligned_strides(ops, layout):
    """Leading dimensions that are not multiples of 4 floats (e.g. 60530 / 52437-gene matrices)."""
    M, N, K = 70, 45, 131
    a, b = rnd(M, K, seed=5), rnd(K, N, seed=6)
    ref = a.double() @ b.double()
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b

    def pad(t, extra):  # view with ld = cols + extra
        buf = torch.zeros(t.shape[0], t.shape[1] + extra)
        buf[:, : t.shape[1]] = t
        return dev(buf)[:, : t.shape[1]]

    out = ops.gemm(layout, pad(A, 3), pad(Bm, 1))
    assert rel_l2(out, ref) < 2e-6
    outbuf = torch.zeros(M, N + 5, device="cuda")
    ops.gemm(layout, pad(A, 3), pad(Bm, 1), out=outbuf[:, :N])
    assert rel_l2(outbuf[:, :N], ref) < 2e-6
    assert float(outbuf[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("layout", [0, 1, 2])
def test_gemm_raw_slabs(ops, layout):
    M, N, K = 100, 72, 700
    a, b = rnd(M, K, seed=7), rnd(K, N, seed=8)
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    slabs = ops.gemm_slabs(layout, dev(A), dev(Bm), splitk=6)
    assert slabs.shape == (6, M, N)
    assert rel_l2(slabs.sum(0), a.double() @ b.double()) < 2e-6


@pytest.mark.parametrize("layout", [0, 1, 2])
@pytest.mark.parametrize("M,N,K", [(512, 1024, 30000), (512, 1024, 20016), (512, 1024, 60532), (5120, 1024, 3000),
                                   (2560, 1024, 30002)])
def test_gemm_raw_slabs_with_k_tail(ops, layout, M, N, K):
    """The engine's K = G reductions (encoder forward, decoder dX) at gene counts that are not a multiple of the k-tile:
    the planner's slab count includes the tail slab; slabs sum to the product; the tail slab holds only the last
    K % 32 columns' contribution."""
    a, b = rnd(M, K, seed=7, scale=0.1), rnd(K, N, seed=8, scale=0.1)
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    slabs = ops.gemm_slabs(layout, dev(A), dev(Bm), splitk=0)
    assert rel_l2(slabs.sum(0), a.double() @ b.double()) < 2e-6
    km = K // 32 * 32
    assert rel_l2(slabs[-1], a[:, km:].double() @ b[km:].double()) < 2e-6


@pytest.mark.parametrize("R,B,G,H", [(8, 8, 64, 48), (33, 33, 257, 72), (128, 128, 2000, 256), (66, 33, 131, 40),
                                     (512, 512, 19996, 72), (1024, 512, 20000, 64)])
def test_decoder_recon(ops, R, B, G, H):
    h, W, bias = rnd(R, H, seed=1), rnd(G, H, seed=2, scale=0.2), rnd(G, seed=3, scale=0.1)
    x = rnd(B, G, seed=4).abs()
    P = h.double() @ W.double().t() + bias.double()
    xh = P.clamp_min(0)
    xx = x.double().repeat(R // B, 1)
    d = xh - xx
    xhat, dP, se_part = ops.decoder_recon(dev(h), dev(W), dev(bias), dev(x))
    assert rel_l2(xhat, xh) < 2e-6
    assert rel_l2(se_part.sum(0), (d * d).sum(1)) < 1e-5
    # dP = 2 d 1[P>0]; entries with |P| within rounding of 0 may flip the mask -> compare where P is clearly signed
    ref_dp = 2 * d * (P > 0)
    clear = P.abs() > 1e-4
    assert rel_l2(dP.cpu().double()[clear], ref_dp[clear]) < 1e-5
    # optional outputs off
    _, _, se2 = ops.decoder_recon(dev(h), dev(W), dev(bias), dev(x), want_xhat=False, want_dP=False)
    assert torch.equal(se2, se_part)
    # bias-gradient partials out of the same epilogue: column sums of the dP it stored, per 128-row tile
    col_part = torch.full((ops.recon_row_tiles(R), G), float("nan"), device="cuda")
    _, dP2, se4 = ops.decoder_recon(dev(h), dev(W), dev(bias), dev(x), col_part=col_part)
    assert torch.equal(dP2, dP) and torch.equal(se4, se_part)
    for t in range(col_part.shape[0]):
        want = dP[128 * t:128 * (t + 1)].double().sum(0)
        assert rel_l2(col_part[t], want) < 1e-6, t
    # every row of se_part is defined by the call (tile rows the chosen tiling does not use are written as zeros)
    poisoned = torch.full((ops.recon_tiles(G), R), float("nan"), device="cuda")
    _, _, se3 = ops.decoder_recon(dev(h), dev(W), dev(bias), dev(x), want_xhat=False, want_dP=False, se_part=poisoned)
    assert torch.equal(se3, se_part)


@pytest.mark.parametrize("R,B,G,H", [(512, 512, 20000, 1000), (512, 512, 20000, 1016), (256, 128, 5008, 72)])
def test_decoder_recon_hidden_width_off_the_k_tile(ops, R, B, G, H):
    """A hidden width that is not a multiple of 32 on the pipelined kernels (mmvae_recon_set_h_kpad): h in a buffer padded
    with zero columns, W's rows read on into the next row (the last one into readable slack) -- same results as the
    guarded loop (to rounding) and as fp64."""
    h, bias = rnd(R, H, seed=1), rnd(G, seed=3, scale=0.1)
    x = rnd(B, G, seed=4).abs()
    Hp = (H + 31) // 32 * 32
    Wbuf = torch.zeros(G * H + 32)                      # an arena: 32 readable floats behind the last row
    Wbuf[:G * H] = rnd(G, H, seed=2, scale=0.2).reshape(-1)
    Wbuf[G * H:] = 3.0                                  # (finite garbage there must not reach the result)
    Wd = dev(Wbuf)[:G * H].view(G, H)
    hpad = torch.zeros(R, Hp, device="cuda")
    hpad[:, :H] = dev(h)
    P = h.double() @ Wbuf[:G * H].view(G, H).double().t() + bias.double()
    xh = P.clamp_min(0)
    d = xh - x.double().repeat(R // B, 1)
    col_part = torch.zeros(ops.recon_row_tiles(R), G, device="cuda")
    xhat, dP, se_part = ops.decoder_recon(hpad[:, :H], Wd, dev(bias), dev(x), col_part=col_part, h_kpad=True)
    assert rel_l2(xhat, xh) < 2e-6
    assert rel_l2(se_part.sum(0), (d * d).sum(1)) < 1e-5
    clear = P.abs() > 1e-4
    assert rel_l2(dP.cpu().double()[clear], (2 * d * (P > 0))[clear]) < 1e-5
    assert rel_l2(col_part.sum(0), dP.double().sum(0)) < 1e-6
    xhat_g, _, se_g = ops.decoder_recon(dev(h), Wd, dev(bias), dev(x))  # the guarded loop
    assert rel_l2(xhat, xhat_g) < 2e-6 and rel_l2(se_part.sum(0), se_g.sum(0)) < 1e-5


# --------------------------------------------------------------------------------------------------- FC epilogues
@pytest.mark.parametrize("B,N,S", [(8, 48, 1), (33, 72, 3), (512, 1024, 4), (128, 100, 1)])
@pytest.mark.parametrize("has_bn", [True, False])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_fc_epilogue_fwd_bwd(ops, B, N, S, has_bn, p):
    slabs = rnd(S, B, N, seed=1)
    bias = rnd(N, seed=2, scale=0.1)
    gamma, beta = 1 + rnd(N, seed=3, scale=0.1), rnd(N, seed=4, scale=0.1)
    rm, rv = rnd(N, seed=5, scale=0.1), 1 + rnd(N, seed=6, scale=0.1).abs()
    mask = (torch.rand(B, N, generator=torch.Generator().manual_seed(7)) >= p).to(torch.uint8) if p > 0 else None
    dd = rnd(2, B, N, seed=8)
    addend = rnd(B, N, seed=9)
    addend_a = rnd(B, N, seed=10)  # gradient on the pre-dropout activation (a hidden representation): bypasses the mask
    # ---- CPU reference with autograd
    sl = slabs.clone().requires_grad_(True)
    bi, ga, be = (t.clone().requires_grad_(True) for t in (bias, gamma, beta))
    z = sl.sum(0) + bi
    if has_bn:
        mean, var = z.mean(0), z.var(0, unbiased=False)
        y = (z - mean) / torch.sqrt(var + 1e-3) * ga + be
    else:
        y = z
    a = torch.relu(y)
    d = a * mask.float() / (1 - p) if mask is not None else a
    ((d * (dd.sum(0) + addend)).sum() + (a * addend_a).sum()).backward()
    # ---- HIP
    bn = None
    rm_d, rv_d = dev(rm.clone()), dev(rv.clone())
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    if has_bn:
        bn = dict(gamma=dev(gamma), beta=dev(beta), running_mean=rm_d, running_var=rv_d, num_batches_tracked=nbt,
                  momentum=0.01, eps=1e-3)
    f = ops.fc_epilogue_fwd(dev(slabs), dev(bias), bn=bn, training=True, relu=True,
                            keep_mask=dev(mask) if mask is not None else None, dropout_p=p)
    torch.testing.assert_close(f["a"].cpu(), a.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(f["d"].cpu(), d.detach(), rtol=1e-5, atol=1e-5)
    if has_bn:
        torch.testing.assert_close(f["z"].cpu(), z.detach(), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(rm_d.cpu(), 0.99 * rm + 0.01 * mean.detach(), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(rv_d.cpu(), 0.99 * rv + 0.01 * z.detach().var(0, unbiased=True), rtol=1e-5, atol=1e-6)
        assert int(nbt) == 1
    dz, dbias, dgamma, dbeta = ops.fc_epilogue_bwd(
        dev(dd), addend=dev(addend), addend_a=dev(addend_a), keep_mask=dev(mask) if mask is not None else None,
        dropout_p=p, relu=True,
        a=f["a"], z=f["z"], gamma=dev(gamma) if has_bn else None, mean=f["mean"], invstd=f["invstd"], has_bn=has_bn)
    assert rel_l2(dz, sl.grad[0]) < 2e-5
    if has_bn:
        assert rel_l2(dgamma, ga.grad) < 2e-5
        assert rel_l2(dbeta, be.grad) < 2e-5
        assert float(dbias.abs().max()) < 1e-3 * float(dd.abs().max()) * B  # exactly 0 in exact arithmetic
    else:
        assert rel_l2(dbias, bi.grad) < 2e-5
    # ---- eval mode (running stats)
    if has_bn:
        fe = ops.fc_epilogue_fwd(dev(slabs), dev(bias), bn=dict(bn, running_mean=dev(rm), running_var=dev(rv)),
                                 training=False, relu=True)
        ye = torch.relu((z.detach() - rm) / torch.sqrt(rv + 1e-3) * gamma + beta)
        torch.testing.assert_close(fe["d"].cpu(), ye, rtol=1e-5, atol=1e-5)


def test_fc_epilogue_bwd_row_scale_and_colsum(ops):
    B, N = 40, 70
    dd, rs = rnd(B, N, seed=1), rnd(B, seed=2)
    dz, dbias, _, _ = ops.fc_epilogue_bwd(dev(dd), row_scale=dev(rs))
    ref = dd * rs.unsqueeze(1)
    assert rel_l2(dz, ref) < 1e-6
    assert rel_l2(dbias, ref.double().sum(0)) < 1e-5
    dz2, dbias2, _, _ = ops.fc_epilogue_bwd(dev(dd), want_dz=False)
    assert dz2 is None and rel_l2(dbias2, dd.double().sum(0)) < 1e-5


@pytest.mark.parametrize("B,N", [(7, 128), (33, 100), (64, 768)])
def test_layernorm(ops, B, N):
    x = rnd(B, N, seed=1).requires_grad_(True)
    y = torch.nn.functional.layer_norm(x, (N,))
    g = rnd(B, N, seed=2)
    (y * g).sum().backward()
    yd, invstd = ops.layernorm_fwd(dev(x.detach()))
    torch.testing.assert_close(yd.cpu(), y.detach(), rtol=1e-5, atol=1e-5)
    dx = ops.layernorm_bwd(dev(g), yd, invstd)
    assert rel_l2(dx, x.grad) < 2e-5


@pytest.mark.parametrize("bn,relu,p,hidden", [(False, True, 0.0, False), (False, True, 0.2, True), (True, True, 0.1, True),
                                              (False, False, 0.3, False), (True, False, 0.0, False)])
def test_fcblock_layernorm_then_relu_and_dropout(ops, bn, relu, p, hidden):
    """Linear -> [BN] -> LayerNorm(no affine) -> [ReLU] -> [Dropout] (components.py:279-288, the core of
    configs/model/configV3.yaml:25-36) on the HIP module path, forward and backward, against the same stack of torch
    modules in fp64 with the same keep masks; the hidden representation is the post-activation, pre-dropout tensor."""
    import torch.nn as nn

    from mmvae_amd.modules.base import FCBlock, FCBlockConfig

    B, dims = 33, [40, 56, 24]
    cfg = FCBlockConfig(layers=dims, dropout_rate=p, use_batch_norm=bn, use_layer_norm=True,
                        activation_fn=nn.ReLU if relu else None, return_hidden=hidden)
    torch.manual_seed(3)
    blk = FCBlock(cfg)
    ref = FCBlock(FCBlockConfig(layers=dims, dropout_rate=p, use_batch_norm=bn, use_layer_norm=True,
                                activation_fn=nn.ReLU if relu else None, return_hidden=hidden)).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in blk.state_dict().items()})
    blk = blk.cuda().train()
    ref.train()
    x = rnd(B, dims[0], seed=5)
    masks = {i: (torch.rand(B, n, generator=torch.Generator().manual_seed(9 + i)) >= p).to(torch.uint8)
             for i, n in enumerate(dims[1:])} if p > 0 else None
    # fp64 reference: the reference's own module order, dropout applied with the explicit masks
    xr = x.double().requires_grad_(True)
    h, hid_ref = xr, []
    for i, layer in enumerate(ref.fc_layers):
        for name, sub in layer.named_children():
            h = h * masks[i].double() / (1 - p) if name == "dr" else sub(h)
            if name == "af" and hidden:
                hid_ref.append(h)
    gy = rnd(B, dims[-1], seed=6)
    gh = [rnd(B, n, seed=20 + i) for i, n in enumerate(dims[1:])] if (hidden and relu) else []
    (h * gy.double()).sum().add(sum((a * g.double()).sum() for a, g in zip(hid_ref, gh))).backward()
    # HIP module path
    xd = dev(x).requires_grad_(True)
    blk.explicit_masks = {i: dev(m) for i, m in masks.items()} if masks else None
    out = blk(xd)
    y, hid = (out if isinstance(out, tuple) else (out, []))
    assert rel_l2(y, h.detach()) < 2e-6
    assert len(hid) == len(hid_ref)
    for a, b in zip(hid, hid_ref):
        assert rel_l2(a, b.detach()) < 2e-6
    (y * dev(gy)).sum().add(sum((a * dev(g)).sum() for a, g in zip(hid, gh))).backward()
    assert rel_l2(xd.grad, xr.grad) < 2e-5
    for (n, pd_), (_, pr) in zip(blk.named_parameters(), ref.named_parameters()):
        if bn and n.endswith("lin.bias"):
            continue  # feeds a BatchNorm: exactly-zero true gradient
        assert rel_l2(pd_.grad, pr.grad) < 2e-5, n


# ------------------------------------------------------------------------------------------------ reparam / losses
@pytest.mark.parametrize("B,Z,K", [(8, 8, 1), (33, 10, 1), (512, 128, 1), (16, 128, 3), (5, 200, 2)])
def test_reparam_kl(ops, B, Z, K):
    mu = rnd(B, Z, seed=1).requires_grad_(True)
    a = rnd(B, Z, seed=2, scale=0.5).requires_grad_(True)
    eps = rnd(K, B, Z, seed=3)
    var = torch.exp(a) + 1e-4
    std = var.sqrt()
    z = mu + std * eps
    klr = (0.5 * (std**2 + mu**2 - 1 - (std**2).log())).sum(-1)
    gz, gkl = rnd(K, B, Z, seed=4), rnd(B, seed=5)
    ((z * gz).sum() + 0.37 * (klr * gkl).sum()).backward()
    sd, zd, kl_row, stat = ops.reparam_kl_fwd(dev(mu.detach()), dev(a.detach()), dev(eps))
    torch.testing.assert_close(zd.cpu(), z.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(sd.cpu(), std.detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(kl_row.cpu(), klr.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stat.cpu()[0], mu.detach().sum(1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(stat.cpu()[1], var.detach().sum(1), rtol=1e-5, atol=1e-5)
    dmu, da = ops.reparam_kl_bwd(dev(mu.detach()), sd, dev(eps), dev(gz), dkl_row=dev(gkl), kl_scale=0.37)
    assert rel_l2(dmu, mu.grad) < 1e-5
    assert rel_l2(da, a.grad) < 1e-5
    kdev = torch.tensor([0.37], device="cuda")
    dmu2, da2 = ops.reparam_kl_bwd(dev(mu.detach()), sd, dev(eps), dev(gz), dkl_row=dev(gkl), kl_scale_dev=kdev)
    assert torch.equal(dmu2, dmu) and torch.equal(da2, da)


@pytest.mark.parametrize("B,G", [(8, 64), (33, 257), (128, 2000)])
def test_mse_sum(ops, B, G):
    xhat, x = rnd(B, G, seed=1), rnd(B, G, seed=2)
    se, dx = ops.mse_sum_fwd_bwd(dev(xhat), dev(x), gscale=0.5)
    d = xhat.double() - x.double()
    assert rel_l2(se, (d * d).sum(1)) < 1e-6
    assert rel_l2(dx, d) < 1e-6  # 0.5 * 2 d


@pytest.mark.parametrize("K", [1, 3, 10])
def test_elbo_finalize(ops, K):
    B, T, Z = 37, 5, 16
    se_part = rnd(T, K * B, seed=1).abs() * 3
    kl_row = rnd(B, seed=2).abs()
    stat = rnd(2, B, seed=3)
    out, w = ops.elbo_finalize(dev(se_part), dev(kl_row), dev(stat), B=B, K=K, Z=Z, kl_weight=0.8, want_w=True)
    se = se_part.double().sum(0).reshape(K, B)
    if K == 1:
        recon = se.sum()
        wref = torch.ones(B)
    else:
        recon = (-(torch.logsumexp(-se, 0) - np.log(K))).sum()
        wref = torch.softmax(-se, 0).reshape(-1)
    kl = kl_row.double().mean()
    o = out.cpu().double()
    assert abs(float(o[1]) - float(recon)) <= 1e-5 * abs(float(recon))
    assert abs(float(o[2]) - float(kl)) <= 1e-6 * abs(float(kl))
    assert abs(float(o[0]) - float(recon + 0.8 * kl)) <= 1e-5 * abs(float(recon))
    assert abs(float(o[3]) - 0.8) < 1e-7
    assert abs(float(o[4]) - float(stat[0].double().sum() / (B * Z))) < 1e-6
    assert abs(float(o[5]) - float(stat[1].double().sum() / (B * Z))) < 1e-6
    torch.testing.assert_close(w.cpu().double(), wref.double(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B,C", [(16, 2), (33, 8), (64, 273), (32, 4644), (512, 4644), (33, 1027), (16, 8192), (7, 8193)])
def test_cross_entropy(ops, B, C):
    logits = rnd(B, C, seed=1, scale=2.0).requires_grad_(True)
    y = torch.randint(0, C, (B,), generator=torch.Generator().manual_seed(2))
    loss = torch.nn.functional.cross_entropy(logits, y, reduction="sum")
    loss.backward()
    rows, dl = ops.cross_entropy_sum(dev(logits.detach()), dev(y), gscale=-25.0)
    tot = ops.sum_f32(rows)
    assert abs(float(tot) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach()))
    assert rel_l2(dl, -25.0 * logits.grad) < 1e-5
    gdev = torch.tensor(-5.0, device="cuda")
    _, dl2 = ops.cross_entropy_sum(dev(logits.detach()), dev(y), gscale=5.0, gscale_dev=gdev)
    assert rel_l2(dl2, -25.0 * logits.grad) < 1e-5


def test_clip_adam_matches_torch(ops):
    n = 70001
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2, scale=3.0), rnd(n, seed=3, scale=0.001)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=5e-3, weight_decay=1e-6)
    p, m, v = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = torch.zeros(8, device="cuda")
    partials = torch.empty(ops.sqnorm_partials(n), device="cuda")
    for g in (g1, g2):
        pt.grad = g.clone()
        norm = torch.nn.utils.clip_grad_norm_([pt], 10.0)
        opt.step()
        ops.clip_adam_step(p, dev(g), m, v, state, partials, max_norm=10.0)
        assert abs(float(state[1]) - float(norm)) <= 1e-5 * float(norm)
        assert rel_l2(p, pt.detach()) < 1e-6
    assert float(state[0]) == 2.0
    # torch's clip coefficient comes from an fp32 norm; ours from an fp64-accumulated one: v ~ clip^2
    assert rel_l2(m, opt.state[pt]["exp_avg"]) < 5e-5
    assert rel_l2(v, opt.state[pt]["exp_avg_sq"]) < 5e-5


def test_adam_confined_to_n_compute_units_is_bit_identical(ops):
    """mmvae_adam_set_workgroups: the elementwise update on N fat workgroups leaves the same bits as the chip-filling
    grid (odd length: vector body + scalar tail; the copy rider too)."""
    from mmvae_amd import _lib

    lib = _lib.load()
    n = 1_000_003
    s = torch.cuda.current_stream().cuda_stream
    state = torch.tensor([4.0, 0, 0.7, 0.9, 0.95, 0.0, 0, 0], device="cuda")  # step, norm, clip, bias corrections, clip value
    outs = []
    try:
        for wg in (0, 7, 64):
            assert lib.mmvae_adam_set_workgroups(wg) == 0 and lib.mmvae_adam_get_workgroups() == wg
            p, g = dev(rnd(n, seed=1)), dev(rnd(n, seed=2, scale=0.1))
            m, v = dev(rnd(n, seed=3, scale=0.01)), dev(rnd(n, seed=4, scale=0.01)).abs()
            src, dst = dev(rnd(256, seed=5)), torch.zeros(256, device="cuda")
            rc = lib.mmvae_adam_step_copy(n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(),
                                          5e-3, 0.9, 0.999, 1e-8, 1e-6, 0.5, 256, src.data_ptr(), dst.data_ptr(), s)
            assert rc == 0
            assert torch.equal(dst, src)
            outs.append((p, m, v))
    finally:
        lib.mmvae_adam_set_workgroups(0)
    assert lib.mmvae_adam_set_workgroups(257) != 0
    for p, m, v in outs[1:]:
        assert torch.equal(p, outs[0][0]) and torch.equal(m, outs[0][1]) and torch.equal(v, outs[0][2])


def test_norm_ranges_and_prepare_in_one_launch_equal_the_separate_launches(ops):
    """mmvae_grad_sqnorm_ranges_prepare: the partials of up to 4 arena ranges + adam_prepare by the workgroup that finishes
    last -- bit for bit what mmvae_grad_sqnorm per range followed by mmvae_adam_prepare leave (partials and state words),
    with partials a fused GEMM epilogue wrote earlier in front; the ticket word is back at zero (replayable)."""
    import ctypes as C

    from mmvae_amd import _lib

    lib = _lib.load()
    arena = dev(rnd(1_300_000, seed=5, scale=2.0))
    ranges = [(16, 200_000), (400_004, 65_536), (600_000, 3), (700_001, 555_555)]  # offsets / lengths in floats
    pre = dev(rnd(7, seed=6).abs())  # partials left by earlier kernels (slots 0..6)
    s = torch.cuda.current_stream().cuda_stream
    for nr in (1, 2, 4):
        rs = ranges[:nr]
        nparts = [lib.mmvae_sqnorm_partials(n) for _, n in rs]
        total = pre.numel() + sum(nparts)
        # separate launches
        p_ref = torch.zeros(total, device="cuda")
        p_ref[: pre.numel()] = pre
        slot = pre.numel()
        for (o, n), k in zip(rs, nparts):
            assert lib.mmvae_grad_sqnorm(n, arena.data_ptr() + 4 * o, p_ref.data_ptr() + 4 * slot, s) == 0
            slot += k
        st_ref = torch.tensor([3.0, 0, 0, 0, 0, 0, 0, 0], device="cuda")
        flags = _lib.PREPARE_NORM | _lib.PREPARE_ADVANCE
        assert lib.mmvae_adam_prepare(total, p_ref.data_ptr(), 10.0, 0.5, 0.9, 0.999, st_ref.data_ptr(), flags, s) == 0
        # one launch, twice (the second run proves the ticket was reset)
        for rep in range(2):
            p_got = torch.zeros(total, device="cuda")
            p_got[: pre.numel()] = pre
            st_got = torch.tensor([3.0, 0, 0, 0, 0, 0, 0, 0], device="cuda")
            ticket = torch.zeros(1, dtype=torch.int32, device="cuda") if rep == 0 else ticket
            gp = (C.c_void_p * nr)(*[arena.data_ptr() + 4 * o for o, _ in rs])
            ln = (C.c_int64 * nr)(*[n for _, n in rs])
            rc = lib.mmvae_grad_sqnorm_ranges_prepare(nr, C.addressof(gp), C.addressof(ln),
                                                      p_got.data_ptr() + 4 * pre.numel(), ticket.data_ptr(), total,
                                                      p_got.data_ptr(), 10.0, 0.5, 0.9, 0.999, st_got.data_ptr(),
                                                      _lib.PREPARE_ADVANCE, s)
            assert rc == 0
            torch.cuda.synchronize()
            assert torch.equal(p_got, p_ref), (nr, rep)
            assert torch.equal(st_got, st_ref), (nr, rep, st_got.tolist(), st_ref.tolist())
            assert int(ticket) == 0


def test_job_segments_pack_unpack_and_flags(ops):
    """mmvae_jobs_pack / _unpack (the exchange of the tensors that took part: segments laid out back to back in units of
    128 floats, position in the upper bits of the job's `reserved` word) and the flag semantics of the job kernels:
    reserved & 3 == 1 -> zeroed ahead of the exchange, == 2 -> zeroed and skipped by the norm / Adam job kernels."""
    import numpy as np

    from mmvae_amd import _lib
    from mmvae_amd.optim import HipAdam

    lib = _lib.load()
    s = torch.cuda.current_stream().cuda_stream
    arena = dev(rnd(200_000, seed=11))
    segs = [(16, 16384, 0), (40_000, 128, 1), (50_000, 7, 0), (60_000, 16384, 2), (90_004, 300, 0)]  # offset, len, flag
    jobs = np.zeros(len(segs), dtype=np.dtype(HipAdam.JOB_DTYPE))
    pos = 0
    for j, (o, n, f) in enumerate(segs):
        jobs[j]["offset"], jobs[j]["len"], jobs[j]["bc1"], jobs[j]["bc2"] = o, n, 0.1, 0.001
        jobs[j]["reserved"] = f | (pos << 2)
        pos += (n + 127) // 128
    jobs_dev = torch.frombuffer(bytearray(jobs.tobytes()), dtype=torch.uint8).cuda()
    staging = torch.full((pos * 128,), float("nan"), device="cuda")
    assert lib.mmvae_jobs_pack(len(segs), jobs_dev.data_ptr(), arena.data_ptr(), staging.data_ptr(), s) == 0
    p = 0
    for o, n, _ in segs:
        assert torch.equal(staging[p * 128: p * 128 + n], arena[o:o + n])
        pad = (n + 127) // 128 * 128 - n
        assert torch.equal(staging[p * 128 + n: p * 128 + n + pad], torch.zeros(pad, device="cuda"))
        p += (n + 127) // 128
    back = torch.zeros_like(arena)
    assert lib.mmvae_jobs_unpack(len(segs), jobs_dev.data_ptr(), back.data_ptr(), (2 * staging).data_ptr(), s) == 0
    touched = torch.zeros_like(arena, dtype=torch.bool)
    for o, n, _ in segs:
        assert torch.equal(back[o:o + n], 2 * arena[o:o + n])
        touched[o:o + n] = True
    assert float(back[~touched].abs().max()) == 0.0
    # flags
    g = arena.clone()
    assert lib.mmvae_grad_zero_flagged_jobs(len(segs), jobs_dev.data_ptr(), g.data_ptr(), s) == 0
    for o, n, f in segs:
        assert torch.equal(g[o:o + n], torch.zeros(n, device="cuda") if f else arena[o:o + n])
    partials = torch.full((len(segs),), float("nan"), device="cuda")
    assert lib.mmvae_grad_sqnorm_jobs(len(segs), jobs_dev.data_ptr(), arena.data_ptr(), partials.data_ptr(), s) == 0
    for j, (o, n, f) in enumerate(segs):
        want = 0.0 if f == 2 else float((arena[o:o + n].double() ** 2).sum())
        assert abs(float(partials[j]) - want) <= 1e-5 * max(want, 1.0), j


def test_philox_streams(ops):
    rng = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    m1 = ops.philox_keep_mask((512, 1024), 0.1, rng)
    m2 = ops.philox_keep_mask((512, 1024), 0.1, rng)
    assert int(rng[1]) == 2 * (512 * 1024 // 4)
    assert not torch.equal(m1, m2)
    assert abs(float(m1.float().mean()) - 0.9) < 3e-3
    e = ops.philox_normal((10, 512, 128), rng)
    assert abs(float(e.mean())) < 5e-3 and abs(float(e.std()) - 1.0) < 5e-3
    assert abs(float((e**4).mean()) - 3.0) < 0.1
    rng2 = torch.tensor([1234, 0], dtype=torch.int64, device="cuda")
    assert torch.equal(ops.philox_keep_mask((512, 1024), 0.1, rng2), m1)  # reproducible from (seed, offset)


def test_philox_fill_jobs_gives_the_numbers_of_single_fills(ops):
    """mmvae_philox_fill_jobs (the keep-masks and the noise of one step in one launch) == one fill per launch."""
    from mmvae_amd import _lib

    rng = torch.tensor([1234, 77], dtype=torch.int64, device="cuda")
    shapes = [((512, 1024), 0.1, 5), ((512, 512), 0.25, 6), ((33, 7), 0.5, 9)]
    want = [ops.philox_keep_mask(sh, p, rng, stream_id=sid, advance=False) for sh, p, sid in shapes]
    want_n = ops.philox_normal((3, 512, 128), rng, stream_id=2, advance=False)
    got = [torch.zeros(sh, dtype=torch.uint8, device="cuda") for sh, _, _ in shapes]
    got_n = torch.zeros((3, 512, 128), device="cuda")
    jobs = [_lib.PhiloxJob(g.data_ptr(), g.numel(), sid, p, 0) for g, (_, p, sid) in zip(got, shapes)]
    jobs.append(_lib.PhiloxJob(got_n.data_ptr(), got_n.numel(), 2, 0.0, 1))
    arr = (_lib.PhiloxJob * len(jobs))(*jobs)
    jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    lib = _lib.load()
    _lib.check(lib.mmvae_philox_fill_jobs(len(jobs), jobs_dev.data_ptr(), max(j.n for j in jobs), rng.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "mmvae_philox_fill_jobs")
    torch.cuda.synchronize()
    for g, w in zip(got, want):
        assert torch.equal(g, w)
    assert torch.equal(got_n, want_n)
    assert rng.tolist() == [1234, 77]  # no advance


def test_small_utils(ops):
    x, y = rnd(1000, seed=1), rnd(1000, seed=2)
    out = ops.axpby(2.0, dev(x), -0.5, dev(y.clone()))
    torch.testing.assert_close(out.cpu(), 2.0 * x - 0.5 * y)
    a, rs = rnd(9, 33, seed=3), rnd(9, seed=4)
    torch.testing.assert_close(ops.scale_rows(dev(a), dev(rs)).cpu(), a * rs.unsqueeze(1))


@pytest.mark.parametrize("B,N,ld", [(5120, 20000, 20000), (300, 1003, 1004), (7, 5, 8), (256, 256, 256), (513, 260, 272), (70, 33, 33)])
def test_weighted_colsum(ops, B, N, ld):
    """w^T x of a wide matrix in one read-only pass (the K-sample bound's decoder-bias gradient): chunk partials whose
    sum matches fp64; identical on a second run (fixed summation order)."""
    full = rnd(B, ld, seed=5)
    x = dev(full)[:, :N]
    w = dev(torch.rand(B, generator=torch.Generator().manual_seed(6)) / B)
    parts = ops.weighted_colsum(x, w)
    assert parts.shape == ((B + 255) // 256, N)
    want = (w.double().cpu() @ full[:, :N].double())
    got = parts.double().sum(0).cpu()
    assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max() + (full[:, :N].abs().double().T @ w.double().cpu()).max())
    assert torch.equal(parts, ops.weighted_colsum(x, w))


def test_bad_arguments_are_rejected_on_host(ops):
    from mmvae_amd import _lib

    a, b = dev(rnd(8, 16)), dev(rnd(8, 12))
    with pytest.raises(ValueError):
        ops.gemm(0, a, b)  # K mismatch
    with pytest.raises(_lib.HipLibraryError):
        ops.gemm(0, rnd(8, 16), rnd(8, 16))  # CPU tensors: no fallback
    lib = _lib.load()
    assert lib.mmvae_gemm_f32(7, 8, 8, 8, 1.0, a.data_ptr(), 16, a.data_ptr(), 16, a.data_ptr(), 8, None, 0, 1, None, 0,
                              None) == _lib.ERR_ARG


# ------------------------------------------------------------------------------------------ batched partial sums
def test_sum_parts_batch(ops):
    """One launch, several fixed-order reductions of different shapes (split-K slabs, column partials), with and
    without accumulate / alpha, aligned and odd sizes."""
    jobs, refs = [], []
    for k, (P, R, Cn, alpha, acc) in enumerate([(4, 64, 128, 1.0, False), (3, 33, 47, 0.5, True), (16, 1, 20000, 1.0, False),
                                                (1, 8, 12, 2.0, False), (7, 1, 5, 1.0, True)]):
        src = dev(rnd(P, R, Cn, seed=20 + k))
        base = dev(rnd(R, Cn + 3, seed=40 + k))
        dst = base[:, :Cn] if k % 2 else dev(rnd(R, Cn, seed=60 + k))
        before = dst.clone()
        s = torch.zeros(R, Cn, device="cuda")
        for p in range(P):  # same order, same fp32 arithmetic: bitwise
            s = s + src[p]
        ref = s * alpha + (before if acc else 0)
        jobs.append((src, dst, alpha, acc))
        refs.append((dst, ref, alpha == 1.0 and not acc))
    ops.sum_parts_batch(jobs)
    for dst, ref, exact in refs:  # alpha / accumulate may be contracted into one fma by the compiler
        assert torch.equal(dst, ref) if exact else torch.allclose(dst, ref, rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------------------------------------- grouped small GEMMs
def test_gemm_batch(ops):
    """One launch, several independent GEMMs of all three layouts and different shapes, with bias / relu / alpha /
    accumulate epilogues and strided outputs; exact on integer data, rel-L2 <= 2e-6 vs fp64 on random data."""
    from mmvae_amd import _lib

    shapes = [(2, 512, 1024, 512), (2, 128, 256, 512), (0, 512, 128, 256), (1, 512, 256, 128), (2, 64, 64, 8),
              (2, 1024, 512, 512), (0, 36, 44, 100), (1, 260, 68, 36)]
    for exact in (True, False):
        jobs, refs = [], []
        for k, (layout, M, N, K) in enumerate(shapes):
            a = _asym(M, K) if exact else rnd(M, K, seed=100 + k)
            b = (_asym(K, N) + 1.0) if exact else rnd(K, N, seed=200 + k)
            A = a if layout != 2 else a.t().contiguous()
            Bm = b.t().contiguous() if layout == 0 else b
            bias = None if (exact or k % 2) else rnd(N, seed=300 + k)
            acc = (k % 3 == 0) and not exact
            alpha = 1.0 if exact else (1.0, -0.5, 2.0)[k % 3]
            relu = (k % 4 == 1) and not exact
            base = dev(rnd(M, N + 4, seed=400 + k))
            out = base[:, :N]  # leading dimension N + 4
            c0 = out.clone().cpu().double()
            ref = alpha * (a.double() @ b.double())
            if bias is not None:
                ref = ref + bias.double()
            if acc:
                ref = ref + c0
            if relu:
                ref = ref.clamp_min(0)
            jobs.append(dict(layout=layout, a=dev(A), b=dev(Bm), out=out, bias=None if bias is None else dev(bias),
                             alpha=alpha, relu=relu, accumulate=acc))
            refs.append((out, ref, base, N))
        ops.gemm_batch(jobs)
        for out, ref, base, N in refs:
            if exact:
                assert torch.equal(out.cpu().double(), ref)
            else:
                assert rel_l2(out, ref) < 2e-6
    # requirements are checked on the host: an odd leading dimension is refused before anything is launched
    a = dev(rnd(64, 66))[:, :64]
    bad = _lib.GemmJob(a.data_ptr(), a.data_ptr(), a.data_ptr(), None, 66, 66, 66, 0, 64, 64, 64, 1.0, 0, 0, 0)
    import ctypes as C
    assert _lib.load().mmvae_gemm_batch_job_ok(C.addressof(bad)) == 0


# ------------------------------------------------------------------------------------------------ CSR -> dense (f1)
@pytest.mark.parametrize("B,G,density", [(33, 257, 0.1), (8, 20000, 0.09), (1, 5, 1.0), (16, 8200, 0.0), (64, 16384, 0.3)])
def test_csr_to_dense(ops, B, G, density):
    """Bit-exact against torch's own to_dense on the CPU; empty rows, full rows, row strides with padding."""
    g = torch.Generator().manual_seed(B * 1000 + G)
    d = torch.rand(B, G, generator=g)
    d = torch.where(torch.rand(B, G, generator=g) < density, d, torch.zeros(()))
    if B > 2:
        d[1] = 0  # an empty row
    x = d.to_sparse_csr()
    out = ops.csr_to_dense(x.cuda())
    assert torch.equal(out.cpu(), d)
    base = torch.full((B, G + 3), 7.0, device="cuda")
    ops.csr_to_dense(x.cuda(), out=base[:, :G])  # leading dimension that is not a multiple of 4
    assert torch.equal(base[:, :G].cpu(), d) and bool((base[:, G:] == 7.0).all())
    # int32 index arrays: what a torch.sparse_csr_tensor built from a scipy slice carries (the reference's batches)
    x32 = torch.sparse_csr_tensor(x.crow_indices().int(), x.col_indices().int(), x.values(), size=x.shape).cuda()
    assert x32.col_indices().dtype == torch.int32
    base.fill_(7.0)
    ops.csr_to_dense(x32, out=base[:, :G])
    assert torch.equal(base[:, :G].cpu(), d) and bool((base[:, G:] == 7.0).all())


@pytest.mark.parametrize("B,G,N,density", [(33, 257, 72, 0.1), (64, 20000, 1024, 0.08), (7, 1000, 2048, 0.0), (16, 5000, 1028, 0.3)])
def test_csr_spmm_against_dense(ops, B, G, N, density):
    """SURVEY 8 f1 measurement kernel: the first layer's product straight from the CSR batch (gather of transposed
    weight rows) equals the dense product of the densified batch (fp64 reference, rel-L2 <= 2e-6); empty rows give the bias."""
    g = torch.Generator().manual_seed(B + G)
    d = torch.rand(B, G, generator=g)
    d = torch.where(d < density, torch.rand(B, G, generator=g) * 9.0, torch.zeros(()))
    if B > 2:
        d[1] = 0.0  # an empty row
    w, bias = rnd(N, G, seed=3, scale=0.05), rnd(N, seed=4, scale=0.1)
    x = d.to_sparse_csr()
    x = torch.sparse_csr_tensor(x.crow_indices().int(), x.col_indices().int(), x.values(), size=(B, G)).cuda()
    y = ops.csr_spmm_wt(x, dev(w.t().contiguous()), dev(bias))
    ref = d.double() @ w.double().t() + bias.double()
    assert rel_l2(y, ref) < 2e-6
    if B > 2:
        assert torch.equal(y[1].cpu(), bias)


@pytest.mark.parametrize("layout,M,N,K", [(2, 1024, 20000, 64), (2, 20000, 1024, 32), (0, 512, 20000, 96),
                                           (2, 2048, 2004, 40), (1, 640, 20000, 64)])
def test_gemm_with_fused_sum_of_squares(ops, layout, M, N, K):
    """mmvae_gemm_f32_sq: same product as mmvae_gemm_f32 (bitwise) plus per-tile partial sums of squares of the stored
    values whose total is the squared Frobenius norm (rtol 1e-5: fp32 partials, any tile shape the planner picks)."""
    a, b = rnd(M, K, seed=11), rnd(K, N, seed=12)
    A = a if layout != 2 else a.t().contiguous()
    Bm = b.t().contiguous() if layout == 0 else b
    ref = ops.gemm(layout, dev(A), dev(Bm), alpha=0.5, splitk=1)
    out, parts = ops.gemm_sq(layout, dev(A), dev(Bm), alpha=0.5)
    assert torch.equal(out, ref)
    total = float(parts.double().sum())
    want = float((ref.double() ** 2).sum())
    assert abs(total - want) <= 1e-5 * want, (total, want)
    out2, parts2 = ops.gemm_sq(layout, dev(A), dev(Bm), alpha=0.5)
    assert torch.equal(parts, parts2), "partials must be bitwise reproducible"


# ------------------------------------------------------------------------------------------ conditional layers (f2)
@pytest.mark.parametrize("B,n_in,n_out,C", [(512, 128, 128, 37), (512, 128, 128, 1), (300, 128, 128, 3), (33, 8, 8, 5),
                                            (64, 24, 40, 200), (16, 300, 260, 3), (100, 130, 70, 2)])
def test_cond_linear_fwd_bwd(ops, B, n_in, n_out, C):
    """Per-cell conditional Linear against a per-condition torch loop (the reference's ConditionalLayer.forward):
    forward / dx to 1e-5, dW / db of the present conditions to 1e-5, absent conditions untouched."""
    g = torch.Generator().manual_seed(B + C)
    x = torch.randn(B, n_in, generator=g)
    dy = torch.randn(B, n_out, generator=g)
    cond = torch.randint(0, C, (B,), generator=g)
    # arena: blocks in shuffled order with gaps, weight and bias of a block not adjacent
    sizes = n_out * n_in + 12, n_out + 4
    perm = torch.randperm(C, generator=g).tolist()
    w_off, b_off, pos = [0] * C, [0] * C, 8
    for c in perm:
        w_off[c] = pos
        pos += sizes[0]
    for c in perm:
        b_off[c] = pos
        pos += sizes[1]
    params = torch.randn(pos, generator=g) * 0.3
    W = torch.stack([params[w_off[c]:w_off[c] + n_out * n_in].view(n_out, n_in) for c in range(C)])
    bias = torch.stack([params[b_off[c]:b_off[c] + n_out] for c in range(C)])
    y_ref = torch.einsum("boi,bi->bo", W[cond], x) + bias[cond]
    dx_ref = torch.einsum("boi,bo->bi", W[cond], dy)
    from mmvae_amd import cond_tables as CT

    t = CT.group_tables(cond.numpy().astype("int32"))
    present = torch.from_numpy(t["present"])
    tables = ops.cond_tables_to_device(t, "cuda")
    dv = lambda t_, dt: t_.to(dt).cuda()
    P, G = params.cuda(), torch.full((pos,), 7.0, device="cuda")
    wo, bo, cd = dv(torch.tensor(w_off), torch.int64), dv(torch.tensor(b_off), torch.int64), tables["cond"]
    y = ops.cond_linear_fwd(dev(x), P, wo, bo, cd, n_out)
    assert rel_l2(y, y_ref) < 1e-5
    # the condition-sorted forward (a shared block read once per 8 cells) does the same arithmetic: same bits
    y_sorted = ops.cond_linear_fwd(dev(x), P, wo, bo, cd, n_out, rows=tables["rows"])
    assert torch.equal(y_sorted, y)
    dx = ops.cond_linear_bwd(dev(dy), dev(x), P, G, wo, bo, tables)
    assert rel_l2(dx, dx_ref) < 1e-5
    # the one-workgroup-per-cell dx kernel agrees (different summation tree: tolerance, not bits)
    dx_plain = ops.cond_linear_bwd(dev(dy), dev(x), P, torch.full((pos,), 7.0, device="cuda"), wo, bo, tables,
                                   sorted_dx=False)
    assert rel_l2(dx_plain, dx_ref) < 1e-5 and rel_l2(dx_plain, dx) < 1e-6
    # fixed-size launch of a captured program: chunk / reduction lists padded to their maxima with -1; dx accumulated
    seg = np.zeros(CT.words(B), dtype=np.int32)
    CT.fill_padded(seg, t, B)
    seg_dev, lay = torch.from_numpy(seg).cuda(), CT.layout(B)
    sizes_of = dict(cond=B, rows=B, chunk_dst=CT.max_chunks(B), chunk_beg=CT.max_chunks(B), chunk_end=CT.max_chunks(B),
                    red_cond=CT.max_reductions(B), red_slot=CT.max_reductions(B), red_n=CT.max_reductions(B))
    padded = {k: seg_dev[lay[k]:lay[k] + n] for k, n in sizes_of.items()}
    G2 = torch.full((pos,), 7.0, device="cuda")
    dx2 = ops.cond_linear_bwd(dev(dy), dev(x), P, G2, wo, bo, padded, dx=dx.clone(), accumulate=True)
    assert torch.equal(G2, G) and torch.equal(dx2, dx + dx)
    Gc = G.cpu()
    touched = torch.zeros(pos, dtype=torch.bool)
    for c in present.tolist():
        rows = (cond == c).nonzero().flatten()
        dW_ref = dy[rows].t() @ x[rows]
        assert rel_l2(Gc[w_off[c]:w_off[c] + n_out * n_in].view(n_out, n_in), dW_ref) < 1e-5
        assert rel_l2(Gc[b_off[c]:b_off[c] + n_out], dy[rows].sum(0)) < 1e-5
        touched[w_off[c]:w_off[c] + n_out * n_in] = True
        touched[b_off[c]:b_off[c] + n_out] = True
    assert bool((Gc[~touched] == 7.0).all()), "gradients of absent conditions (and the gaps) must not be written"


@pytest.mark.parametrize("B,Z,counts", [(512, 128, [8, 2, 273, 4644, 4, 1]), (100, 24, [3, 1, 7]), (33, 8, [5, 2])])
def test_cond_linear_multi_positions_match_single_launches(ops, B, Z, counts):
    """mmvae_cond_linear_{fwd,bwd_dw,bwd_dx}_multi (ABI 9): the positions of a "parallel" selection order in ONE launch
    each (ConditionalLayers.forward, components.py:586-631) against one launch per position of the single-position entry
    points: forward and weight / bias gradients bit for bit, dx (another summation tree) to 1e-6 and against fp64."""
    import ctypes as C  # noqa: F401

    from mmvae_amd import _lib, cond_tables as CT

    lib = _lib.load()
    n_pos = len(counts)
    g = torch.Generator().manual_seed(B + Z)
    x = torch.randn(B, Z, generator=g).cuda()
    dy = torch.randn(B, n_pos * Z, generator=g).cuda()          # [B, n_pos Z]: position j owns columns j Z .. (j + 1) Z
    # one arena with every position's blocks; global block index = base_j + local index
    w_off, b_off, pos, bases = [], [], 8, []
    for c in counts:
        bases.append(len(w_off))
        for _ in range(c):
            w_off.append(pos)
            pos += Z * Z + 4
        for _ in range(c):
            b_off.append(pos)
            pos += Z + 4
    params = (torch.randn(pos, generator=g) * 0.3).cuda()
    wo, bo = torch.tensor(w_off, dtype=torch.int64).cuda(), torch.tensor(b_off, dtype=torch.int64).cuda()
    P = CT.words(B)
    lay = CT.layout(B)
    pack = np.zeros(n_pos * P, dtype=np.int32)
    conds = []
    for j, c in enumerate(counts):
        local = torch.randint(0, c, (B,), generator=g).numpy().astype("int32")
        conds.append(local)
        CT.fill_padded(pack[j * P:(j + 1) * P], CT.group_tables(local, bases[j]), B)
    tbl = torch.from_numpy(pack).cuda()
    ptr = lambda j, name: tbl.data_ptr() + 4 * (j * P + lay[name])
    st = torch.cuda.current_stream().cuda_stream
    n_chunks, n_red = CT.max_chunks(B), CT.max_reductions(B)
    slot = CT.partial_slots(B) * (Z * Z + Z)
    # ---- one launch per position
    y1 = torch.zeros(B, n_pos * Z, device="cuda")
    G1 = torch.full((pos,), 7.0, device="cuda")
    dx1 = torch.zeros(B, Z, device="cuda")
    part = torch.zeros(n_pos * slot, device="cuda")
    for j in range(n_pos):
        assert lib.mmvae_cond_linear_fwd(B, Z, Z, x.data_ptr(), Z, params.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                         ptr(j, "cond"), ptr(j, "rows"), y1.data_ptr() + 4 * j * Z, n_pos * Z, st) == 0
    for j in range(n_pos - 1, -1, -1):
        assert lib.mmvae_cond_linear_bwd_dw(n_chunks, ptr(j, "chunk_dst"), ptr(j, "chunk_beg"), ptr(j, "chunk_end"),
                                            ptr(j, "rows"), Z, Z, dy.data_ptr() + 4 * j * Z, n_pos * Z, x.data_ptr(), Z,
                                            G1.data_ptr(), wo.data_ptr(), bo.data_ptr(), n_red, ptr(j, "red_cond"),
                                            ptr(j, "red_slot"), ptr(j, "red_n"), part.data_ptr(), st) == 0
        assert lib.mmvae_cond_linear_bwd_dx(B, Z, Z, dy.data_ptr() + 4 * j * Z, n_pos * Z, params.data_ptr(), wo.data_ptr(),
                                            ptr(j, "cond"), ptr(j, "rows"), dx1.data_ptr(), Z, int(j != n_pos - 1), st) == 0
    # ---- all positions per launch
    y2 = torch.zeros(B, n_pos * Z, device="cuda")
    G2 = torch.full((pos,), 7.0, device="cuda")
    dx2 = torch.zeros(B, Z, device="cuda")
    assert lib.mmvae_cond_linear_fwd_multi(n_pos, B, Z, Z, x.data_ptr(), Z, 0, params.data_ptr(), wo.data_ptr(), bo.data_ptr(),
                                           ptr(0, "cond"), ptr(0, "rows"), P, y2.data_ptr(), n_pos * Z, Z, st) == 0
    assert lib.mmvae_cond_linear_bwd_dw_multi(n_pos, n_chunks, ptr(0, "chunk_dst"), ptr(0, "chunk_beg"), ptr(0, "chunk_end"),
                                              ptr(0, "rows"), P, Z, Z, dy.data_ptr(), n_pos * Z, Z, x.data_ptr(), Z, 0,
                                              G2.data_ptr(), wo.data_ptr(), bo.data_ptr(), n_red, ptr(0, "red_cond"),
                                              ptr(0, "red_slot"), ptr(0, "red_n"), part.data_ptr(), slot, st) == 0
    assert lib.mmvae_cond_linear_bwd_dx_multi(n_pos, B, Z, Z, dy.data_ptr(), n_pos * Z, Z, params.data_ptr(), wo.data_ptr(),
                                              ptr(0, "cond"), P, dx2.data_ptr(), Z, 0, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(y1, y2) and torch.equal(G1, G2)
    assert rel_l2(dx2, dx1) < 1e-6
    # fp64 reference of y and dx
    Pd = params.double().cpu()
    y_ref = torch.zeros(B, n_pos * Z, dtype=torch.float64)
    dx_ref = torch.zeros(B, Z, dtype=torch.float64)
    for j in range(n_pos):
        for b in range(0, B, max(1, B // 16)):  # a sample of the cells
            c = bases[j] + int(conds[j][b])
            W = Pd[w_off[c]:w_off[c] + Z * Z].view(Z, Z)
            y_ref[b, j * Z:(j + 1) * Z] = W @ x[b].double().cpu() + Pd[b_off[c]:b_off[c] + Z]
    rows = list(range(0, B, max(1, B // 16)))
    assert rel_l2(y2.cpu()[rows], y_ref[rows]) < 1e-5
    for b in rows:
        for j in range(n_pos):
            c = bases[j] + int(conds[j][b])
            dx_ref[b] += Pd[w_off[c]:w_off[c] + Z * Z].view(Z, Z).t() @ dy[b, j * Z:(j + 1) * Z].double().cpu()
    assert rel_l2(dx2.cpu()[rows], dx_ref[rows]) < 1e-5
    # accumulate flag of the multi dx
    assert lib.mmvae_cond_linear_bwd_dx_multi(n_pos, B, Z, Z, dy.data_ptr(), n_pos * Z, Z, params.data_ptr(), wo.data_ptr(),
                                              ptr(0, "cond"), P, dx2.data_ptr(), Z, 1, st) == 0
    torch.cuda.synchronize()
    assert rel_l2(dx2, 2 * dx1) < 1e-6


def test_cross_entropy_heads_one_launch(ops):
    """mmvae_cross_entropy_heads: all heads of an adversary on one packed logits matrix == one launch per head."""
    from mmvae_amd import _lib

    B, widths = 512, [8, 2, 273, 4644]
    Ct, H = sum(widths), len(widths)
    g = torch.Generator().manual_seed(5)
    logits = (torch.randn(B, Ct, generator=g) * 2).cuda()
    labels = torch.stack([torch.randint(0, w, (B,), generator=g) for w in widths]).cuda()
    cols = [sum(widths[:k]) for k in range(H)]
    lib, s = _lib.load(), torch.cuda.current_stream().cuda_stream
    rows_ref, d_ref = torch.zeros(H, B, device="cuda"), torch.zeros(B, Ct, device="cuda")
    for h, (c0, w) in enumerate(zip(cols, widths)):
        _lib.check(lib.mmvae_cross_entropy_sum(B, w, logits.data_ptr() + 4 * c0, Ct, labels[h].data_ptr(),
                                               rows_ref[h].data_ptr(), d_ref.data_ptr() + 4 * c0, Ct, None, 25.0, s), "ce")
    rows, d = torch.zeros(H, B, device="cuda"), torch.zeros(B, Ct, device="cuda")
    cw = torch.tensor(cols + widths, dtype=torch.int32, device="cuda")
    _lib.check(lib.mmvae_cross_entropy_heads(B, H, max(widths), cw.data_ptr(), cw.data_ptr() + 4 * H, logits.data_ptr(), Ct,
                                             labels.data_ptr(), rows.data_ptr(), d.data_ptr(), Ct, 25.0, s), "ce heads")
    torch.cuda.synchronize()
    want = torch.stack([torch.nn.functional.cross_entropy(logits[:, c0:c0 + w].double().cpu(), labels[h].cpu(),
                                                          reduction="none") for h, (c0, w) in enumerate(zip(cols, widths))])
    assert rel_l2(rows, want) < 1e-6 and rel_l2(rows, rows_ref) < 1e-6 and rel_l2(d, d_ref) < 1e-6
    assert torch.equal(rows[3], rows_ref[3]) and torch.equal(d[:, cols[3]:], d_ref[:, cols[3]:])  # wide head: same kernel body


def test_upload_words_reads_page_locked_host_memory_in_place():
    """mmvae_upload_words (ABI 10): per-step host tables into device memory by a kernel reading the page-locked source over
    the host link -- bit-exact for every length / alignment (16-byte groups + tail, element path), in stream order with
    the kernels that read the table; pageable memory is refused (MMVAE_ERR_ARG), never read."""
    from mmvae_amd import _lib

    lib = _lib.load()
    stream = torch.cuda.current_stream().cuda_stream
    for n in (1, 3, 4, 5, 1023, 4096, 131075):
        src = torch.randint(-2**31, 2**31 - 1, (n + 4,), dtype=torch.int64).to(torch.int32).pin_memory()
        for off in (0, 1):  # 16-byte aligned and not
            dst = torch.full((n + 8,), 7, dtype=torch.int32, device="cuda")
            rc = lib.mmvae_upload_words(n, src.data_ptr() + 4 * off, dst.data_ptr() + 4 * off, stream)
            assert rc == 0
            torch.cuda.synchronize()
            assert torch.equal(dst[off:off + n].cpu(), src[off:off + n])
            assert (dst[:off] == 7).all() and (dst[off + n:] == 7).all()
    pageable = torch.zeros(64, dtype=torch.int32)
    dst = torch.zeros(64, dtype=torch.int32, device="cuda")
    assert lib.mmvae_upload_words(64, pageable.data_ptr(), dst.data_ptr(), stream) == _lib.ERR_ARG
    assert lib.mmvae_upload_words(0, pageable.data_ptr(), dst.data_ptr(), stream) == _lib.ERR_ARG
