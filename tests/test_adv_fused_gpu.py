"""Row-owner adversary kernels (mmvae_adv_pass_f32 / mmvae_adv_dw_f32 / mmvae_adam_step_multi) through the C-ABI against
torch autograd in fp64 on the CPU: one adversarial phase of `CMMVAEModel.grf` (reference models/cmmvae_model.py:59-101;
Adversarial, GradientReversalFunction: modules/base/components.py:638-674, 879-899) -- encoder FCBlock (Linear -> ReLU
-> Dropout), every head, CrossEntropyLoss(sum), gradients of every parameter and of the hidden representation.
Tolerances: losses rtol 2e-6; logits / gradients rel-L2 <= 2e-6 (exact-f32 MFMA, fp32 softmax); labels exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import rel_l2  # noqa: E402


def _pad4(v):
    return (v + 3) // 4 * 4


class _Pool:
    def __init__(self, device):
        self.device, self.t = device, {}

    def __call__(self, name, shape, dtype):
        key = (name, tuple(shape), dtype)
        if key not in self.t:
            self.t[key] = torch.zeros(tuple(shape), dtype=dtype, device=self.device)
        return self.t[key]


class _FakeOpt:
    """What AdvProgram needs of a HipAdam: flat arenas, state words, one param group."""

    def __init__(self, numel, device):
        class A:
            pass

        self.arena = A()
        self.arena.numel = numel
        self.arena.data = torch.zeros(numel, device=device)
        self.arena.grad = torch.zeros(numel, device=device)
        self.arena.exp_avg = torch.zeros(numel, device=device)
        self.arena.exp_avg_sq = torch.zeros(numel, device=device)
        self.state_dev = torch.zeros(8, device=device)
        self.param_groups = [dict(lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6)]


def _build(widths, classes, B, seed, p_drop, device):
    """Random adversary with its parameters laid out in flat arenas the way HipAdam(pack=) does (heads stacked, each
    head's rows starting on a multiple of 4, holes zero).  Returns (AdvNet, cpu fp64 parameter copies, masks)."""
    from mmvae_amd.adv_program import AdvLayer, AdvNet

    g = torch.Generator().manual_seed(seed)
    H = len(classes)
    ne = widths[-1]
    col = [sum(_pad4(c) for c in classes[:h]) for h in range(H)] if H > 1 else [0]
    Ct = sum(_pad4(c) for c in classes) if H > 1 else classes[0]
    n_par = sum(_pad4(widths[l + 1] * widths[l]) + _pad4(widths[l + 1]) for l in range(len(widths) - 1))
    n_par += _pad4(Ct * ne) + _pad4(Ct)
    opt = _FakeOpt(n_par, device)
    off = 0

    def take(shape):
        nonlocal off
        n = int(np.prod(shape))
        d, gr = opt.arena.data[off:off + n].view(shape), opt.arena.grad[off:off + n].view(shape)
        off += _pad4(n)
        return d, gr

    layers, cpu = [], dict(W=[], b=[])
    masks = []
    for l in range(len(widths) - 1):
        W, gW = take((widths[l + 1], widths[l]))
        b, gb = take((widths[l + 1],))
        W.copy_(torch.randn(W.shape, generator=g) * (2.0 / widths[l + 1]) ** 0.5)
        b.copy_(torch.randn(b.shape, generator=g) * 0.1)
        lay = AdvLayer(W=W, b=b, gW=gW, gb=gb, relu=True, p_drop=p_drop)
        if p_drop > 0:
            m = (torch.rand(B, widths[l + 1], generator=g) >= p_drop).to(torch.uint8)
            lay.masks["phase"] = m.to(device)
            masks.append(m)
        else:
            masks.append(None)
        layers.append(lay)
        cpu["W"].append(W.cpu().double())
        cpu["b"].append(b.cpu().double())
    Wh, gWh = take((Ct, ne))
    bh, gbh = take((Ct,))
    for h, c in enumerate(classes):
        Wh[col[h]:col[h] + c].copy_(torch.randn(c, ne, generator=g) * (2.0 / c) ** 0.5)
        bh[col[h]:col[h] + c].copy_(torch.randn(c, generator=g) * 0.1)
    x = (torch.randn(B, widths[0], generator=g) * 1.5).to(device)
    net = AdvNet(x=x, ldx=widths[0], layers=layers, Wh=Wh, bh=bh, gWh=gWh, gbh=gbh, col=col, classes=list(classes), opt=opt)
    cpu.update(Wh=Wh.cpu().double(), bh=bh.cpu().double(), x=x.cpu().double())
    return net, cpu, masks


def _reference(cpu, masks, labels, widths, classes, col, p_drop, gscale):
    x = cpu["x"].clone().requires_grad_(True)
    Ws = [w.clone().requires_grad_(True) for w in cpu["W"]]
    bs = [b.clone().requires_grad_(True) for b in cpu["b"]]
    Wh = cpu["Wh"].clone().requires_grad_(True)
    bh = cpu["bh"].clone().requires_grad_(True)
    h = x
    for l, (W, b) in enumerate(zip(Ws, bs)):
        h = torch.relu(h @ W.t() + b)
        if masks[l] is not None:
            h = h * masks[l].double() / (1.0 - p_drop)
    logits = h @ Wh.t() + bh
    losses = []
    for hd, c in enumerate(classes):
        lg = logits[:, col[hd]:col[hd] + c]
        losses.append(torch.nn.functional.cross_entropy(lg, labels[hd], reduction="sum"))
    total = torch.stack(losses).sum()
    (gscale * total).backward()
    return dict(losses=[float(v) for v in losses], total=float(total), logits=logits.detach(), gx=-x.grad,
                gW=[w.grad for w in Ws], gb=[b.grad for b in bs], gWh=Wh.grad, gbh=bh.grad)


CASES = [
    # widths, classes, B, dropout
    ([256, 128, 64], [8, 2, 273, 4644], 512, 0.0),   # config C4, adversary on h1
    ([128, 64], [8, 2, 273, 4644], 512, 0.0),        # config C4, adversary on z
    ([24, 16, 8], [5, 2, 37], 16, 0.0),              # golden "adversarial"
    ([12, 8], [5, 2, 37], 16, 0.25),                 # golden "adv_dropout"
    ([12, 8, 6], [4, 3], 12, 0.0),                   # golden "cond_adv": widths off every tile, B off the cell tile
    ([40, 33, 20], [7], 45, 0.1),                    # one head, class count not a multiple of 4
    ([64, 128], [300, 17], 100, 0.0),                # 128-wide encoded features (8 head tiles)
]


@pytest.mark.parametrize("widths,classes,B,p_drop", CASES)
@pytest.mark.parametrize("reverse", [False, True])
def test_adv_pass_and_dw_match_autograd(widths, classes, B, p_drop, reverse):
    from mmvae_amd import _lib
    from mmvae_amd.adv_program import AdvProgram, supported

    lib = _lib.load()
    dev = torch.device("cuda")
    gscale = 25.0 if reverse else 1.0
    net, cpu, masks = _build(widths, classes, B, seed=len(widths) * 100 + B, p_drop=p_drop, device=dev)
    assert supported(lib, net, B) is not None
    g = torch.Generator().manual_seed(7)
    labels = torch.stack([torch.randint(0, c, (B,), generator=g) for c in classes])
    labels_dev = labels.to(dev)
    metrics = torch.zeros(64, device=dev)
    metrics[40] = 3.0  # total loss so far
    prog = AdvProgram(lib, _Pool(dev), [net], B, labels_dev, dev)
    H = len(classes)
    cfg = dict(gscale=gscale, reverse=reverse, loss_each=[metrics.data_ptr()], loss_total=[metrics.data_ptr() + 4 * H],
               total_loss=metrics.data_ptr() + 160 if reverse else None, total_scale=gscale,
               opts=[dict(flags=_lib.PREPARE_NORM | _lib.PREPARE_ADVANCE, max_norm=10.0, norm_out=metrics.data_ptr() + 200)])
    prog.build_phase("phase", cfg)
    for rep in range(2):  # tickets must come back to zero: the second run gives the same numbers
        prog.launch_pass("phase")
        prog.launch_dw("phase")
    torch.cuda.synchronize()
    ref = _reference(cpu, masks, list(labels), widths, classes, net.col, p_drop, gscale)
    m = metrics.cpu().double()
    np.testing.assert_allclose(m[:H].numpy(), ref["losses"], rtol=2e-6)
    np.testing.assert_allclose(float(m[H]), ref["total"], rtol=2e-6)
    if reverse:
        np.testing.assert_allclose(float(m[40]), gscale * ref["total"], rtol=4e-6)  # (stored, not accumulated)
    b = prog.bufs[0]
    lg = b["logits"].cpu().double()
    for h, c in enumerate(classes):
        sl = slice(net.col[h], net.col[h] + c)
        assert rel_l2(lg[:, sl], ref["logits"][:, sl]) <= 2e-6
        lse_ref = torch.logsumexp(ref["logits"][:, sl], dim=1)
        assert rel_l2(b["lse"][h].cpu().double(), lse_ref) <= 2e-6
    if reverse:
        assert rel_l2(b["gx"].cpu().double(), ref["gx"]) <= 4e-6
    sq = 0.0
    for l, lay in enumerate(net.layers):
        assert rel_l2(lay.gW.cpu().double(), ref["gW"][l]) <= 4e-6, f"dW of layer {l}"
        assert rel_l2(lay.gb.cpu().double(), ref["gb"][l]) <= 4e-6, f"db of layer {l}"
        sq += float((ref["gW"][l] ** 2).sum() + (ref["gb"][l] ** 2).sum())
    assert rel_l2(net.gWh.cpu().double(), ref["gWh"]) <= 4e-6
    assert rel_l2(net.gbh.cpu().double(), ref["gbh"]) <= 4e-6
    sq += float((ref["gWh"] ** 2).sum() + (ref["gbh"] ** 2).sum())
    # padding rows of the stacked heads stay exactly zero
    if H > 1:
        for h, c in enumerate(classes):
            assert not net.gWh[net.col[h] + c:net.col[h] + _pad4(c)].any()
    norm = sq ** 0.5
    st = net.opt.state_dev.cpu()
    np.testing.assert_allclose(float(st[1]), norm, rtol=4e-6)
    np.testing.assert_allclose(float(m[50]), norm, rtol=4e-6)
    assert float(st[0]) == 2.0  # two launches advanced the step twice
    np.testing.assert_allclose(float(st[2]), min(1.0, 10.0 / (norm + 1e-6)), rtol=4e-6)
    # the gradient arena's own norm is the reported one (the jobs cover the arena)
    np.testing.assert_allclose(float(net.opt.arena.grad.double().norm()), norm, rtol=4e-6)


def test_adv_two_jobs_one_launch_and_adam_multi():
    """Two adversaries (C4's) as two jobs of the same launches: each equals its single-job run bit for bit; total_loss
    accumulates in job order; mmvae_adam_step_multi equals mmvae_adam_step per arena bit for bit."""
    from mmvae_amd import _lib
    from mmvae_amd.adv_program import AdvProgram

    lib = _lib.load()
    dev = torch.device("cuda")
    B, classes = 512, [8, 2, 273, 4644]
    H = len(classes)
    g = torch.Generator().manual_seed(11)
    labels = torch.stack([torch.randint(0, c, (B,), generator=g) for c in classes]).to(dev)

    def run(which):
        nets = [_build(w, classes, B, seed=s, p_drop=0.0, device=dev)[0] for w, s in which]
        metrics = torch.zeros(256, device=dev)
        prog = AdvProgram(lib, _Pool(dev), nets, B, labels, dev, splits=4)  # (the merge order follows the split count)
        n = len(nets)
        cfg = dict(gscale=25.0, reverse=True, loss_each=[metrics.data_ptr() + 4 * 8 * i for i in range(n)],
                   loss_total=[metrics.data_ptr() + 4 * (8 * i + H) for i in range(n)], total_loss=metrics.data_ptr() + 4 * 100,
                   total_scale=25.0,
                   opts=[dict(flags=_lib.PREPARE_NORM | _lib.PREPARE_ADVANCE, max_norm=10.0, norm_out=None)] * n)
        prog.build_phase("g", cfg)
        prog.build_adam()
        prog.launch_pass("g")
        prog.launch_dw("g")
        before = [(nt.opt.arena.data.clone(), nt.opt.arena.grad.clone(), nt.opt.state_dev.clone()) for nt in nets]
        prog.launch_adam()
        torch.cuda.synchronize()
        return nets, prog, metrics.cpu(), before

    both = [([256, 128, 64], 1), ([128, 64], 2)]
    nets2, prog2, m2, before = run(both)
    for i, w in enumerate(both):
        nets1, prog1, m1, _ = run([w])
        assert torch.equal(m1[:H + 1], m2[8 * i:8 * i + H + 1])
        assert torch.equal(nets1[0].opt.arena.grad, before[i][1])
        assert torch.equal(prog1.bufs[0]["gx"], prog2.bufs[i]["gx"])
        assert torch.equal(nets1[0].opt.arena.data, nets2[i].opt.arena.data)
    assert float(m2[100]) == 25.0 * float(m2[H]) + 25.0 * float(m2[8 + H]) or abs(
        float(m2[100]) - 25.0 * (float(m2[H]) + float(m2[8 + H]))) <= 1e-6 * abs(float(m2[100]))
    # Adam: the multi-arena launch against the single-arena entry point
    from mmvae_amd import ops  # noqa: F401

    for i, nt in enumerate(nets2):
        p, gr, st = before[i]
        mm, vv = torch.zeros_like(p), torch.zeros_like(p)
        gp = nt.opt.param_groups[0]
        _lib.check(lib.mmvae_adam_step(p.numel(), p.data_ptr(), gr.data_ptr(), mm.data_ptr(), vv.data_ptr(), st.data_ptr(),
                                       gp["lr"], gp["betas"][0], gp["betas"][1], gp["eps"], gp["weight_decay"], 1.0,
                                       torch.cuda.current_stream().cuda_stream), "mmvae_adam_step")
        torch.cuda.synchronize()
        assert torch.equal(p, nt.opt.arena.data)
        assert torch.equal(mm, nt.opt.arena.exp_avg) and torch.equal(vv, nt.opt.arena.exp_avg_sq)


def test_adv_pass_ignores_out_of_range_labels():
    """A label outside [0, classes) contributes neither loss nor gradient (mmvae_cross_entropy_sum's rule)."""
    from mmvae_amd import _lib
    from mmvae_amd.adv_program import AdvProgram

    lib = _lib.load()
    dev = torch.device("cuda")
    widths, classes, B = [24, 16, 8], [5, 37], 16
    net, cpu, masks = _build(widths, classes, B, seed=3, p_drop=0.0, device=dev)
    labels = torch.tensor([[0, 1, 2, 3, 4, -1, 5, 99] * 2, list(range(16))])
    metrics = torch.zeros(64, device=dev)
    prog = AdvProgram(lib, _Pool(dev), [net], B, labels.to(dev), dev)
    cfg = dict(gscale=1.0, reverse=True, loss_each=[metrics.data_ptr()], loss_total=[metrics.data_ptr() + 8],
               total_loss=None, total_scale=0.0, opts=None)
    prog.build_phase("p", cfg)
    prog.launch_pass("p")
    prog.launch_dw("p")
    torch.cuda.synchronize()
    rows = prog.bufs[0]["loss_rows"].cpu()
    bad = (labels[0] < 0) | (labels[0] >= 5)
    assert bool((rows[0][bad] == 0).all()) and bool((rows[0][~bad] > 0).all())
    # gradient of the head bias: softmax column sums minus the one-hot counts of the VALID labels only
    lg = prog.bufs[0]["logits"].cpu().double()[:, :5]
    p = torch.softmax(lg, dim=1)
    onehot = torch.zeros(B, 5, dtype=torch.float64)
    for r in range(B):
        if not bad[r]:
            onehot[r, labels[0][r]] = 1.0
    assert rel_l2(net.gbh[:5].cpu().double(), (p - onehot).sum(0)) <= 4e-6


# ------------------------------------------------------------------------------------------------ sparse weight gradient
@pytest.mark.parametrize("B,G,M,density", [(512, 2000, 1024, 0.1), (33, 257, 70, 0.3), (512, 640, 128, 0.0), (300, 1000, 64, 1.0)])
def test_sparse_first_layer_weight_gradient(B, G, M, density):
    """mmvae_ell_from_dense_f32 + mmvae_dw_sparse_ell_f32: dW = dY^T x from the gene-major ELL form of a
    sparse batch against fp64 (rel-L2 <= 1e-6: an fp32 FMA chain per element); the ELL lists hold every stored entry once,
    cells ascending, padded with zeros to a multiple of 8.  Measured against the dense GEMM in profiles/r4_sparse_dw.txt."""
    from mmvae_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(B + G)
    x = torch.where(torch.rand(B, G, generator=g) < density, torch.rand(B, G, generator=g) * 9.0 + 0.1, torch.zeros(()))
    dY = torch.randn(B, M, generator=g)
    xd, dYd = x.cuda(), dY.cuda()
    cap = (B + 7) // 8 * 8
    rows = torch.full((G * cap,), -1, dtype=torch.int32, device="cuda")
    vals = torch.full((G * cap,), float("nan"), device="cuda")
    cnt = torch.zeros(G, dtype=torch.int32, device="cuda")
    dW = torch.empty(M, G, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.mmvae_ell_from_dense_f32(B, G, xd.data_ptr(), G, cap, rows.data_ptr(), vals.data_ptr(), cnt.data_ptr(), st),
               "mmvae_ell_from_dense_f32")
    _lib.check(lib.mmvae_dw_sparse_ell_f32(B, G, M, dYd.data_ptr(), M, rows.data_ptr(), vals.data_ptr(), cnt.data_ptr(), cap,
                                           dW.data_ptr(), G, st), "mmvae_dw_sparse_ell_f32")
    torch.cuda.synchronize()
    assert torch.equal(cnt.cpu().long(), (x != 0).sum(0))
    r, v, c = rows.cpu().view(G, cap), vals.cpu().view(G, cap), cnt.cpu()
    for gi in (0, G // 2, G - 1):
        n = int(c[gi])
        cells = torch.nonzero(x[:, gi]).flatten()
        assert torch.equal(r[gi, :n].long(), cells * 64) and torch.equal(v[gi, :n], x[cells, gi])
        pad = (n + 7) // 8 * 8
        assert not r[gi, n:pad].any() and not v[gi, n:pad].any()
    ref = dY.double().t() @ x.double()
    if density == 0.0:
        assert not dW.any()
    else:
        assert rel_l2(dW.cpu().double(), ref) <= 1e-6
