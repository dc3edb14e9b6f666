"""torch.autograd.Function wrappers: the module-level (flexible) execution path of the mirror.

Every tensor-sized operation below -- forward and backward -- is a call into libmmvae_hip.so through mmvae_amd.ops.
Autograd only sequences them.  0-dim loss bookkeeping (adding two scalar losses, dividing by the batch size) is left
to torch scalar ops; the graph-captured step engine (mmvae_amd.engine) does even that inside kernels.

Reference call sites these replace are cited per class (paths relative to /root/reference/src/cmmvae/).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


class FCLayerFn(torch.autograd.Function):
    """One FCBlock layer: Linear -> [BatchNorm1d] -> [ReLU] -> [Dropout]  (modules/base/components.py:275-290).

    Returns (d, a): d = layer output (post dropout), a = post-activation / pre-dropout tensor, which is what
    FCBlock.forward collects as a hidden representation (components.py:312-313).  a is d when there is no dropout.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn_state, training, relu, keep_mask, dropout_p):
        x2 = x if x.is_contiguous() else x.contiguous()
        B, N = x2.shape[0], weight.shape[0]
        has_bn = bn_state is not None
        ctx.has_bn, ctx.relu, ctx.p = has_bn, relu, dropout_p
        ctx.training = training
        use_mask = keep_mask is not None and training and dropout_p > 0
        _, splitk = ops.gemm_plan(ops.GEMM_NT, B, N, x2.shape[1])
        if not has_bn and not use_mask and splitk == 1:
            d = ops.gemm(ops.GEMM_NT, x2, weight, bias=bias, relu=relu, splitk=1)  # bias/ReLU in the GEMM epilogue
            a = d
            z = mean = invstd = None
        else:
            slabs = ops.gemm_slabs(ops.GEMM_NT, x2, weight, splitk=splitk)
            bn = None
            if has_bn:
                bn = dict(gamma=gamma, beta=beta, running_mean=bn_state["running_mean"],
                          running_var=bn_state["running_var"],
                          num_batches_tracked=bn_state.get("num_batches_tracked"),
                          momentum=bn_state["momentum"], eps=bn_state["eps"])
            f = ops.fc_epilogue_fwd(slabs, bias, bn=bn, training=training, relu=relu,
                                    keep_mask=keep_mask if use_mask else None, dropout_p=dropout_p)
            d, a, z, mean, invstd = f["d"], f["a"], f["z"], f["mean"], f["invstd"]
        ctx.use_mask = use_mask
        ctx.save_for_backward(x2, weight, gamma, a, z, mean, invstd, keep_mask if use_mask else None)
        if a is d:
            return d  # no dropout: the hidden representation IS the layer output
        return d, a

    @staticmethod
    def backward(ctx, gd, ga=None):
        x, weight, gamma, a, z, mean, invstd, mask = ctx.saved_tensors
        if ctx.has_bn and not ctx.training:
            raise RuntimeError("backward through eval-mode BatchNorm is not part of the training step")
        gd = gd.contiguous() if gd is not None else None
        ga = ga.contiguous() if ga is not None else None
        if gd is None and ga is None:
            return (None,) * 10
        # gd: gradient on the layer output d (through the keep mask); ga: gradient on the pre-dropout activation, which
        # bypasses the mask (addend_a of the layer-tail kernel)
        if gd is None:
            gd, ga, mask_arg, p = ga, None, None, 0.0
        else:
            mask_arg, p = (mask, ctx.p) if ctx.use_mask else (None, 0.0)
        dz, dbias, dgamma, dbeta = ops.fc_epilogue_bwd(
            gd, addend_a=ga, keep_mask=mask_arg, dropout_p=p, relu=ctx.relu, a=a if ctx.relu else None, z=z,
            gamma=gamma, mean=mean, invstd=invstd, has_bn=ctx.has_bn)
        dw = ops.gemm(ops.GEMM_TN, dz, x) if ctx.needs_input_grad[1] else None
        dx = ops.gemm(ops.GEMM_NN, dz, weight) if ctx.needs_input_grad[0] else None
        return dx, dw, dbias if ctx.needs_input_grad[2] else None, dgamma, dbeta, None, None, None, None, None


class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm(elementwise_affine=False) (components.py:281)."""

    @staticmethod
    def forward(ctx, x, eps):
        y, invstd = ops.layernorm_fwd(x.contiguous(), eps)
        ctx.save_for_backward(y, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        y, invstd = ctx.saved_tensors
        return ops.layernorm_bwd(gy.contiguous(), y, invstd), None


class LayerTailFn(torch.autograd.Function):
    """[ReLU] -> [Dropout] as a pass of its own (the tail of an FCBlock layer whose LayerNorm separates it from the
    Linear's column kernel, components.py:281-288).  Returns (d, a): output and pre-dropout activation."""

    @staticmethod
    def forward(ctx, x, relu, keep_mask, dropout_p):
        use_mask = keep_mask is not None and dropout_p > 0
        f = ops.fc_epilogue_fwd(x.contiguous(), None, relu=relu, keep_mask=keep_mask if use_mask else None,
                                dropout_p=dropout_p if use_mask else 0.0)
        d, a = f["d"], f["a"]
        ctx.relu, ctx.p, ctx.use_mask = relu, dropout_p, use_mask
        ctx.save_for_backward(a if relu else None, keep_mask if use_mask else None)
        if a is d:
            return d
        return d, a

    @staticmethod
    def backward(ctx, gd, ga=None):
        a, mask = ctx.saved_tensors
        gd = gd.contiguous() if gd is not None else None
        ga = ga.contiguous() if ga is not None else None
        if gd is None and ga is None:
            return None, None, None, None
        if gd is None:
            gd, ga, mask_arg, p = ga, None, None, 0.0
        else:
            mask_arg, p = (mask, ctx.p) if ctx.use_mask else (None, 0.0)
        dx, _, _, _ = ops.fc_epilogue_bwd(gd, addend_a=ga, keep_mask=mask_arg, dropout_p=p, relu=ctx.relu,
                                          a=a if ctx.relu else None, want_dbias=False)
        return dx, None, None, None


def layer_tail(x, *, relu: bool, keep_mask=None, dropout_p: float = 0.0):
    out = LayerTailFn.apply(x, relu, keep_mask, dropout_p)
    return out if isinstance(out, tuple) else (out, out)


class CondLinearFn(torch.autograd.Function):
    """Per-cell conditional Linear, y[b] = W[c_b] x[b] + bias[c_b] (ConditionalLayer.forward, components.py:365-413,
    for single-Linear condition blocks), on the grouped HIP kernels.  The thousands of condition blocks are not autograd
    inputs: they are addressed through their offsets in the optimiser's arena (`bank`), and backward writes the
    gradients of the conditions PRESENT in the batch straight into the gradient arena and tells the optimiser which
    parameters took part (the others keep "no gradient", like under torch autograd)."""

    @staticmethod
    def forward(ctx, x, bank, tables, present_params):
        x2 = x if x.is_contiguous() else x.contiguous()
        a = bank["opt"].arena
        y = ops.cond_linear_fwd(x2, a.data, bank["w_off"], bank["b_off"], tables["cond"], bank["n_out"], rows=tables["rows"])
        ctx.save_for_backward(x2)
        ctx.bank, ctx.tables, ctx.present_params = bank, tables, present_params
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        bank = ctx.bank
        opt = bank["opt"]
        a = opt.arena
        gy = gy if gy.is_contiguous() else gy.contiguous()
        dx = ops.cond_linear_bwd(gy, x, a.data, a.grad, bank["w_off"], bank["b_off"], ctx.tables)
        opt.note_direct_grads(ctx.present_params)
        return dx, None, None, None


class ReparamKLFn(torch.autograd.Function):
    """Encoder tail (components.py:795-801) fused with the Gaussian KL of BaseVAE.elbo (modules/vae.py:136-137).

    (mu, a_raw, eps) -> z = mu + sqrt(exp(a_raw) + var_eps) * eps, std, kl_sum = sum_b sum_j KL_bj.
    eps: [B,Z] or [K,B,Z] (K-sample extension).  Also caches, non-differentiably, sum(mu) and sum(var) for the
    Mean / Variance log lines (models/cmmvae_model.py:170-171)."""

    @staticmethod
    def forward(ctx, mu, a_raw, eps, var_eps):
        mu, a_raw, eps = mu.contiguous(), a_raw.contiguous(), eps.contiguous()
        std, z, kl_row, stat = ops.reparam_kl_fwd(mu, a_raw, eps, var_eps)
        kl_sum = ops.sum_f32(kl_row).reshape(())
        ctx.var_eps = var_eps
        ctx.save_for_backward(mu, std, eps)
        ctx.mark_non_differentiable(stat)
        return z, std, kl_sum, stat

    @staticmethod
    def backward(ctx, gz, gstd, gkl, _gstat):
        mu, std, eps = ctx.saved_tensors
        gz = gz.contiguous() if gz is not None else None
        gstd = gstd.contiguous() if gstd is not None else None
        if gkl is not None:
            dmu, da = ops.reparam_kl_bwd(mu, std, eps, gz, dstd_extra=gstd, kl_scale_dev=gkl.contiguous(),
                                         var_eps=ctx.var_eps)
        else:
            dmu, da = ops.reparam_kl_bwd(mu, std, eps, gz, dstd_extra=gstd, kl_scale=0.0, var_eps=ctx.var_eps)
        return dmu, da, None, None


class MseSumFn(torch.autograd.Function):
    """F.mse_loss(xhat, x, reduction="sum") (modules/vae.py:143): per-cell squared error rows + their sum."""

    @staticmethod
    def forward(ctx, xhat, x):
        se_row, _ = ops.mse_sum_fwd_bwd(xhat, x, want_grad=False)
        ctx.save_for_backward(xhat, x)
        return ops.sum_f32(se_row).reshape(())

    @staticmethod
    def backward(ctx, g):
        xhat, x = ctx.saved_tensors
        _, dx = ops.mse_sum_fwd_bwd(xhat, x, want_grad=True, gscale_dev=g.contiguous())
        return dx, None


class KSampleReconFn(torch.autograd.Function):
    """K-sample reconstruction term (extension): sum_b -logmeanexp_k(-||x_b - xhat_bk||^2).
    xhat: [K*B, G] (sample-major), x: [B, G].  Reduces to MseSumFn at K = 1."""

    @staticmethod
    def forward(ctx, xhat, x, K):
        B = x.shape[0]
        xr = x.repeat(K, 1) if K > 1 else x  # plumbing copy; the engine path compares against x[r % B] in-kernel
        se_row, _ = ops.mse_sum_fwd_bwd(xhat, xr, want_grad=False)
        out, w = ops.elbo_finalize(se_row.reshape(1, -1), None, None, B=B, K=K, want_w=True)
        ctx.save_for_backward(xhat, xr, w)
        return out[1].reshape(())

    @staticmethod
    def backward(ctx, g):
        xhat, xr, w = ctx.saved_tensors
        _, d = ops.mse_sum_fwd_bwd(xhat, xr, want_grad=True, gscale_dev=g.contiguous())
        return ops.scale_rows(d, w), None, None


class CrossEntropySumFn(torch.autograd.Function):
    """nn.CrossEntropyLoss(reduction="sum") on adversarial head logits (models/cmmvae_model.py:54,85)."""

    @staticmethod
    def forward(ctx, logits, labels):
        rows, _ = ops.cross_entropy_sum(logits, labels, want_grad=False)
        ctx.save_for_backward(logits, labels)
        return ops.sum_f32(rows).reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, labels = ctx.saved_tensors
        _, dl = ops.cross_entropy_sum(logits, labels, want_grad=True, gscale_dev=g.contiguous())
        return dl, None


class GradientReversalFunction(torch.autograd.Function):
    """Gradient reversal layer (components.py:879-899): identity forward, -alpha * grad backward."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        if grad_output.is_cuda:
            g = grad_output.contiguous()
            return ops.axpby(-float(ctx.alpha), g, 0.0, torch.empty_like(g)), None
        return grad_output.neg() * ctx.alpha, None  # CPU plumbing


def fc_layer(x, weight, bias, *, gamma=None, beta=None, bn_state: Optional[dict] = None, training=True, relu=False,
             keep_mask=None, dropout_p=0.0):
    """Returns (d, a): layer output and post-activation / pre-dropout tensor (the same object without dropout)."""
    out = FCLayerFn.apply(x, weight, bias, gamma, beta, bn_state, training, relu, keep_mask, dropout_p)
    return out if isinstance(out, tuple) else (out, out)
