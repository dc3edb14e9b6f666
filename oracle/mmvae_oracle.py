"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU (torch fp32) restatement of the MMVAE training step of zdebruine/MMVAE (`cmmvae` 0.1.2.dev2), used only as
the checker: by tests/, by __graft_entry__.smoke() and by bench.py's `cpu_baseline` leg.  Nothing under mmvae_amd/
may import this module; the product path is the HIP library and fails loudly without it.

Parity status
  * K = 1 (the only regime the reference has): PINNED.  tests/golden/*.npz are outputs of the reference's own
    modules, produced in the build container by tests/golden/make_golden.py (imports /root/reference/src/cmmvae);
    tests/test_oracle_golden.py checks this file against every one of them.
  * K > 1 (K-sample log-mean-exp ELBO, BASELINE configs 3 and 5): build-defined extension, *** parity unpinned ***
    -- the reference has no multi-sample ELBO.  It reduces exactly to K = 1.

Each function cites the reference lines it restates (paths relative to /root/reference/src/cmmvae/).
Parameters live in a flat dict keyed like the reference's CMMVAE.state_dict()
(`experts.human.encoder.fc_layers.0.lin.weight`, `vae.encoder.mean_encoder.bias`, ...).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------------------- specs
@dataclass
class FCSpec:
    """One FCBlock (modules/base/components.py:40-174 FCBlockConfig after broadcasting)."""

    layers: List[int]
    dropout_rate: List[float]
    use_batch_norm: List[bool]
    use_layer_norm: List[bool]
    relu: List[bool]  # activation_fn is torch.nn.ReLU (True) or None (False) per layer
    return_hidden: List[bool]

    @staticmethod
    def make(layers, dropout_rate=0.0, use_batch_norm=False, use_layer_norm=False, relu=False, return_hidden=False):
        layers = list(layers)
        if len(layers) == 1:  # components.py:121-122
            layers = layers * 2
        n = len(layers) - 1

        def b(v):
            return list(v) if isinstance(v, (list, tuple)) else [v] * n

        return FCSpec(layers, b(dropout_rate), b(use_batch_norm), b(use_layer_norm), b(relu), b(return_hidden))

    @property
    def n_layers(self) -> int:
        return len(self.layers) - 1


@dataclass
class AdvSpec:
    """One Adversarial (components.py:638-674): encoder FCBlock + one Linear head per condition."""

    encoder: FCSpec
    heads: Dict[str, int]  # condition -> number of classes


@dataclass
class CondSpec:
    """ConditionalLayers of a CLVAE (components.py:466-631, clvae.py:31-111): per key either a shared ConditionalLayer
    (one FCBlock per condition), a per-species dict of ConditionalLayers, or -- key "species" -- one FCBlock per
    species.  Condition names are the ModuleDict keys ('.' already replaced by '_', components.py:353-363)."""

    fc: FCSpec  # conditional_config (the same for every block)
    keys: List[str]  # `conditionals` (incl. "species"); also the order when not parallel
    shared: Dict[str, List[str]]  # key -> condition names
    species_specific: Dict[str, Dict[str, List[str]]]  # key -> species -> condition names
    species_blocks: List[str]  # species that own a block under layers.species
    parallel: bool = False  # selection_order == ["parallel"]: outputs concatenated, shuffled order

    def block_prefixes(self) -> List[str]:
        out = []
        for key, names in self.shared.items():
            out += [f"vae.conditionals.layers.{key}.conditions.{c}" for c in names]
        for key, by_species in self.species_specific.items():
            for sp, names in by_species.items():
                out += [f"vae.conditionals.layers.{key}.{sp}.conditions.{c}" for c in names]
        if "species" in self.keys:
            out += [f"vae.conditionals.layers.species.{sp}" for sp in self.species_blocks]
        return out


@dataclass
class ModelSpec:
    experts: Dict[str, Tuple[FCSpec, FCSpec]]  # id -> (encoder, decoder)   components.py:812-876
    vae_encoder: FCSpec  # vae.encoder.fc         components.py:726
    vae_decoder: FCSpec  # vae.decoder            vae.py:204
    latent_dim: int
    var_eps: float = 1e-4  # components.py:704
    hidden_z: bool = False  # components.py:803-804
    softmax_z: bool = False  # Encoder(distribution="ln"): z = softmax(rsample) (components.py:740-741,801)
    adversarials: List[AdvSpec] = field(default_factory=list)
    conditionals: Optional[CondSpec] = None  # clvae.py:42-49


@dataclass
class HParams:
    """cmmvae_model.py:299-324 (Adam lr 5e-3, wd 1e-6), config.py (clip 10/norm), cmmvae_model.py:56 (adv_weight)."""

    lr: float = 5e-3
    weight_decay: float = 1e-6
    beta1: float = 0.9
    beta2: float = 0.999
    adam_eps: float = 1e-8
    vae_clip: Optional[float] = 10.0
    expert_clip: Optional[float] = 10.0
    adversarial_clip: Optional[float] = 10.0
    adv_weight: float = 1.0
    bn_momentum: float = 0.01  # components.py:279
    bn_eps: float = 1e-3
    world_size: int = 1  # DDP gradient averaging (Lightning DDP semantics): grads are divided by world_size
    clip_algorithm: str = "norm"  # "value": GradientClipConfig(algorithm="value"), config.py:8 -> clip_grad_value_
    elbo_mode: str = "analytic"  # "iwae": the opt-in full-IWAE objective of the K-sample extension (elbo_iwae)


# --------------------------------------------------------------------------------------------------- init helpers
def fc_param_names(prefix: str, spec: FCSpec) -> List[Tuple[str, Tuple[int, ...]]]:
    """Names/shapes in nn.Module.parameters() order (components.py:275-290)."""
    out = []
    for i, (n_in, n_out) in enumerate(zip(spec.layers[:-1], spec.layers[1:])):
        out.append((f"{prefix}.fc_layers.{i}.lin.weight", (n_out, n_in)))
        out.append((f"{prefix}.fc_layers.{i}.lin.bias", (n_out,)))
        if spec.use_batch_norm[i]:
            out.append((f"{prefix}.fc_layers.{i}.bn.weight", (n_out,)))
            out.append((f"{prefix}.fc_layers.{i}.bn.bias", (n_out,)))
    return out


def group_param_names(spec: ModelSpec) -> Dict[str, List[Tuple[str, Tuple[int, ...]]]]:
    """Optimiser groups of configure_optimizers (cmmvae_model.py:299-324): one per expert, vae, one per adversary."""
    groups: Dict[str, List] = {}
    for eid, (enc, dec) in spec.experts.items():
        groups[f"expert_{eid}"] = fc_param_names(f"experts.{eid}.encoder", enc) + fc_param_names(
            f"experts.{eid}.decoder", dec
        )
    h = spec.vae_encoder.layers[-1]
    groups["vae"] = (
        fc_param_names("vae.encoder.fc", spec.vae_encoder)
        + [
            ("vae.encoder.mean_encoder.weight", (spec.latent_dim, h)),
            ("vae.encoder.mean_encoder.bias", (spec.latent_dim,)),
            ("vae.encoder.var_encoder.weight", (spec.latent_dim, h)),
            ("vae.encoder.var_encoder.bias", (spec.latent_dim,)),
        ]
        + fc_param_names("vae.decoder", spec.vae_decoder)
    )
    if spec.conditionals is not None:  # registered after super().__init__ (clvae.py:87): last in the vae group
        for prefix in spec.conditionals.block_prefixes():
            groups["vae"] += fc_param_names(prefix, spec.conditionals.fc)
    for i, adv in enumerate(spec.adversarials):
        names = fc_param_names(f"adversarials.{i}.encoder", adv.encoder)
        for cond, ncls in adv.heads.items():
            names.append((f"adversarials.{i}.heads.{cond}.fc_layers.0.lin.weight", (ncls, adv.encoder.layers[-1])))
            names.append((f"adversarials.{i}.heads.{cond}.fc_layers.0.lin.bias", (ncls,)))
        groups[f"adversarial_{i + 1}"] = names
    return groups


def bn_buffer_names(spec: ModelSpec) -> List[Tuple[str, int]]:
    out = []

    def add(prefix, fc):
        for i, n_out in enumerate(fc.layers[1:]):
            if fc.use_batch_norm[i]:
                out.append((f"{prefix}.fc_layers.{i}.bn", n_out))

    for eid, (enc, dec) in spec.experts.items():
        add(f"experts.{eid}.encoder", enc)
        add(f"experts.{eid}.decoder", dec)
    add("vae.encoder.fc", spec.vae_encoder)
    add("vae.decoder", spec.vae_decoder)
    for i, adv in enumerate(spec.adversarials):
        add(f"adversarials.{i}.encoder", adv.encoder)
    return out


def init_state(spec: ModelSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """He init (modules/base/init.py:4-9: kaiming_normal_ fan_out/relu on every Linear weight, bias 0), BN affine 1/0,
    running stats 0/1 (nn.BatchNorm1d defaults)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for names in group_param_names(spec).values():
        for name, shape in names:
            if name.endswith("lin.weight") or name.endswith("_encoder.weight"):
                std = (2.0 / shape[0]) ** 0.5  # fan_out = out_features
                sd[name] = torch.randn(shape, generator=g) * std
            elif name.endswith("bn.weight"):
                sd[name] = torch.ones(shape)
            else:
                sd[name] = torch.zeros(shape)
    for name, n in bn_buffer_names(spec):
        sd[name + ".running_mean"] = torch.zeros(n)
        sd[name + ".running_var"] = torch.ones(n)
        sd[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


# ------------------------------------------------------------------------------------------------------- forward
class _ReluWithGivenSlope(torch.autograd.Function):
    """relu(y) whose backward multiplies by a GIVEN 0/1 slope instead of 1[y > 0].  At a pre-activation within rounding
    distance of zero the slope an implementation takes is decided by the last bit of a 1000-term dot product; a full-size
    step has 10^7-10^8 ReLU inputs, so two correct fp32 implementations disagree on a handful of them, and each
    disagreement moves the gradients by ~1e-3 (measured).  Parity tests therefore hand the oracle
    the slopes the implementation under test took and check separately that they differ from 1[y > 0] only at
    pre-activations next to zero."""

    @staticmethod
    def forward(ctx, y, slope):
        ctx.save_for_backward(slope)
        return torch.relu(y)

    @staticmethod
    def backward(ctx, g):
        (slope,) = ctx.saved_tensors
        return g * slope.to(g.dtype), None


def fcblock_forward(sd, prefix: str, spec: FCSpec, x, training: bool, masks: Optional[Dict[str, torch.Tensor]],
                    hp: HParams, bn_updates: Optional[dict] = None, relu_slopes: Optional[dict] = None,
                    relu_inputs: Optional[dict] = None):
    """FCBlock.forward (components.py:292-314): per layer Linear -> BN -> LN -> act -> Dropout; hidden collected
    after "af" (pre-dropout) for layers with return_hidden.  `relu_slopes` {layer prefix: 0/1 tensor}: backward slopes of
    the ReLUs given by the caller (_ReluWithGivenSlope); `relu_inputs`: filled with every ReLU's input (detached)."""
    hidden = []
    for i in range(spec.n_layers):
        p = f"{prefix}.fc_layers.{i}"
        x = F.linear(x, sd[p + ".lin.weight"], sd[p + ".lin.bias"])  # :276
        if spec.use_batch_norm[i]:  # :279 BatchNorm1d(momentum=0.01, eps=0.001)
            if training:
                mean = x.mean(0)
                var_b = x.var(0, unbiased=False)
                if bn_updates is not None:
                    n = x.shape[0]
                    var_u = var_b * (n / (n - 1)) if n > 1 else var_b
                    bn_updates[p + ".bn.running_mean"] = (
                        (1 - hp.bn_momentum) * sd[p + ".bn.running_mean"] + hp.bn_momentum * mean.detach()
                    )
                    bn_updates[p + ".bn.running_var"] = (
                        (1 - hp.bn_momentum) * sd[p + ".bn.running_var"] + hp.bn_momentum * var_u.detach()
                    )
                    bn_updates[p + ".bn.num_batches_tracked"] = sd[p + ".bn.num_batches_tracked"] + 1
            else:
                mean = sd[p + ".bn.running_mean"]
                var_b = sd[p + ".bn.running_var"]
            x = (x - mean) / torch.sqrt(var_b + hp.bn_eps) * sd[p + ".bn.weight"] + sd[p + ".bn.bias"]
        if spec.use_layer_norm[i]:  # :281 LayerNorm(elementwise_affine=False)
            x = F.layer_norm(x, (x.shape[-1],))
        if spec.relu[i]:  # :282-286
            if relu_inputs is not None:
                relu_inputs[p] = x.detach()
            if relu_slopes is not None and p in relu_slopes:
                x = _ReluWithGivenSlope.apply(x, relu_slopes[p])
            else:
                x = torch.relu(x)
            if spec.return_hidden[i]:  # :312-313
                hidden.append(x)
        if spec.dropout_rate[i] > 0 and training:  # :287-288
            key = p + ".dr"
            if masks is None or key not in masks:
                raise KeyError(f"oracle needs an explicit keep mask for {key}")
            x = x * masks[key].to(x.dtype) / (1.0 - spec.dropout_rate[i])
    return x, hidden


def conditional_layer(spec: CondSpec, sd, prefix: str, z, names: List[str], training: bool, hp: HParams):
    """ConditionalLayer.forward (components.py:365-413): every row goes through the FCBlock of its own condition."""
    out = torch.empty(z.shape[0], spec.fc.layers[-1], dtype=z.dtype)
    rows_of: Dict[str, List[int]] = {}
    for r, n in enumerate(names):
        rows_of.setdefault(n, []).append(r)
    parts = []
    for n, rows in rows_of.items():
        idx = torch.tensor(rows)
        y, _ = fcblock_forward(sd, f"{prefix}.conditions.{n}", spec.fc, z.index_select(0, idx), training, None, hp)
        parts.append((idx, y))
    for idx, y in parts:
        out = out.index_copy(0, idx, y)
    return out


def conditional_layers(spec: CondSpec, sd, z, cond: Dict[str, List[str]], species: str, order: List[str],
                       training: bool, hp: HParams):
    """ConditionalLayers.forward (components.py:586-631) for a given selection order."""
    outs = []
    for key in order:
        if key == "species":
            y, _ = fcblock_forward(sd, f"vae.conditionals.layers.species.{species}", spec.fc, z, training, None, hp)
        elif key in spec.shared:
            y = conditional_layer(spec, sd, f"vae.conditionals.layers.{key}", z, cond[key], training, hp)
        else:
            y = conditional_layer(spec, sd, f"vae.conditionals.layers.{key}.{species}", z, cond[key], training, hp)
        if spec.parallel:
            outs.append(y)
        else:
            z = y
    return torch.cat(outs, dim=1) if outs else z


def model_forward(spec: ModelSpec, sd, x, expert_id: str, eps, training: bool, masks, hp: HParams, bn_updates=None,
                  cond: Optional[Dict[str, List[str]]] = None, cond_order: Optional[List[str]] = None,
                  relu_slopes: Optional[dict] = None, relu_inputs: Optional[dict] = None):
    """CMMVAE.forward (modules/cmmvae.py:85-113) with BaseVAE.forward (modules/vae.py:98-102) and Encoder.forward
    (components.py:783-809).  eps: [B,Z] (K = 1, the reference) or [K,B,Z] (extension).  cond / cond_order: per-row
    condition names per key and the selection order of this forward (models with conditional layers)."""
    enc, dec = spec.experts[expert_id]
    rk = dict(relu_slopes=relu_slopes, relu_inputs=relu_inputs)
    shared, _ = fcblock_forward(sd, f"experts.{expert_id}.encoder", enc, x, training, masks, hp, bn_updates, **rk)
    q, hidden = fcblock_forward(sd, "vae.encoder.fc", spec.vae_encoder, shared, training, masks, hp, bn_updates, **rk)
    mu = F.linear(q, sd["vae.encoder.mean_encoder.weight"], sd["vae.encoder.mean_encoder.bias"])  # :791
    var = torch.exp(F.linear(q, sd["vae.encoder.var_encoder.weight"], sd["vae.encoder.var_encoder.bias"])) + spec.var_eps
    std = var.sqrt()  # :798 Normal(q_m, q_v.sqrt())
    z = mu + std * eps  # :801 rsample == loc + eps * scale ; broadcasts over K
    if spec.softmax_z:  # :801 z_transformation (nn.Softmax(dim=-1) for distribution "ln", :740-741)
        z = torch.softmax(z, dim=-1)
    if spec.hidden_z:  # :803-804 (K = 1 sample)
        hidden = hidden + [z if z.dim() == 2 else z[0]]
    if spec.conditionals is not None:  # vae.py:100 after_reparameterize -> clvae.py:107-111; the returned z is its output
        assert z.dim() == 2 and cond is not None, "conditional layers: K = 1 and per-row conditions required"
        z = conditional_layers(spec.conditionals, sd, z, cond, expert_id, cond_order or spec.conditionals.keys,
                               training, hp)
    zk = z.reshape(-1, z.shape[-1])
    sh, _ = fcblock_forward(sd, "vae.decoder", spec.vae_decoder, zk, training, masks, hp, bn_updates, **rk)
    xhat, _ = fcblock_forward(sd, f"experts.{expert_id}.decoder", dec, sh, training, masks, hp, bn_updates, **rk)
    return {"mu": mu, "std": std, "z": z, "xhat": xhat, "hidden": hidden, "shared_xhat": sh}


def elbo_iwae(mu, std, x, xhat, kl_weight: float, K: int, eps, z):
    """Opt-in "full IWAE" mode of the K-sample extension (not in the reference, parity unpinned): the
    analytic KL is replaced by the sampled log-density ratio inside the log-mean-exp,
        r[k,b]  = log q(z_kb | x_b) - log p(z_kb) = sum_j ( -log s_bj - eps_kbj^2 / 2 + z_kbj^2 / 2 )
        loss    = sum_b -logmeanexp_k( -SE[k,b] - (kl_weight / B) r[k,b] )
    (the 1/B keeps the default objective's scaling: E_eps[r] = KL, so K = 1 equals vae.py:136-152 in expectation).
    Reported beside it: recon_loss = sum_b sum_k w SE, kl_loss = mean_b sum_k w r, w = softmax_k of the log-weights."""
    B = x.shape[0]
    se = ((xhat.reshape(K, B, -1) - x.unsqueeze(0)) ** 2).sum(-1)  # [K,B]
    epsk = eps.reshape(K, B, -1)
    zk = z.reshape(K, B, -1)
    r = (-std.log().unsqueeze(0) - 0.5 * epsk.pow(2) + 0.5 * zk.pow(2)).sum(-1)  # [K,B]
    lw = -se - (kl_weight / B) * r
    loss = (-(torch.logsumexp(lw, dim=0) - torch.log(torch.tensor(float(K))))).sum()
    w = torch.softmax(lw, dim=0).detach()
    return {"loss": loss, "recon_loss": (w * se).sum().detach(), "kl_loss": ((w * r).sum() / B).detach(),
            "kl_weight": kl_weight}


def elbo(mu, std, x, xhat, kl_weight: float, K: int = 1):
    """BaseVAE.elbo (modules/vae.py:136-152): kl_divergence(Normal(mu,std), Normal(0,1)).sum(-1).mean();
    F.mse_loss(xhat, x, reduction="sum"); loss = recon + kl_weight * kl.
    K > 1 (extension, parity unpinned): recon = sum_b -log mean_k exp(-SE[b,k])."""
    var_ratio = std.pow(2)
    kl = (0.5 * (var_ratio + mu.pow(2) - 1 - var_ratio.log())).sum(-1).mean()
    if K == 1:
        recon = F.mse_loss(xhat, x, reduction="sum")
    else:
        B = x.shape[0]
        se = ((xhat.reshape(K, B, -1) - x.unsqueeze(0)) ** 2).sum(-1)  # [K,B]
        recon = (-(torch.logsumexp(-se, dim=0) - torch.log(torch.tensor(float(K))))).sum()
    return {"loss": recon + kl_weight * kl, "recon_loss": recon, "kl_loss": kl, "kl_weight": kl_weight}


class _GRL(torch.autograd.Function):
    """GradientReversalFunction (components.py:879-899)."""

    @staticmethod
    def forward(ctx, x, alpha):
        ctx.alpha = alpha
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.neg() * ctx.alpha, None


def adversarial_losses(spec: ModelSpec, sd, hidden, labels: Dict[str, torch.Tensor], detach: bool, hp: HParams,
                       masks: Optional[dict] = None, relu_slopes: Optional[dict] = None,
                       relu_inputs: Optional[dict] = None):
    """CMMVAEModel.grf (models/cmmvae_model.py:59-101): per adversary, CE(sum) per head then summed.
    Heads are iterated in the order of `labels` (cmmvae_model.py:83: `for condition, label in labels.items()`).
    `masks`: explicit dropout keep masks of adversary encoders with dropout (the adversaries run in training mode);
    `relu_slopes` / `relu_inputs`: as for fcblock_forward (keys "adversarials.<i>.encoder.fc_layers.<j>")."""
    out = []
    for i, (h, adv) in enumerate(zip(hidden, spec.adversarials)):
        h = h.detach() if detach else _GRL.apply(h, 1)
        e, _ = fcblock_forward(sd, f"adversarials.{i}.encoder", adv.encoder, h, True, masks, hp, None, relu_slopes,
                               relu_inputs)
        heads = {}
        for cond, y in labels.items():
            p = f"adversarials.{i}.heads.{cond}.fc_layers.0.lin"
            logits = F.linear(e, sd[p + ".weight"], sd[p + ".bias"])
            heads[cond] = F.cross_entropy(logits, y, reduction="sum")  # cmmvae_model.py:54
        summed = torch.sum(torch.stack(list(heads.values())))
        out.append({"heads": heads, "summed": summed})
    return out


# ------------------------------------------------------------------------------------------------------ optimiser
def grad_norm(grads: List[torch.Tensor]) -> torch.Tensor:
    return torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()


def clip_and_adam(names, sd, grads, opt_state, group: str, clip: Optional[float], hp: HParams):
    """clip_grad_norm_(params, clip) (Lightning clip_gradients "norm", cmmvae_model.py:126-129,203-209) followed by
    torch.optim.Adam.step (single-tensor formulas, amsgrad=False, coupled weight decay)."""
    # parameters without a gradient (conditions absent from the batch) are skipped by clip_grad_norm_ and by
    # torch.optim.Adam alike, and every parameter counts its own steps (state["step"] is per parameter)
    names = [n for n in names if grads.get(n) is not None]
    gl = [grads[n] / hp.world_size for n in names]
    norm = grad_norm(gl)
    coef = 1.0
    if clip is not None and hp.clip_algorithm == "value":  # Lightning clip_gradients(..., "value") = clip_grad_value_
        gl = [g.clamp(-float(clip), float(clip)) for g in gl]
    elif clip is not None:
        coef = min(1.0, float(clip) / (float(norm) + 1e-6))
    st = opt_state.setdefault(group, {"steps": {}, "exp_avg": {}, "exp_avg_sq": {}})
    new = {}
    for n, g in zip(names, gl):
        t = st["steps"].get(n, 0) + 1
        st["steps"][n] = t
        bc1 = 1 - hp.beta1**t
        bc2 = 1 - hp.beta2**t
        p = sd[n]
        g = g * coef + hp.weight_decay * p
        m = st["exp_avg"].get(n, torch.zeros_like(p))
        v = st["exp_avg_sq"].get(n, torch.zeros_like(p))
        m = m + (1 - hp.beta1) * (g - m)
        v = hp.beta2 * v + (1 - hp.beta2) * g * g
        st["exp_avg"][n], st["exp_avg_sq"][n] = m, v
        denom = v.sqrt() / (bc2**0.5) + hp.adam_eps
        new[n] = p - (hp.lr / bc1) * (m / denom)
    return new, norm


# ------------------------------------------------------------------------------------------------------ the step
def train_step(spec: ModelSpec, sd: Dict[str, torch.Tensor], opt_state: dict, x: torch.Tensor, expert_id: str,
               eps: torch.Tensor, masks: Optional[dict], labels: Optional[Dict[str, torch.Tensor]], kl_weight: float,
               hp: HParams, cond: Optional[Dict[str, List[str]]] = None, cond_order: Optional[List[str]] = None,
               relu_slopes: Optional[dict] = None):
    """CMMVAEModel.training_step (models/cmmvae_model.py:138-217).  Returns (outputs, new_sd); opt_state is updated
    in place.  Order: forward, elbo, [D phase: backward/clip/Adam per adversary], [G phase with updated adversaries],
    backward of loss + adv_weight * sum(adv), clip vae, clip expert, Adam vae, Adam expert.
    `relu_slopes`: see _ReluWithGivenSlope; the outputs then also carry every ReLU's input as out["relu_inputs"]."""
    groups = group_param_names(spec)
    live = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    bn_updates: dict = {}
    K = eps.shape[0] if eps.dim() == 3 else 1
    relu_inputs = {} if relu_slopes is not None else None
    fwd = model_forward(spec, live, x, expert_id, eps, True, masks, hp, bn_updates, cond, cond_order, relu_slopes,
                        relu_inputs)
    if hp.elbo_mode == "iwae":
        e = elbo_iwae(fwd["mu"], fwd["std"], x, fwd["xhat"], kl_weight, K, eps, fwd["z"])
    else:
        e = elbo(fwd["mu"], fwd["std"], x, fwd["xhat"], kl_weight, K)
    out = {
        "loss": e["loss"].detach(), "recon_loss": e["recon_loss"].detach(), "kl_loss": e["kl_loss"].detach(),
        "kl_weight": kl_weight, "Mean": fwd["mu"].mean().detach(), "Variance": fwd["std"].pow(2).mean().detach(),
        "z": fwd["z"].detach(), "xhat": fwd["xhat"].detach(), "hidden": [h.detach() for h in fwd["hidden"]],
        "mu": fwd["mu"].detach(), "std": fwd["std"].detach(), "grad_norms": {},
    }
    if relu_inputs is not None:
        out["relu_inputs"] = relu_inputs
    new_sd = dict(sd)
    total = e["loss"]
    if spec.adversarials:
        assert labels is not None
        # (given slopes are the generator phase's: the discriminator phase runs on the weights before its own update)
        d = adversarial_losses(spec, live, fwd["hidden"], labels, True, hp, masks)
        out["discriminator"] = [{"heads": {c: v.detach() for c, v in a["heads"].items()}, "summed": a["summed"].detach()}
                                for a in d]
        for i, a in enumerate(d):  # cmmvae_model.py:120-131
            names = [n for n, _ in groups[f"adversarial_{i + 1}"]]
            gl = torch.autograd.grad(a["summed"], [live[n] for n in names])
            upd, norm = clip_and_adam(names, sd, dict(zip(names, gl)), opt_state, f"adversarial_{i + 1}",
                                      hp.adversarial_clip, hp)
            out["grad_norms"][f"discriminator_{i + 1}"] = norm
            new_sd.update(upd)
            for n in names:
                live[n] = upd[n].detach().clone().requires_grad_(True)
        g = adversarial_losses(spec, live, fwd["hidden"], labels, False, hp, masks, relu_slopes, relu_inputs)  # :134
        out["generator"] = [{"heads": {c: v.detach() for c, v in a["heads"].items()}, "summed": a["summed"].detach()}
                            for a in g]
        for a in g:  # :182-184
            total = total + a["summed"] * hp.adv_weight
    out["total_loss"] = total.detach()
    vae_names = [n for n, _ in groups["vae"]]
    exp_names = [n for n, _ in groups[f"expert_{expert_id}"]]
    adv_names = [n for i in range(len(spec.adversarials)) for n, _ in groups[f"adversarial_{i + 1}"]]
    all_names = vae_names + exp_names + adv_names
    gl = torch.autograd.grad(total, [live[n] for n in all_names], allow_unused=True)
    grads = dict(zip(all_names, gl))  # None: the parameter did not take part in this step (an absent condition)
    out["grads"] = {n: grads[n].detach() for n in vae_names + exp_names if grads[n] is not None}
    for i in range(len(spec.adversarials)):  # logged only (cmmvae_model.py:196-200)
        names = [n for n, _ in groups[f"adversarial_{i + 1}"]]
        out["grad_norms"][f"generator_{i + 1}"] = grad_norm([grads[n] for n in names if grads[n] is not None])
    upd_v, norm_v = clip_and_adam(vae_names, sd, grads, opt_state, "vae", hp.vae_clip, hp)
    upd_e, norm_e = clip_and_adam(exp_names, sd, grads, opt_state, f"expert_{expert_id}", hp.expert_clip, hp)
    out["grad_norms"]["vae"] = norm_v
    out["grad_norms"][f"expert_{expert_id}"] = norm_e
    new_sd.update(upd_v)
    new_sd.update(upd_e)
    new_sd.update(bn_updates)
    return out, new_sd


def eval_step(spec: ModelSpec, sd, x, expert_id: str, eps, kl_weight: float, hp: HParams, cond=None, cond_order=None):
    """CMMVAEModel.validation_step (models/cmmvae_model.py:219-248): eval-mode forward + elbo."""
    with torch.no_grad():
        fwd = model_forward(spec, sd, x, expert_id, eps, False, None, hp, None, cond, cond_order)
        e = elbo(fwd["mu"], fwd["std"], x, fwd["xhat"], kl_weight, 1)
    return {**e, "z": fwd["z"], "xhat": fwd["xhat"], "mu": fwd["mu"]}


def cross_generate(spec: ModelSpec, sd, x, expert_id: str, eps, hp: HParams):
    """CMMVAE.forward(..., cross_generate=True) in eval mode (modules/cmmvae.py:95-107): the shared decoder output is
    decoded through EVERY expert.  Returns {expert: xhat}."""
    with torch.no_grad():
        fwd = model_forward(spec, sd, x, expert_id, eps, False, None, hp)
        return {other: fcblock_forward(sd, f"experts.{other}.decoder", dec, fwd["shared_xhat"], False, None, hp)[0]
                for other, (_, dec) in spec.experts.items()}


def latent_embeddings(spec: ModelSpec, sd, x, expert_id: str, eps, hp: HParams):
    """CMMVAE.get_latent_embeddings (modules/cmmvae.py:115-142): expert encoder + VAE encoder, one rsample -> z."""
    with torch.no_grad():
        return model_forward(spec, sd, x, expert_id, eps, False, None, hp)["z"]


def linear_kl_weight(step_count: int, min_kl=1e-7, max_kl=1e-5, warmup_steps=1e3, climax_steps=1e4) -> float:
    """LinearKLAnnealingFn (modules/base/annealing_fn.py:17-42) after `step_count` calls of .step()."""
    x = -warmup_steps + step_count
    if step_count == 0 or x < 0:
        return min_kl
    m = (max_kl - min_kl) / climax_steps
    return min(max(m * x + min_kl, min_kl), max_kl)


# ---------------------------------------------------------------------------------------------------- synthetic data
def synthetic_counts(B: int, G: int, seed: int = 1234) -> torch.Tensor:
    """SURVEY 8d synthetic input: c ~ Poisson(lambda_g), lambda_g = 0.15 LogNormal(0,1); x = log1p(1e4 c / rowsum)
    (mirrors scripts/data-preprocessing/data_processing_functions.py:12-31)."""
    g = torch.Generator().manual_seed(seed)
    lam = 0.15 * torch.exp(torch.randn(G, generator=g))
    c = torch.poisson(lam.expand(B, G), generator=g)
    rs = c.sum(1, keepdim=True).clamp_min(1.0)
    return torch.log1p(1e4 * c / rs)


# ------------------------------------------------------------------------------------- in-place stepping (timing leg)
class InPlaceStepper:
    """The same training step as `train_step`, kept the way a trainer keeps it: parameters are leaf tensors updated in
    place by `torch.optim.Adam` (one per expert, one for the VAE, one per adversary -- cmmvae_model.py:299-324),
    gradients come from `backward()`, clipping is `clip_grad_norm_`.  `train_step` is the functional statement the parity
    tests compare against (it clones the whole state every step: fine for checking, wrong for timing); this class is what
    bench.py's `cpu_baseline` times.  tests/test_oracle_golden.py checks that both give the same numbers."""

    def __init__(self, spec: ModelSpec, sd: Dict[str, torch.Tensor], hp: HParams):
        self.spec, self.hp = spec, hp
        self.sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not _is_buffer(k)
                       else v.detach().clone()) for k, v in sd.items()}
        self.groups = {g: [n for n, _ in names] for g, names in group_param_names(spec).items()}
        self.optim = {g: torch.optim.Adam([self.sd[n] for n in names], lr=hp.lr, betas=(hp.beta1, hp.beta2),
                                          eps=hp.adam_eps, weight_decay=hp.weight_decay)
                      for g, names in self.groups.items()}

    def _clip(self, group: str, max_norm: Optional[float]) -> torch.Tensor:
        params = [self.sd[n] for n in self.groups[group] if self.sd[n].grad is not None]
        if max_norm is None:
            return grad_norm([p.grad for p in params])
        return torch.nn.utils.clip_grad_norm_(params, max_norm)

    def step(self, x, expert_id: str, eps, masks, labels, kl_weight: float) -> dict:
        spec, hp, sd = self.spec, self.hp, self.sd
        live = [f"expert_{expert_id}", "vae"] + [g for g in self.groups if g.startswith("adversarial_")]
        for g in live:
            self.optim[g].zero_grad(set_to_none=True)
        bn_updates: dict = {}
        K = eps.shape[0] if eps.dim() == 3 else 1
        fwd = model_forward(spec, sd, x, expert_id, eps, True, masks, hp, bn_updates)
        e = elbo(fwd["mu"], fwd["std"], x, fwd["xhat"], kl_weight, K)
        out = {"loss": e["loss"].detach(), "recon_loss": e["recon_loss"].detach(), "kl_loss": e["kl_loss"].detach(),
               "grad_norms": {}}
        total = e["loss"]
        if spec.adversarials:
            d = adversarial_losses(spec, sd, fwd["hidden"], labels, True, hp, masks)
            for i, a in enumerate(d, start=1):
                g = f"adversarial_{i}"
                a["summed"].backward()
                out["grad_norms"][f"discriminator_{i}"] = self._clip(g, hp.adversarial_clip)
                self.optim[g].step()
                self.optim[g].zero_grad(set_to_none=True)
            for a in adversarial_losses(spec, sd, fwd["hidden"], labels, False, hp, masks):
                total = total + a["summed"] * hp.adv_weight
        out["total_loss"] = total.detach()
        total.backward()
        out["grad_norms"]["vae"] = self._clip("vae", hp.vae_clip)
        out["grad_norms"][f"expert_{expert_id}"] = self._clip(f"expert_{expert_id}", hp.expert_clip)
        self.optim["vae"].step()
        self.optim[f"expert_{expert_id}"].step()
        with torch.no_grad():
            for k, v in bn_updates.items():
                sd[k].copy_(v)
        return out

    def state(self) -> Dict[str, torch.Tensor]:
        return {k: v.detach() for k, v in self.sd.items()}


def _is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))
