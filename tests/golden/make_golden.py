#!/usr/bin/env python3
"""Generate golden input/output vectors by running the REFERENCE's own modules (zdebruine/MMVAE, cmmvae).

Runs only in the build container, where /root/reference exists:   python tests/golden/make_golden.py
It imports `cmmvae.modules` from /root/reference/src (nothing is copied), drives them with stock torch calls in the
order of CMMVAEModel.training_step (models/cmmvae_model.py:138-217; that class itself needs `lightning`, which is
not installed, so its ~80 lines of orchestration are driven here: manual_backward == loss.backward(),
clip_gradients(..., "norm") == torch.nn.utils.clip_grad_norm_, optimizers == torch.optim.Adam(lr=5e-3, wd=1e-6)),
and writes small .npz fixtures (inputs, explicit noise / dropout keep-masks, initial parameters, expected outputs).
The fixtures are data only; they travel to the GPU box, the reference does not.

Randomness is made explicit so any implementation can replay it:
  * nn.Dropout modules of the constructed reference model are swapped for a module applying a given keep mask
    (same arithmetic as F.dropout: x * mask / (1 - p));
  * torch.distributions.Normal.rsample is patched to `loc + eps * scale` with a given eps (what it computes anyway).
"""
import json
import os
import random
import sys
import tempfile
import warnings

import numpy as np
import pandas as pd
import torch
import torch.nn as nn

REF_SRC = "/root/reference/src"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(OUT_DIR))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class ExplicitDropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p
        self.mask = None

    def forward(self, x):
        if not self.training:
            return x
        return x * self.mask.to(x.dtype) / (1.0 - self.p)


def fc_cfg(base, layers, dropout=0.0, bn=False, relu=True, return_hidden=False):
    return base.FCBlockConfig(
        layers=list(layers), dropout_rate=dropout, use_batch_norm=bn, use_layer_norm=False,
        activation_fn=nn.ReLU if relu else None, return_hidden=return_hidden,
    )


def condition_value(key, species, i):
    """Raw metadata value of condition i of `key` (dots on purpose: ConditionalLayer replaces them, components.py:353-363)."""
    return f"{key}.{species}.{i}" if species else f"{key}.{i}"


def write_condition_files(case, tmpdir):
    """unique_expression_{key}.csv under shared/ and {species}/ (components.py:420-464)."""
    root = os.path.join(tmpdir, "conditionals")
    cond = case["cond"]
    os.makedirs(os.path.join(root, "shared"), exist_ok=True)
    for key, n in cond["shared"].items():
        pd.Series([condition_value(key, None, i) for i in range(n)]).to_csv(
            os.path.join(root, "shared", f"unique_expression_{key}.csv"), header=False, index=False)
    for key, by_species in cond["species_specific"].items():
        for species, n in by_species.items():
            os.makedirs(os.path.join(root, species), exist_ok=True)
            pd.Series([condition_value(key, species, i) for i in range(n)]).to_csv(
                os.path.join(root, species, f"unique_expression_{key}.csv"), header=False, index=False)
    return root


def conditional_kwargs(base, case, tmpdir):
    cond = case["cond"]
    root = write_condition_files(case, tmpdir)
    Z = case["Z"]
    kw = dict(
        conditional_config=base.FCBlockConfig(layers=[Z], dropout_rate=0.0, use_batch_norm=False,
                                              use_layer_norm=cond["layer_norm"], activation_fn=None),
        conditionals_directory=root, conditionals=list(cond["keys"]),
        # (selection_order: null, as configs/model/compare/adversarial-conditional.yaml writes it, does not instantiate at
        # the reference's HEAD: components.py:519 indexes it -- so the shuffled order is only reachable as "parallel")
        selection_order=["parallel"] if cond["parallel"] else list(cond["keys"]),
    )
    if cond["parallel"]:
        kw["concat_config"] = base.ConcatBlockConfig(dropout_rate=0.0, use_batch_norm=False, use_layer_norm=False,
                                                     activation_fn=nn.ReLU)
    return kw


def build_reference(case, tmpdir):
    from cmmvae.modules import CMMVAE, CLVAE, base
    from cmmvae.modules.base.init import he_init_weights

    base.Adversarial.labels.clear()
    experts = []
    for eid, G in case["experts"].items():
        enc = [G] + case["expert_hidden"]
        dec = case["expert_hidden"][::-1] + [G]
        experts.append(base.Expert(eid, fc_cfg(base, enc, dropout=case["dropout"], bn=True),
                                   fc_cfg(base, dec)))
    cond_kwargs = {}
    if case.get("cond"):
        cond_kwargs = conditional_kwargs(base, case, tmpdir)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = CLVAE(
            latent_dim=case["Z"],
            encoder_config=fc_cfg(base, [case["expert_hidden"][-1]] + case["vae_hidden"], bn=True, return_hidden=True,
                                  dropout=case.get("vae_dropout", 0.0)),
            decoder_config=fc_cfg(base, [case["Z"]] + case["vae_hidden"][::-1] + [case["expert_hidden"][-1]]),
            hidden_z=case["hidden_z"], **cond_kwargs,
            **({"distribution": case["distribution"]} if case.get("distribution") else {}),
        )
    advs = None
    if case.get("adversarials"):
        os.makedirs(os.path.join(tmpdir, "human"), exist_ok=True)
        for cond, n in case["conditions"].items():
            pd.Series([f"{cond}_{i}" for i in range(n)]).to_csv(
                os.path.join(tmpdir, "human", f"unique_expression_{cond}.csv"), header=False, index=False)
        advs = []
        for enc_layers in case["adversarials"]:
            advs.append(base.Adversarial(
                encoder=fc_cfg(base, enc_layers, dropout=case.get("adv_dropout", 0.0)),
                heads=fc_cfg(base, [enc_layers[-1]], relu=False),
                conditions=list(case["conditions"].keys()), labels_dir=tmpdir))
    module = CMMVAE(vae, base.Experts(experts), advs)
    torch.manual_seed(case["seed"])
    he_init_weights(module)  # BaseModel.init_weights (models/base_model.py:106-109)
    # make BN affine / biases non-trivial so that their gradients and Adam updates are exercised
    g = torch.Generator().manual_seed(case["seed"] + 1)
    for name, p in module.named_parameters():
        if name.endswith("bias") or name.endswith("bn.weight"):
            with torch.no_grad():
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    # swap nn.Dropout for explicit-mask modules
    drops = {}
    for name, m in list(module.named_modules()):
        for cname, child in list(m.named_children()):
            if isinstance(child, nn.Dropout):
                ed = ExplicitDropout(child.p)
                setattr(m, cname, ed)
                drops[f"{name}.{cname}"] = ed
    return module, drops


def total_norm(params):
    gs = [p.grad for p in params if p.grad is not None]
    return torch.sqrt(sum((g.double() ** 2).sum() for g in gs)).float()


def run_case(case):
    from cmmvae.modules.base import GradientReversalFunction, Adversarial
    from torch.distributions import Normal

    out = {}
    with tempfile.TemporaryDirectory() as tmpdir:
        module, drops = build_reference(case, tmpdir)
        module.train()
        for k, v in module.state_dict().items():
            out[f"sd0/{k}"] = v.numpy().copy()
        optims = {f"expert_{eid}": torch.optim.Adam(m.parameters(), lr=5e-3, weight_decay=1e-6)
                  for eid, m in module.experts.items()}
        optims["vae"] = torch.optim.Adam(module.vae.parameters(), lr=5e-3, weight_decay=1e-6)
        has_adv = bool(case.get("adversarials"))
        if has_adv:
            for i, adv in enumerate(module.adversarials, start=1):
                optims[f"adversarial_{i}"] = torch.optim.Adam(adv.parameters(), lr=5e-3, weight_decay=1e-6)
        ce = nn.CrossEntropyLoss(reduction="sum")
        adv_weight = case.get("adv_weight") or 1.0  # cmmvae_model.py:56
        g = torch.Generator().manual_seed(case["seed"] + 2)
        if case.get("regen"):
            from tests import helpers as H

            regen = H.RegenStream(case)
        orig_rsample = Normal.rsample
        eps_holder = {}
        Normal.rsample = lambda self, sample_shape=torch.Size(): self.loc + eps_holder["eps"] * self.scale
        try:
            for t, eid in enumerate(case["schedule"]):
                G = case["experts"][eid]
                B = case["B"]
                lam = 0.15 * torch.exp(torch.randn(G, generator=g))
                c = torch.poisson(lam.expand(B, G) * case.get("lam_scale", 8.0), generator=g)
                x = torch.log1p(1e4 * c / c.sum(1, keepdim=True).clamp_min(1.0))
                eps = torch.randn(B, case["Z"], generator=g)
                eps_holder["eps"] = eps
                out[f"step{t}/in/x"] = x.numpy()
                out[f"step{t}/in/eps"] = eps.numpy()
                kl_weight = case["kl_weights"][t]
                meta = {}
                labels = {}
                if has_adv:
                    for cond, n in case["conditions"].items():
                        idx = torch.randint(0, n, (B,), generator=g)
                        meta[cond] = [f"{cond}_{int(i)}" for i in idx]
                        out[f"step{t}/in/labels/{cond}"] = idx.numpy().astype(np.int64)
                if case.get("cond"):
                    cond = case["cond"]
                    for key, n in cond["shared"].items():
                        idx = torch.randint(0, n, (B,), generator=g)
                        meta[key] = [condition_value(key, None, int(i)) for i in idx]
                        out[f"step{t}/in/cond/{key}"] = idx.numpy().astype(np.int64)
                    for key, by_species in cond["species_specific"].items():
                        idx = torch.randint(0, by_species[eid], (B,), generator=g)
                        meta[key] = [condition_value(key, eid, int(i)) for i in idx]
                        out[f"step{t}/in/cond/{key}"] = idx.numpy().astype(np.int64)
                    # "parallel" / unordered selection shuffles with Python's random (components.py:601-603)
                    random.seed(case["seed"] * 100 + t)
                metadata = pd.DataFrame(meta if meta else {"dummy": [0] * B})
                for name, ed in drops.items():
                    if name.startswith((f"experts.{eid}.", "vae.", "adversarials.")):
                        # size of this dropout's input = out_features of the Linear in the same layer
                        lin = dict(module.named_modules())[name.rsplit(".", 1)[0] + ".lin"]
                        ed.mask = (torch.rand(B, lin.out_features, generator=g) >= ed.p).to(torch.uint8)
                        out[f"step{t}/in/mask/{name}"] = ed.mask.numpy()

                if case.get("regen"):  # the tests re-draw these inputs with tests/helpers.RegenStream: same numbers?
                    rx, reps, rmasks, rlabels = regen.step(t, eid)
                    assert torch.equal(rx, x) and torch.equal(reps, eps)
                    assert all(torch.equal(rlabels[k], torch.from_numpy(out[f"step{t}/in/labels/{k}"])) for k in rlabels)
                    assert {f"step{t}/in/mask/{k}" for k in rmasks} == {k for k in out if k.startswith(f"step{t}/in/mask/")}
                    assert all(torch.equal(m, torch.from_numpy(out[f"step{t}/in/mask/{k}"])) for k, m in rmasks.items())

                if case.get("regen"):  # independent single steps: every step starts from a regenerated state
                    from cmmvae.modules.base.init import he_init_weights as ref_init

                    count, moments = H.regen_state(case, t, module, eid, init_fn=ref_init)
                    by_name = dict(module.named_parameters())
                    for key in ["vae", f"expert_{eid}"] + [k for k in optims if k.startswith("adversarial_")]:
                        optims[key].state.clear()
                        for p_ in optims[key].param_groups[0]["params"]:
                            n_ = next(n for n, q in by_name.items() if q is p_)
                            if n_ in moments:
                                optims[key].state[p_] = {"step": torch.tensor(float(count)), "exp_avg": moments[n_][0].clone(),
                                                         "exp_avg_sq": moments[n_][1].clone()}
                    for k_, v_ in module.state_dict().items():
                        out[f"step{t}/sd_in/{k_}"] = v_.numpy().copy()

                # ---- training_step, cmmvae_model.py:138-217
                metadata["species"] = eid
                for o in optims.values():
                    o.zero_grad()  # :151-155 (all adversarial optimisers too)
                qz, pz, z, xhats, hidden = module(x=x, metadata=metadata, expert_id=eid)
                loss_dict = module.vae.elbo(qz, pz, x, xhats[eid], kl_weight)
                res = {
                    "loss": loss_dict["loss"], "recon_loss": loss_dict["recon_loss"], "kl_loss": loss_dict["kl_loss"],
                    "Mean": qz.mean.mean(), "Variance": qz.variance.mean(), "z": z, "xhat": xhats[eid],
                    "mu": qz.loc, "std": qz.scale,
                }
                for i, h in enumerate(hidden):
                    res[f"hidden/{i}"] = h
                total = loss_dict["loss"]
                if has_adv:
                    # gradient_reversal_domain_classifier, :103-136
                    lab = {}
                    for cond, mp in Adversarial.labels.items():
                        lab[cond] = torch.tensor([mp[v] for v in metadata[cond].values])

                    def grf(detach):  # :59-101
                        sums = []
                        for i, (h, adv) in enumerate(zip(hidden, module.adversarials), start=1):
                            h = h.detach() if detach else GradientReversalFunction.apply(h, 1)
                            enc = adv.encoder(h)
                            heads = []
                            for cond, y in lab.items():
                                l_ = ce(adv.heads[cond](enc), y)
                                heads.append(l_)
                                res[f"{'discriminator' if detach else 'generator'}_{i}/{cond}"] = l_
                            s = torch.sum(torch.stack(heads))
                            res[f"{'discriminator' if detach else 'generator'}_{i}/summed"] = s
                            sums.append(s)
                        return sums

                    d_losses = grf(True)
                    for i, (dl, adv) in enumerate(zip(d_losses, module.adversarials), start=1):
                        opt = optims[f"adversarial_{i}"]
                        dl.backward()
                        res[f"grad_norms/discriminator_{i}"] = total_norm(list(adv.parameters()))
                        torch.nn.utils.clip_grad_norm_(list(adv.parameters()), 10.0)
                        opt.step()
                        opt.zero_grad()
                    for gl_ in grf(False):
                        total = total + gl_ * adv_weight  # :182-184
                total.backward()  # :187
                res["total_loss"] = total
                res["grad_norms/vae"] = total_norm(list(module.vae.parameters()))
                res[f"grad_norms/expert_{eid}"] = total_norm(list(module.experts[eid].parameters()))
                if has_adv:
                    for i, adv in enumerate(module.adversarials, start=1):
                        res[f"grad_norms/generator_{i}"] = total_norm(list(adv.parameters()))
                for n_, p in list(module.vae.named_parameters()):
                    if p.grad is not None:  # parameters of conditions absent from the batch have no gradient
                        out[f"step{t}/grad/vae.{n_}"] = p.grad.numpy().copy()
                for n_, p in list(module.experts[eid].named_parameters()):
                    out[f"step{t}/grad/experts.{eid}.{n_}"] = p.grad.numpy().copy()
                if case.get("clip_value"):  # GradientClipConfig(val, "value") (config.py:4-26) -> clip_grad_value_
                    torch.nn.utils.clip_grad_value_(list(module.vae.parameters()), case["clip_value"])
                    torch.nn.utils.clip_grad_value_(list(module.experts[eid].parameters()), case["clip_value"])
                else:
                    torch.nn.utils.clip_grad_norm_(list(module.vae.parameters()), 10.0)  # :203-204
                    torch.nn.utils.clip_grad_norm_(list(module.experts[eid].parameters()), 10.0)  # :206-209
                optims["vae"].step()  # :212-213
                optims[f"expert_{eid}"].step()
                for k, v in res.items():
                    out[f"step{t}/out/{k}"] = v.detach().numpy().copy()
                for k, v in module.state_dict().items():
                    out[f"step{t}/sd/{k}"] = v.numpy().copy()

            # ---- validation_step (cmmvae_model.py:219-248): eval-mode forward + elbo on the last batch
            module.eval()
            if case.get("cond"):
                random.seed(case["seed"] * 100 + 99)
            with torch.no_grad():
                qz, pz, z, xhats, hidden = module(x, metadata, eid)
                ld = module.vae.elbo(qz, pz, x, xhats[eid], 1.0)
            out["eval/expert_id"] = np.array(eid)
            for k in ("loss", "recon_loss", "kl_loss"):
                out[f"eval/out/{k}"] = ld[k].numpy().copy()
            out["eval/out/z"] = z.numpy().copy()
            out["eval/out/xhat"] = xhats[eid].numpy().copy()
            out["eval/out/mu"] = qz.loc.numpy().copy()
            # ---- cross-generation (cmmvae.py:95-107; runners/cross_generation.py:87-152): decode the shared latent
            #      through every expert; and the predict path, get_latent_embeddings (cmmvae.py:115-142)
            if case.get("cond"):
                random.seed(case["seed"] * 100 + 99)
            with torch.no_grad():
                _, _, _, xh_all, _ = module(x, metadata, eid, cross_generate=True)
                emb = module.get_latent_embeddings(x, metadata, eid)
            for other, xh in xh_all.items():
                out[f"eval/out/xhat_cross/{other}"] = xh.numpy().copy()
            out["eval/out/embedding_z"] = emb["z"][0].numpy().copy()
        finally:
            Normal.rsample = orig_rsample
    if case.get("regen"):
        out = compact_outputs(out)
    out["case_json"] = np.array(json.dumps(case))
    return out


def compact_outputs(out):
    """Full-size ("regen") cases: inputs and initial parameters are regenerated from the seeds by the tests, so only
    checksums + sampled entries of every tensor are kept (tests/helpers.compact); scalars stay as they are."""
    from tests import helpers as H

    small = {}
    for k, v in out.items():
        v = np.asarray(v)
        if "/in/" in k:
            if k.endswith("/in/x"):  # guards the regeneration of the batch
                small[k + "/sumsq"] = np.array((v.astype(np.float64) ** 2).sum())
            continue
        if v.ndim == 0:
            small[k] = v
            continue
        if k.startswith("sd0/"):
            continue  # step 0 starts from step0/sd_in like every other step
        name = k.split("/", 2)[2]
        for field, arr in H.compact(name, v, light="/sd_in/" in k).items():
            small[f"{k}/{field}"] = arr
    return small


def annealing_vectors():
    from cmmvae.modules.base import LinearKLAnnealingFn, KLAnnealingFn

    out = {}
    for i, kw in enumerate([dict(), dict(min_kl_weight=0.0, max_kl_weight=1.0, warmup_steps=3, climax_steps=5)]):
        fn = LinearKLAnnealingFn(**kw)
        vals = [fn.kl_weight]
        for _ in range(20):
            fn.step()
            vals.append(fn.kl_weight)
        out[f"linear{i}/kwargs"] = np.array(json.dumps(kw))
        out[f"linear{i}/values"] = np.array(vals, dtype=np.float64)
    c = KLAnnealingFn(0.25)
    c.step()
    out["const/values"] = np.array([c.kl_weight])
    return out


CASES = {
    "c1_small": dict(seed=11, experts={"human": 64}, expert_hidden=[48, 24], vae_hidden=[16], Z=8, B=8, dropout=0.1,
                     hidden_z=False, schedule=["human", "human"], kl_weights=[1.0, 0.5]),
    "two_mod_odd": dict(seed=23, experts={"human": 257, "mouse": 131}, expert_hidden=[72, 40], vae_hidden=[24], Z=10,
                        B=33, dropout=0.1, hidden_z=True, schedule=["human", "mouse", "human"],
                        kl_weights=[1.0, 1.0, 1.0]),
    "adversarial": dict(seed=37, experts={"human": 96, "mouse": 80}, expert_hidden=[64, 32], vae_hidden=[24], Z=12, B=16,
                        dropout=0.1, hidden_z=True, schedule=["human", "mouse"], kl_weights=[1.0, 1.0],
                        adversarials=[[24, 16, 8], [12, 8]], conditions={"assay": 5, "sex": 2, "donor_id": 37},
                        adv_weight=25),
    # dropout outside the expert encoder: in the VAE encoder and in both adversary encoders.  One keep mask per Dropout
    # module and step: the two adversarial phases (discriminator / generator) see the same mask in this fixture
    "adv_dropout": dict(seed=71, experts={"human": 96, "mouse": 80}, expert_hidden=[64, 32], vae_hidden=[24], Z=12, B=16,
                        dropout=0.1, vae_dropout=0.1, adv_dropout=0.25, hidden_z=True, schedule=["human", "mouse", "human"],
                        kl_weights=[1.0, 1.0, 1.0], adversarials=[[24, 16, 8], [12, 8]],
                        conditions={"assay": 5, "sex": 2, "donor_id": 37}, adv_weight=25),
    # conditional layers after the reparameterisation: shared and species-specific ConditionalLayers plus
    # the per-species block, applied in a fixed order / concatenated in shuffled order ("parallel"); some conditions
    # are absent from a batch, so their parameters have no gradient and torch's Adam skips them (own step counts)
    "cond_seq": dict(seed=41, experts={"human": 48, "mouse": 40}, expert_hidden=[32, 16], vae_hidden=[12], Z=8, B=12,
                     dropout=0.1, hidden_z=False, schedule=["human", "mouse", "human"], kl_weights=[1.0, 1.0, 1.0],
                     cond=dict(keys=["assay", "donor_id", "species", "sex"], shared={"assay": 4, "sex": 2},
                               species_specific={"donor_id": {"human": 5, "mouse": 3}}, layer_norm=True,
                               parallel=False)),
    "cond_par": dict(seed=43, experts={"human": 48, "mouse": 40}, expert_hidden=[32, 16], vae_hidden=[12], Z=8, B=12,
                     dropout=0.1, hidden_z=True, schedule=["human", "mouse", "human"], kl_weights=[1.0, 0.5, 1.0],
                     cond=dict(keys=["assay", "donor_id", "species", "sex"], shared={"assay": 4, "sex": 2},
                               species_specific={"donor_id": {"human": 5, "mouse": 3}}, layer_norm=True,
                               parallel=True)),
    # the reference's adversarial-conditional configuration in small: conditional layers AND two adversaries (on h1
    # and on z, which is taken BEFORE the conditional layers)
    "cond_adv": dict(seed=47, experts={"human": 48, "mouse": 40}, expert_hidden=[32, 16], vae_hidden=[12], Z=8, B=12,
                     dropout=0.1, hidden_z=True, schedule=["human", "mouse", "human"], kl_weights=[1.0, 1.0, 0.5],
                     adversarials=[[12, 8, 6], [8, 6]], conditions={"tissue": 4, "dev_stage": 3}, adv_weight=25,
                     cond=dict(keys=["assay", "donor_id", "species", "sex"], shared={"assay": 4, "sex": 2},
                               species_specific={"donor_id": {"human": 5, "mouse": 3}}, layer_norm=True,
                               parallel=False)),
    # Encoder(distribution="ln"): softmax over the latent sample (components.py:740-741,801)
    "ln_dist": dict(seed=29, experts={"human": 96, "mouse": 72}, expert_hidden=[64, 32], vae_hidden=[24], Z=12, B=20,
                    dropout=0.1, hidden_z=True, distribution="ln", schedule=["human", "mouse", "human"],
                    kl_weights=[1.0, 1.0, 0.5],
                    # the softmax damps the encoder's gradients to rounding level in places, and a cold Adam step is
                    # sign-like (+-lr whatever the magnitude): post-step parameters are held to 5e-4 instead of 1e-4
                    # (gradients themselves to 1e-4 as everywhere)
                    param_tol=5e-4),
    # GradientClipConfig(algorithm="value") (config.py:8): Lightning clip_gradients(..., "value") = clip_grad_value_
    "clip_value": dict(seed=31, experts={"human": 96, "mouse": 72}, expert_hidden=[64, 32], vae_hidden=[24], Z=12, B=20,
                       dropout=0.1, hidden_z=False, clip_value=0.05, schedule=["human", "mouse", "human"],
                       kl_weights=[1.0, 1.0, 1.0]),
    # ---- full-size ("regen") cases: the kernels the benchmark times, pinned to the reference.  Initial parameters and
    # inputs are regenerated from the seeds (tests/helpers.regen_initial_state / RegenStream); the fixture keeps
    # checksums and sampled entries.  lam_scale 1.0 = the benchmark's count distribution (~10 % non-zeros).
    # mid_odd: gene counts / batch that are multiples of nothing (K tails, edge groups, zero-slack rows of the bf16x3 kernels)
    "mid_odd": dict(seed=53, experts={"human": 10001, "mouse": 8190}, expert_hidden=[1024, 512], vae_hidden=[256], Z=128,
                    B=500, dropout=0.1, hidden_z=False, schedule=["human", "mouse", "human", "mouse"],
                    kl_weights=[1.0, 1.0, 0.5, 0.5], regen=True, lam_scale=1.0),
    # BASELINE config 2: the configuration the metric is quoted on
    "c2_full": dict(seed=59, experts={"human": 20000, "mouse": 20000}, expert_hidden=[1024, 512], vae_hidden=[256],
                    Z=128, B=512, dropout=0.1, hidden_z=False, schedule=["human", "mouse", "human", "mouse"],
                    kl_weights=[1.0, 1.0, 1.0, 1.0], regen=True, lam_scale=1.0),
    # BASELINE config 4 per GPU: + two adversaries (on h1 and z) with the class counts of data/conditional_layers/*.csv
    "c4_full": dict(seed=61, experts={"human": 20000, "mouse": 20000}, expert_hidden=[1024, 512], vae_hidden=[256],
                    Z=128, B=512, dropout=0.1, hidden_z=True, schedule=["human", "mouse", "human"],
                    kl_weights=[1.0, 1.0, 1.0], adversarials=[[256, 128, 64], [128, 64]],
                    conditions={"assay": 8, "sex": 2, "dataset_id": 273, "donor_id": 4644}, adv_weight=25,
                    regen=True, lam_scale=1.0),
    # BASELINE config 5 per GPU at K = 1 (the reference has no K): three expert channels, 30 000 genes, batch 1024
    "c5_three_mod": dict(seed=67, experts={"human": 30000, "mouse": 30000, "macaque": 30000}, expert_hidden=[1024, 512],
                         vae_hidden=[256], Z=128, B=1024, dropout=0.1, hidden_z=False,
                         schedule=["human", "mouse", "macaque", "human"], kl_weights=[1.0, 1.0, 1.0, 1.0],
                         regen=True, lam_scale=1.0),
}


def main():
    if not os.path.isdir(REF_SRC):
        print(f"{REF_SRC} not present: golden vectors can only be generated in the build container; nothing done.")
        return 0
    sys.path.insert(0, REF_SRC)
    only = sys.argv[1:]
    for name, case in CASES.items():
        if only and name not in only:
            continue
        torch.set_num_threads(8 if case.get("regen") else 1)
        out = run_case(dict(case))
        path = os.path.join(OUT_DIR, f"{name}.npz")
        if os.path.exists(path):  # regenerated vectors must reproduce the committed ones bit for bit
            old = np.load(path)
            for k in old.files:
                assert k in out and np.array_equal(np.asarray(old[k]), np.asarray(out[k])), f"{name}: {k} changed"
        np.savez_compressed(path, **out)
        print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
    if only:
        return 0
    ann = annealing_vectors()
    np.savez_compressed(os.path.join(OUT_DIR, "annealing.npz"), **ann)
    print("wrote annealing.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
