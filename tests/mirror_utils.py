"""Builds the mmvae_amd mirror for a golden case and replays the golden training steps through it."""
import os
import tempfile
import warnings

import numpy as np
import pandas as pd
import torch
import torch.nn as nn

from tests import helpers as H


def build_mirror(case, device, tmpdir, use_engine=False):
    from mmvae_amd.models import CMMVAEModel
    from mmvae_amd.modules import CMMVAE, CLVAE, base
    from mmvae_amd.config import AutogradConfig, GradientClipConfig

    def cfg(layers, dropout=0.0, bn=False, relu=True, return_hidden=False):
        return base.FCBlockConfig(layers=list(layers), dropout_rate=dropout, use_batch_norm=bn, use_layer_norm=False,
                                  activation_fn=nn.ReLU if relu else None, return_hidden=return_hidden)

    base.Adversarial.labels.clear()
    eh, vh, Z = case["expert_hidden"], case["vae_hidden"], case["Z"]
    experts = [base.Expert(eid, cfg([G] + eh, dropout=case["dropout"], bn=True), cfg(eh[::-1] + [G]))
               for eid, G in case["experts"].items()]
    cond_kwargs = {}
    if case.get("cond"):
        c = case["cond"]
        root = os.path.join(tmpdir, "conditionals")
        os.makedirs(os.path.join(root, "shared"), exist_ok=True)
        for key, n in c["shared"].items():
            pd.Series([H.cond_value(key, None, i) for i in range(n)]).to_csv(
                os.path.join(root, "shared", f"unique_expression_{key}.csv"), header=False, index=False)
        for key, by_species in c["species_specific"].items():
            for species, n in by_species.items():
                os.makedirs(os.path.join(root, species), exist_ok=True)
                pd.Series([H.cond_value(key, species, i) for i in range(n)]).to_csv(
                    os.path.join(root, species, f"unique_expression_{key}.csv"), header=False, index=False)
        cond_kwargs = dict(
            conditional_config=base.FCBlockConfig(layers=[Z], dropout_rate=0.0, use_batch_norm=False,
                                                  use_layer_norm=c["layer_norm"], activation_fn=None),
            conditionals_directory=root, conditionals=list(c["keys"]),
            selection_order=["parallel"] if c["parallel"] else list(c["keys"]))
        if c["parallel"]:
            cond_kwargs["concat_config"] = base.ConcatBlockConfig(dropout_rate=0.0, use_batch_norm=False,
                                                                  use_layer_norm=False, activation_fn=nn.ReLU)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = CLVAE(latent_dim=Z, encoder_config=cfg([eh[-1]] + vh, bn=True, return_hidden=True),
                    decoder_config=cfg([Z] + vh[::-1] + [eh[-1]]), hidden_z=case["hidden_z"], **cond_kwargs)
    advs = None
    if case.get("adversarials"):
        os.makedirs(os.path.join(tmpdir, "human"), exist_ok=True)
        for cond, n in case["conditions"].items():
            pd.Series([f"{cond}_{i}" for i in range(n)]).to_csv(
                os.path.join(tmpdir, "human", f"unique_expression_{cond}.csv"), header=False, index=False)
        advs = [base.Adversarial(encoder=cfg(enc), heads=cfg([enc[-1]], relu=False),
                                 conditions=list(case["conditions"].keys()), labels_dir=tmpdir)
                for enc in case["adversarials"]]
    clip = lambda: GradientClipConfig(val=10, algorithm="norm")
    model = CMMVAEModel(CMMVAE(vae, base.Experts(experts), advs), adv_weight=case.get("adv_weight"),
                        autograd_config=AutogradConfig(clip(), clip(), clip()), use_engine=use_engine)
    return model.to(device)


def load_state(model, z, prefix):
    sd = {k: v for k, v in H.sd_from(z, prefix).items()}
    missing, unexpected = model.module.load_state_dict(sd, strict=True), None
    return missing


def replay_training(name, device, use_engine=False, check=True, prepare=None):
    """Runs the golden schedule through CMMVAEModel.training_step; returns list of per-step result dicts."""
    from mmvae_amd import backend

    case, z = H.load_case(name)
    results = []
    with tempfile.TemporaryDirectory() as tmpdir, backend.cpu_plumbing(device == "cpu"):
        model = build_mirror(case, device, tmpdir, use_engine=use_engine)
        load_state(model, z, "sd0/")
        model.train()
        model.trainer.set_stage("training")
        if prepare is not None:
            prepare(model)
        for t, eid in enumerate(case["schedule"]):
            x, eps, masks, labels = H.step_inputs(z, t)
            model.kl_annealing_fn.kl_weight = case["kl_weights"][t]
            model.module.vae.encoder.explicit_eps = eps.to(device)
            enc = model.module.experts[eid].encoder
            enc.explicit_masks = {}
            for k, m in masks.items():
                if k.startswith(f"experts.{eid}.encoder.fc_layers."):
                    enc.explicit_masks[int(k.split(".")[4])] = m.to(device)
            meta = {cond: [f"{cond}_{int(i)}" for i in idx] for cond, idx in labels.items()}
            if case.get("cond"):
                import random

                raw, _ = H.cond_inputs(case, z, t, eid)
                meta.update(raw)
                random.seed(case["seed"] * 100 + t)  # the shuffle of "parallel" selection (components.py:601-603)
            metadata = pd.DataFrame(meta if meta else {"dummy": [0] * x.shape[0]})
            model.logged.clear()
            model.training_step((x.to(device), metadata, eid), t)
            if device != "cpu":
                torch.cuda.synchronize()
            logged = {k: (float(v.detach()) if torch.is_tensor(v) else v) for k, v in model.logged.items()}
            sd = {k: v.detach().cpu().clone() for k, v in model.module.state_dict().items()}
            results.append({"logged": logged, "sd": sd, "eid": eid})
        replay_training.last_engine = model._engine
    return case, z, results


def check_against_golden(case, z, results, rtol_loss=2e-5, tol_param=1e-4):
    from oracle import mmvae_oracle as O  # checker only

    spec = H.spec_from_case(case)
    skip = H.bn_fed_biases(spec)
    lr, mom = 5e-3, 0.01
    for t, r in enumerate(results):
        eid = r["eid"]
        g = lambda k: float(np.array(z[f"step{t}/out/{k}"]))
        L = r["logged"]

        def close(a, b, what, rtol=rtol_loss, atol=1e-5):
            assert abs(a - b) <= rtol * abs(b) + atol, f"step{t} {what}: got {a}, reference {b}"

        close(L[f"loss/training/{eid}"], g("total_loss"), "loss")
        close(L[f"recon_loss/training/{eid}"], g("recon_loss"), "recon_loss")
        close(L[f"kl_loss/training/{eid}"], g("kl_loss"), "kl_loss")
        close(L[f"Mean/training/{eid}"], g("Mean"), "Mean", atol=1e-6)
        close(L[f"Variance/training/{eid}"], g("Variance"), "Variance", atol=1e-6)
        close(L["grad_norms/vae"], g("grad_norms/vae"), "grad_norms/vae", rtol=5e-5)
        close(L[f"grad_norms/expert_{eid}"], g(f"grad_norms/expert_{eid}"), "grad_norms/expert", rtol=5e-5)
        for i in range(len(case.get("adversarials", []) or [])):
            for phase in ("discriminator", "generator"):
                close(L[f"{phase}_{i + 1}/training/{eid}/adversarial_loss/summed"], g(f"{phase}_{i + 1}/summed"),
                      f"{phase}_{i + 1}/summed")
                for cond in case["conditions"]:
                    close(L[f"{phase}_{i + 1}/training/{eid}/adversarial_loss/{cond}"], g(f"{phase}_{i + 1}/{cond}"),
                          f"{phase}_{i + 1}/{cond}")
                close(L[f"grad_norms/{phase}_{i + 1}"], g(f"grad_norms/{phase}_{i + 1}"), f"grad_norms/{phase}",
                      rtol=5e-5)
        for n, v in r["sd"].items():
            ref = np.array(z[f"step{t}/sd/{n}"])
            if n in skip:
                assert np.abs(v.numpy() - ref).max() <= 2 * lr * (t + 1) + 1e-6, n
            elif v.dtype == torch.int64:
                assert int(v) == int(ref), n
            elif n.endswith("running_mean") and t > 0:
                assert np.abs(v.numpy() - ref).max() <= mom * lr * (t + 1) * (t + 2) + 1e-6, n
            else:
                assert H.rel_l2(v, ref) < tol_param, f"step{t} param {n}: rel-L2 {H.rel_l2(v, ref)}"
