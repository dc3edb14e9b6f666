"""Shared test helpers: golden-fixture loading and case -> oracle ModelSpec mapping."""
import json
import os

import numpy as np
import torch

from oracle import mmvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c1_small", "two_mod_odd", "adversarial"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    case = json.loads(str(z["case_json"]))
    return case, z


def spec_from_case(case) -> O.ModelSpec:
    eh, vh, Z = case["expert_hidden"], case["vae_hidden"], case["Z"]
    experts = {}
    for eid, G in case["experts"].items():
        enc = O.FCSpec.make([G] + eh, dropout_rate=case["dropout"], use_batch_norm=True, relu=True)
        dec = O.FCSpec.make(eh[::-1] + [G], relu=True)
        experts[eid] = (enc, dec)
    advs = []
    for enc_layers in case.get("adversarials", []) or []:
        advs.append(O.AdvSpec(O.FCSpec.make(enc_layers, relu=True), dict(case["conditions"])))
    return O.ModelSpec(
        experts=experts,
        vae_encoder=O.FCSpec.make([eh[-1]] + vh, use_batch_norm=True, relu=True, return_hidden=True),
        vae_decoder=O.FCSpec.make([Z] + vh[::-1] + [eh[-1]], relu=True),
        latent_dim=Z,
        hidden_z=case["hidden_z"],
        adversarials=advs,
    )


def hparams_from_case(case) -> O.HParams:
    return O.HParams(adv_weight=float(case.get("adv_weight") or 1.0))


def sd_from(z, prefix):
    out = {}
    for k in z.files:
        if k.startswith(prefix):
            out[k[len(prefix):]] = torch.from_numpy(np.array(z[k]))
    return out


def step_inputs(z, t):
    x = torch.from_numpy(z[f"step{t}/in/x"])
    eps = torch.from_numpy(z[f"step{t}/in/eps"])
    masks = {k[len(f"step{t}/in/mask/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"step{t}/in/mask/")}
    labels = {k[len(f"step{t}/in/labels/"):]: torch.from_numpy(z[k]) for k in z.files
              if k.startswith(f"step{t}/in/labels/")}
    return x, eps, masks, labels


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().flatten().cpu()
    b = torch.as_tensor(b).detach().double().flatten().cpu()
    d = (a - b).norm()
    n = b.norm()
    return float(d / n) if n > 0 else float(d)


def bn_fed_biases(spec: O.ModelSpec):
    """Linear biases directly followed by BatchNorm: their true gradient is exactly zero, so the computed one is
    rounding noise and Adam (scale-free) turns it into +-lr-sized chaotic updates.  The reference's own values for
    these are noise-determined; they are excluded from post-step parameter parity."""
    names = set()

    def add(prefix, fc):
        for i in range(fc.n_layers):
            if fc.use_batch_norm[i]:
                names.add(f"{prefix}.fc_layers.{i}.lin.bias")

    for eid, (enc, dec) in spec.experts.items():
        add(f"experts.{eid}.encoder", enc)
        add(f"experts.{eid}.decoder", dec)
    add("vae.encoder.fc", spec.vae_encoder)
    add("vae.decoder", spec.vae_decoder)
    return names
