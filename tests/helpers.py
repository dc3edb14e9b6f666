"""Shared test helpers: golden-fixture loading and case -> oracle ModelSpec mapping."""
import json
import os

import numpy as np
import torch

from oracle import mmvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c1_small", "two_mod_odd", "adversarial"]
# conditional layers after the reparameterisation; cond_adv: + two adversaries
COND_CASES = ["cond_seq", "cond_par", "cond_adv"]


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
    conds = None
    dec_layers = [Z] + vh[::-1] + [eh[-1]]
    if case.get("cond"):
        c = case["cond"]
        species = list(case["experts"].keys())
        conds = O.CondSpec(
            fc=O.FCSpec.make([Z], use_layer_norm=c["layer_norm"]),
            keys=list(c["keys"]),
            shared={k: [cond_name(k, None, i) for i in range(n)] for k, n in c["shared"].items()},
            species_specific={k: {sp: [cond_name(k, sp, i) for i in range(n)] for sp, n in by.items()}
                              for k, by in c["species_specific"].items()},
            species_blocks=species, parallel=c["parallel"])
        if c["parallel"]:  # clvae.py:55-79: one more decoder layer, concat_dim -> Z, described by concat_config (ReLU)
            dec_layers = [len(c["keys"]) * Z] + dec_layers
    return O.ModelSpec(
        experts=experts,
        vae_encoder=O.FCSpec.make([eh[-1]] + vh, use_batch_norm=True, relu=True, return_hidden=True),
        vae_decoder=O.FCSpec.make(dec_layers, relu=True),
        latent_dim=Z,
        hidden_z=case["hidden_z"],
        adversarials=advs,
        conditionals=conds,
    )


def cond_value(key, species, i):
    """Raw metadata value of condition i (as tests/golden/make_golden.py writes them: dots on purpose)."""
    return f"{key}.{species}.{i}" if species else f"{key}.{i}"


def cond_name(key, species, i):
    """ModuleDict key of that condition (ConditionalLayer.format_condition_key, components.py:353-363)."""
    return cond_value(key, species, i).replace(".", "_")


def cond_inputs(case, z, t, eid):
    """Per-row condition indices of step t -> (raw metadata columns, ModuleDict names per key)."""
    raw, names = {}, {}
    c = case["cond"]
    for key in list(c["shared"]) + list(c["species_specific"]):
        idx = [int(i) for i in np.array(z[f"step{t}/in/cond/{key}"])]
        sp = None if key in c["shared"] else eid
        raw[key] = [cond_value(key, sp, i) for i in idx]
        names[key] = [cond_name(key, sp, i) for i in idx]
    return raw, names


def cond_order(case, t):
    """Selection order of step t: the configured order, or -- "parallel" -- the shuffle Python's random produces from
    the seed the generator used (components.py:601-603: random.sample(selection_order, len))."""
    import random

    c = case["cond"]
    if not c["parallel"]:
        return list(c["keys"])
    # ConditionalLayers.__init__ removes "species" from the list and appends it again (components.py:509-511), so the
    # list that is shuffled has it last
    base = [k for k in c["keys"] if k != "species"] + ["species"]
    return random.Random(case["seed"] * 100 + t).sample(base, len(base))


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
