"""Shared test helpers: golden-fixture loading and case -> oracle ModelSpec mapping."""
import json
import os

import numpy as np
import torch

from oracle import mmvae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c1_small", "two_mod_odd", "adversarial", "adv_dropout", "ln_dist", "clip_value"]
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
        advs.append(O.AdvSpec(O.FCSpec.make(enc_layers, relu=True, dropout_rate=case.get("adv_dropout", 0.0)),
                              dict(case["conditions"])))
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
        vae_encoder=O.FCSpec.make([eh[-1]] + vh, use_batch_norm=True, relu=True, return_hidden=True,
                                  dropout_rate=case.get("vae_dropout", 0.0)),
        vae_decoder=O.FCSpec.make(dec_layers, relu=True),
        latent_dim=Z, softmax_z=case.get("distribution") == "ln",
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
    hp = O.HParams(adv_weight=float(case.get("adv_weight") or 1.0))
    if case.get("clip_value"):
        v = float(case["clip_value"])
        hp.clip_algorithm, hp.vae_clip, hp.expert_clip, hp.adversarial_clip = "value", v, v, v
    return hp


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


# ------------------------------------------------------------------------------------------------ full-size cases
# Golden cases at the BASELINE sizes ("regen" cases: c2_full, c4_full, ...) cannot commit 40-190 M initial parameters or
# 41 MB batches.  Inputs and initial parameters are REGENERATED from the case's seeds by the functions below -- the very
# functions tests/golden/make_golden.py drew them with when it ran the reference -- and the fixture holds, per tensor,
# fp64 checksums plus a fixed sample of entries (small tensors in full).  sd0 checksums guard the regeneration itself.
REGEN_CASES = ["mid_odd", "c2_full", "c4_full", "c5_three_mod"]
SAMPLES = 256
FULL_BELOW = 2048


def sample_index(name: str, numel: int, n: int = SAMPLES) -> np.ndarray:
    """Fixed pseudo-random entry positions of the tensor called `name` (a 64-bit LCG seeded by the name's CRC)."""
    import zlib

    state = np.uint64(zlib.crc32(name.encode()) + 0x9E3779B97F4A7C15)
    out = np.empty(n, dtype=np.int64)
    with np.errstate(over="ignore"):
        for i in range(n):
            state = state * np.uint64(6364136223846793005) + np.uint64(1442695040888963407)
            out[i] = int(state >> np.uint64(33)) % numel
    return out


def compact(name: str, t, light: bool = False) -> dict:
    """What a regen fixture keeps of tensor `name`: everything when small, else checksums + sampled entries
    (`light`: the norm and 16 samples -- enough to guard a regenerated input state)."""
    a = np.ascontiguousarray(np.asarray(t))
    if a.size <= (64 if light else FULL_BELOW) or not np.issubdtype(a.dtype, np.floating):
        return {"full": a}
    flat = a.reshape(-1)
    d = flat.astype(np.float64)
    return {"sumsq": np.array((d * d).sum()), "samples": flat[sample_index(name, flat.size, 16 if light else SAMPLES)].copy()}


# Sampled-entry tolerance of gradients / parameters in cases with adversaries: the discriminators' own Adam step sits
# between the forward pass and the generator-phase gradients, and a cold first Adam step is sign-like (+-lr whatever the
# magnitude), so rounding-level differences in the discriminator gradients reach the generator gradients amplified by
# adv_weight = 25.  Measured oracle-vs-reference at C4 size: 1.0e-4 on 256 sampled entries while the norms agree to 1e-7.
ADVERSARIAL_SAMPLE_TOL = 5e-4


def compare_compact(name: str, got, z, prefix: str, tol: float, what: str = "", tol_samples: float = None):
    """`got` (tensor) against the compact record stored under `prefix` in fixture `z`: full tensors and samples by
    rel-L2 (`tol_samples`, default `tol`), the norm through sumsq (`tol`).  Returns the worst relative deviation seen."""
    tol_samples = tol if tol_samples is None else tol_samples
    g = torch.as_tensor(got).detach().cpu()
    if prefix in z.files:  # 0-dim entries (BatchNorm step counters) are stored as they are
        ref = np.array(z[prefix])
        assert float(g) == float(ref) or abs(float(g) - float(ref)) <= tol * abs(float(ref)), f"{what}{name}: {float(g)} vs {ref}"
        return 0.0
    if f"{prefix}/full" in z.files:
        ref = torch.from_numpy(np.array(z[f"{prefix}/full"]))
        if not ref.is_floating_point():
            assert torch.equal(g.to(ref.dtype).reshape(ref.shape), ref), f"{what}{name}: integer buffer differs"
            return 0.0
        e = rel_l2(g.reshape(ref.shape), ref)
        assert e < tol_samples, f"{what}{name}: rel-L2 {e:.3g} (full tensor)"
        return e
    flat = g.reshape(-1)
    ref_s = torch.from_numpy(np.array(z[f"{prefix}/samples"]))
    e1 = rel_l2(flat[torch.from_numpy(sample_index(name, flat.numel(), ref_s.numel()))], ref_s)
    n_got, n_ref = float(flat.double().pow(2).sum().sqrt()), float(np.sqrt(np.array(z[f"{prefix}/sumsq"])))
    e2 = abs(n_got - n_ref) / max(n_ref, 1e-30)
    assert e1 < tol_samples and e2 < tol, f"{what}{name}: sampled rel-L2 {e1:.3g}, norm deviation {e2:.3g}"
    return max(e1, e2)


def regen_state(case, t: int, module, eid: str, init_fn=None):
    """State a regen case's step t starts from, written into `module` (a CPU module: the same CPU torch build draws
    the same numbers; the fixture's checksums verify).  Steps are INDEPENDENT single steps: at these sizes and this
    learning rate the reference's own trajectory amplifies a 1e-6 perturbation tenfold per step (measured with the
    oracle: 4e-6 after step 0, 2e-2 after step 3), so a sequential schedule could only be compared loosely.
      t = 0: cold start -- He init under the case seed, then +0.1 N(0,1) on every bias and BatchNorm weight
             (parameters() order), optimiser state empty;
      t > 0: warm start -- the same construction under seed + 100 t, BatchNorm running statistics perturbed, and Adam
             moments of the optimisers that step (shared VAE, expert `eid`, adversaries) drawn at a realistic scale with
             step count 5 + t: the update then exercises bias corrections, moment mixing and weight decay.
    Returns (step_count, {parameter name: (exp_avg, exp_avg_sq)}) -- (0, {}) for the cold start."""
    if init_fn is None:
        from mmvae_amd.modules.base.init import he_init_weights as init_fn

    seed = case["seed"] + 100 * t
    torch.manual_seed(seed)
    init_fn(module)  # every Linear: weight ~ He, bias = 0
    g = torch.Generator().manual_seed(seed + 1)
    moments = {}
    with torch.no_grad():
        for name, p in module.named_parameters():  # everything that is not a Linear weight is defined here
            if name.endswith("bn.weight"):
                p.fill_(1.0)
            elif name.endswith("bn.bias"):
                p.zero_()
            if name.endswith("bias") or name.endswith("bn.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for name, b in module.named_buffers():
            if name.endswith("running_mean") or name.endswith("num_batches_tracked"):
                b.zero_()
            elif name.endswith("running_var"):
                b.fill_(1.0)
        if t == 0:
            return 0, moments
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
            elif name.endswith("num_batches_tracked"):
                b.fill_(5 + t)
        for name, p in module.named_parameters():
            if name.startswith(("vae.", f"experts.{eid}.", "adversarials.")):
                moments[name] = (1e-3 * torch.randn(p.shape, generator=g),
                                 1e-6 * (0.5 + torch.rand(p.shape, generator=g)))
    return 5 + t, moments


class RegenStream:
    """Per-step inputs of a regen case drawn from the case's generator in the generator script's order: gene rates,
    counts, x, eps, adversarial class indices, dropout keep masks of the active expert's encoder."""

    def __init__(self, case):
        self.case = case
        self.g = torch.Generator().manual_seed(case["seed"] + 2)

    def step(self, t: int, eid: str):
        c = self.case
        G, B = c["experts"][eid], c["B"]
        lam = 0.15 * torch.exp(torch.randn(G, generator=self.g))
        counts = torch.poisson(lam.expand(B, G) * c.get("lam_scale", 8.0), generator=self.g)
        x = torch.log1p(1e4 * counts / counts.sum(1, keepdim=True).clamp_min(1.0))
        eps = torch.randn(B, c["Z"], generator=self.g)
        labels = {}
        if c.get("adversarials"):
            for cond, n in c["conditions"].items():
                labels[cond] = torch.randint(0, n, (B,), generator=self.g)
        masks = {}
        for i, width in enumerate(c["expert_hidden"]):
            if c["dropout"] > 0:
                masks[f"experts.{eid}.encoder.fc_layers.{i}.dr"] = (
                    torch.rand(B, width, generator=self.g) >= c["dropout"]).to(torch.uint8)
        return x, eps, masks, labels



def run_in_child(module: str, func: str, env: dict, timeout: int = 900) -> None:
    """Run `module.func()` in a child process (fresh interpreter, `env` on top of this process's environment) and hold
    it to the marker line it prints when all its checks have passed.  For tests that bring up a real RCCL communicator
    inside the test session: the runtime's communicator teardown has aborted the process once in a while on this pool
    (`Fatal Python error: Aborted` inside destroy_process_group, behind green checks) -- in a child that cannot take the
    session down; a teardown crash behind the marker is reported as a warning."""
    import os
    import subprocess
    import sys
    import warnings

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (f"import sys; sys.path.insert(0, {root!r}); import importlib; "
            f"m = importlib.import_module({module!r}); getattr(m, {func!r})()")
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), cwd=root, capture_output=True, text=True,
                       timeout=timeout)
    if "CHILD_CASE_OK" not in p.stdout:
        raise AssertionError(f"{module}.{func} failed in its child process (exit code {p.returncode})\n"
                             f"--- stdout\n{p.stdout[-3000:]}\n--- stderr\n{p.stderr[-6000:]}")
    if p.returncode != 0:
        warnings.warn(f"{module}.{func}: checks passed, the child then exited with code {p.returncode} "
                      f"(communicator teardown): {p.stderr[-400:]}")
