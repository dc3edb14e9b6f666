"""TEST INFRASTRUCTURE (checker only; never imported by the product package): one step of the program the engine
actually runs -- the captured, replayed hipGraph with device Philox noise and its side branches -- against
oracle.train_step (the CPU restatement of cmmvae_model.py:138-217, pinned to the reference's golden vectors).

The full-size parity tests feed explicit noise through freshly built plans; what `bench.py` times is a REPLAYED plan
keyed by the resident input pointer whose dropout masks and rsample noise are drawn by the library's Philox streams at
the head of the program.  Here the step is taken exactly that way, and afterwards
  * the pre-step state (parameters, BatchNorm buffers, Adam moments and step counts, KL weight) -- snapshotted before,
  * the noise the Philox fill left in the plan's buffers (keep masks, eps) -- read after,
  * the ReLU slopes the step took (tests/mirror_utils.engine_relu_slopes' rule: read off its activations)
go into oracle.train_step on the host, and losses, gradient norms, gradients and post-step parameters are compared.

Used by tests/test_bench_program_gpu.py and by bench.py's `parity` block (outside the timed region)."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from oracle import mmvae_oracle as O

TOL = {"loss": 1e-4, "grad_norm": 5e-5, "grad": 1e-4, "param": 1e-4, "param_cold": 1e-3}


def _fc_spec(block) -> O.FCSpec:
    c = block.config
    relu = [getattr(a, "__name__", str(a)).split(".")[-1] == "ReLU" if a is not None else False for a in c.activation_fn]
    return O.FCSpec.make(list(c.layers), dropout_rate=list(c.dropout_rate), use_batch_norm=list(c.use_batch_norm),
                         use_layer_norm=list(c.use_layer_norm), relu=relu, return_hidden=list(c.return_hidden))


def spec_of(model, eid: str) -> O.ModelSpec:
    """ModelSpec of the active expert + the shared VAE (+ the adversaries: BASELINE config 4) of a CMMVAEModel."""
    from mmvae_amd.modules.base.components import Adversarial

    m = model.module
    if getattr(m.vae, "conditionals", None) is not None:
        raise ValueError("program_check covers the expert + VAE (+ adversaries) step; conditional layers draw their "
                         "selection order on the host: see tests/test_step_gpu.py for those programs")
    advs = []
    for adv in m.adversarials:
        enc = _fc_spec(adv.encoder)
        if any(p > 0 for p in enc.dropout_rate):
            raise ValueError("program_check: adversary dropout draws one keep mask per phase; the oracle takes one per layer")
        advs.append(O.AdvSpec(encoder=enc, heads={c: adv.heads[c].fc_layers[0].lin.out_features
                                                  for c in Adversarial.labels.keys()}))
    exp = m.experts[eid]
    return O.ModelSpec(experts={eid: (_fc_spec(exp.encoder), _fc_spec(exp.decoder))}, vae_encoder=_fc_spec(m.vae.encoder.fc),
                       vae_decoder=_fc_spec(m.vae.decoder), latent_dim=m.vae.encoder.mean_encoder.out_features,
                       var_eps=float(m.vae.encoder.var_eps), hidden_z=bool(m.vae.encoder.hidden_z), adversarials=advs)


def _rel_l2(a, b) -> float:
    a, b = a.detach().double().flatten().cpu(), b.detach().double().flatten().cpu()
    n = b.norm()
    return float((a - b).norm() / n) if n > 0 else float((a - b).norm())


def _names(model):
    return {id(p): n for n, p in model.module.named_parameters()}


def snapshot(model, eid: str):
    """(state_dict of the VAE + expert `eid` on the host, oracle optimiser state, common Adam step count)."""
    model._flush_engine()
    torch.cuda.synchronize()
    keep = ("vae.", f"experts.{eid}.", "adversarials.")
    sd = {k: v.detach().cpu().clone() for k, v in model.module.state_dict().items() if k.startswith(keep)}
    names = _names(model)
    opts = model.get_optimizers()
    state, counts = {}, []
    groups = [("vae", opts["vae"]), (f"expert_{eid}", opts["experts"][eid])]
    groups += [(f"adversarial_{i + 1}", o) for i, o in enumerate(opts.get("adversarials", {}).values())]
    for group, opt in groups:
        a = opt.arena
        count = int(round(float(opt.state_dev[0])))
        counts.append(count)
        st = {"steps": {}, "exp_avg": {}, "exp_avg_sq": {}}
        for i, p in enumerate(a.params):
            n = names[id(p)]
            st["steps"][n] = count
            st["exp_avg"][n] = a.view(a.exp_avg, i).detach().cpu().clone()
            st["exp_avg_sq"][n] = a.view(a.exp_avg_sq, i).detach().cpu().clone()
        state[group] = st
    return sd, state, min(counts)


def _slopes(model, eid: str) -> Dict[str, torch.Tensor]:
    """The 0/1 slope every ReLU of the last engine step took, read from the activations it left in its buffers."""
    plan = model._engine.last_plan
    n_vae_dec = len(model.module.vae.decoder.fc_layers)
    slopes = {}
    for i, l in enumerate(plan.enc_layers):
        name = (f"experts.{eid}.encoder.fc_layers.{i}" if i < plan.n_expert_enc
                else f"vae.encoder.fc.fc_layers.{i - plan.n_expert_enc}")
        if l.relu:
            act = l.a if l.a is not None else l.d
            slopes[name] = (act[: l.rows] > 0).cpu()
    for j, l in enumerate(plan.dec_layers):
        name = f"vae.decoder.fc_layers.{j}" if j < n_vae_dec else f"experts.{eid}.decoder.fc_layers.{j - n_vae_dec}"
        if not l.relu:
            continue
        if l is plan.dec_layers[-1]:  # fused with the reconstruction epilogue: dP = 2 (xhat - x) 1[P > 0]
            slopes[name] = (plan.dP[: plan.R] != 0).cpu()
        else:
            act = l.a if l.a is not None else l.d
            slopes[name] = (act[: l.rows] > 0).cpu()
    # the adversaries' encoders, generator phase (fused passes: its activations are what the step left behind; the oracle's
    # discriminator phase keeps its own slopes)
    prog = getattr(plan, "adv_prog", None)
    if prog is not None:
        for i, (net, bufs) in enumerate(zip(prog.nets, prog.bufs)):
            for j, lay in enumerate(net.layers):
                if lay.relu and lay.p_drop == 0:
                    slopes[f"adversarials.{i}.encoder.fc_layers.{j}"] = (bufs["act"][j][: plan.B] > 0).cpu()
    return slopes


def check_step(model, eid: str, x: torch.Tensor, meta, step_index: int, strict: bool = True, next_batch=None) -> dict:
    """Take ONE training step of `model` on (x, meta, eid) the way the caller's loop does (captured program, Philox
    noise) and compare it with the oracle.  Returns the measured deviations; `strict` asserts the tolerances TOL.
    next_batch: the (x, meta, eid) the caller's loop will step next -- passed on as the loop does
    (CMMVAEModel.hint_next_batch: the pipelined program computes that step's first product ahead, and THIS step may be
    consuming the product the previous step computed for it; the oracle works from the snapshotted pre-step weights, so
    a stale product would show)."""
    from mmvae_amd.modules.base.components import Adversarial

    spec = spec_of(model, eid)
    hp = O.HParams(adv_weight=float(model.adv_weight))
    labels = None
    if spec.adversarials:  # cmmvae_model.py:111-115
        labels = {c: torch.tensor([table[v] for v in meta[c].values]) for c, table in Adversarial.labels.items()}
    sd_in, opt_state, count = snapshot(model, eid)
    kl_weight = float(model.kl_annealing_fn.kl_weight)
    model.logged.clear()
    if next_batch is not None:
        model.hint_next_batch(next_batch)
    model.training_step((x, meta, eid), step_index)
    model._flush_engine()
    torch.cuda.synchronize()
    plan = model._engine.last_plan
    replayed = plan._graphs is not None
    # the noise this step drew (Philox fill at the head of the program), from the plan's buffers
    masks = {}
    for i, l in enumerate(plan.enc_layers):
        if l.mask is not None:
            prefix = (f"experts.{eid}.encoder.fc_layers.{i}" if i < plan.n_expert_enc
                      else f"vae.encoder.fc.fc_layers.{i - plan.n_expert_enc}")
            masks[prefix + ".dr"] = l.mask[: l.rows].detach().cpu().bool()
    eps = plan.eps.detach().cpu().clone()
    eps = eps[0] if eps.shape[0] == 1 else eps
    slopes = _slopes(model, eid)
    ref, sd_new = O.train_step(spec, sd_in, opt_state, x.detach().cpu(), eid, eps, masks, labels, kl_weight, hp,
                               relu_slopes=slopes)
    # slopes that differ from the oracle's own 1[y > 0] must be kinks: |y| within rounding distance of zero
    kinks = 0
    last = f"experts.{eid}.decoder.fc_layers.{len(model.module.experts[eid].decoder.fc_layers) - 1}"
    assert last in slopes
    for name, slope in slopes.items():
        y = ref["relu_inputs"][name]
        diff = (y > 0) != slope.reshape(y.shape)
        keep = masks.get(name + ".dr")
        if keep is not None:  # a dropped unit's slope never reaches a gradient
            diff &= keep.reshape(y.shape)
        if name == last:  # dP is also zero where xhat == x exactly (the step's xhat: within rounding of the oracle's)
            xr = x.detach().cpu()
            diff &= ~((y > 0) & ~slope.reshape(y.shape) & torch.isclose(torch.relu(y), xr, rtol=1e-5, atol=1e-6))
        n = int(diff.sum())
        if n:
            rms = float(y.double().pow(2).mean().sqrt())
            worst = float(y[diff].abs().max())
            # An adversary's encoder in the generator phase: the discriminator phase has just stepped the adversary's
            # weights, and Adam's step is sign-like (+-lr whatever the size of the gradient) wherever the second moment
            # is ~0 -- every weight on a cold step, the rows of (nearly) dead units on any step.  A gradient entry within
            # rounding of zero there (a unit on the edge for one cell; the oracle's discriminator phase keeps its own
            # slopes) takes either value on the two sides, and the generator-phase pre-activations of that unit then
            # differ by O(lr |h|_1) ~ 6e-3, not by rounding.  Units that close to zero on that scale count as kinks
            # there; the caller sees them in `kinks`.
            bound = 1e-2 if name.startswith("adversarials.") else 1e-4
            assert worst <= bound * rms, f"{name}: a ReLU slope differs at |y| = {worst:.3e} (rms {rms:.3e}): not a kink"
            kinks += n
    out = {"kinks": kinks, "cold": count == 0, "replayed": bool(replayed), "adam_step": count + 1,
           "first_product_from_previous_step": getattr(plan, "slabs_ahead", None) is not None,
           "computes_next_first_product": getattr(plan, "prefetch_slabs", None) is not None,
           "forked": bool(getattr(plan, "_forked", False)), "philox": not plan.explicit}
    logged = {k: float(v.detach() if torch.is_tensor(v) else v) for k, v in model.logged.items()}
    for k, key in (("loss", "total_loss"), ("recon_loss", "recon_loss"), ("kl_loss", "kl_loss")):
        r = float(ref[key])
        out[k] = abs(logged[f"{k}/training/{eid}"] - r) / max(abs(r), 1e-30)
    for i in range(len(spec.adversarials)):  # both phases: every head's loss, their sum, the (pre-clip) gradient norm
        worst = 0.0
        for phase in ("discriminator", "generator"):
            got = {c: logged[f"{phase}_{i + 1}/training/{eid}/adversarial_loss/{c}"] for c in list(labels) + ["summed"]}
            want = dict({c: float(v) for c, v in ref[phase][i]["heads"].items()}, summed=float(ref[phase][i]["summed"]))
            worst = max([worst] + [abs(got[c] - want[c]) / max(abs(want[c]), 1e-30) for c in want])
            gn = float(ref["grad_norms"][f"{phase}_{i + 1}"])
            out[f"grad_norm_{phase}_{i + 1}"] = abs(logged[f"grad_norms/{phase}_{i + 1}"] - gn) / gn
        out[f"adversarial_loss_{i + 1}"] = worst
    out["grad_norm_vae"] = abs(logged["grad_norms/vae"] - float(ref["grad_norms"]["vae"])) / float(ref["grad_norms"]["vae"])
    ge = float(ref["grad_norms"][f"expert_{eid}"])
    out["grad_norm_expert"] = abs(logged[f"grad_norms/expert_{eid}"] - ge) / ge
    skip = set()  # Linear biases that feed a BatchNorm: exactly-zero true gradient, rounding noise on both sides
    for prefix, fc in ((f"experts.{eid}.encoder", spec.experts[eid][0]), (f"experts.{eid}.decoder", spec.experts[eid][1]),
                       ("vae.encoder.fc", spec.vae_encoder), ("vae.decoder", spec.vae_decoder)):
        for i in range(fc.n_layers):
            if fc.use_batch_norm[i]:
                skip.add(f"{prefix}.fc_layers.{i}.lin.bias")
    names = _names(model)
    opts = model.get_optimizers()
    wg = wp = 0.0
    for opt in (opts["vae"], opts["experts"][eid]):
        for i, p in enumerate(opt.arena.params):
            n = names[id(p)]
            if n in skip or n not in ref["grads"]:
                continue
            e = _rel_l2(opt.arena.grad_view(i), ref["grads"][n])
            if e > wg:
                wg, out["worst_grad"] = e, n
    sd_got = {k: v.detach().cpu() for k, v in model.module.state_dict().items() if k in sd_new}
    for n, v in sd_got.items():
        if n in skip or not v.is_floating_point() or n.endswith("running_mean"):
            continue
        e = _rel_l2(v, sd_new[n])
        if e > wp:
            wp, out["worst_param"] = e, n
    out["grad"], out["param"] = wg, wp
    if strict:
        for k in ["loss", "recon_loss", "kl_loss"] + [k for k in out if k.startswith("adversarial_loss_")]:
            assert out[k] <= TOL["loss"], (k, out)
        for k in out:
            if k.startswith(("grad_norm_discriminator", "grad_norm_generator")):
                assert out[k] <= TOL["grad_norm"], (k, out)
        assert out["grad_norm_vae"] <= TOL["grad_norm"] and out["grad_norm_expert"] <= TOL["grad_norm"], out
        assert out["grad"] <= TOL["grad"], out
        assert out["param"] <= (TOL["param_cold"] if out["cold"] else TOL["param"]), out
    return out
