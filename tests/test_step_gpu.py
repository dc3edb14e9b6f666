"""Whole-step parity on the GPU: CMMVAEModel.training_step (module path and graph-captured engine) through the HIP
library vs the golden vectors produced by the reference's own modules.  Tolerances (fp32, stated): scalar losses
rtol 2e-5; gradient norms rtol 5e-5; post-Adam parameters rel-L2 <= 1e-4; integer buffers exact."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from tests import helpers as H  # noqa: E402
from tests import mirror_utils as MU  # noqa: E402


@pytest.mark.parametrize("name", H.CASES + H.COND_CASES)
def test_module_path_matches_reference(name):
    case, z, results = MU.replay_training(name, "cuda", use_engine=False)
    MU.check_against_golden(case, z, results)


@pytest.mark.parametrize("name", H.CASES + H.COND_CASES)
def test_engine_path_matches_reference(name):
    case, z, results = MU.replay_training(name, "cuda", use_engine=True)
    MU.check_against_golden(case, z, results)
    engine = MU.replay_training.last_engine
    if case.get("distribution") == "ln":
        # Encoder(distribution="ln") (softmax over the latent sample, components.py:740-741): the captured engine
        # declines it; the step ran -- and matched -- on the module path (HIP kernels sequenced by autograd)
        assert not engine
        return
    assert engine, "the captured engine must have taken this configuration"
    if name in H.COND_CASES:  # conditional layers ran inside the captured program, not on the module path
        assert all(p.cond is not None for p in engine._plans.values())


@pytest.mark.parametrize("name", H.CASES)
def test_engine_overlapped_exchange_program_matches_reference(name, monkeypatch):
    """The data-parallel program of the engine (early shared-VAE exchange on a second stream, the expert's exchange and
    clip + Adam deferred onto the communication stream, next use of the expert gated by an event), forced on one rank."""
    monkeypatch.setenv("MMVAE_DP_OVERLAP", "1")
    case, z, results = MU.replay_training(name, "cuda", use_engine=True)
    MU.check_against_golden(case, z, results)


def test_engine_overlapped_exchange_with_single_rank_rccl():
    """Same program with a real process group (RCCL, one rank): every collective is issued, on both communicators,
    beside the graph replays.  In a child process (tests.helpers.run_in_child)."""
    from tests.helpers import run_in_child

    run_in_child("tests.test_step_gpu", "_body_single_rank_rccl",
                 {"MMVAE_SINGLE_RANK_COLLECTIVES": "1", "MASTER_PORT": "29611", "MASTER_ADDR": "127.0.0.1"})


def _body_single_rank_rccl():
    import torch.distributed as td

    from mmvae_amd import dist as mdist

    mdist.init_from_env()
    try:
        assert mdist.collectives_active()

        def prepare(model):
            model.optimizers()
            mdist.broadcast_parameters(model)
            red = mdist.attach(model)
            assert red.small_group is not red.group

        for name in ("two_mod_odd", "adversarial"):
            case, z, results = MU.replay_training(name, "cuda", use_engine=True, prepare=prepare)
            MU.check_against_golden(case, z, results)
        print("CHILD_CASE_OK", flush=True)
    finally:
        torch.cuda.synchronize()
        td.destroy_process_group()


@pytest.mark.parametrize("name", ["two_mod_odd", "adversarial"])
def test_engine_follows_settings_changed_after_capture(name):
    """Learning rate, clip value and adversarial weight edited between steps (param_groups, autograd_config, adv_weight):
    the captured programs freeze them at build time, so the engine compares a settings signature on every step and
    rebuilds its plans -- engine and module path must stay together bit for bit in the logged losses."""
    def edit(model, t):
        if t == 1:
            for o in model.optimizers():
                o.param_groups[0]["lr"] = 1e-3
            model.autograd_config.expert_gradient_clip.val = 0.5
        if t == 2:
            model.adv_weight = 3.0
            model.autograd_config.vae_gradient_clip.val = 2.0

    _, _, ra = MU.replay_training(name, "cuda", use_engine=True, before_step=edit)
    _, _, rb = MU.replay_training(name, "cuda", use_engine=False, before_step=edit)
    for a, b in zip(ra, rb):
        for k, v in a["sd"].items():
            if v.is_floating_point() and not k.endswith("lin.bias"):
                # batch means of pre-activations sit next to zero: their relative distance is the least stable word
                tol = 5e-5 if k.endswith("running_mean") else 1e-5
                assert H.rel_l2(v, b["sd"][k]) < tol, k
        la, lb = a["logged"][f"loss/training/{a['eid']}"], b["logged"][f"loss/training/{b['eid']}"]
        assert abs(la - lb) <= 2e-5 * abs(lb)


def _trained_mirror(name, tmpdir, use_engine):
    """The mirror holding the reference's own post-training state of a golden case, in eval mode."""
    case, z = H.load_case(name)
    model = MU.build_mirror(case, "cuda", tmpdir, use_engine=use_engine)
    T = len(case["schedule"]) - 1
    MU.load_state(model, z, f"step{T}/sd/")
    model.eval()
    model.trainer.set_stage("validation")
    x, eps, _, labels = H.step_inputs(z, T)
    meta = {cond: [f"{cond}_{int(i)}" for i in idx] for cond, idx in labels.items()}
    import pandas as pd

    if case.get("cond"):
        meta.update(H.cond_inputs(case, z, T, str(z["eval/expert_id"]))[0])
    metadata = pd.DataFrame(meta if meta else {"dummy": [0] * x.shape[0]})
    model.module.vae.encoder.explicit_eps = eps.cuda()
    return case, z, model, x.cuda(), metadata, str(z["eval/expert_id"])


@pytest.mark.parametrize("use_engine", [False, True])
@pytest.mark.parametrize("name", H.CASES + H.COND_CASES)
def test_eval_and_predict_paths_match_reference(name, use_engine, tmp_path):
    """validation_step (eval-mode forward + ELBO, cmmvae_model.py:219-248), predict_step /
    get_latent_embeddings (cmmvae.py:115-142) and cross-generation (cmmvae.py:95-107) on the HIP path against the
    reference's own outputs.  Tolerances: losses rtol 1e-4 (eval mode right after 2-3 steps runs on barely-warmed
    running statistics: activations are huge), tensors rel-L2 <= 2e-5."""
    import numpy as np

    import random

    case, z, model, x, metadata, eid = _trained_mirror(name, str(tmp_path), use_engine)
    reseed = (lambda: random.seed(case["seed"] * 100 + 99)) if case.get("cond") else (lambda: None)
    with torch.no_grad():
        reseed()  # the shuffled order of the conditional layers (components.py:601-603), as the generator drew it
        ld = model.validation_step((x, metadata, eid))
        torch.cuda.synchronize()
        for k in ("loss", "recon_loss", "kl_loss"):
            ref = float(np.array(z[f"eval/out/{k}"]))
            assert abs(float(ld[k]) - ref) <= 1e-4 * abs(ref) + 1e-5, (k, float(ld[k]), ref)
        assert f"loss/validation/{eid}" in model.logged
        emb = model.predict_step((x, metadata, eid))
        assert H.rel_l2(emb["z"][0], z["eval/out/embedding_z"]) < 2e-5
        assert (emb["z"][1]["species"] == eid).all()
        if use_engine and case.get("cond"):  # the validation program ran the conditional layers inside the engine
            assert all(p.cond is not None for k, p in model._engine._plans.items() if k[0] == "validate")
        reseed()
        qz, pz, zz, xhats, hidden = model.module(x, metadata, eid, cross_generate=True)
        assert set(xhats) == set(case["experts"])
        for other, xh in xhats.items():
            assert H.rel_l2(xh, z[f"eval/out/xhat_cross/{other}"]) < 2e-5
        assert H.rel_l2(zz, z["eval/out/z"]) < 2e-5


def test_predictions_file_holds_the_reference_embeddings(tmp_path):
    """SURVEY 8(f4), writer side, end to end: Trainer.predict -> engine predict_step -> predictions.h5; the stored
    embeddings are the reference's (rel-L2 <= 2e-5) and the metadata rows carry the species written by predict_step."""
    from mmvae_amd import predictions as P
    from mmvae_amd.trainer import Trainer

    case, z, model, x, metadata, eid = _trained_mirror("two_mod_odd", str(tmp_path), True)
    writer = P.PredictionWriter(str(tmp_path), "exp", "run")
    Trainer().predict(model, [(x, metadata.copy(), eid), (x, metadata.copy(), eid)], writer=writer)
    data, meta, _ = P.load_from_hdf5(writer.hdf5_filepath, "z")
    B = x.shape[0]
    assert data.shape[0] == 2 * B and len(meta) == 2 * B
    assert H.rel_l2(torch.from_numpy(data[:B]), z["eval/out/embedding_z"]) < 2e-5
    assert (data[:B] == data[B:]).all()
    assert set(meta["species"]) == {eid.encode()}


@pytest.mark.parametrize("use_engine", [False, True])
def test_csr_batches_train_like_dense_ones(use_engine, monkeypatch):
    """a `torch.sparse_csr` batch (what the datapipes yield with return_dense: false) is densified by
    the HIP pass and must give the reference's results bit for bit the same way the dense batch does."""
    import tests.helpers as HH

    orig = HH.step_inputs

    def csr_inputs(z, t):
        x, eps, masks, labels = orig(z, t)
        return x.to_sparse_csr(), eps, masks, labels

    monkeypatch.setattr(HH, "step_inputs", csr_inputs)
    monkeypatch.setattr(MU.H, "step_inputs", csr_inputs)
    case, z, results = MU.replay_training("two_mod_odd", "cuda", use_engine=use_engine)
    MU.check_against_golden(case, z, results)


@pytest.mark.parametrize("name", ["two_mod_odd", "adversarial", "cond_adv"])
def test_slack_behind_engine_buffers_stays_zero(name):
    """The engine's GEMMs rely on it (StepEngine.buf): 16 elements + 32 rows of zeros behind every buffer, which no
    kernel ever writes -- a weight-gradient GEMM over B = 33 cells reads rows 33..63 of its operands as zeros."""
    MU.replay_training(name, "cuda", use_engine=True)
    engine = MU.replay_training.last_engine
    torch.cuda.synchronize()
    checked = 0
    for key, t in engine._pool.items():
        full = torch.empty(0, dtype=t.dtype, device=t.device).set_(t.untyped_storage())
        tail = full[t.storage_offset() + t.numel():]
        assert tail.numel() >= 16
        assert not bool(tail.ne(0).any()), f"slack behind engine buffer {key} was written"
        checked += 1
    assert checked > 20


@pytest.mark.parametrize("name", ["two_mod_odd", "adversarial", "cond_adv"])
def test_checkpoint_and_resume_continue_bit_for_bit(name, tmp_path):
    """Stop after the first steps, save module + optimiser state_dicts, rebuild everything from the files and continue:
    the resumed run must end on exactly the parameters of the uninterrupted one (engine path; optimiser moments, step
    counts -- per parameter for the conditional blocks --, BatchNorm buffers all travel through the state_dicts)."""
    import copy
    import random

    import pandas as pd

    case, z = H.load_case(name)
    T = len(case["schedule"])

    def run_steps(model, first, last):
        for t in range(first, last):
            eid = case["schedule"][t]
            x, eps, masks, labels = H.step_inputs(z, t)
            model.kl_annealing_fn.kl_weight = case["kl_weights"][t]
            model.module.vae.encoder.explicit_eps = eps.cuda()
            enc = model.module.experts[eid].encoder
            enc.explicit_masks = {int(k.split(".")[4]): m.cuda() for k, m in masks.items()
                                  if k.startswith(f"experts.{eid}.encoder.fc_layers.")}
            meta = {c: [f"{c}_{int(i)}" for i in idx] for c, idx in labels.items()}
            if case.get("cond"):
                meta.update(H.cond_inputs(case, z, t, eid)[0])
                random.seed(case["seed"] * 100 + t)
            model.training_step((x.cuda(), pd.DataFrame(meta if meta else {"dummy": [0] * x.shape[0]}), eid), t)

    def fresh(sub):
        model = MU.build_mirror(case, "cuda", str(tmp_path / sub), use_engine=True)
        MU.load_state(model, z, "sd0/")
        model.train()
        model.trainer.set_stage("training")
        model.optimizers()
        return model

    (tmp_path / "a").mkdir(), (tmp_path / "b").mkdir(), (tmp_path / "c").mkdir()
    straight = fresh("a")
    run_steps(straight, 0, T)
    want = {k: v.detach().clone() for k, v in straight.state_dict().items()}

    first = fresh("b")
    run_steps(first, 0, T - 1)
    torch.save({"model": first.state_dict(), "optim": [copy.deepcopy(o.state_dict()) for o in first.optimizers()]},
               tmp_path / "ckpt.pt")
    del first

    ckpt = torch.load(tmp_path / "ckpt.pt", weights_only=False)
    resumed = fresh("c")
    resumed.load_state_dict(ckpt["model"])
    for o, sd in zip(resumed.optimizers(), ckpt["optim"]):
        o.load_state_dict(sd)
    run_steps(resumed, T - 1, T)
    got = resumed.state_dict()
    assert got.keys() == want.keys()
    for k in want:
        assert torch.equal(got[k], want[k]), f"{k} differs after resume"


@pytest.mark.parametrize("name", ["two_mod_odd", "cond_adv"])
def test_reloading_a_snapshot_into_a_live_engine_repeats_the_step(name, tmp_path):
    """Rollback with the captured programs alive: snapshot (module + optimisers), take a step, load the snapshot back into
    the SAME model and take the step again -- bit for bit the same parameters (state_dicts copy in place into the arenas
    the graphs point at; optimiser step counts are re-read from the loaded state)."""
    import copy
    import random

    import pandas as pd

    case, z = H.load_case(name)
    T = len(case["schedule"])
    model = MU.build_mirror(case, "cuda", str(tmp_path), use_engine=True)
    MU.load_state(model, z, "sd0/")
    model.train()
    model.trainer.set_stage("training")
    model.optimizers()

    def step(t):
        eid = case["schedule"][t]
        x, eps, masks, labels = H.step_inputs(z, t)
        model.kl_annealing_fn.kl_weight = case["kl_weights"][t]
        model.module.vae.encoder.explicit_eps = eps.cuda()
        enc = model.module.experts[eid].encoder
        enc.explicit_masks = {int(k.split(".")[4]): m.cuda() for k, m in masks.items()
                              if k.startswith(f"experts.{eid}.encoder.fc_layers.")}
        meta = {c: [f"{c}_{int(i)}" for i in idx] for c, idx in labels.items()}
        if case.get("cond"):
            meta.update(H.cond_inputs(case, z, t, eid)[0])
            random.seed(case["seed"] * 100 + t)
        model.training_step((x.cuda(), pd.DataFrame(meta if meta else {"dummy": [0] * x.shape[0]}), eid), t)

    for t in range(T):  # every plan built and captured (the last expert of the schedule has run at least once before)
        step(t)
    snap = {"model": copy.deepcopy(model.state_dict()), "optim": [copy.deepcopy(o.state_dict()) for o in model.optimizers()]}
    step(T - 1)
    want = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(snap["model"])
    for o, sd in zip(model.optimizers(), snap["optim"]):
        o.load_state_dict(sd)
    step(T - 1)
    got = model.state_dict()
    for k in want:
        assert torch.equal(got[k], want[k]), f"{k} differs after the rollback"


def test_closing_the_engine_releases_its_programs_and_training_goes_on(tmp_path, monkeypatch):
    """StepEngine.close() destroys every captured graph (a plan holds runtime objects that only a cyclic collection
    would otherwise free); the next step rebuilds and re-captures its plan and continues bit for bit.  The measurement
    hook of bench.py's roofline leg rides along: in eager runs with a probe set, the tagged forward GEMM of the expert
    encoder is bracketed by a timing event pair."""
    import copy

    import pandas as pd

    case, z = H.load_case("two_mod_odd")
    T = len(case["schedule"])

    def run(close_at):
        model = MU.build_mirror(case, "cuda", str(tmp_path), use_engine=True)
        MU.load_state(model, z, "sd0/")
        model.train()
        model.trainer.set_stage("training")
        model.optimizers()
        probe = {}
        for rep in range(3):
            for t in range(T):
                eid = case["schedule"][t]
                x, eps, masks, labels = H.step_inputs(z, t)
                model.kl_annealing_fn.kl_weight = case["kl_weights"][t]
                model.module.vae.encoder.explicit_eps = eps.cuda()
                model.module.experts[eid].encoder.explicit_masks = {
                    int(k.split(".")[4]): m.cuda() for k, m in masks.items()
                    if k.startswith(f"experts.{eid}.encoder.fc_layers.")}
                if close_at is not None and (rep, t) == close_at:
                    eng = model._engine
                    graphs = sum(1 for p in eng._plans.values() for g in (p._graphs or []) if not isinstance(g, tuple))
                    assert graphs > 0, "plans should have been captured by now"
                    eng.close()
                    assert not eng._plans
                if close_at is not None and rep == 2:  # last round eagerly, with the probe set
                    model._engine.eager_only = True
                    for p in model._engine._plans.values():
                        p.probe = probe
                model.training_step((x.cuda(), pd.DataFrame({"dummy": [0] * x.shape[0]}), eid), t)
                if close_at is not None and rep == 2:
                    for p in model._engine._plans.values():
                        p.probe = probe
        return copy.deepcopy(model.state_dict()), probe

    want, _ = run(None)
    got, probe = run((1, 1))
    for k in want:
        assert torch.equal(got[k], want[k]), f"{k} differs after close()"
    pairs = probe.get("enc_l1_fwd", [])
    assert pairs, "the probe hook did not see the expert encoder's forward GEMM"
    torch.cuda.synchronize()
    assert all(e0.elapsed_time(e1) > 0 and flops > 0 and e1.elapsed_time(e2) >= 0 for e0, e1, flops, e2 in pairs)


def test_ln_dist_module_path_trajectory_without_resync():
    """`ln_dist` on the HIP module path over its OWN trajectory (no re-synchronisation with the reference's
    post-step parameters between the steps): logged losses to 5e-5, post-step parameters to 1e-3 (the cold, sign-like
    Adam step on softmax-damped gradients is felt by the later steps)."""
    import numpy as np

    case, z, results = MU.replay_training("ln_dist", "cuda", use_engine=False, resync=False)
    spec = H.spec_from_case(case)
    skip = H.bn_fed_biases(spec)
    for t, r in enumerate(results):
        eid = r["eid"]
        for k in ("loss", "recon_loss", "kl_loss"):
            ref = float(np.array(z[f"step{t}/out/{k if k != 'loss' else 'total_loss'}"]))
            assert abs(r["logged"][f"{k}/training/{eid}"] - ref) <= 5e-5 * abs(ref), (t, k)
        for n, v in r["sd"].items():
            key = f"step{t}/sd/{n}"
            if key in z.files and n not in skip and v.is_floating_point() and not n.endswith("running_mean"):
                assert H.rel_l2(v, z[key]) < 1e-3, (t, n, H.rel_l2(v, z[key]))


def test_parallel_conditional_layers_batched_per_launch_match_one_launch_per_position(monkeypatch):
    """SURVEY 8 f2 at the reference's scale: CLVAE with selection_order = ["parallel"] (human_only.yaml:78-79), Z = 128, the
    conditionals assay (8) / sex (2) / dataset_id (273) / donor_id (4 644) + tissue + species -- through the captured engine
    with all positions per launch (mmvae_cond_linear_*_multi, the default) against one launch per position
    (MMVAE_COND_BATCHED=0): same Philox noise, same shuffled concatenation order, the same initial parameters.  One step
    (from identical state: later steps start from parameters a cold, sign-like Adam step has moved by +-lr on rounding
    noise): the loss to 1e-7, the gradients the step left in the shared VAE's and the expert's arenas to 1e-5 -- forward and
    the blocks' weight gradients are the same kernels' arithmetic, the input gradient sums the positions in another tree."""
    import importlib.util
    import random
    import tempfile

    from mmvae_amd import rng, synthetic
    from tests.helpers import rel_l2

    spec = importlib.util.spec_from_file_location("bench_conditional", os.path.join(ROOT, "tools", "bench_conditional.py"))
    BC = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(BC)
    G, B = 2048, 512
    runs = {}
    for batched in ("1", "0"):
        monkeypatch.setenv("MMVAE_COND_BATCHED", batched)
        with tempfile.TemporaryDirectory() as d:
            torch.manual_seed(0)
            model = BC.build(d, G, use_engine=True, parallel=True)
            model.train()
            model.trainer.set_stage("training")
            rng.state(torch.device("cuda", 0))
            rng.reseed(4321)
            random.seed(7)
            x = synthetic.synthetic_counts(B, G, seed=3, device="cuda")
            model.training_step((x, BC.metadata(B, "human", 0), "human"), 0)
            model._flush_engine()
            torch.cuda.synchronize()
            plans = [p for p in model._engine._plans.values() if p.cond is not None]
            assert plans and all(p.cond.batched == (batched == "1") and p.cond.parallel for p in plans)
            opts = model.get_optimizers()
            runs[batched] = (float(model.logged["loss/training/human"]), opts["vae"].arena.grad.detach().cpu().clone(),
                             opts["experts"]["human"].arena.grad.detach().cpu().clone())
            model._engine.close()
    (la, va, ea), (lb, vb, eb) = runs["1"], runs["0"]
    assert abs(la - lb) <= 1e-7 * abs(lb), (la, lb)
    assert rel_l2(va.double(), vb.double()) <= 1e-5 and rel_l2(ea.double(), eb.double()) <= 1e-5
    assert float(va.abs().max()) > 0 and float(ea.abs().max()) > 0


def test_categorical_metadata_columns_give_the_same_conditional_step():
    """The conditional programs' host side maps each cell's metadata value to a condition block: str columns through
    dictionary look-ups (csrc/pylookup.c), `category` columns (what the census' obs frames hold) through one category ->
    block table per categories object and a gather of the codes (a big categories object gets its table the second time it
    is seen; the first time its cells are looked up one by one).  Both must build the same index tables: four steps (two
    per expert; sequential selection order, shuffled) with the same frames as str and as categorical columns sharing one
    dtype per key -- bit-identical losses and gradient arenas.  A categorical value no layer knows is refused like a str one."""
    import importlib.util
    import random
    import tempfile

    import pandas as pd

    from mmvae_amd import rng, synthetic

    spec = importlib.util.spec_from_file_location("bench_conditional", os.path.join(ROOT, "tools", "bench_conditional.py"))
    BC = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(BC)
    G, B = 2048, 256
    runs = {}
    for kind in ("str", "category"):
        with tempfile.TemporaryDirectory() as d:
            torch.manual_seed(0)
            model = BC.build(d, G, use_engine=True)
            model.train()
            model.trainer.set_stage("training")
            rng.state(torch.device("cuda", 0))
            rng.reseed(99)
            random.seed(3)
            out = []
            # categories: every label of the key (more than a batch holds), in another order, one dtype object per key
            dtypes = {k: pd.CategoricalDtype([f"{k}_{j}" for j in reversed(range(n))]) for k, n in BC.SIZES.items()}
            for i, eid in enumerate(("human", "mouse", "human", "mouse")):
                md = BC.metadata(B, eid, i)
                if kind == "category":
                    for k in BC.SIZES:
                        md[k] = md[k].astype(dtypes[k])
                    md["tissue"] = md["tissue"].astype("category")
                x = synthetic.synthetic_counts(B, G, seed=3 + i, device="cuda")
                model.training_step((x, md, eid), i)
                model._flush_engine()
                torch.cuda.synchronize()
                opts = model.get_optimizers()
                out.append((float(model.logged[f"loss/training/{eid}"]), opts["vae"].arena.grad.detach().cpu().clone(),
                            opts["experts"][eid].arena.grad.detach().cpu().clone()))
            if kind == "category":
                plans = [p for p in model._engine._plans.values() if p.cond is not None]
                assert all(len(p.cond.entries["donor_id"]["cat_maps"]) == 1 for p in plans)  # the table path was taken
                bad = BC.metadata(B, "human", 5)
                bad["assay"] = pd.Categorical(["assay_0"] * (B - 1) + ["unheard of"])
                with pytest.raises(KeyError):
                    model.training_step((x, bad, "human"), 4)
            runs[kind] = out
            model._engine.close()
    for (la, va, ea), (lb, vb, eb) in zip(runs["str"], runs["category"]):
        assert la == lb and torch.equal(va, vb) and torch.equal(ea, eb)
        assert float(va.abs().max()) > 0


def test_sequential_conditional_layers_deferred_weight_gradients_are_bit_identical(monkeypatch):
    """Sequential selection order (compare/conditional.yaml: selection_order null): the engine keeps LayerNorm backward +
    input gradient per position on the chain and computes the positions' weight gradients behind it in two launches
    (positions 1 .. n-1 batched, position 0 alone) instead of one per position inside the chain.  Same kernels on the same
    operands: loss and both gradient arenas bit for bit those of MMVAE_COND_BATCHED=0 (a launch per position)."""
    import importlib.util
    import random
    import tempfile

    from mmvae_amd import rng, synthetic

    spec = importlib.util.spec_from_file_location("bench_conditional", os.path.join(ROOT, "tools", "bench_conditional.py"))
    BC = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(BC)
    G, B = 2048, 512
    runs = {}
    for batched in ("1", "0"):
        monkeypatch.setenv("MMVAE_COND_BATCHED", batched)
        with tempfile.TemporaryDirectory() as d:
            torch.manual_seed(0)
            model = BC.build(d, G, use_engine=True)
            model.train()
            model.trainer.set_stage("training")
            rng.state(torch.device("cuda", 0))
            rng.reseed(77)
            random.seed(11)
            x = synthetic.synthetic_counts(B, G, seed=3, device="cuda")
            model.training_step((x, BC.metadata(B, "human", 0), "human"), 0)
            model._flush_engine()
            torch.cuda.synchronize()
            plans = [p for p in model._engine._plans.values() if p.cond is not None]
            assert plans and all(p.cond.defer_dw == (batched == "1") and not p.cond.parallel for p in plans)
            opts = model.get_optimizers()
            runs[batched] = (float(model.logged["loss/training/human"]), opts["vae"].arena.grad.detach().cpu().clone(),
                             opts["experts"]["human"].arena.grad.detach().cpu().clone())
            model._engine.close()
    (la, va, ea), (lb, vb, eb) = runs["1"], runs["0"]
    assert la == lb and torch.equal(va, vb) and torch.equal(ea, eb)
    assert float(va.abs().max()) > 0
