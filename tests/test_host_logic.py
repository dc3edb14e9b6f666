"""Host-logic tests (no GPU): config validation mirrors the reference's tests/test_components.py behaviours, YAML
instantiation with the reference's schema, the trainer loop on CPU plumbing (BASELINE config C1), KL annealing,
log-key tagging, and the C-ABI surface (every symbol the header declares is exported and bound)."""
import os
import re

import numpy as np
import pandas as pd
import pytest
import torch
import torch.nn as nn

from mmvae_amd import backend
from mmvae_amd.modules.base import Encoder, Expert, Experts, FCBlock, FCBlockConfig
from mmvae_amd.modules.base.components import ConditionalLayer, collect_species_files, is_iterable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- FCBlockConfig / FCBlock (reference tests/test_components.py:22-93)
def test_is_iterable():
    assert is_iterable([1]) and is_iterable("s") and is_iterable({"k": 1})
    assert not is_iterable(123) and not is_iterable(None)


def test_fcblock_config_broadcast_and_shapes():
    cfg = FCBlockConfig(layers=[10, 20, 30], dropout_rate=0.5)
    block = FCBlock(cfg)
    assert cfg.n_layers == 2 and block.input_dim == 10 and block.output_dim == 30
    assert cfg.dropout_rate == [0.5, 0.5]
    with backend.cpu_plumbing():
        assert block(torch.randn(5, 10)).shape == (5, 30)
    assert FCBlock(FCBlockConfig(layers=[10, 20, 30])).can_bypass
    assert not FCBlock(FCBlockConfig(layers=[10, 20, 30], return_hidden=True)).can_bypass
    assert FCBlockConfig(layers=[10]).layers == [10, 10] and FCBlockConfig(layers=[10]).n_layers == 1
    cfg = FCBlockConfig(layers=[10, 20], activation_fn=nn.ReLU)
    assert all(issubclass(a, nn.ReLU) for a in cfg.activation_fn)


@pytest.mark.parametrize("kwargs", [dict(layers=[-10, 20]), dict(layers=[10, 20, 30], dropout_rate=[0.5]),
                                    dict(layers=(10, 20)), dict(layers=[10, 20], dropout_rate=1),
                                    dict(layers=[10, 20], use_batch_norm="yes")])
def test_fcblock_config_rejects_bad_input(kwargs):
    with pytest.raises(ValueError):
        FCBlockConfig(**kwargs)


def test_encoder_expert_surface():
    with backend.cpu_plumbing():
        enc = Encoder(latent_dim=5, fc_block_config=FCBlockConfig(layers=[10]))
        assert enc.n_layers == 1 and enc.var_eps == 1e-4
        q_m, q_v, latent, hidden = enc(torch.randn(5, 10))  # 4-tuple when return_dist is False
        assert q_m.shape == q_v.shape == latent.shape == (5, 5) and isinstance(hidden, list)
        ex = Expert("e1", FCBlockConfig(layers=[10], return_hidden=[True]), FCBlockConfig(layers=[10]))
        encoded, hid = ex.encode(torch.randn(5, 10))
        assert encoded.shape == (5, 10) and ex.decode(encoded).shape == (5, 10)
        with pytest.raises(NotImplementedError):
            ex(torch.randn(5, 10))
    cfg = FCBlockConfig(layers=[10, 20])
    exps = Experts([Expert("a", cfg, cfg), Expert("b", cfg, cfg)])
    assert len(exps) == 2 and exps.labels == {"a": 0, "b": 1}


def test_conditional_layer_and_species_files(tmp_path):
    csv = tmp_path / "unique_expression_assay.csv"
    pd.Series(["10x 3' v3", "microwell-seq", "a.b"]).to_csv(csv, header=False, index=False)
    layer = ConditionalLayer("assay", str(csv), FCBlockConfig(layers=[6]))
    assert len(layer.conditions) == 3 and "a_b" in layer.conditions
    x = torch.randn(4, 6)
    with backend.cpu_plumbing():
        out = layer(x, pd.DataFrame({"assay": ["microwell-seq", "a.b", "10x 3' v3", "microwell-seq"]}))
    assert out.shape == x.shape
    (tmp_path / "shared").mkdir()
    (tmp_path / "human").mkdir()
    (tmp_path / "shared" / "unique_expression_assay.csv").write_text("x\n")
    (tmp_path / "human" / "unique_expression_assay.csv").write_text("x\n")
    (tmp_path / "human" / "unique_expression_sex.csv").write_text("x\n")
    files = collect_species_files(str(tmp_path), ["assay", "sex"])
    assert set(files["shared"]) == {"assay"} and set(files["human"]) == {"sex"}


# ---- log keys (reference tests/test_tag_log_dict.py)
def test_tag_log_dict():
    from mmvae_amd.models import tag_log_dict

    d = {"loss": 1}
    assert tag_log_dict(d, ["training", "human"]) == {"loss/training/human": 1}
    assert tag_log_dict(d, ["a", "b"], key_pos="last") == {"a/b/loss": 1}
    assert tag_log_dict(d, [], sep="-") == {"loss": 1}
    with pytest.raises(ValueError):
        tag_log_dict(d, ["a"], key_pos="middle")


def test_linear_kl_annealing_matches_reference(golden_dir):
    import json
    from mmvae_amd.modules.base import KLAnnealingFn, LinearKLAnnealingFn

    z = np.load(os.path.join(golden_dir, "annealing.npz"))
    for i in range(2):
        fn = LinearKLAnnealingFn(**json.loads(str(z[f"linear{i}/kwargs"])))
        vals = [fn.kl_weight]
        for _ in range(20):
            fn.step()
            vals.append(fn.kl_weight)
        np.testing.assert_allclose(vals, z[f"linear{i}/values"], rtol=1e-12)
    c = KLAnnealingFn(0.25)
    c.step()
    assert c.kl_weight == 0.25


# ---- YAML schema + trainer loop on CPU plumbing (BASELINE config C1)
def test_yaml_instantiation_and_c1_fit_on_cpu_plumbing():
    from mmvae_amd import instantiate, synthetic
    from mmvae_amd.models import CMMVAEModel
    from mmvae_amd.trainer import MultiModalBatches, Trainer

    torch.manual_seed(0)
    model = instantiate.load_yaml(os.path.join(ROOT, "configs", "model", "c1_core_vae.yaml"))
    assert isinstance(model, CMMVAEModel) and len(model.module.adversarials) == 0
    assert model.autograd_config.vae_gradient_clip.val == 10
    keys = model.state_dict().keys()
    assert "module.experts.human.encoder.fc_layers.0.lin.weight" in keys
    assert "module.vae.encoder.mean_encoder.bias" in keys
    assert model.module.experts["human"].encoder.fc_layers[0].lin.weight.shape == (1024, 2000)
    w0 = model.module.vae.encoder.mean_encoder.weight.detach().clone()
    trainer = Trainer.from_yaml(os.path.join(ROOT, "configs", "trainer", "config.yaml"))
    x = synthetic.synthetic_counts(128 * 4, 2000)
    batches = [(x[i * 128:(i + 1) * 128], pd.DataFrame({"dummy": [0] * 128}), "human") for i in range(4)]
    with backend.cpu_plumbing():
        hist = trainer.fit(model, MultiModalBatches({"human": batches}, seed=1), val_batches=batches[:2])
        preds = trainer.predict(model, batches[:1])
    train = [h for h in hist if h["stage"] == "training"]
    assert len(train) == 2 and trainer.global_step == 8
    assert all(np.isfinite(h["loss/training/human"]) and h["kl_weight/training/human"] == 1.0 for h in train)
    assert not torch.equal(w0, model.module.vae.encoder.mean_encoder.weight.detach())  # Adam stepped
    assert any("loss/validation/human" in h for h in hist)
    z, meta = preds[0]["z"]
    assert z.shape == (128, 128) and (meta["species"] == "human").all()


def test_instantiate_reference_style_adversarial_yaml(tmp_path):
    """The reference's human_only.yaml layout: adversarials as Adversarial objects, activation as dotted string."""
    import yaml
    from mmvae_amd import instantiate, synthetic

    labels = synthetic.write_label_dir(str(tmp_path), {"assay": 3, "sex": 2})
    node = {"class_path": "cmmvae.modules.base.Adversarial", "init_args": {
        "encoder": {"class_path": "cmmvae.modules.base.FCBlockConfig",
                    "init_args": {"layers": [16, 8], "activation_fn": "torch.nn.ReLU"}},
        "heads": {"class_path": "cmmvae.modules.base.FCBlockConfig", "init_args": {"layers": [8], "activation_fn": None}},
        "conditions": ["assay", "sex"], "labels_dir": labels}}
    from mmvae_amd.modules.base import Adversarial

    Adversarial.labels.clear()
    adv = instantiate.build(yaml.safe_load(yaml.safe_dump(node)))
    assert isinstance(adv, Adversarial)
    assert adv.heads["assay"].fc_layers[0].lin.weight.shape == (3, 8)
    assert Adversarial.labels["sex"] == {"sex_0": 0, "sex_1": 1}


# ---- C-ABI surface
def test_abi_header_symbols_are_exported_and_bound():
    import ctypes
    from mmvae_amd import _lib

    header = open(os.path.join(ROOT, "include", "mmvae_hip.h")).read()
    declared = set(re.findall(r"\b(mmvae_[a-z0-9_]+)\s*\(", header))
    declared -= {"mmvae_bn_params"}
    lib = _lib.load()  # loads on a GPU-less host too (no compute call is made)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"
    unbound = sorted(declared - set(_lib.PROTOTYPES))
    assert not unbound, f"declared in the header but not bound in _lib.PROTOTYPES: {unbound}"
    extra = sorted(set(_lib.PROTOTYPES) - declared)
    assert not extra, f"bound but not declared in the header: {extra}"
    header_version = int(re.search(r"#define\s+MMVAE_ABI_VERSION\s+(\d+)", header).group(1))
    assert lib.mmvae_abi_version() == header_version >= 3 and lib.mmvae_build_arch() == b"gfx950"
    # pure host helpers may be called without a GPU
    t, s = ctypes.c_int(), ctypes.c_int()
    assert lib.mmvae_gemm_plan(0, 512, 1024, 20000, ctypes.byref(t), ctypes.byref(s)) == 0 and s.value >= 8
    assert lib.mmvae_gemm_plan(9, 1, 1, 1, None, None) == _lib.ERR_ARG
    assert lib.mmvae_gemm_get_precision() == _lib.GEMM_PRECISION_BF16X3  # default
    assert lib.mmvae_recon_tiles(20000) == 157 and lib.mmvae_sqnorm_partials(1 << 20) == 16
    assert lib.mmvae_gemm_set_precision(_lib.GEMM_PRECISION_F32) == 0 and lib.mmvae_recon_tiles(20000) == 125
    assert lib.mmvae_gemm_set_precision(7) == _lib.ERR_ARG
    assert lib.mmvae_gemm_set_precision(_lib.GEMM_PRECISION_BF16X3) == 0


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mmvae_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


# ---------------------------------------------------------------------------- conditional-layer index tables (8 f2)
@pytest.mark.parametrize("B,C", [(512, 1), (512, 2), (512, 8), (512, 273), (512, 4644), (33, 5), (1, 1), (65, 2)])
def test_cond_tables_cover_every_cell_once_in_batch_order(B, C):
    """mmvae_amd.cond_tables: chunks partition the cells, stay inside one block, keep batch order, never exceed the
    chunk size or the padded maxima; multi-chunk blocks get consecutive scratch slots and one reduction entry."""
    import numpy as np

    from mmvae_amd import cond_tables as CT

    rng = np.random.default_rng(B * 7 + C)
    local = rng.integers(0, C, B).astype(np.int32)
    t = CT.group_tables(local, base=11)
    assert np.array_equal(t["cond"], local + 11)
    assert np.array_equal(np.sort(t["rows"]), np.arange(B))
    seen, slots = np.zeros(B, dtype=int), {}
    for d, b, e in zip(t["chunk_dst"], t["chunk_beg"], t["chunk_end"]):
        assert 0 < e - b <= CT.CHUNK
        cells = t["rows"][b:e]
        assert np.all(np.diff(cells) > 0), "cells of a block stay in batch order"
        blocks = set(local[cells].tolist())
        assert len(blocks) == 1
        seen[cells] += 1
        if d >= 0:
            assert d - 11 in blocks
        else:
            assert d <= -2
            slots[-2 - d] = blocks.pop()
    assert (seen == 1).all()
    assert sorted(slots) == list(range(len(slots))) and len(slots) <= CT.partial_slots(B)
    for c, s0, n in zip(t["red_cond"], t["red_slot"], t["red_n"]):
        assert n > 1 and all(slots[s] == c - 11 for s in range(s0, s0 + n))
    assert sum(t["red_n"]) == len(slots)
    assert len(t["chunk_dst"]) <= CT.max_chunks(B) and len(t["red_cond"]) <= CT.max_reductions(B)
    assert np.array_equal(t["present"], np.unique(local))
    seg = np.full(CT.words(B), 12345, dtype=np.int32)
    CT.fill_padded(seg, t, B)
    lay = CT.layout(B)
    n = len(t["chunk_dst"])
    assert np.array_equal(seg[lay["chunk_dst"]:lay["chunk_dst"] + n], t["chunk_dst"])
    assert (seg[lay["chunk_dst"] + n:lay["chunk_dst"] + CT.max_chunks(B)] == -1).all()
    assert (seg[lay["red_cond"] + len(t["red_cond"]):lay["red_cond"] + CT.max_reductions(B)] == -1).all()
    assert not (seg == 12345).any(), "every word of the padded set is written"


@pytest.mark.parametrize("R", [512, 1, 7, 33, 64, 1000])
def test_native_cond_tables_equal_the_numpy_statement(R):
    """libmmvae_feed.so's mmvae_feed_cond_tables (what the captured conditional programs build their per-step tables
    with: cond_tables.fill_all, every position in one call) writes exactly the words of group_tables + fill_padded, for
    dense banks (counting sort), one-block banks, and huge sparse indices (comparison sort)."""
    from mmvae_amd import cond_tables as CT

    rng = np.random.default_rng(R)
    sizes = [8, 2, 273, 4644, 4, 1]
    base = np.cumsum([0] + sizes[:-1]).astype(np.int32)
    P = CT.words(R)
    for trial in range(6):
        local = np.stack([rng.integers(0, s, R) for s in sizes]).astype(np.int32)
        if trial == 1:
            local[3] = rng.integers(0, 2**30, R)
        if trial == 2:
            local[2] = 5
        stride = P + trial % 2 * 3
        seg = np.full(len(sizes) * stride + 5, 12345, dtype=np.int32)
        present = CT.fill_all(seg, stride, local, base, R)
        for j in range(len(sizes)):
            t = CT.group_tables(local[j], int(base[j]))
            ref = np.zeros(P, dtype=np.int32)
            CT.fill_padded(ref, t, R)
            assert np.array_equal(seg[j * stride:j * stride + P], ref), (trial, j)
            assert np.array_equal(present[j], t["present"])
        assert (seg[len(sizes) * stride:] == 12345).all()
    with pytest.raises(ValueError):
        CT.fill_all(np.zeros(P, dtype=np.int32), P, -np.ones((1, R), dtype=np.int32), base[:1], R)


def test_metadata_lookup_helper_matches_the_interpreter():
    """csrc/pylookup.c (CPython API, ctypes.PyDLL): table[value] for a list of metadata values into an int32 array; the
    index of the first unknown value instead of an exception; wrong argument types raise."""
    from mmvae_amd import cond_tables as CT

    rng = np.random.default_rng(0)
    keys = [f"donor_{i}" for i in range(300)] + [7, 7.5, ("a", 1)]
    table = {k: i for i, k in enumerate(keys)}
    values = [keys[i] for i in rng.integers(0, len(keys), 512)]
    out = np.full(600, -5, dtype=np.int32)
    assert CT.lookup_i32(table, values, out) == 512
    assert out[:512].tolist() == [table[v] for v in values] and (out[512:] == -5).all()
    values[100] = "never seen"
    assert CT.lookup_i32(table, values, out) == 100
    assert CT.lookup_i32(table, [], out) == 0
    with pytest.raises(TypeError):
        CT.lookup_i32(table, [[1, 2]], out)  # unhashable
    with pytest.raises(OverflowError):
        CT.lookup_i32({"a": 2**40}, ["a"], out)
    with pytest.raises(ValueError):
        CT.lookup_i32(table, values, np.zeros(3, dtype=np.int32))


def test_graft_entry_build_passes():
    """`__graft_entry__.build()` is the driver's "does it build" check: compile both libraries (a no-op when they are up
    to date), load them, verify ABI versions and the target architecture."""
    import __graft_entry__ as entry

    entry.build()


def test_conditional_layers_refuse_more_than_one_latent_sample():
    """K > 1 is this build's extension, conditional layers are the reference's (one latent row per cell through the
    cell's blocks): the combination has no definition and is refused instead of being mis-indexed."""
    import pandas as pd
    import torch

    from mmvae_amd.modules.clvae import CLVAE

    class _Stub:
        conditionals = object()

    with pytest.raises(ValueError, match="one latent sample"):
        CLVAE.after_reparameterize(_Stub(), torch.zeros(3, 4, 8), pd.DataFrame({"a": [0] * 4}))


def test_forked_programs_are_fenced_to_validated_runtimes():
    """The multi-stream captured programs run only on HIP runtimes they were validated on (the hipGraphLaunch hazard of
    DESIGN.md has no root cause); MMVAE_SIDE_DW_ANY=1 overrides."""
    import dataclasses

    from mmvae_amd.engine import EngineSettings, forks_allowed

    st = EngineSettings()
    assert forks_allowed(st, "7.0.51831")
    assert not forks_allowed(st, "7.2.26015") and not forks_allowed(st, "")
    assert forks_allowed(dataclasses.replace(st, side_dw_any=True), "7.2.26015")


def test_lookahead_announces_the_next_batch_and_the_model_consumes_the_hint():
    """mmvae_amd.trainer.Lookahead (the loop side of the step engine's software pipelining across steps): batches come out
    in order and unchanged, and before each one the model is told which one follows (None behind the last)."""
    from mmvae_amd.trainer import Lookahead

    class Model:
        def __init__(self):
            self.seen = []

        def hint_next_batch(self, b):
            self.seen.append(b)

    m = Model()
    batches = [("x0", "m0", "human"), ("x1", "m1", "mouse"), ("x2", "m2", "human")]
    out = []
    for b in Lookahead(batches, m):
        out.append((b, m.seen[-1]))
    assert [b for b, _ in out] == batches
    assert [h for _, h in out] == [batches[1], batches[2], None]
    assert list(Lookahead([], m)) == [] and len(Lookahead(batches, m)) == 3
    # CMMVAEModel keeps one hint and hands it to exactly one training step
    from mmvae_amd.models import CMMVAEModel

    class Probe(CMMVAEModel):
        def __init__(self):  # (no module: only the hint bookkeeping is exercised)
            pass

    p = Probe()
    p.hint_next_batch(("x", "meta", "mouse"))
    assert p._next_hint == ("x", "mouse")
    p.hint_next_batch(None)
    assert p._next_hint is None
