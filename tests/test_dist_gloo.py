"""Data-parallel path on CPU with the gloo backend, world_size 2 (the N > 1 logic of bench.py / mmvae_amd.dist):
bucketed all-reduce of the flat gradient arenas, gradient averaging folded into the optimiser, identical parameters
on every rank, and equality with a hand-computed average of the per-shard gradients."""
import os
import socket
import tempfile

import pandas as pd
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers as H
from tests import mirror_utils as MU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _build(case, tmpdir):
    from mmvae_amd import backend

    with backend.cpu_plumbing():
        torch.manual_seed(0)
        model = MU.build_mirror(case, "cpu", tmpdir)
        model.train()
        model.trainer.set_stage("training")
        model.optimizers()
    return model


def _step(model, x, eps, mask, eid):
    from mmvae_amd import backend

    with backend.cpu_plumbing():
        model.module.vae.encoder.explicit_eps = eps
        model.module.experts[eid].encoder.explicit_masks = mask
        model.training_step((x, pd.DataFrame({"dummy": [0] * x.shape[0]}), eid), 0)


def _flat_params(model, case):
    """All parameters except Linear biases that feed a BatchNorm (zero true gradient -> chaotic under Adam, and the
    workers run single-threaded matmuls with a different rounding than the 8-thread main process)."""
    skip = H.bn_fed_biases(H.spec_from_case(case))
    return torch.cat([p.detach().flatten() for n, p in model.module.named_parameters() if n not in skip])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MMVAE_DIST_TIMEOUT_S="120")
    torch.set_num_threads(1)
    from mmvae_amd import backend, dist as mdist

    assert mdist.init_from_env("gloo") == world
    case, z = H.load_case("c1_small")
    x, eps, masks, _ = H.step_inputs(z, 0)
    B = x.shape[0] // world
    rows = slice(rank * B, (rank + 1) * B)
    mask = {int(k.split(".")[4]): m[rows] for k, m in masks.items()}
    with tempfile.TemporaryDirectory() as d:
        model = _build(case, d)
        MU.load_state(model, z, "sd0/")
        mdist.broadcast_parameters(model)
        red = mdist.attach(model, mdist.GradAllReducer(bucket_bytes=4096, side_stream=False))
        assert all(abs(o.grad_scale - 1.0 / world) < 1e-12 and o.reducer is red for o in model.optimizers())
        _step(model, x[rows], eps[rows], mask, "human")
        flat = _flat_params(model, case)
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
        if rank == 0:
            torch.save(flat, os.path.join(out_dir, "dp.pt"))
        # the engine's exchange primitives: default attach() builds a second communicator for the small arenas;
        # both reduce a flat arena in place, in buckets
        red2 = mdist.attach(model)
        assert red2.small_group is not red2.group
        for small in (False, True):
            t = torch.full((5000,), float(rank + 1))
            red2.bucket_bytes = 4096
            red2.reduce_here(t, small=small)
            assert torch.equal(t, torch.full((5000,), float(sum(range(1, world + 1)))))
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_matches_manual_average(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    dp = torch.load(os.path.join(tmp_path, "dp.pt"))
    # reference computation in one process: per-shard gradients, averaged, then the same clip + Adam
    case, z = H.load_case("c1_small")
    x, eps, masks, _ = H.step_inputs(z, 0)
    B = x.shape[0] // world
    with tempfile.TemporaryDirectory() as d:
        from mmvae_amd import backend

        shard_grads = []
        for r in range(world):
            m = _build(case, d)
            MU.load_state(m, z, "sd0/")
            rows = slice(r * B, (r + 1) * B)
            mask = {int(k.split(".")[4]): v[rows] for k, v in masks.items()}
            for o in m.optimizers():
                o.step = lambda closure=None: None  # gradients only
            _step(m, x[rows], eps[rows], mask, "human")
            shard_grads.append([o.arena.grad.clone() for o in m.optimizers()])
        m = _build(case, d)
        MU.load_state(m, z, "sd0/")
        with backend.cpu_plumbing():
            for i, o in enumerate(m.optimizers()):
                o.arena.grad.copy_(sum(g[i] for g in shard_grads))
                o.grad_scale = 1.0 / world
                o.set_clip(10.0)
                o.arena.gather_grads = lambda *a, **k: []
                if i in (m.optimizer_map["vae"], m.optimizer_map["experts"]["human"]):
                    o.step()
        ref = _flat_params(m, case)
    assert H.rel_l2(dp, ref) < 1e-5


def test_bucket_ranges_cover_the_arena_exactly():
    from mmvae_amd.dist import bucket_ranges

    for n, bb in ((10, 16), (1000, 4096), (4097, 4096), (1, 4)):
        rs = bucket_ranges(n, bb)
        assert rs[0].start == 0 and rs[-1].stop == n
        assert all(a.stop == b.start for a, b in zip(rs, rs[1:]))
        assert all(len(r) <= max(bb // 4, 1) for r in rs)


def _presence_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MMVAE_DIST_TIMEOUT_S="120")
    torch.set_num_threads(1)
    from mmvae_amd import backend, dist as mdist
    from mmvae_amd.optim import HipAdam

    assert mdist.init_from_env("gloo") == world
    with backend.cpu_plumbing():
        torch.manual_seed(3)
        params = [torch.nn.Parameter(torch.randn(5, 3)) for _ in range(4)]
        opt = HipAdam(params, lr=1e-2, weight_decay=0.0)
        opt.sparse_presence = True
        opt.reducer = mdist.GradAllReducer(side_stream=False)
        opt.grad_scale = 1.0 / world
        before = [p.detach().clone() for p in params]
        g = torch.Generator().manual_seed(100)
        grads = [torch.randn(5, 3, generator=g) for _ in range(4)]  # the same numbers on both ranks
        mine = {0: (0, 1), 1: (1, 2)}[rank]  # rank 0 saw blocks 0 and 1, rank 1 blocks 1 and 2; nobody saw block 3
        for i in mine:
            params[i].grad = grads[i].clone() * (rank + 1)
        opt.step()
        torch.save({"after": [p.detach().clone() for p in params], "before": before, "steps": [int(v) for v in opt.host_steps()]},
                   os.path.join(out_dir, f"presence{rank}.pt"))
    dist.destroy_process_group()


def test_parameters_seen_by_any_rank_step_on_every_rank(tmp_path):
    """Condition blocks: which parameters got a gradient is a per-rank fact.  Under data parallelism a parameter steps
    when ANY rank produced a gradient for it (ranks that did not contribute zeros), identically on every rank; a
    parameter nobody saw is skipped (no moment decay, no step count) -- torch.optim.Adam under DDP with unused
    parameters."""
    world, port = 2, _free_port()
    mp.spawn(_presence_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(os.path.join(tmp_path, "presence0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "presence1.pt"))
    for a, b in zip(r0["after"], r1["after"]):
        assert torch.equal(a, b), "ranks diverged"
    for i in range(3):
        assert not torch.equal(r0["after"][i], r0["before"][i]), f"block {i} was seen by a rank and must step"
    assert torch.equal(r0["after"][3], r0["before"][3]), "block 3 was seen by nobody and must not move"
    assert list(r0["steps"]) == [1, 1, 1, 0] and list(r1["steps"]) == [1, 1, 1, 0]
    # the update itself: Adam's first step moves every entry by lr * sign(averaged gradient)
    g = torch.Generator().manual_seed(100)
    grads = [torch.randn(5, 3, generator=g) for _ in range(4)]
    avg = [grads[0] * 1 / 2, grads[1] * (1 + 2) / 2, grads[2] * 2 / 2]
    for i in range(3):
        want = r0["before"][i] - 1e-2 * torch.sign(avg[i])
        assert torch.allclose(r0["after"][i], want, atol=1e-6), i


def _cond_batch(case, z, t=0):
    eid = case["schedule"][t]
    x, eps, masks, labels = H.step_inputs(z, t)
    meta = {c: [f"{c}_{int(i)}" for i in idx] for c, idx in labels.items()}
    meta.update(H.cond_inputs(case, z, t, eid)[0])
    return eid, x, eps, masks, pd.DataFrame(meta)


def _cond_step(model, case, x, eps, masks, meta, eid, rows):
    import random

    from mmvae_amd import backend

    with backend.cpu_plumbing():
        model.kl_annealing_fn.kl_weight = case["kl_weights"][0]
        model.module.vae.encoder.explicit_eps = eps[rows]
        model.module.experts[eid].encoder.explicit_masks = {
            int(k.split(".")[4]): m[rows] for k, m in masks.items() if k.startswith(f"experts.{eid}.encoder.fc_layers.")}
        random.seed(case["seed"] * 100)
        model.training_step((x[rows], meta.iloc[rows].reset_index(drop=True), eid), 0)


def _cond_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MMVAE_DIST_TIMEOUT_S="120")
    torch.set_num_threads(1)
    from mmvae_amd import dist as mdist

    assert mdist.init_from_env("gloo") == world
    case, z = H.load_case("cond_seq")
    eid, x, eps, masks, meta = _cond_batch(case, z)
    B = x.shape[0] // world
    rows = slice(rank * B, (rank + 1) * B)
    with tempfile.TemporaryDirectory() as d:
        model = _build(case, d)
        MU.load_state(model, z, "sd0/")
        mdist.broadcast_parameters(model)
        mdist.attach(model, mdist.GradAllReducer(bucket_bytes=1 << 16, side_stream=False))
        _cond_step(model, case, x, eps, masks, meta, eid, rows)
        flat = torch.cat([p.detach().flatten() for _, p in model.module.named_parameters()])
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
        if rank == 0:
            torch.save({n: p.detach().clone() for n, p in model.module.named_parameters()},
                       os.path.join(out_dir, "cond_dp.pt"))
    dist.destroy_process_group()


def test_two_rank_step_of_a_conditional_model_matches_the_manual_union(tmp_path):
    """Conditional layers under data parallelism (module path, CPU plumbing over gloo): the two ranks see different
    conditions; every block seen by either steps on both with the averaged gradient (zeros from the rank that did not
    see it), blocks seen by neither are skipped -- equal to a one-process computation of exactly that."""
    world, port = 2, _free_port()
    mp.spawn(_cond_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    dp = torch.load(os.path.join(tmp_path, "cond_dp.pt"))
    case, z = H.load_case("cond_seq")
    eid, x, eps, masks, meta = _cond_batch(case, z)
    B = x.shape[0] // world
    with tempfile.TemporaryDirectory() as d:
        from mmvae_amd import backend

        shard = []
        for r in range(world):
            m = _build(case, d)
            MU.load_state(m, z, "sd0/")
            for o in m.optimizers():
                o.step = lambda closure=None: None  # gradients only
            _cond_step(m, case, x, eps, masks, meta, eid, slice(r * B, (r + 1) * B))
            shard.append([(o.arena.grad.clone(), set(o._inactive)) for o in m.optimizers()])
        m = _build(case, d)
        MU.load_state(m, z, "sd0/")
        seen_by_one_only = 0
        with backend.cpu_plumbing():
            for i, o in enumerate(m.optimizers()):
                if i not in (m.optimizer_map["vae"], m.optimizer_map["experts"][eid]):
                    continue
                o.arena.grad.copy_(sum(g for g, _ in (s[i] for s in shard)))
                absent_everywhere = sorted(set.intersection(*[s[i][1] for s in shard]))
                seen_by_one_only += len(set.union(*[s[i][1] for s in shard])) - len(absent_everywhere)
                o.grad_scale = 1.0 / world
                o.set_clip(10.0)
                o._gather = lambda absent=absent_everywhere: list(absent)
                o.step()
        assert seen_by_one_only > 0, "the shards must differ in the conditions they hold for this test to mean anything"
        skip = H.bn_fed_biases(H.spec_from_case(case))
        for n, p in m.module.named_parameters():
            if n not in skip:
                assert H.rel_l2(dp[n], p.detach()) < 1e-5, n


def _adv_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MMVAE_DIST_TIMEOUT_S="120")
    torch.set_num_threads(1)
    from mmvae_amd import backend, dist as mdist

    assert mdist.init_from_env("gloo") == world
    case, z = H.load_case("adversarial")
    with tempfile.TemporaryDirectory() as d:
        model = _build(case, d)
        MU.load_state(model, z, "sd0/")
        mdist.broadcast_parameters(model)
        mdist.attach(model, mdist.GradAllReducer(bucket_bytes=1 << 16, side_stream=False))
        for t in range(2):  # discriminator + generator phases, two modalities
            eid = case["schedule"][t]
            x, eps, masks, labels = H.step_inputs(z, t)
            B = x.shape[0] // world
            rows = slice(rank * B, (rank + 1) * B)
            meta = pd.DataFrame({c: [f"{c}_{int(i)}" for i in idx] for c, idx in labels.items()}).iloc[rows]
            with backend.cpu_plumbing():
                model.kl_annealing_fn.kl_weight = case["kl_weights"][t]
                model.module.vae.encoder.explicit_eps = eps[rows]
                model.module.experts[eid].encoder.explicit_masks = {
                    int(k.split(".")[4]): m[rows] for k, m in masks.items()
                    if k.startswith(f"experts.{eid}.encoder.fc_layers.")}
                model.training_step((x[rows], meta.reset_index(drop=True), eid), t)
        flat = torch.cat([p.detach().flatten() for _, p in model.module.named_parameters()])
        assert torch.isfinite(flat).all()
        gathered = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], g) for g in gathered), "ranks diverged"
    dist.destroy_process_group()


def test_two_rank_adversarial_steps_keep_the_replicas_identical():
    """Discriminator and generator phases under data parallelism (module path, gloo): every optimiser of the step --
    adversaries, shared VAE, active expert -- exchanges its gradients; the replicas stay bit-identical."""
    world, port = 2, _free_port()
    mp.spawn(_adv_worker, args=(world, port, ""), nprocs=world, join=True)


# ------------------------------------------------------------------------------------------- sharded expert update
@pytest.mark.parametrize("numel,world", [(42_003_456, 8), (1000, 8), (1001, 2), (7, 8), (4, 1), (170_000_123, 4)])
def test_arena_shards_partition_the_padded_arena(numel, world):
    """ParamArena.shard: equal slices of a multiple of 4 elements, in rank order, covering the arena once; the rounding
    stays inside the 32 spare elements behind an arena (else None: the engine falls back to the all-reduce)."""
    from mmvae_amd.optim import ParamArena

    a = ParamArena.__new__(ParamArena)
    a.numel = numel
    sh = [a.shard(world, r) for r in range(world)]
    assert all(s is not None for s in sh)
    per = sh[0][0]
    assert per % 4 == 0 and per * world >= numel and per * world <= numel + 32
    assert [s[1] for s in sh] == [r * per for r in range(world)]
    assert sum(s[2] for s in sh) == numel and all(0 <= s[2] <= per for s in sh)


def _shard_worker(rank, world, port, out_dir):
    """The arithmetic of the engine's sharded update (engine._optimizer_sharded) replayed with torch ops over gloo:
    reduce-scatter -> slice norm, all-gathered -> clip + Adam on the slice -> all-gather of the parameters; against the
    unsharded all-reduce + full update (HipAdam CPU plumbing).  Then HipAdam.sync_sharded_state()."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      MMVAE_DIST_TIMEOUT_S="120")
    torch.set_num_threads(1)
    from mmvae_amd import backend, dist as mdist
    from mmvae_amd.optim import HipAdam

    assert mdist.init_from_env("gloo") == world
    with backend.cpu_plumbing():
        torch.manual_seed(5)
        params = [torch.nn.Parameter(torch.randn(n)) for n in (1001, 37, 4096, 3)]
        ref_params = [torch.nn.Parameter(p.detach().clone()) for p in params]
        opt, ref = HipAdam(params, max_grad_norm=10.0), HipAdam(ref_params, max_grad_norm=10.0)
        a, ar = opt.arena, ref.arena
        per, lo, n_loc = a.shard(world, rank)
        g_cfg = opt.param_groups[0]
        b1, b2 = g_cfg["betas"]
        for step in range(1, 4):
            torch.manual_seed(100 * step + rank)
            local = torch.randn(a.numel) * 3.0  # this rank's gradients
            # unsharded: all-reduce, then the full update on every rank
            ar.grad.copy_(local)
            dist.all_reduce(ar.grad)
            ref.grad_scale = 1.0 / world
            ar.gather_grads = lambda *x, **k: []
            ref.step()
            # sharded
            a.grad.copy_(local)
            full = a.grad_full[:per * world]
            dist.reduce_scatter_tensor(full[lo:lo + per], full)
            gs = 1.0 / world
            mine = (a.grad[lo:lo + n_loc].double() ** 2).sum().float().reshape(1)
            allsq = torch.zeros(world)
            dist.all_gather_into_tensor(allsq, mine)
            norm = float(allsq.double().sum().sqrt()) * gs
            clip = min(1.0, 10.0 / (norm + 1e-6))
            sl = slice(lo, lo + n_loc)
            gr = a.grad[sl] * gs * clip + g_cfg["weight_decay"] * a.data[sl]
            a.exp_avg[sl].lerp_(gr, 1 - b1)
            a.exp_avg_sq[sl].mul_(b2).addcmul_(gr, gr, value=1 - b2)
            denom = a.exp_avg_sq[sl].sqrt() / ((1 - b2 ** step) ** 0.5) + g_cfg["eps"]
            a.data[sl].addcdiv_(a.exp_avg[sl], denom, value=-g_cfg["lr"] / (1 - b1 ** step))
            dfull = a.data_full[:per * world]
            dist.all_gather_into_tensor(dfull, dfull[lo:lo + per])
            assert H.rel_l2(a.data, ar.data) < 2e-6, (step, H.rel_l2(a.data, ar.data))
            gathered = [torch.empty_like(a.data) for _ in range(world)]
            dist.all_gather(gathered, a.data.clone())
            assert all(torch.equal(gathered[0], t) for t in gathered), "replicas diverged"
        # the moments of the other slices are stale until they are gathered
        stale = a.exp_avg.clone()
        opt.sharded = True
        opt.sync_sharded_state()
        assert not opt.sharded
        assert H.rel_l2(a.exp_avg, ar.exp_avg) < 2e-6 and H.rel_l2(a.exp_avg_sq, ar.exp_avg_sq) < 2e-6
        assert world == 1 or not torch.equal(stale, a.exp_avg)
        assert torch.equal(a.exp_avg[sl], stale[sl])
    dist.destroy_process_group()


def test_sharded_update_equals_all_reduce_update_and_moments_gather(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_shard_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
