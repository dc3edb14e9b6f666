"""Conditional layers inside a captured step program (used by mmvae_amd.engine._Plan)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib, cond_tables, dist as mdist
from .engine_common import _PinnedRing, _p
from .modules.base.components import ConditionalLayer, FCBlock
from .optim import arena_of


class CondProgram:
    """The conditional layers of a CLVAE (reference `ConditionalLayers.forward`, components.py:586-631, over
    `ConditionalLayer.forward` :365-413) inside a captured program.

    Every layer is a bank of Linear(Z, Z) (+ LayerNorm without affine) blocks that live in the shared-VAE optimiser's
    arena; the kernels address a block through per-condition element offsets (mmvae_cond_linear_*), so ONE table of all
    banks of this species serves every layer: position j of the (per step shuffled) selection order simply reads the
    GLOBAL block index of each cell from its own static array.  Per step the host derives from the metadata, for every
    position: cond[R] (global block per cell), the cells sorted by block and cut into chunks, the reduction list of
    the blocks with several chunks (mmvae_amd.cond_tables, padded to fixed sizes), and the optimiser's job table -- dense parameters + the blocks present, with each
    tensor's own bias corrections (torch.optim.Adam semantics for parameters without a gradient: skipped, per-parameter
    step counts) -- packs them into one page-locked array and uploads it with one copy before the replay."""

    @staticmethod
    def _resolve(cl, key, eid):
        layer = cl.layers[key]
        if isinstance(layer, nn.ModuleDict):
            if eid not in layer:
                raise RuntimeError(f"'species' must be set to access non-shared conditional layer for batch_key '{key}'")
            layer = layer[eid]
        return layer

    @staticmethod
    def _blocks(layer):
        return list(layer.conditions.values()) if isinstance(layer, ConditionalLayer) else [layer]

    @staticmethod
    def _all_blocks(cl):
        for layer in cl.layers.values():
            for sub in (layer.values() if isinstance(layer, nn.ModuleDict) else [layer]):
                yield from CondProgram._blocks(sub)

    @staticmethod
    def supported(cl, opt_vae, Z: int) -> bool:
        has_ln = None
        for blk in CondProgram._all_blocks(cl):
            if not isinstance(blk, FCBlock) or len(blk.fc_layers) != 1:
                return False
            names = [n for n, _ in blk.fc_layers[0].named_children()]
            if any(n not in ("lin", "ln") for n in names):
                return False
            lin = blk.fc_layers[0].lin
            if lin.in_features != Z or lin.out_features != Z or lin.bias is None:
                return False
            ln = "ln" in names
            if has_ln is None:
                has_ln = ln
            if ln != has_ln:
                return False
            for p in (lin.weight, lin.bias):
                hit = arena_of(p)
                if hit is None or hit[0] is not opt_vae:
                    return False
        return has_ln is not None

    def __init__(self, plan, cl, eid: str, train: bool):
        import numpy as np
        import pandas as pd

        self.np, self.pd = np, pd
        self.plan, self.cl, self.eid, self.train = plan, cl, eid, train
        eng = plan.eng
        self.eng = eng
        self.opt = eng.opts["vae"]
        a = self.opt.arena
        R = self.R = plan.R
        Z = self.Z = plan.Z
        self.parallel = bool(cl.is_parallel)
        self.keys = list(cl.selection_order)
        self.n_pos = len(self.keys)
        # ---- one table of every block this species can meet
        w_off, b_off = [], []
        self.entries = {}
        for key in self.keys:
            layer = self._resolve(cl, key, eid)
            blocks = self._blocks(layer)
            lins = [b.fc_layers[0].lin for b in blocks]
            w_idx = np.array([arena_of(l.weight)[1] for l in lins], dtype=np.int64)
            b_idx = np.array([arena_of(l.bias)[1] for l in lins], dtype=np.int64)
            # (value -> block caches are shared by every plan of this expert: a second plan -- another batch size, an
            # evaluation program -- does not start from empty tables)
            shared = eng.__dict__.setdefault("_cond_lookup", {}).setdefault((eid, key), dict(raw_index={}, cat_maps={}, last=None))
            ent = dict(base=len(w_off), w_idx=w_idx, b_idx=b_idx, layer=layer if isinstance(layer, ConditionalLayer) else None,
                       raw_index=shared["raw_index"], cat_maps=shared["cat_maps"], shared=shared)
            if ent["layer"] is not None:
                ent["index"] = {k: i for i, k in enumerate(layer.conditions.keys())}
            w_off += [a.offsets[i] for i in w_idx]
            b_off += [a.offsets[i] for i in b_idx]
            self.entries[key] = ent
        first = next(self._all_blocks(cl)).fc_layers[0]
        self.ln_eps = float(first.ln.eps) if hasattr(first, "ln") else None
        dev = eng.device
        self.w_off = torch.tensor(w_off, dtype=torch.int64, device=dev)
        self.b_off = torch.tensor(b_off, dtype=torch.int64, device=dev)
        # ---- optimiser bookkeeping: dense parameters (always stepped) vs condition blocks (any species)
        managed = set()
        for blk in self._all_blocks(cl):
            lin = blk.fc_layers[0].lin
            managed.update((arena_of(lin.weight)[1], arena_of(lin.bias)[1]))
        self.dense = np.array([i for i in range(len(a.params)) if i not in managed], dtype=np.int64)
        jpb = (Z * Z + 16383) // 16384 + 1  # jobs of one block: weight chunks + bias
        b1, b2 = self.opt.param_groups[0]["betas"]
        n_dense_jobs = len(self.opt.job_table(self.dense, b1, b2)) if train else 0
        # blocks that can step: at most one per cell of the batch -- of EVERY rank's batch under data parallelism
        cells = R * (mdist.world_size() if mdist.collectives_active() else 1)
        self.max_jobs = n_dense_jobs + sum(min(cells, len(e["w_idx"])) for e in self.entries.values()) * jpb if train else 0
        self.n_exchange, self.exchange_floats, self.staging = 0, 0, None
        if mdist.collectives_active() and train:
            # gradient exchange over the union's segments only: staging for every job of a full table
            # (sized by what a full table can hold, in the 128-float units of mmvae_jobs_pack: the dense parameters and
            # one weight + bias per block that can step -- not 64 KB per job: 4 644 donor blocks at world 8 would have
            # reserved gigabytes)
            pad128 = lambda v: (int(v) + 127) // 128 * 128  # noqa: E731
            n_blocks = (self.max_jobs - n_dense_jobs) // jpb
            dense_floats = sum(pad128(a.params[i].numel()) for i in self.dense) + 128 * n_dense_jobs
            self.staging = eng.buf("cond.exchange", (dense_floats + n_blocks * (pad128(Z * Z) + 128 * jpb + pad128(Z)),))
        if mdist.collectives_active():
            self.max_jobs += self.max_jobs - n_dense_jobs  # + the segments retired from the previous step's union
        # ---- static device tables, filled by load(): one padded cond_tables set per position
        self.P = cond_tables.words(R)
        self.lay = cond_tables.layout(R)
        self.n_chunks, self.n_red = cond_tables.max_chunks(R), cond_tables.max_reductions(R)
        # (r5) "parallel" selection order: the positions read one input and are independent of each other -- ONE launch per
        # kernel for all of them (mmvae_cond_linear_*_multi; LayerNorm over the [R, n_pos Z] matrices as R n_pos rows of Z)
        # instead of n_pos launches of each in sequence (6 positions: 6 x 62 us of latency-bound launches per step)
        self.batched = bool(self.parallel and self.n_pos > 1 and Z <= 256 and eng.settings.cond_batched)
        # (r5) sequential order: a position's forward and input gradient depend on the previous position, its WEIGHT
        # gradient on nothing downstream -- the chain keeps LayerNorm backward + dx per position, the weight gradients of
        # positions 1 .. n-1 (inputs: the previous positions' outputs, one strided buffer) are ONE batched launch behind
        # the chain, position 0's (input: z) one more: 2 x 2 launches instead of n x 2 on the critical path
        self.defer_dw = bool(not self.parallel and self.n_pos > 2 and Z <= 256 and eng.settings.cond_batched and train)
        self.part_stride = cond_tables.partial_slots(R) * (Z * Z + Z)
        self.dw_partials = eng.buf("cond.dw_partials", ((self.n_pos if (self.batched or self.defer_dw) else 1) * self.part_stride,)) if train else None
        self.idx_words = (self.n_pos * self.P + 1) // 2 * 2
        words = self.idx_words + 6 * self.max_jobs
        self.pack_dev = eng.buf(f"cond.pack.{eid}.{int(train)}", (words,), torch.int32)
        self.ring = _PinnedRing(words)
        self._scratch = np.zeros(words, dtype=np.int32)
        self._local = np.zeros((self.n_pos, R), dtype=np.int32)  # per step: block index of every cell at every position
        self._bases = np.zeros(self.n_pos, dtype=np.int32)
        self.jobs_ptr = self.pack_dev.data_ptr() + 4 * self.idx_words
        self.partials = eng.buf(f"cond.sqparts.{eid}", (max(self.max_jobs, 1),)) if train else None
        self._active = None
        # ---- activations
        self.lin = eng.buf("cond.lin", (R, Z))  # pre-LayerNorm output of the position being computed
        self.gl = eng.buf("cond.gl", (R, Z))    # gradient w.r.t. that pre-LayerNorm output
        wide = self.n_pos * Z if self.parallel else Z
        self.out = eng.buf("cond.out", (R, wide)) if self.parallel else None
        self.d_out = eng.buf("cond.d_out", (R, wide)) if train else None
        self.y = [None] * self.n_pos
        self.invstd = [eng.buf(f"cond.invstd{j}", (R,)) for j in range(self.n_pos)] if self.ln_eps is not None else None
        self.mean = eng.buf("cond.mean", (R,)) if self.ln_eps is not None else None
        self.x_in = [None] * self.n_pos
        if self.batched:
            self.lin_wide = eng.buf("cond.lin_wide", (R, wide)) if self.ln_eps is not None else None
            self.gl_wide = eng.buf("cond.gl_wide", (R, wide)) if (self.ln_eps is not None and train) else None
            self.invstd_all = eng.buf("cond.invstd_all", (R * self.n_pos,)) if self.ln_eps is not None else None
            self.mean_all = eng.buf("cond.mean_all", (R * self.n_pos,)) if self.ln_eps is not None else None

    def _ptr(self, j: int, name: str) -> int:
        """Device address of array `name` (cond_tables.layout) of position j."""
        return self.pack_dev.data_ptr() + 4 * (j * self.P + self.lay[name])

    # ------------------------------------------------------------------------------------------ program emission
    def emit_forward(self, z: torch.Tensor):
        plan, lib, R, Z = self.plan, self.plan.lib, self.R, self.Z
        params = self.opt.arena.data
        if self.batched:
            n, wide = self.n_pos, self.out.shape[1]
            lin_out = self.lin_wide if self.ln_eps is not None else self.out
            for j in range(n):
                self.x_in[j] = z
                self.y[j] = (self.out[:, j * Z:(j + 1) * Z], wide)
            plan._emit(lib.mmvae_cond_linear_fwd_multi, n, R, Z, Z, _p(z), Z, 0, _p(params), _p(self.w_off), _p(self.b_off),
                       self._ptr(0, "cond"), self._ptr(0, "rows"), self.P, _p(lin_out), wide, Z)
            if self.ln_eps is not None:  # row r, position j of [R, n Z] = row r n + j of [R n, Z]
                plan._emit(lib.mmvae_layernorm_fwd, R * n, Z, _p(self.lin_wide), Z, self.ln_eps, _p(self.out), Z,
                           _p(self.mean_all), _p(self.invstd_all))
            return self.out, wide
        cur = z
        y_all = self.eng.buf("cond.y_all", (self.n_pos, R, Z)) if self.defer_dw else None
        for j in range(self.n_pos):
            x = z if self.parallel else cur
            self.x_in[j] = x
            if self.parallel:
                y, ldy = self.out[:, j * Z:(j + 1) * Z], self.out.shape[1]
            elif y_all is not None:  # one buffer: position j + 1 reads its input at y_all[j] (uniform stride for the batched dW)
                y, ldy = y_all[j], Z
            else:
                y, ldy = self.eng.buf(f"cond.y{j}", (R, Z)), Z
            self.y[j] = (y, ldy)
            lin_out, ld_lin = (self.lin, Z) if self.ln_eps is not None else (y, ldy)
            plan._emit(lib.mmvae_cond_linear_fwd, R, Z, Z, _p(x), Z, _p(params), _p(self.w_off), _p(self.b_off),
                       self._ptr(j, "cond"), self._ptr(j, "rows"), _p(lin_out), ld_lin)
            if self.ln_eps is not None:
                plan._emit(lib.mmvae_layernorm_fwd, R, Z, _p(self.lin), Z, self.ln_eps, _p(y), ldy, _p(self.mean),
                           _p(self.invstd[j]))
            cur = y
        return (self.out, self.out.shape[1]) if self.parallel else (cur, Z)

    def emit_backward(self, dz: torch.Tensor):
        """d_out (gradient w.r.t. what the decoder read) -> block gradients straight into the arena + dz."""
        plan, lib, R, Z = self.plan, self.plan.lib, self.R, self.Z
        a = self.opt.arena
        g, ldg = self.d_out, self.d_out.shape[1]
        if self.batched:
            n = self.n_pos
            gl = g
            if self.ln_eps is not None:
                plan._emit(lib.mmvae_layernorm_bwd, R * n, Z, _p(g), Z, _p(self.out), Z, _p(self.invstd_all), _p(self.gl_wide), Z)
                gl = self.gl_wide
            plan._emit(lib.mmvae_cond_linear_bwd_dw_multi, n, self.n_chunks, self._ptr(0, "chunk_dst"), self._ptr(0, "chunk_beg"),
                       self._ptr(0, "chunk_end"), self._ptr(0, "rows"), self.P, Z, Z, _p(gl), ldg, Z, _p(self.x_in[0]), Z, 0,
                       _p(a.grad), _p(self.w_off), _p(self.b_off), self.n_red, self._ptr(0, "red_cond"),
                       self._ptr(0, "red_slot"), self._ptr(0, "red_n"), _p(self.dw_partials), self.part_stride)
            plan._emit(lib.mmvae_cond_linear_bwd_dx_multi, n, R, Z, Z, _p(gl), ldg, Z, _p(a.data), _p(self.w_off),
                       self._ptr(0, "cond"), self.P, _p(dz), Z, 0)
            return
        n = self.n_pos
        gl_all = self.eng.buf("cond.gl_all", (n, R, Z)) if self.defer_dw else None  # every position's pre-LayerNorm gradient
        for j in range(self.n_pos - 1, -1, -1):
            y, ldy = self.y[j]
            gj = g[:, j * Z:(j + 1) * Z] if self.parallel else g
            if self.ln_eps is not None:
                gl_j = gl_all[j] if gl_all is not None else self.gl
                plan._emit(lib.mmvae_layernorm_bwd, R, Z, _p(gj), ldg, _p(y), ldy, _p(self.invstd[j]), _p(gl_j), Z)
                gl, ldgl = gl_j, Z
            elif gl_all is not None:  # (no LayerNorm: the incoming gradient is kept for the batched dW below)
                plan._emit(lib.mmvae_axpby, R * Z, 1.0, _p(gj), 0.0, _p(gl_all[j]))
                gl, ldgl = gl_all[j], Z
            else:
                gl, ldgl = gj, ldg
            if gl_all is None:
                plan._emit(lib.mmvae_cond_linear_bwd_dw, self.n_chunks, self._ptr(j, "chunk_dst"), self._ptr(j, "chunk_beg"),
                           self._ptr(j, "chunk_end"), self._ptr(j, "rows"), Z, Z, _p(gl), ldgl, _p(self.x_in[j]), Z, _p(a.grad),
                           _p(self.w_off), _p(self.b_off), self.n_red, self._ptr(j, "red_cond"), self._ptr(j, "red_slot"),
                           self._ptr(j, "red_n"), _p(self.dw_partials))
            if self.parallel:
                plan._emit(lib.mmvae_cond_linear_bwd_dx, R, Z, Z, _p(gl), ldgl, _p(a.data), _p(self.w_off),
                           self._ptr(j, "cond"), self._ptr(j, "rows"), _p(dz), Z, int(j != self.n_pos - 1))
            else:
                dx = dz if j == 0 else self.eng.buf(f"cond.dx{j % 2}", (R, Z))
                plan._emit(lib.mmvae_cond_linear_bwd_dx, R, Z, Z, _p(gl), ldgl, _p(a.data), _p(self.w_off),
                           self._ptr(j, "cond"), self._ptr(j, "rows"), _p(dx), Z, 0)
                g, ldg = dx, Z
        if gl_all is not None:
            # positions 1 .. n-1: input = the previous position's output (y_all[j - 1]); then position 0 (input z)
            plan._emit(lib.mmvae_cond_linear_bwd_dw_multi, n - 1, self.n_chunks, self._ptr(1, "chunk_dst"),
                       self._ptr(1, "chunk_beg"), self._ptr(1, "chunk_end"), self._ptr(1, "rows"), self.P, Z, Z,
                       _p(gl_all[1]), Z, R * Z, self.y[0][0].data_ptr(), Z, R * Z, _p(a.grad), _p(self.w_off), _p(self.b_off),
                       self.n_red, self._ptr(1, "red_cond"), self._ptr(1, "red_slot"), self._ptr(1, "red_n"),
                       self.dw_partials.data_ptr() + 4 * self.part_stride, self.part_stride)
            plan._emit(lib.mmvae_cond_linear_bwd_dw, self.n_chunks, self._ptr(0, "chunk_dst"), self._ptr(0, "chunk_beg"),
                       self._ptr(0, "chunk_end"), self._ptr(0, "rows"), Z, Z, _p(gl_all[0]), Z, _p(self.x_in[0]), Z, _p(a.grad),
                       _p(self.w_off), _p(self.b_off), self.n_red, self._ptr(0, "red_cond"), self._ptr(0, "red_slot"),
                       self._ptr(0, "red_n"), _p(self.dw_partials))

    # ------------------------------------------------------------------------------------------------ per step
    def _local_indices(self, ent, key, metadata, out):
        """out[:] = block index (inside the layer's bank) of every cell, from the layer's metadata column."""
        if ent["layer"] is None:  # the species block: every cell goes through the one block of this expert
            out[:] = 0
            return
        raw_index, layer = ent["raw_index"], ent["layer"]
        col = metadata[layer.batch_key]
        arr = col.array  # (the ExtensionArray itself: the `.cat` accessor costs 20 us per column, this 3)
        cat = arr if isinstance(arr, self.pd.Categorical) else None
        if cat is not None:
            # categorical column (what obs frames of the census hold; row slices of one chunk share the categories object):
            # category -> block once per categories object, then one gather per step -- the look-ups below cost ~0.1 ms per
            # layer and step on freshly unpickled strings (a cache miss per cell) in a program whose host side is the limit
            cats = cat.dtype.categories
            hit = ent["cat_maps"].get(id(cats))
            if (hit is None or hit[0] is not cats) and len(cats) > 64 and ent["shared"]["last"] is not cats:
                # a big categories object seen for the first time: building its table costs a Python loop over every
                # category -- worth it for the categories a chunk's row slices share, not for a frame that was
                # factorised for this batch alone (it takes the per-cell look-ups below; seen again, it gets its table)
                ent["shared"]["last"] = cats
                cat = None
            elif hit is None or hit[0] is not cats:
                if len(ent["cat_maps"]) > 16:
                    ent["cat_maps"].clear()
                to_block = self.np.full(len(cats) + 1, -1, dtype=self.np.int32)  # (last entry: code -1 = missing value)
                for i, v in enumerate(cats.tolist()):
                    k = layer.format_condition_key(str(v))
                    if k in ent["index"]:
                        to_block[i] = ent["index"][k]
                hit = ent["cat_maps"][id(cats)] = (cats, to_block)
        if cat is not None:
            self.np.take(hit[1], cat.codes, out=out, mode="wrap")  # (code -1 wraps to the last entry)
            if (out < 0).any():
                bad = col.iloc[int(self.np.flatnonzero(out < 0)[0])]
                raise KeyError(f"{layer.batch_key}: {bad!r} is not a condition of this layer")
            return
        values = col.tolist()
        n = len(values)
        while True:
            hit = cond_tables.lookup_i32(raw_index, values, out)
            if hit == n:
                return
            v = values[hit]  # first sight of this raw value: its block (KeyError: unknown condition)
            raw_index[v] = ent["index"][layer.format_condition_key(str(v))]

    def load(self, metadata) -> None:
        import random

        np = self.np
        R, P = self.R, self.P
        order = self.keys
        if self.cl.shuffle_selection_order:  # the same draw the module path makes (components.py:601-603)
            order = random.sample(order, len(order))
        # all host arithmetic first, into an ordinary array: the first runtime call after a graph launch waits until the
        # launch has been handed to the device queue (~0.8 ms for this program), and that wait should overlap this work
        pack = self._scratch
        pack[self.idx_words:] = 0  # unused job slots: empty jobs
        active = [self.dense]
        if len(metadata) != R:
            raise _lib.HipLibraryError(f"engine: metadata has {len(metadata)} rows, the batch {R}")
        local, bases = self._local, self._bases
        for j, key in enumerate(order):
            ent = self.entries[key]
            self._local_indices(ent, key, metadata, local[j])
            bases[j] = ent["base"]
        # every position's padded table set in one native call (cond_tables.fill_all; the numpy statement of the same
        # tables, group_tables + fill_padded, took 47 us per position of a host-bound program)
        for key, present in zip(order, cond_tables.fill_all(pack, P, local, bases, R)):
            ent = self.entries[key]
            active.append(ent["w_idx"][present])
            active.append(ent["b_idx"][present])
        if self.train:
            act = np.concatenate(active)
            b1, b2 = self.opt.param_groups[0]["betas"]
            absent_here = None
            if mdist.collectives_active():
                # Data parallelism: a parameter steps when ANY rank produced a gradient for it (the others contribute
                # zeros; DDP's semantics for unused parameters, HipAdam._allreduce on the module path).  One MAX
                # all-reduce of presence flags per step, on the host path ahead of the replay; every rank then builds
                # the same job table, and marks the segments it did not write itself for zeroing.
                # (on the HOST -- the flags come from host metadata; a device all-reduce + read-back was a
                # full host-device synchronisation ahead of every replay)
                n = len(self.opt.arena.params)
                present = np.zeros(n, dtype=np.int32)
                present[act] = 1
                mdist.host_all_reduce_max(present)
                union = np.flatnonzero(present).astype(np.int64)
                absent_here = np.setdiff1d(union, act, assume_unique=False)
                act = union
            jobs, owner = self.opt.job_table(act, b1, b2, with_owner=True)
            if len(jobs) > self.max_jobs:
                raise _lib.HipLibraryError("engine: conditional job table overflow")
            if absent_here is not None and len(absent_here):
                jobs["reserved"][np.isin(act[owner], absent_here)] = 1  # zeroed ahead of the exchange
            self.n_exchange = 0
            if absent_here is not None:
                # the exchange moves the union's segments only: each job's place in the staging buffer, back to back in
                # units of 128 floats, rides in the upper bits of its `reserved` word (mmvae_jobs_pack)
                units = (jobs["len"].astype(np.int64) + 127) // 128
                pos = np.cumsum(units) - units
                self.exchange_floats = int(units.sum()) * 128
                if self.exchange_floats > self.staging.numel() or int(pos[-1] if len(pos) else 0) >= (1 << 29):
                    raise _lib.HipLibraryError("engine: conditional exchange staging overflow")
                jobs["reserved"] |= (pos.astype(np.int64) << 2).astype(np.int32)
                self.n_exchange = len(jobs)
            if absent_here is not None:
                # segments that stepped last time and do not now: zeroed once (the dense all-reduce of the arena would
                # otherwise sum their stale values on every step), skipped by the norm / Adam job kernels
                prev = getattr(self.eng, "_cond_prev_union", None)  # engine-wide: every plan steps the same VAE arena
                carry = np.empty(0, dtype=np.int64)
                if prev is not None:
                    retired = np.setdiff1d(prev, act)
                    if len(retired):
                        rj, r_owner = self.opt.job_table(retired, b1, b2, with_owner=True)
                        room = self.max_jobs - len(jobs)
                        if len(rj) > room:  # (another species' plan left more than this table holds: the rest next time)
                            fits = r_owner < (r_owner[room] if room > 0 else 0)
                            carry = retired[(r_owner[room] if room > 0 else 0):]
                            rj = rj[fits]
                        rj["reserved"] = 2
                        jobs = np.concatenate([jobs, rj])
                self.eng._cond_prev_union = np.union1d(act, carry)
            pack[self.idx_words:self.idx_words + 6 * len(jobs)] = jobs.view(np.int32)
            self._active = act
        self.ring.take()[:] = pack
        self.ring.upload(self.pack_dev)

    def commit(self) -> None:
        """The step ran: the tensors of its job table have taken one more step."""
        if not self.train:
            return
        steps = self.opt.host_steps()
        steps[self._active] += 1
