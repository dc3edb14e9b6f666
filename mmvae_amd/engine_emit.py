"""Program-building primitives of a step plan (mixin of mmvae_amd.engine._Plan): emitting launches into the current
segment, GEMM placement (grouped / side-branch / fused norm partials), fork / join edges of the captured graph, one
FCBlock layer forward / backward, deferred reductions, the fused clip + Adam of one optimiser (in order, overlapped or
sharded under data parallelism)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, dist as mdist
from .engine_common import ACC, NN, NT, RAW, RELU, SLACK, SQ_FUSED_SLOTS, TN, _LayerRef, _NOPL, _PlaneBuf, _p, _planes_desc, _s
from .optim import HipAdam


class PlanEmit:
    def _emit(self, fn, *args, probe=None):
        """probe: (tag, work) -- see _probed."""
        lib_fn = fn

        def call():
            rc = lib_fn(*args, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"{lib_fn.__name__} failed with code {rc}")

        self._cur.append(self._probed(probe[0], probe[1], call) if probe else call)

    def _mark_call(self, name: str):
        """Diagnostics (MMVAE_STAMPS=1): a marker launch that writes the device wall clock under `name` on whatever stream
        is current when it is enqueued; None otherwise.  tools/stamps_timeline.py prints the milestones of a replay."""
        if not self.eng.stamps:
            return None
        slot = len(self.stamp_names)
        self.stamp_names.append(name)
        buf = self.eng.buf("debug.stamps", (256,), torch.int64)
        lib = self.lib

        def call():
            lib.mmvae_debug_stamp(buf.data_ptr(), slot, _s())

        return call

    def _mark(self, name: str):
        c = self._mark_call(name)
        if c is not None:
            self._cur.append(c)

    def _probed(self, tag, work, call, **meta):
        """Measurement hook (bench.py's roofline leg): in an EAGER run with plan.probe set, `call` is bracketed by a
        timing event pair on the stream it launches on (e0 -> e1; e1 -> e2 is an empty pair: what one event marker costs
        there); never active under capture.  `work`: algorithmic FLOPs (or bytes) of the launch; `meta` (kernel name,
        workgroup cap, bound) is kept in plan.probe_meta[tag]."""
        if tag is None:
            return call
        plan = self
        self.probe_meta[tag] = dict(meta, work=work)

        def wrapped():
            pr = plan.probe
            if pr is None:
                return call()
            st = torch.cuda.current_stream()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record(st)
            call()
            e1.record(st)
            e2.record(st)
            pr.setdefault(tag, []).append((e0, e1, work, e2))

        return wrapped

    def _cut(self, marker):
        self._join()  # a captured segment may not end with work outstanding on the side branch
        self.segments.append(self._cur)
        self.segments.append(marker)
        self._cur = []

    def _host_marker(self, marker):
        """Cut the program here; `marker` -- ("lane", stream, items) or ("lane_join", stream) -- is acted on by the host
        between two captured segments (PlanRun._run_program)."""
        self._join()  # a captured segment may not end with work outstanding on a branch
        if self._cur:
            self.segments.append(self._cur)
            self._cur = []
        self.segments.append(marker)

    def slot(self, name: str) -> int:
        if name not in self.metric_slots:
            self.metric_slots[name] = 8 + sum(1 for v in self.metric_slots.values() if 8 <= v < 192)
            assert self.metric_slots[name] < 192
        return self.metric_slots[name]

    def mptr(self, name: str) -> int:
        return self.metrics.data_ptr() + 4 * self.slot(name)

    @staticmethod
    def kpad(rows: int) -> int:
        """K of a weight-gradient GEMM over `rows` batch rows: the next multiple of 32.  Both operands are engine
        buffers with 32 zero rows of slack behind them (StepEngine.buf), so the extra rows contribute exact zeros and
        the GEMM stays on the pipelined whole-k-tile kernel for any batch size."""
        return (rows + 31) // 32 * 32

    def _plan_gemm(self, layout, M, N, K):
        tile, sk = C.c_int(0), C.c_int(0)
        self.lib.mmvae_gemm_plan(layout, M, N, K, C.byref(tile), C.byref(sk))
        return sk.value

    def _fuse_sqnorm(self, layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags, side_cap: int = 0,
                     on_side: bool = True, planes=None, fork: bool = True, stream=None) -> bool:
        """Unsplit weight-gradient GEMM straight into a gradient arena: let its epilogue also leave the partial sums of
        squares of what it stores (mmvae_gemm_f32_sq), so that the clip's norm pass does not read the 82 MB back.  Only
        without a gradient exchange: under data parallelism the norm is that of the REDUCED gradients."""
        eng = self.eng
        if eng.overlap or eng.world > 1 or ldc != N or (flags & ~ACC):
            return False
        regular = int(_p(A) % 16 == 0 and _p(Bm) % 16 == 0 and lda % 4 == 0 and ldb % 4 == 0)
        self.lib.mmvae_gemm_set_workgroup_cap(side_cap)  # the tile (and so the partial count) is planned under the cap
        n_part = self.lib.mmvae_gemm_sq_partials(layout, M, N, K, regular)
        self.lib.mmvae_gemm_set_workgroup_cap(0)
        if n_part <= 0:
            return False
        hit = eng.locate_grad(Cm)
        if hit is None:
            return False
        opt, off = hit
        if opt.reducer is not None or (self.cond is not None and opt is self.opt_vae):
            return False
        buf = eng.sq_buffer(opt)
        base = self._sq_used.get(id(opt), 0)
        if base + n_part > SQ_FUSED_SLOTS:
            return False
        self._sq_used[id(opt)] = base + n_part
        self._sq_cover.setdefault(id(opt), []).append((off, M * N))
        plan = self

        ap, bp = (planes[0].args() if planes and planes[0] else _NOPL), (planes[1].args() if planes and planes[1] else _NOPL)
        tag, self._probe_next = self._probe_next, None

        def launch_gemm():
            if planes:
                rc = plan.lib.mmvae_gemm_planes_f32(layout, M, N, K, alpha, _p(A), lda, *ap, _p(Bm), ldb, *bp, _p(Cm), ldc,
                                                    _p(bias), flags | SLACK, 1, None, 0, buf.data_ptr() + 4 * base, n_part,
                                                    _s())
            else:
                rc = plan.lib.mmvae_gemm_f32_sq(layout, M, N, K, alpha, _p(A), lda, _p(Bm), ldb, _p(Cm), ldc, _p(bias),
                                                flags | SLACK, buf.data_ptr() + 4 * base, n_part, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"mmvae_gemm_f32_sq failed with code {rc} (layout {layout}, {M}x{N}x{K})")

        launch = self._probed(tag, 2.0 * M * N * K, launch_gemm, bound="mfma", cus=side_cap, planes=_planes_desc(planes),
                              shape=f"{('NT', 'NN', 'TN')[layout]} {M}x{N}x{K}")

        if side_cap:  # persistent grid capped to `side_cap` workgroups: the CUs left over serve another branch
            side = (stream if stream is not None else eng.side_stream) if on_side else None
            if on_side and fork:
                self._fork(side)

            def call():
                plan.lib.mmvae_gemm_set_workgroup_cap(side_cap)
                try:
                    if side is not None:
                        with torch.cuda.stream(side):
                            launch()
                    else:
                        launch()
                finally:
                    plan.lib.mmvae_gemm_set_workgroup_cap(0)
        else:
            call = launch
        self._cur.append(call)
        return True

    def _side_capped_gemm(self, layout, M, N, K, A, lda, Bm, ldb, Cm, ldc, cap: int, planes=None, flags: int = 0,
                          sk: int = 1, fork: bool = True, stream=None) -> None:
        """Unsplit GEMM on the side stream with its persistent grid capped to `cap` workgroups (no fused norm partials:
        under a gradient exchange the clip's norm is that of the REDUCED gradients); joined by the next cut / _join()."""
        plan = self
        ap, bp = (planes[0].args() if planes and planes[0] else _NOPL), (planes[1].args() if planes and planes[1] else _NOPL)
        tag, self._probe_next = self._probe_next, None

        def launch_gemm():
            if planes:
                rc = plan.lib.mmvae_gemm_planes_f32(layout, M, N, K, 1.0, _p(A), lda, *ap, _p(Bm), ldb, *bp, _p(Cm), ldc,
                                                    None, flags | SLACK, sk, None, 0, None, 0, _s())
            else:
                rc = plan.lib.mmvae_gemm_f32(layout, M, N, K, 1.0, _p(A), lda, _p(Bm), ldb, _p(Cm), ldc, None,
                                             flags | SLACK, sk, None, 0, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"capped side GEMM failed with code {rc} (layout {layout}, {M}x{N}x{K})")

        launch = self._probed(tag, 2.0 * M * N * K, launch_gemm, bound="mfma", cus=cap, planes=_planes_desc(planes),
                              shape=f"{('NT', 'NN', 'TN')[layout]} {M}x{N}x{K}" + (f" split-K {sk}" if sk > 1 else ""))
        side = stream if stream is not None else self.eng.side_stream
        if fork:
            self._fork(side)

        def call():
            plan.lib.mmvae_gemm_set_workgroup_cap(cap)
            try:
                with torch.cuda.stream(side):
                    launch()
            finally:
                plan.lib.mmvae_gemm_set_workgroup_cap(0)

        self._cur.append(call)

    def _queue_gemm(self, layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags) -> bool:
        """Weight-gradient GEMMs of the core layers (the planner's 64x64-tile class) are independent of each other and
        only feed the optimiser: queue them for ONE grouped launch (_flush_gemms) instead of a launch each."""
        tile, sk = C.c_int(0), C.c_int(0)
        self.lib.mmvae_gemm_plan(layout, M, N, K, C.byref(tile), C.byref(sk))
        if tile.value != 2:
            return False
        job = _lib.GemmJob(_p(A), _p(Bm), _p(Cm), _p(bias), lda, ldb, ldc, layout, M, N, K, float(alpha), int(flags), 0, 0)
        if not self.lib.mmvae_gemm_batch_job_ok(C.addressof(job)):
            return False
        if layout == TN and K >= 1024 and bias is None and not (flags & ~ACC):
            # K x B sample rows (K-sample programs: 5120; batches of 1024 cells): a grouped job runs its whole K in one
            # workgroup per 64 x 64 tile -- 160 k-tiles one after the other, 161 us at C3 for 2.5 GFLOP (C5's encoder side
            # at 1024 rows: 5.355 -> 5.326 ms).  Slices of 512 rows as jobs of their
            # own into slabs, summed by the deferred reduction that runs behind the grouped launch anyway.
            n_sl = (K + 511) // 512
            slabs = self.eng.buf(f"dwslabs.{self._next_defer_id()}", (n_sl, M, N))
            ok = True
            sub = []
            for i in range(n_sl):
                k0 = 512 * i
                kk = min(512, K - k0)
                j = _lib.GemmJob(_p(A) + 4 * k0 * lda, _p(Bm) + 4 * k0 * ldb, _p(slabs[i]), None, lda, ldb, N, layout, M, N,
                                 kk, 1.0, 0, 0, 0)
                ok = ok and bool(self.lib.mmvae_gemm_batch_job_ok(C.addressof(j)))
                sub.append(j)
            if ok:
                self._gemm_jobs.extend(sub)
                self._sum_keep.append((A, Bm, Cm, slabs))
                self._defer_sum(slabs, n_sl, M * N, M, N, N, Cm, ldc, alpha, flags & ACC)
                return True
        self._gemm_jobs.append(job)
        self._sum_keep.append((A, Bm, Cm, bias))
        return True

    def _gemm_group(self, jobs) -> bool:
        """Independent GEMMs of the planner's 64x64-tile class in ONE launch, in place (not deferred): the two heads of
        the encoder forward and backward.  jobs: (layout, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, alpha).  False
        (nothing emitted) when a job is not of that class."""
        arr = []
        for layout, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias, flags, alpha in jobs:
            tile, sk = C.c_int(0), C.c_int(0)
            self.lib.mmvae_gemm_plan(layout, M, N, K, C.byref(tile), C.byref(sk))
            job = _lib.GemmJob(_p(A), _p(Bm), _p(Cm), _p(bias), lda, ldb, ldc, layout, M, N, K, float(alpha), int(flags), 0, 0)
            if tile.value != 2 or not self.lib.mmvae_gemm_batch_job_ok(C.addressof(job)):
                return False
            arr.append(job)
            self._sum_keep.append((A, Bm, Cm, bias))
        table = (_lib.GemmJob * len(arr))(*arr)
        total = C.c_int(0)
        _lib.check(self.lib.mmvae_gemm_batch_prepare(len(arr), C.addressof(table), C.byref(total)), "mmvae_gemm_batch_prepare")
        jobs_dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.eng.device)
        self._job_tables.append(jobs_dev)
        self._emit(self.lib.mmvae_gemm_batch_f32, len(arr), jobs_dev.data_ptr(), total.value)
        return True

    def _flush_gemms(self):
        if not self._gemm_jobs:
            return
        n = len(self._gemm_jobs)
        arr = (_lib.GemmJob * n)(*self._gemm_jobs)
        total = C.c_int(0)
        _lib.check(self.lib.mmvae_gemm_batch_prepare(n, C.addressof(arr), C.byref(total)), "mmvae_gemm_batch_prepare")
        jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.eng.device)
        self._job_tables.append(jobs_dev)
        self._emit(self.lib.mmvae_gemm_batch_f32, n, jobs_dev.data_ptr(), total.value)
        self._gemm_jobs = []

    def gemm(self, layout, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias=None, flags=0, alpha=1.0, side=False, planes=None):
        """Complete GEMM (internal split-K reduce through a workspace when the plan asks for it).  side=True runs it
        on the engine's side stream (weight gradients: off the backward critical path) with its own workspace."""
        if side and self._queue_gemm(layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags):
            return
        sk = self._plan_gemm(layout, M, N, K)
        if side and sk > 1 and not (flags & ~ACC) and bias is None:
            # weight gradient with a split: raw slabs into a buffer of its own, summed later together with every other
            # pending reduction of the backward pass in ONE mmvae_sum_parts_batch launch (a reduce launch per GEMM is
            # ~5 us of pure launch cost)
            slabs = self.eng.buf(f"dwslabs.{self._next_defer_id()}", (sk, M, N))
            self._emit_gemm(layout, M, N, K, 1.0, A, lda, Bm, ldb, slabs, N, None, RAW, sk, False, planes=planes)
            self._defer_sum(slabs, sk, M * N, M, N, N, Cm, ldc, alpha, flags & ACC)
            return
        if side and sk == 1 and self._fuse_sqnorm(layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags, planes=planes):
            return
        nbytes = self.lib.mmvae_gemm_workspace_bytes(layout, M, N, K, sk)
        self._ws_bytes = max(self._ws_bytes, nbytes)
        self._emit_gemm(layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags, sk, True, planes=planes)

    def _edge(self, src, dst):
        """dst waits for everything enqueued so far on src (None = the current stream at run time): a graph edge under
        capture.  The event lives as long as the plan (torch's Stream.wait_stream would create one and drop it at once,
        in the middle of the capture: legal, but one variable less in a multi-stream capture on a runtime whose graph
        launches are fragile -- DESIGN.md section 4, "a runtime hazard")."""
        ev = torch.cuda.Event()
        self._events.append(ev)

        def call():
            s = src if src is not None else torch.cuda.current_stream()
            d = dst if dst is not None else torch.cuda.current_stream()
            ev.record(s)
            d.wait_event(ev)

        self._cur.append(call)

    def _next_x_split_job(self, l: _LayerRef, rows: int):
        """The next piece of the input batch's split for a layer whose tail is a column-kernel launch (fwd_layer's slab
        path), or None."""
        jobs = getattr(self, "_x_split_jobs", None)
        if not jobs:
            return None
        p_drop = l.p if self.mode == "train" else 0.0
        if l.bn is None and p_drop == 0 and self._plan_gemm(NT, rows, l.n_out, l.n_in) == 1:
            return None  # this layer's tail is fused into its GEMM
        return jobs.pop(0)

    def _fork(self, stream=None):
        """A branch stream (default: the side stream) waits for everything enqueued so far on the main stream."""
        side = stream if stream is not None else self.eng.side_stream
        self._edge(None, side)
        self._forked = True
        if side not in self._dirty:
            self._dirty.append(side)

    def _join(self, only=None):
        """Main stream waits for every branch with outstanding work (before the optimiser reads the gradient arenas);
        `only`: for that branch stream alone."""
        for side in self._dirty:
            if only is None or side is only:
                self._edge(side, None)
        self._dirty = [] if only is None else [s for s in self._dirty if s is not only]

    def _take(self, start: int) -> list:
        """Remove and return the calls emitted since position `start`."""
        calls = self._cur[start:]
        del self._cur[start:]
        return calls

    def _branch(self, stream, calls):
        """Run `calls` on `stream` as a branch of the captured graph, behind the last _fork(stream) (the point of the
        main stream it depends on) and joined by the next _join().  Emit it AFTER the main-stream work it should run
        beside: the graph executor enqueues in emission order, and a main-stream kernel enqueued behind a branch waited
        for the branch's node(s) ahead of it (profiles/r2_branch_order.txt).  The calls must not fork or join."""
        if not calls:
            return

        def call():
            with torch.cuda.stream(stream):
                for c in calls:
                    c()

        self._cur.append(call)
        if stream not in self._dirty:
            self._dirty.append(stream)

    def _next_defer_id(self) -> int:
        # position in this plan's program: the same geometry built again (another input pointer) shares the buffers
        self._defer_id = getattr(self, "_defer_id", 0) + 1
        return self._defer_id

    def _defer_sum(self, src, n_parts, part_stride, rows, cols, ld_src, dst, ld_dst, alpha=1.0, flags=0):
        """Queue dst[rows, cols] (+)= alpha * sum of n_parts partial results at src; see _flush_sums."""
        self._sum_jobs.append(_lib.SumJob(_p(src), _p(dst), part_stride, ld_src, ld_dst, n_parts, rows, cols, float(alpha),
                                          int(flags), 0))
        self._sum_keep.append((src, dst))

    def _flush_sums(self):
        """One launch for every reduction queued since the last flush (before anything reads those gradients)."""
        self._flush_gemms()
        if not self._sum_jobs:
            return
        arr = (_lib.SumJob * len(self._sum_jobs))(*self._sum_jobs)
        jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.eng.device)
        self._job_tables.append(jobs_dev)  # lives as long as the plan (the captured graph reads it on every replay)
        self._emit(self.lib.mmvae_sum_parts_batch, len(self._sum_jobs), jobs_dev.data_ptr(),
                   max(int(j.rows) * int(j.cols) for j in self._sum_jobs))
        self._sum_jobs = []

    def gemm_raw(self, layout, M, N, K, A, lda, Bm, ldb, planes=None) -> int:
        """Raw split-K slabs into the shared slab buffer; returns the slab count."""
        sk = self._plan_gemm(layout, M, N, K)
        self._slab_floats = max(self._slab_floats, sk * M * N)
        self._emit_gemm(layout, M, N, K, 1.0, A, lda, Bm, ldb, None, N, None, RAW, sk, False, planes=planes)
        return sk

    def _emit_gemm(self, layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags, sk, use_ws, planes=None):
        """planes: (A planes or None, B planes or None) -- pre-split forms of the operands (_PlaneBuf); the fp32 pointers
        stay the library's fallback for shapes the planes kernels do not take."""
        plan = self
        tag, self._probe_next = self._probe_next, None
        ap, bp = (planes[0].args() if planes and planes[0] else _NOPL), (planes[1].args() if planes and planes[1] else _NOPL)

        def launch_gemm():
            ws = plan.ws
            c_ptr = _p(Cm) if Cm is not None else plan.slab.data_ptr()
            if planes:
                rc = plan.lib.mmvae_gemm_planes_f32(layout, M, N, K, alpha, _p(A), lda, *ap, _p(Bm), ldb, *bp, c_ptr, ldc,
                                                    _p(bias), flags | SLACK, sk, ws.data_ptr() if use_ws else None,
                                                    ws.numel() * 4 if use_ws else 0, None, 0, _s())
            else:
                rc = plan.lib.mmvae_gemm_f32(layout, M, N, K, alpha, _p(A), lda, _p(Bm), ldb, c_ptr, ldc, _p(bias),
                                             flags | SLACK, sk, ws.data_ptr() if use_ws else None,
                                             ws.numel() * 4 if use_ws else 0, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"mmvae_gemm_f32 failed with code {rc} (layout {layout}, {M}x{N}x{K})")

        launch = self._probed(tag, 2.0 * M * N * K, launch_gemm, bound="mfma", cus=0, planes=_planes_desc(planes),
                              shape=f"{('NT', 'NN', 'TN')[layout]} {M}x{N}x{K}" + (f" split-K {sk}" if sk > 1 else ""))

        call = launch
        self._cur.append(call)

    # ---- one FCBlock layer forward: cur [rows, n_in] -> l.d
    def fwd_layer(self, tag: str, l: _LayerRef, cur: torch.Tensor, ld_cur: int, rows: int, training: bool = True,
                  mask_tag: Optional[str] = None, mask_stream: Optional[int] = None, planes_out: Optional[_PlaneBuf] = None,
                  split_job=None, slabs_from: Optional[torch.Tensor] = None):
        """`mask_tag`: name of the keep-mask buffer when it must differ from the layer's other buffers (the two phases
        of an adversary share activations but draw fresh masks); `mask_stream`: its Philox stream id.  `slabs_from`
        [S, rows, n_out]: the layer's split-K partial products already exist there (the previous step computed them
        beside its forward chain: StepEngine.training_step, "prefetch") -- no GEMM is emitted, the tail sums those."""
        eng = self.eng
        l.inp, l.ld_inp, l.rows = cur, ld_cur, rows
        l.d = eng.buf(f"{tag}.d", (rows, l.n_out))
        l.z = eng.buf(f"{tag}.z", (rows, l.n_out)) if l.bn is not None else None
        l.mean = eng.buf(f"{tag}.mean", (l.n_out,)) if l.bn is not None else None
        l.invstd = eng.buf(f"{tag}.invstd", (l.n_out,)) if l.bn is not None else None
        l.mask = eng.buf(f"{mask_tag or tag}.mask", (rows, l.n_out), torch.uint8) if (l.p > 0 and training) else None
        if l.mask is not None:
            self._mask_layers.append((l, mask_stream if mask_stream is not None else len(self._mask_layers)))
        l.a = eng.buf(f"{tag}.a", (rows, l.n_out)) if (l.p > 0 and l.return_hidden and training) else None
        l.dz = eng.buf(f"{tag}.dz", (rows, l.n_out)) if training else None
        p_drop = l.p if training else 0.0
        self._fcws_bytes = max(getattr(self, "_fcws_bytes", 0), self.lib.mmvae_fc_workspace_bytes(rows, l.n_out))
        sk = self._plan_gemm(NT, rows, l.n_out, l.n_in)
        if l.bn is None and p_drop == 0 and sk == 1:
            self.gemm(NT, rows, l.n_out, l.n_in, cur, ld_cur, l.W, l.n_in, l.d, l.n_out, bias=l.b,
                      flags=RELU if l.relu else 0)
            if planes_out is not None:  # no column kernel behind this GEMM: a split pass of its own
                self._emit(self.lib.mmvae_split_planes_f32, rows, l.n_out, _p(l.d), l.n_out, *planes_out.args())
            return l.d
        if slabs_from is not None:
            assert tuple(slabs_from.shape) == (sk, rows, l.n_out) and slabs_from.is_contiguous()
            S = sk
        else:
            S = self.gemm_raw(NT, rows, l.n_out, l.n_in, cur, ld_cur, l.W, l.n_in)
        bnp = None
        if l.bn is not None:
            bn = l.bn
            bnp = _lib.BnParams(_p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                _p(bn.num_batches_tracked), float(bn.momentum), float(bn.eps))
            l._bnp = bnp  # keep the struct alive for the lifetime of the plan
        plan = self

        def call():
            args = (rows, l.n_out, slabs_from.data_ptr() if slabs_from is not None else plan.slab.data_ptr(), l.n_out, S, _p(l.b),
                    C.byref(bnp) if bnp is not None else None, int(training), int(l.relu),
                    _p(l.mask), p_drop, _p(l.z), _p(l.a), _p(l.d), l.n_out, _p(l.mean), _p(l.invstd),
                    plan.fcws.data_ptr(), plan.fcws.numel() * 4)
            if split_job is not None:  # extra workgroups of the tail split an unrelated matrix (the input batch)
                rc = plan.lib.mmvae_fc_epilogue_fwd_split(*args, *split_job, _s())
            elif planes_out is not None:  # the layer tail also leaves the bf16 planes of its output
                rc = plan.lib.mmvae_fc_epilogue_fwd_planes(*args, *planes_out.args(), _s())
            else:
                rc = plan.lib.mmvae_fc_epilogue_fwd(*args, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"mmvae_fc_epilogue_fwd failed with code {rc}")

        self._cur.append(call)
        return l.d

    # ---- one layer backward.  din: tensor [rows, n_out] or None (= shared slab buffer holding S_in raw slabs)
    def bwd_layer(self, l: _LayerRef, din, S_in: int, addend=None, need_dx: str = "raw", dx_out=None, dx_flags=0,
                  dx_alpha=1.0, dz_planes: Optional[_PlaneBuf] = None, inp_planes: Optional[_PlaneBuf] = None,
                  row_scale=None):
        """dz_planes / inp_planes: pre-split forms of this layer's output gradient (written by its column kernel) and of
        its input -- both operands of its weight-gradient GEMM.  row_scale: per-row factor of the incoming gradient
        (the K-sample bound's weights, applied where the slabs are summed)."""
        rows = l.rows
        dw_planes = (dz_planes, inp_planes) if (dz_planes is not None and (inp_planes is not None or
                                                                         getattr(self, "x_fp32_dw1", False))) else None
        plan = self
        relu_src = l.a if l.a is not None else l.d
        has_bn = l.bn is not None

        own_ws = None
        if not has_bn and l.gb is not None:
            own_ws = self._bias_partials(rows, l.n_out, l.gb)

        def call():
            din_ptr = _p(din) if din is not None else plan.slab.data_ptr()
            ws = own_ws if own_ws is not None else plan.fcws
            # `addend` is a gradient on the hidden representation = the activation BEFORE dropout: it bypasses the mask
            args = (rows, l.n_out, din_ptr, l.n_out, S_in, None, _p(addend), _p(row_scale), _p(l.mask), l.p, int(l.relu),
                    _p(relu_src) if l.relu else None, _p(l.z), _p(l.bn.weight) if has_bn else None, _p(l.mean),
                    _p(l.invstd), int(has_bn), _p(l.dz), l.n_out, _p(l.gb) if own_ws is None else None,
                    _p(l.ggamma) if has_bn else None, _p(l.gbeta) if has_bn else None, ws.data_ptr(), ws.numel() * 4)
            if dw_planes is not None:
                rc = plan.lib.mmvae_fc_epilogue_bwd_planes(*args, *dz_planes.args(), _s())
            else:
                rc = plan.lib.mmvae_fc_epilogue_bwd(*args, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"mmvae_fc_epilogue_bwd failed with code {rc}")

        self._cur.append(call)
        # dW[n_out, n_in] = dz^T[n_out, rows] . inp[rows, n_in]  -> straight into the gradient arena
        # (an adversary reading the first of K > 1 samples: the rows behind its B input rows are the next sample, not slack)
        k_rows = rows if (self.K > 1 and l.inp is self.z and rows != self.R) else self.kpad(rows)
        if getattr(self, "_defer_next_dw", False):
            # (side-branch mode) the first layer's chip-filling weight gradient is emitted behind the shared VAE's
            # optimiser: by then the decoder's weight gradient on the side branch has released its CUs
            self._deferred_dw = (TN, l.n_out, l.n_in, k_rows, l.dz, l.n_out, l.inp, l.ld_inp, l.gW, l.n_in)
            self._deferred_dw_planes = dw_planes
            self._defer_next_dw = False
        else:
            big_first = l is self.enc_layers[0] and 2.0 * l.n_out * l.n_in * k_rows >= 5e9
            self._probe_next = "enc_l1_dw" if big_first else None
            self.gemm(TN, l.n_out, l.n_in, k_rows, l.dz, l.n_out, l.inp, l.ld_inp, l.gW, l.n_in, side=True, planes=dw_planes)
            self._probe_next = None
        if need_dx == "raw":
            return self.gemm_raw(NN, rows, l.n_in, l.n_out, l.dz, l.n_out, l.W, l.n_in)
        if need_dx == "full":
            # a small complete product: one grouped launch without split-K instead of slabs + a reduction launch
            if not self._gemm_group([(NN, rows, l.n_in, l.n_out, l.dz, l.n_out, l.W, l.n_in, dx_out, l.n_in, None,
                                      dx_flags, dx_alpha)]):
                self.gemm(NN, rows, l.n_in, l.n_out, l.dz, l.n_out, l.W, l.n_in, dx_out, l.n_in, flags=dx_flags,
                          alpha=dx_alpha)
        return 0

    def _bias_partials(self, rows, N, dbias):
        """A [ceil(rows/32), N] partial-column-sum buffer of its own for one layer + the deferred sum into dbias."""
        RC = (rows + 31) // 32
        nfl = max(self.lib.mmvae_fc_workspace_bytes(rows, N) // 4, RC * N)
        ws = self.eng.buf(f"biasparts.{self._next_defer_id()}", (nfl,))
        self._defer_sum(ws, RC, N, 1, N, N, dbias, N)
        return ws

    def _emit_colsum_pair(self, B, N, pair, dbias0, dbias1):
        """Column sums of two stacked [B, N] matrices (pair: [2, B, N], B a multiple of the 32-row chunk) in one pass;
        the two halves of the chunk partials are summed into dbias0 / dbias1 by the deferred reduction."""
        plan = self
        rows = 2 * B
        RC = rows // 32
        self._fcws_bytes = max(getattr(self, "_fcws_bytes", 0), self.lib.mmvae_fc_workspace_bytes(rows, N))
        ws = self.eng.buf(f"biasparts.{self._next_defer_id()}",
                          (max(self.lib.mmvae_fc_workspace_bytes(rows, N) // 4, RC * N),))
        self._defer_sum(ws, RC // 2, N, 1, N, N, dbias0, N)
        self._defer_sum(ws[(RC // 2) * N:], RC // 2, N, 1, N, N, dbias1, N)

        def call():
            rc = plan.lib.mmvae_fc_epilogue_bwd(rows, N, _p(pair), N, 1, None, None, None, None, 0.0, 0, None, None,
                                                None, None, None, 0, None, N, None, None, None, ws.data_ptr(),
                                                ws.numel() * 4, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"mmvae_fc_epilogue_bwd (stacked column sums) failed with code {rc}")

        self._cur.append(call)

    def _emit_fc_bwd(self, rows, N, din, addend, row_scale, dz_out, dbias):
        """Plain (no BN / ReLU / mask) column pass: dz = row_scale * (din + addend) (optional), dbias = column sums."""
        plan = self
        self._fcws_bytes = max(getattr(self, "_fcws_bytes", 0), self.lib.mmvae_fc_workspace_bytes(rows, N))
        own_ws = self._bias_partials(rows, N, dbias) if dbias is not None else None

        def call():
            ws = own_ws if own_ws is not None else plan.fcws
            rc = plan.lib.mmvae_fc_epilogue_bwd(rows, N, _p(din), N, 1, _p(addend), None, _p(row_scale), None, 0.0, 0, None, None,
                                                None, None, None, 0, _p(dz_out), N, _p(dbias) if own_ws is None else None,
                                                None, None, ws.data_ptr(), ws.numel() * 4, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"mmvae_fc_epilogue_bwd (column sum) failed with code {rc}")

        self._cur.append(call)

    def optimizer(self, opt: HipAdam, max_norm: float, advance: bool = True, step: bool = True, exchange: str = "inline",
                  join: bool = True, tail_copy=None):
        """Fused clip + Adam over one optimiser's arenas.  `exchange` places the gradient all-reduce under data
        parallelism: "inline" (here, on the main stream), "wait" (it was begun earlier with _begin_exchange; the main
        stream joins it here) or "deferred" (it and everything after it run on the communication stream, overlapped
        with the next step).  `join=False`: none of this optimiser's gradients come from the side branch."""
        if join:
            self._join()
        self._flush_sums()
        a = opt.arena
        g = opt.param_groups[0]
        b1, b2 = g["betas"]
        gs = 1.0 / self.eng.world
        npart = self.lib.mmvae_sqnorm_partials(a.numel)
        sh = a.shard(self.eng.shard_sim_world or self.eng.world, mdist.rank()) if (
            self.eng.shard and opt is self.opt_exp and self.cond is None
                                                       and exchange in ("inline", "deferred")
                                                       and opt.reducer is not None) else None
        if sh is not None:
            return self._optimizer_sharded(opt, sh, max_norm, advance, step, exchange)
        if (opt.reducer is not None or self.eng.overlap) and exchange != "done":  # ("done": the caller has cut already)
            self._cut(("ar_" + exchange, opt))
        if self.cond is not None and opt is self.opt_vae:
            # only the tensors that took part: the dense parameters + the condition blocks present in the batch, from
            # the job table uploaded for this step (fixed launch size, empty jobs return at once)
            c = self.cond
            self._emit(self.lib.mmvae_grad_sqnorm_jobs, c.max_jobs, c.jobs_ptr, _p(a.grad), _p(c.partials))
            flags = _lib.PREPARE_NORM | (_lib.PREPARE_ADVANCE if (advance and step) else 0)
            self._emit(self.lib.mmvae_adam_prepare, c.max_jobs, _p(c.partials), max_norm, gs, b1, b2, _p(opt.state_dev), flags)
            if step:
                self._emit(self.lib.mmvae_adam_step_jobs, c.max_jobs, c.jobs_ptr, _p(a.data), _p(a.grad), _p(a.exp_avg),
                           _p(a.exp_avg_sq), _p(opt.state_dev), g["lr"], b1, b2, g["eps"], g["weight_decay"], gs)
            return
        cover = sorted(self._sq_cover.pop(id(opt), []))
        flags = _lib.PREPARE_NORM | (_lib.PREPARE_ADVANCE if (advance and step) else 0)
        # ranges of the arena the norm pass still has to read: everything no fused GEMM epilogue has covered (those have
        # left their partials in the first slots of the buffer); adam_prepare sums all partials (fp64, slot order)
        ranges, pos = [], 0
        for off, n in cover + [(a.numel, 0)]:
            if off > pos:
                ranges.append((pos, off - pos))
            pos = max(pos, off + n)
        buf = self.eng.sq_buffer(opt) if cover else opt.partials
        slot = self._sq_used.pop(id(opt)) if cover else 0
        nparts = [self.lib.mmvae_sqnorm_partials(n) for _, n in ranges]
        npart = slot + sum(nparts)
        assert npart <= buf.numel()
        if 1 <= len(ranges) <= 4 and opt.reducer is None and not self.eng.overlap:
            # one launch: the ranges' partials + (last workgroup to finish) the fp64 sum, clip coefficient, step count
            gp = (C.c_void_p * len(ranges))(*[a.grad.data_ptr() + 4 * o for o, _ in ranges])
            ln = (C.c_int64 * len(ranges))(*[n for _, n in ranges])
            ticket = self.eng.buf(f"sqticket.{id(opt)}", (1,), torch.int32)
            self._sum_keep.append((gp, ln, ticket))
            self._emit(self.lib.mmvae_grad_sqnorm_ranges_prepare, len(ranges), C.addressof(gp), C.addressof(ln),
                       buf.data_ptr() + 4 * slot, _p(ticket), npart, _p(buf), max_norm, gs, b1, b2, _p(opt.state_dev), flags)
        else:
            for (o, n), k in zip(ranges, nparts):
                self._emit(self.lib.mmvae_grad_sqnorm, n, a.grad.data_ptr() + 4 * o, buf.data_ptr() + 4 * slot)
                slot += k
            self._emit(self.lib.mmvae_adam_prepare, npart, _p(buf), max_norm, gs, b1, b2, _p(opt.state_dev), flags)
        pr = ("adam_expert", 28.0 * a.numel) if opt is self.opt_exp else None  # bytes: p, g, m, v read; p, m, v written
        if step and tail_copy is not None:  # (n, src, dst): the step's logged scalars ride on this launch
            self._emit(self.lib.mmvae_adam_step_copy, a.numel, _p(a.data), _p(a.grad), _p(a.exp_avg), _p(a.exp_avg_sq),
                       _p(opt.state_dev), g["lr"], b1, b2, g["eps"], g["weight_decay"], gs, tail_copy[0],
                       _p(tail_copy[1]), _p(tail_copy[2]), probe=pr)
        elif step:
            self._emit(self.lib.mmvae_adam_step, a.numel, _p(a.data), _p(a.grad), _p(a.exp_avg), _p(a.exp_avg_sq),
                       _p(opt.state_dev), g["lr"], b1, b2, g["eps"], g["weight_decay"], gs, probe=pr)
        if step and pr:
            self.probe_meta["adam_expert"].update(bound="hbm", cus=0, shape=f"{a.numel} parameters, 28 B each")

    def _optimizer_sharded(self, opt: HipAdam, sh, max_norm, advance, step, exchange):
        """The expert's update under data parallelism, sharded ("prefer direct reduce-scatter + all-gather"):
        reduce-scatter of the gradient arena -> sum of squares of this rank's slice, all-gathered (world floats; every
        rank sums them in rank order: identical norms) -> clip + Adam on the slice -> all-gather of the parameters.
        Replicas stay bit-identical: every parameter is computed once, by its owner."""
        a, g, lib = opt.arena, opt.param_groups[0], self.lib
        b1, b2 = g["betas"]
        gs = 1.0 / self.eng.world
        per, lo, n_loc = sh
        W = self.eng.world
        sim = bool(self.eng.shard_sim_world)
        mine = self.eng.buf(f"shard.sq.{id(opt)}", (1,))
        allsq = self.eng.buf(f"shard.allsq.{id(opt)}", (W,))
        self.shard_info = dict(per=per, lo=lo, n_loc=n_loc, mine=mine, allsq=allsq, sim=sim)
        opt.sharded = not sim  # (the timing diagnostics of --sim-world exchange nothing: there is nothing to gather back)
        self._cut(("rs_" + exchange, opt))
        if n_loc > 0:
            np_loc = int(lib.mmvae_sqnorm_partials(n_loc))
            parts = self.eng.buf(f"shard.parts.{id(opt)}", (np_loc,))
            self._emit(lib.mmvae_grad_sqnorm, n_loc, a.grad.data_ptr() + 4 * lo, _p(parts))
            self._emit(lib.mmvae_sum_f32, np_loc, _p(parts), _p(mine), 0)
        else:  # (more ranks than 4-element groups: this rank owns nothing)
            self._emit(lib.mmvae_axpby, 1, 0.0, _p(mine), 0.0, _p(mine))
        self._cut(("ag_norm", opt))
        flags = _lib.PREPARE_NORM | (_lib.PREPARE_ADVANCE if (advance and step) else 0)
        self._emit(lib.mmvae_adam_prepare, W, _p(allsq), max_norm, gs, b1, b2, _p(opt.state_dev), flags)
        if step and n_loc > 0:
            self._emit(lib.mmvae_adam_step, n_loc, a.data.data_ptr() + 4 * lo, a.grad.data_ptr() + 4 * lo,
                       a.exp_avg.data_ptr() + 4 * lo, a.exp_avg_sq.data_ptr() + 4 * lo, _p(opt.state_dev), g["lr"], b1, b2,
                       g["eps"], g["weight_decay"], gs, probe=("adam_expert", 28.0 * n_loc))
            self.probe_meta["adam_expert"].update(bound="hbm", cus=0, shape=f"{n_loc} parameters (1/{W} of the arena), 28 B each")
        if step:
            self._cut(("ag_params", opt))

    def _begin_exchange(self, opt: HipAdam):
        """All gradients of `opt` are final here: start their all-reduce on the small-message stream."""
        self._flush_sums()
        self._cut(("ar_begin", opt))

    def copy_scalar(self, src_ptr: int, dst_name: str):
        self._emit(self.lib.mmvae_axpby, 1, 1.0, src_ptr, 0.0, self.mptr(dst_name))

    def log_norm(self, opt: HipAdam, name: str, final: bool = True):
        """The pre-clip gradient norm `opt` has just computed, under metric `name`.  Its state word lives in the metrics
        buffer (StepEngine.__init__): when it is not overwritten again within the step (`final`), the metric is that
        word itself -- no launch; otherwise (discriminator phase: the generator phase reuses the optimiser) it is copied."""
        base = self.eng._state_slot.get(id(opt))
        if base is not None and final:
            self.metric_slots[name] = base + 1
        else:
            self.copy_scalar(opt.state_dev.data_ptr() + 4, name)

    # ---------------------------------------------------------------------------------------------------- build
