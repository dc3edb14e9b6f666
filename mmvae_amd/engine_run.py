"""Running a step plan (mixin of mmvae_amd.engine._Plan): explicit noise, the exchange points between captured
segments (all-reduce / reduce-scatter / all-gather placement), eager and replayed execution, release, logging."""
from __future__ import annotations

import torch

from . import _lib, dist as mdist
from .constants import REGISTRY_KEYS as RK
from .engine_common import _p, _s


class PlanRun:
    def load_explicit_noise(self, enc_mod, expert):
        """Parity mode: eps and every dropout keep mask of the program come from the caller -- `explicit_eps` of the
        encoder, `explicit_masks[layer index]` of each FCBlock with dropout (expert encoder, VAE blocks, adversary
        encoders: one mask per layer, used by both adversarial phases, like the module path)."""
        eps = enc_mod.explicit_eps
        if eps is not None:
            self.eps.copy_(eps.reshape(self.eps.shape))
        for l, _ in self._mask_layers:
            masks = (l.block.explicit_masks or {}) if l.block is not None else {}
            if l.index not in masks:
                raise KeyError(f"explicit noise mode: no keep mask for layer {l.index} of a {l.n_in}->{l.n_out} block "
                               "with dropout (set explicit_masks on that FCBlock)")
            l.mask.copy_(masks[l.index])

    def _exchange(self, marker, tail):
        """One data-parallel exchange point between two captured segments.  Returns the stream the rest of the
        program runs on (None = stay on the main stream)."""
        kind, opt = marker
        if kind == "ar_many":  # several small arenas at one exchange point (the adversaries of a phase)
            for o in opt:
                if o.reducer is not None:
                    o.reducer.reduce_here(o.arena.grad, small=True)
            return tail
        red = opt.reducer
        eng = self.eng
        main = torch.cuda.current_stream()

        def reduce_small():
            c = self.cond
            if c is not None and opt is self.opt_vae and c.n_exchange:
                # conditional layers: only the union's segments travel (pack -> all-reduce -> unpack, on this stream)
                lib, st = self.lib, c.staging
                _lib.check(lib.mmvae_jobs_pack(c.n_exchange, c.jobs_ptr, _p(opt.arena.grad), _p(st), _s()), "mmvae_jobs_pack")
                red.reduce_here(st[: c.exchange_floats], small=True)
                _lib.check(lib.mmvae_jobs_unpack(c.n_exchange, c.jobs_ptr, _p(opt.arena.grad), _p(st), _s()), "mmvae_jobs_unpack")
            else:
                red.reduce_here(opt.arena.grad, small=True)

        if kind == "ar_inline":
            if red is not None:
                reduce_small()
        elif kind == "ar_begin":
            eng.small_stream.wait_stream(main)
            if red is not None:
                with torch.cuda.stream(eng.small_stream):
                    reduce_small()
        elif kind == "ar_wait":
            main.wait_stream(eng.small_stream)
        elif kind == "ar_deferred":
            eng.comm_stream.wait_stream(main)
            if red is not None:
                with torch.cuda.stream(eng.comm_stream):
                    red.reduce_here(opt.arena.grad)
            return eng.comm_stream
        elif kind in ("rs_inline", "rs_deferred", "ag_norm", "ag_params"):
            import contextlib
            import torch.distributed as tdist

            si, a, W = self.shard_info, opt.arena, eng.world
            if kind == "rs_deferred":
                eng.comm_stream.wait_stream(main)
                tail = eng.comm_stream
            live = red is not None and not mdist.DRY_RUN and not si["sim"]
            with (torch.cuda.stream(tail) if tail is not None else contextlib.nullcontext()):
                if kind.startswith("rs_"):
                    full = a.grad_full[:si["per"] * W]
                    if live:
                        tdist.reduce_scatter_tensor(full[si["lo"]:si["lo"] + si["per"]], full, op=tdist.ReduceOp.SUM,
                                                    group=red.group)
                elif kind == "ag_norm":
                    if live:
                        tdist.all_gather_into_tensor(si["allsq"], si["mine"], group=red.group)
                    else:  # (no peers to hear from: the slice's own sum; adam_prepare adds up all `world` words)
                        si["allsq"].zero_()
                        si["allsq"][:1].copy_(si["mine"])
                elif live:
                    full = a.data_full[:si["per"] * W]
                    tdist.all_gather_into_tensor(full, full[si["lo"]:si["lo"] + si["per"]], group=red.group)
        else:
            raise _lib.HipLibraryError(f"engine: unknown exchange marker {kind}")
        return tail

    def _run_program(self, items, launch):
        tail = None  # once set, the rest of the program (the deferred update) runs on the communication stream
        lane = None  # a lane whose fork point has been recorded and whose items are not enqueued yet

        def enqueue_lane():
            nonlocal lane
            if lane is not None:
                with torch.cuda.stream(lane[1]):
                    self._run_program(lane[2], launch)
                lane = None

        for idx, it in enumerate(items):
            if isinstance(it, tuple) and it[0] == "lane":
                # A section of the program (segments + exchange points) as a lane of its own on a branch stream, ordered
                # behind everything enqueued so far on the main stream.  Its items are enqueued BEHIND the main stream's
                # next segment: the exchange programs run with the host only just ahead of the GPU (a dozen replays and
                # eight collectives per step), and a main stream left empty while the host enqueues the lane's ~0.4 ms
                # of launches stalls for exactly that long (untraced markers: reconstruction done 518 us behind the
                # fork point instead of 144).
                enqueue_lane()
                it[1].wait_stream(torch.cuda.current_stream())
                lane = it
            elif isinstance(it, tuple) and it[0] == "lane_join":
                enqueue_lane()
                torch.cuda.current_stream().wait_stream(it[1])
            elif isinstance(it, tuple):
                tail = self._exchange(it, tail)
            elif tail is None:
                launch(it)
                enqueue_lane()
            else:
                with torch.cuda.stream(tail):
                    launch(it)
        enqueue_lane()
        if tail is None:
            return None
        with torch.cuda.stream(tail):
            self.exp_norm_log.copy_(self.opt_exp.state_dev[1:2])
            ev = torch.cuda.Event()
            ev.record(tail)
        return ev

    def release(self):
        """Destroy the captured graphs and drop the program's closures (which reference the plan: a cycle only the
        cyclic collector would free).  A graph with forked branches owns runtime-internal streams, and a process that
        piled up dozens of such executables (a test session; plans rebuilt after every settings change) crashed inside
        hipGraphLaunch on some boxes.  The caller has synchronised the device."""
        def reset(items):
            for g in items:
                if not isinstance(g, tuple):
                    g.reset()
                elif g[0] == "lane":
                    reset(g[2])

        reset(self._graphs or [])
        self._graphs = None
        self.segments = []
        self._cur = []
        self._events = []

    @staticmethod
    def _launch_eager(seg):
        for call in seg:
            call()

    def run(self):
        """One step.  Returns the event behind the deferred expert update (overlapped data parallelism) or None."""
        self._runs += 1
        if self._runs == 1 or not self.eng.settings.graphs or self.eng.eager_only:
            # a real step; the first run also loads every code object before capture
            return self._run_program(self.segments, self._launch_eager)
        if self._graphs is None:
            torch.cuda.synchronize()
            def capture(segments):
                graphs = []
                for seg in segments:
                    if isinstance(seg, tuple) and seg[0] == "lane":
                        graphs.append(("lane", seg[1], capture(seg[2])))
                        continue
                    if isinstance(seg, tuple):
                        graphs.append(seg)
                        continue
                    if not seg:  # (nothing between two exchange points)
                        continue
                    g = torch.cuda.CUDAGraph()
                    # thread-local capture mode: a process group's watchdog thread may touch its events meanwhile
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for call in seg:
                            call()
                    graphs.append(g)
                return graphs

            self._graphs = capture(self.segments)
        return self._run_program(self._graphs, lambda g: g.replay())

    def log(self, model, eid: str):
        """Log the step's scalars: views of the plan's log buffer (filled by the captured program itself: no copy, no host
        sync).  The views and tagged dictionaries are built once per (stage, expert): ~25 tensor views per step were a
        tenth of the host's share of an adversarial step."""
        key = (model.stage_name, eid)
        cache = self.__dict__.setdefault("_log_cache", {})
        calls = cache.get(key)
        if calls is None:
            calls = cache[key] = self._log_calls(model, eid)
        for kind, a, b in calls:
            if kind == "auto":
                model.auto_log(a, **b)
            else:
                model.log(a, b)

    def _log_calls(self, model, eid: str):
        m = self.log_buf
        stage = model.stage_name
        calls = []
        main = {RK.LOSS: m[self.slot("total_loss")], RK.RECON_LOSS: m[1], RK.KL_LOSS: m[2], RK.KL_WEIGHT: m[3],
                "Mean": m[4], "Variance": m[5]}
        for i in range(1, getattr(self, "n_adv", 0) + 1):
            for phase in ("discriminator", "generator"):
                tags = [f"{phase}_{i}", stage, eid, RK.ADV_LOSS]
                for c in self.conditions + ["summed"]:
                    calls.append(("auto", {c: m[self.slot(f"{phase}_{i}/{c}")]}, dict(tags=tags, key_pos="last")))
                calls.append(("log", f"grad_norms/{phase}_{i}", m[self.slot(f"grad_norms/{phase}_{i}")]))
        calls.append(("log", "grad_norms/vae", m[self.slot("grad_norms/vae")]))
        # overlapped mode: the norm is produced on the communication stream; the logged tensor is filled when that
        # stream gets there (read it after engine.flush() / a device synchronisation)
        calls.append(("log", f"grad_norms/expert_{eid}", self.exp_norm_log[0] if self.exp_norm_log is not None
                      else m[self.slot("grad_norms/expert")]))
        calls.append(("auto", main, dict(tags=[stage, eid])))
        return calls
