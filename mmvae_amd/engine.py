"""Graph-captured training-step engine: the measured hot path.

One MMVAE training step (reference `CMMVAEModel.training_step`, models/cmmvae_model.py:138-217, with
`gradient_reversal_domain_classifier` :103-136 and `BaseVAE.elbo` modules/vae.py:136-152) is compiled, per expert, into
a FIXED program of libmmvae_hip.so launches over pre-allocated HBM buffers:

    [Philox masks / eps] -> expert encoder (GEMM + fused BN/ReLU/dropout column kernels, split-K slabs summed in the
    epilogue) -> VAE encoder -> mean/var heads -> fused reparameterise+KL -> VAE decoder -> expert decoder, last layer
    fused with the squared-error / dP epilogue -> ELBO finalise -> [adversarial D phase: fwd, CE, bwd, clip+Adam;
    G phase behind gradient reversal] -> backward GEMMs writing weight gradients STRAIGHT into the optimiser's flat
    gradient arena -> fused global-norm clip + Adam over the arenas.

The program is run eagerly once (a real step; loads the code objects), then captured into a hipGraph and replayed:
no tracing compiler, no per-step allocation, no host read-back (loss scalars stay in a device metrics buffer).
Per-step host values (KL weight) live in device scalars the kernels read, so the captured graph stays valid.
Under data parallelism the program is cut at the gradient all-reduce points (RCCL over the flat arenas) into several
graphs.  Conditional layers (CLVAE) run inside the program through the grouped kernels: the per-cell
condition indices, the per-condition row groups and the optimiser's job table of the blocks that took part are
per-step HOST values, uploaded into static device tables before the replay (CondProgram).  Configurations outside this
shape (LayerNorm / non-ReLU activations in the FC blocks, conditional blocks that are not one Linear, conditional
layers under data parallelism) use the module path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib, dist as mdist, rng
from .constants import REGISTRY_KEYS as RK
from .engine_adv import PlanAdversaries
from .engine_common import ACC, NN, NT, SQ_FUSED_SLOTS, TN, _LayerRef, _PlaneBuf, _p, _supported_block
from .engine_cond import CondProgram
from .engine_emit import PlanEmit
from .engine_run import PlanRun
from .modules.base.components import Adversarial, FCBlock, _identity
from .optim import HipAdam

MAX_POINTER_PLANS = 8
SIDE_MAX_ROWS = 640  # the forked programs were measured up to this many rows (see _Plan._build)
ADV_DW_CAP = 170     # adversarial programs: workgroup cap of the decoder's weight gradient on its late branch (_Plan._build)


@dataclass(frozen=True)
class EngineSettings:
    """Every behavioural switch of the step engine, read ONCE from the environment when the engine is built (DESIGN.md
    section 10 lists them with their measurements).  Defaults are the measured-best values."""
    conditionals: bool = True    # MMVAE_ENGINE_CONDITIONALS=0: models with conditional layers take the module path
    graphs: bool = True          # MMVAE_NO_GRAPH=1: every step eagerly (diagnostics)
    planes: bool = True          # MMVAE_PLANES=0: fp32 operands in every G-wide GEMM (no pre-split bf16 planes)
    side_dw: int = 125           # MMVAE_SIDE_DW: workgroup cap of the decoder's weight gradient on its side branch; 0 = one stream
    side_dw2: int = 185          # MMVAE_SIDE_DW2: cap of the encoder's weight gradient beside the shared VAE's optimiser
    side_dw_dp: int = 125        # MMVAE_SIDE_DW_DP: that branch inside the exchange (data-parallel) program; 0 = in order
    side_dw_any: bool = False    # MMVAE_SIDE_DW_ANY=1: fork outside the measured geometry too
    prefetch_adv: int = 86       # MMVAE_PREFETCH_ADV: its cap in adversarial programs (second branch stream, beside the adversaries' lane: 3 rounds of the 256 work items; C4 1.091 -> 1.070 ms, 128: 1.083, 64: 1.094); 0 = off
    prefetch_join: bool = False  # MMVAE_PREFETCH_JOIN=1: join that product ahead of the reconstruction launch (diagnostics)
    cond_batched: bool = True    # MMVAE_COND_BATCHED=0: conditional layers of a "parallel" selection order one launch per position
    x_planes: bool = False       # MMVAE_X_PLANES=1: the batch pre-split for the first layer's weight gradient even where that GEMM can read it as fp32 (see StepEngine._x_planes)
    prefetch: int = 128          # MMVAE_PREFETCH: workgroup cap of the NEXT step's first forward GEMM beside this step's forward chain (software pipelining across steps, needs the caller's hint); 0 = off
    adv_fused: bool = True       # MMVAE_ADV_FUSED=0: the per-layer adversary program (the path of adversaries with BatchNorm)
    adv_aside: int = 2           # MMVAE_ADV_ASIDE: 0 the fused adversary passes in order; 1 on the branch stream; 2 + the decoder's weight gradient on a second branch from where the first is joined
    dp_overlap: Optional[bool] = None  # MMVAE_DP_OVERLAP: None = overlapped exchange whenever gradients are exchanged
    dp_shard: bool = True        # MMVAE_DP_SHARD=0: all-reduce + full update instead of the sharded expert update
    dp_kernels: str = "auto"     # MMVAE_DP_KERNELS: dynamic | persistent | auto (timed on the first multi-rank steps)
    dp_sim_world: int = 0        # MMVAE_DP_SIM_WORLD: timing diagnostics (bench.py --sim-world)
    dp_autotune_force: bool = False  # MMVAE_DP_AUTOTUNE_FORCE=1: run the kernel-family timing on one rank too (tests)
    stamps: bool = False         # MMVAE_STAMPS=1: marker launches inside the captured programs (untraced timelines; bench.py prints them)

    @staticmethod
    def from_env() -> "EngineSettings":
        e = os.environ.get
        ov = e("MMVAE_DP_OVERLAP", "")
        kernels = e("MMVAE_DP_KERNELS", "auto")
        if kernels not in ("auto", "dynamic", "persistent"):
            raise _lib.HipLibraryError(f"MMVAE_DP_KERNELS={kernels!r}: dynamic | persistent | auto")
        return EngineSettings(
            conditionals=e("MMVAE_ENGINE_CONDITIONALS", "1") != "0", graphs=e("MMVAE_NO_GRAPH", "0") == "0",
            planes=e("MMVAE_PLANES", "1") != "0", side_dw=int(e("MMVAE_SIDE_DW", "125")),
            side_dw2=int(e("MMVAE_SIDE_DW2", "185")), side_dw_dp=int(e("MMVAE_SIDE_DW_DP", "125")),
            side_dw_any=e("MMVAE_SIDE_DW_ANY", "0") != "0", prefetch=int(e("MMVAE_PREFETCH", "128")), prefetch_join=e("MMVAE_PREFETCH_JOIN", "0") == "1",
            cond_batched=e("MMVAE_COND_BATCHED", "1") != "0", x_planes=e("MMVAE_X_PLANES", "0") == "1", prefetch_adv=int(e("MMVAE_PREFETCH_ADV", "86")),
            adv_fused=e("MMVAE_ADV_FUSED", "1") != "0",
            adv_aside=int(e("MMVAE_ADV_ASIDE", "2")),
            dp_overlap=None if ov == "" else ov != "0", dp_shard=e("MMVAE_DP_SHARD", "1") != "0", dp_kernels=kernels,
            dp_sim_world=int(e("MMVAE_DP_SIM_WORLD", "0")), dp_autotune_force=e("MMVAE_DP_AUTOTUNE_FORCE", "0") != "0",
            stamps=e("MMVAE_STAMPS", "0") == "1")


# HIP runtimes (prefixes of torch.version.hip) on which the multi-stream captured programs have been validated: > 10^4
# replays per program of the measured geometry without a fault.  The box-dependent crash inside hipGraphLaunch that toy
# shapes showed on 7.0 (tools/debug/graph_crash_stress.py is the reproducer; _Plan._build keeps such shapes on one stream)
# has no root cause: on a runtime nobody has run them on, the programs stay on ONE stream unless MMVAE_SIDE_DW_ANY=1.
FORK_VALIDATED_RUNTIMES = ("7.0.",)


def concurrent_streams(device, n: int, pool: int = 12) -> list:
    """Up to `n` new streams whose work was observed to run BESIDE the current stream's.  HIP maps streams onto a few
    hardware queues, and two streams on one queue run one after the other.  The current stream runs two 0.2 ms spin
    kernels with an event between them; the candidate waits for that event and runs a short spin: timed on the candidate
    itself it is done after ~0.2 ms when it runs beside the second spin, after ~0.4 ms when it queues behind it."""
    main = torch.cuda.current_stream(device)
    spin = 400_000
    out = []
    for _ in range(pool):
        c = torch.cuda.Stream(device=device)
        torch.cuda.synchronize(device)
        t0, t1, m0, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        ev = torch.cuda.Event()
        with torch.cuda.stream(c):
            t0.record()
        m0.record(main)
        torch.cuda._sleep(spin)
        ev.record(main)
        torch.cuda._sleep(spin)
        m1.record(main)
        with torch.cuda.stream(c):
            c.wait_event(ev)
            torch.cuda._sleep(2_000)
            t1.record()
        torch.cuda.synchronize(device)
        side_us, both_us = t0.elapsed_time(t1) * 1e3, m0.elapsed_time(m1) * 1e3
        if side_us < 0.8 * both_us:
            out.append(c)
            if len(out) == n:
                break
    return out


def forks_allowed(settings: "EngineSettings", runtime: Optional[str] = None) -> bool:
    """May captured programs fork onto branch streams on this HIP runtime?"""
    v = (torch.version.hip or "") if runtime is None else runtime
    return bool(settings.side_dw_any) or any(v.startswith(p) for p in FORK_VALIDATED_RUNTIMES)


class StepEngine:
    @staticmethod
    def try_build(model) -> Optional["StepEngine"]:
        m = model.module
        cl = getattr(m.vae, "conditionals", None)
        if cl is not None and not EngineSettings.from_env().conditionals:
            return None
        if m.vae.encoder.z_transformation is not _identity:
            return None
        blocks = [m.vae.encoder.fc, m.vae.decoder]
        for e in m.experts.values():
            blocks += [e.encoder, e.decoder]
        for adv in m.adversarials:
            blocks.append(adv.encoder)
            for h in adv.heads.values():
                if len(h.fc_layers) != 1 or len(list(h.fc_layers[0].children())) != 1:
                    return None
        if not all(_supported_block(b) for b in blocks):
            return None
        opts = model.optimizers()
        if not all(isinstance(o, HipAdam) and o._hip for o in opts):
            return None
        if cl is not None:
            # (ranks see different conditions: the blocks that step are the UNION over the ranks, CondProgram.load)
            if not CondProgram.supported(cl, model.get_optimizers()["vae"], m.vae.encoder.mean_encoder.out_features):
                return None
        return StepEngine(model)

    def __init__(self, model, settings: Optional[EngineSettings] = None):
        self.model = model
        self.settings = settings if settings is not None else EngineSettings.from_env()
        self.lib = _lib.load()
        self.device = next(model.parameters()).device
        self.opts = model.get_optimizers()
        self._grad_of: Dict[int, torch.Tensor] = {}
        for opt in model.optimizers():
            for i, p in enumerate(opt.arena.params):
                self._grad_of[id(p)] = opt.arena.grad_view(i)
        self._pool: Dict[tuple, torch.Tensor] = {}
        self._plans: Dict[tuple, "_Plan"] = {}
        # The optimisers' device state words (step, pre-clip norm, clip coefficient, bias corrections) are moved into the
        # top of the engine's metrics buffer: the logged gradient norms are then part of the one 256-word copy that ends
        # the captured program, instead of a single-word copy launch per optimiser.
        self._state_slot: Dict[int, int] = {}
        opts_list = list(model.optimizers())
        if len(opts_list) <= 8:
            metrics = self.buf("metrics", (256,))
            for i, opt in enumerate(opts_list):
                view = metrics[192 + 8 * i:200 + 8 * i]
                if opt.state_dev.data_ptr() != view.data_ptr():
                    view.copy_(opt.state_dev)
                    opt.state_dev = view
                self._state_slot[id(opt)] = 192 + 8 * i
        self._ptr_seen: Dict[tuple, int] = {}
        self.stamps = self.settings.stamps  # diagnostics: milestone markers inside the programs
        self.eager_only = False  # measurement hook (bench.py's roofline leg): run the programs eagerly, not from their graphs
        self.klw_dev = torch.ones(1, dtype=torch.float32, device=self.device)
        self._klw_host = None
        st = self.settings
        # Operands of the G-wide GEMMs written once as bf16 planes by their producers (include/mmvae_hip.h, "Pre-split
        # operands") instead of being split inside every GEMM tile that reads them; K = 1 training programs: x and dY for
        # the first layer's weight gradient, the last hidden activations for the last layer's.
        self.planes = st.planes
        # The decoder's G-wide weight gradient (dW = dP^T h, ~105 us at C2, needed by the optimiser only) on a second
        # stream beside the backward chain of the core layers (~150 us of latency-bound launches that leave most CUs
        # idle): the persistent GEMM kernel is launched with its grid capped to `side_dw` workgroups = CUs, the chain
        # gets the rest.  Inside the captured graph (a forked branch joined ahead of the expert's optimiser).  Measured
        # at C2 (profiles/r2_side_dw_sweep.txt): cap 125 -> -2.4 %, 140/167 -> -0.8 %, 200 -> +2.6 %; bit-identical
        # results.  0 = one stream.
        self.side_dw = st.side_dw
        self.fork_ok = forks_allowed(st)
        if not self.fork_ok and (st.side_dw or st.side_dw_dp):
            import warnings

            warnings.warn(f"mmvae_amd.engine: HIP runtime {torch.version.hip} is not one the forked step programs were "
                          f"validated on {FORK_VALIDATED_RUNTIMES}: single-stream programs (MMVAE_SIDE_DW_ANY=1 overrides)")
            self.side_dw = 0
        # cap of the expert encoder's weight gradient while the shared VAE's optimiser (and the loss words, the bias
        # column sums) run beside it on the branch stream (553 items at C2: 3 rounds on 185 workgroups as on 256)
        self.side_dw2 = st.side_dw2
        # the decoder's branch inside the exchange program (data parallelism): beside the part of the backward chain that
        # lies ahead of the shared VAE's exchange point (the cut joins it); 0 = in order
        self.side_dw_dp = st.side_dw_dp if self.fork_ok else 0
        self.side_dw_any = st.side_dw_any
        # adversaries without BatchNorm: both phases of all of them as seven launches (_Plan._build_adversaries_fused)
        self.adv_fused = st.adv_fused
        self.side_stream = torch.cuda.Stream(device=self.device) if (self.side_dw or self.side_dw_dp) else None
        # adversarial programs: the first branch stream carries the adversaries' passes from the reparameterisation into
        # the backward chain, the decoder's capped weight gradient takes a second one
        self.side_stream2 = torch.cuda.Stream(device=self.device) if (self.side_dw and st.adv_aside >= 2 and st.adv_fused) else None
        self.comm_stream = self.small_stream = self.lane_stream = None
        self._configure_parallel()
        self._sig = self._signature()
        self._pending: Dict[str, torch.cuda.Event] = {}
        import weakref

        me = weakref.ref(self)
        for opt in opts_list:  # torch-side readers of the moments / parameters wait for updates left on another stream
            opt.settle = lambda me=me: me() is not None and me().flush()
        # software pipelining across steps: what the last training program computed ahead for the next one
        # (training_step, "prefetch"): None or a dict(eid, ptr, shape, stride, version, w_version, slabs)
        self._prefetched: Optional[dict] = None
        self.prefetch_stats = {"issued": 0, "consumed": 0, "discarded": 0, "staged_ahead": 0}
        self._next_seen: Dict[tuple, int] = {}   # announced batches by (expert, pointer, stride, shape): resident or streamed?
        self._staged: Dict[str, tuple] = {}      # expert -> (pointer, shape, stride, version) of the batch in its static input buffer

    def _configure_parallel(self) -> None:
        """Overlapped data parallelism (default whenever gradients are exchanged).  The active expert's parameters are
        not read again until that expert's NEXT step (modalities alternate), so its gradient all-reduce (~170 MB over
        xGMI at C2) and the clip + Adam update that needs the reduced gradients run on a communication stream,
        concurrently with the next step's compute for another modality; the next step of the SAME expert waits for the
        event recorded behind that update.  The shared-VAE gradients (a few MB, needed every step) are final before the
        expert-encoder backward starts: their all-reduce is issued there, on a second stream and a second communicator,
        and hides behind the remaining backward GEMMs.  MMVAE_DP_OVERLAP=0 restores the in-order exchange; =1 forces
        the overlapped program on one rank (measured slower at N = 1: Adam is HBM-bound)."""
        self.world = mdist.world_size()
        st = self.settings
        self.overlap = mdist.collectives_active() if st.dp_overlap is None else bool(st.dp_overlap)
        # sharded expert update under data parallelism: reduce-scatter of the gradient arena, clip + Adam on this rank's
        # 1 / world of it, all-gather of the parameters -- the same bytes on the wire as the all-reduce, the 1.2 GB
        # Adam pass world times shorter (MMVAE_DP_SHARD=0: all-reduce + the full update on every rank)
        self.shard = mdist.collectives_active() and st.dp_shard
        # diagnostics (timing only, wrong numbers): on ONE rank, update the slice a rank of a world of N would own and skip
        # the collectives -- what the compute side of the N-rank program costs (bench.py --sim-world)
        self.shard_sim_world = st.dp_sim_world if self.world == 1 else 0
        # The wave-specialised GEMM kernel runs ONE persistent workgroup per CU with statically dealt work items: a
        # collective's workgroups holding CUs beside it (the previous step's all-reduce under data parallelism) would
        # delay whole workgroups by a round.  Under a gradient exchange the 2 x 4-wave kernel (measured sensitivity:
        # DESIGN.md section 7) is kept unless the caller chose explicitly.
        # (library launch state, set and restored here -- not the process environment: a later single-rank engine of
        # the same process gets the persistent kernel back)
        # MMVAE_DP_KERNELS = dynamic | persistent | auto (default): which of the two the exchange program launches.  With
        # this rank's slice of a world of 8, side branch on (r4_dp_rehearsal.txt): persistent 0.94 ms against 1.04 ms --
        # but only a real run knows how the collectives' resident workgroups treat the static deal, so "auto" times both
        # on the first steps of a multi-rank run and keeps the faster one (_dp_autotune).
        self.dp_kernels = st.dp_kernels
        if not hasattr(self, "_tune"):
            self._tune = None
            self.dp_tuned: Dict[str, float] = {}
        if "MMVAE_X3W" not in os.environ:
            if not mdist.collectives_active():
                self.lib.mmvae_gemm_set_x3w(-1)
            else:
                choice = self.dp_kernels if self.dp_kernels != "auto" else (self.dp_tuned.get("choice") or "dynamic")
                tune = (self.dp_kernels == "auto" and "choice" not in self.dp_tuned
                        and (mdist.world_size() > 1 or st.dp_autotune_force))
                if tune and self._tune is None:
                    self._tune = dict(kind="dynamic", step=0, ev=None)
                elif tune:
                    # re-configured in the middle of a tune (a signature change dropped the plans: every rank sees it at
                    # the same step): the family being timed stays, its window starts again
                    self._tune.update(step=0, ev=None)
                    choice = self._tune["kind"]
                self.lib.mmvae_gemm_set_x3w(1 if choice == "persistent" else 0)
        if self.overlap and self.comm_stream is None:
            # streams of the exchange program: HIP maps streams onto a few hardware queues (4 by default), and two streams
            # on one queue run one after the other -- a lane on a stream that shares the main stream's queue started when
            # the main stream's segment ended (untraced markers), on another stream 400 us earlier.  Take streams that
            # were SEEN to run beside the current stream.
            got = concurrent_streams(self.device, 3)
            # (reported by bench.py: a first multi-GPU run explains itself -- fewer than 3 observed means some stream of the
            # exchange program shares a hardware queue with the main stream and its work will run behind it, not beside it)
            self.streams_probe = {"wanted": 3, "observed_concurrent": len(got)}
            self.comm_stream, self.small_stream, self.lane_stream = (got + [torch.cuda.Stream(device=self.device)
                                                                            for _ in range(3)])[:3]

    def _signature(self) -> tuple:
        """Everything a captured training program freezes at build time: optimiser hyper-parameters, clip values, the
        adversarial weight, world size / gradient exchange.  Compared on every training step; a change (dist.attach()
        after the first step, a new learning rate through param_groups or HipAdam.load_state_dict, an edited
        autograd_config) drops the plans and their graphs, which are then rebuilt with the current values."""
        ac = self.model.autograd_config
        clip = lambda c: (float(c.val), c.algorithm or "norm") if (c and c.val) else (0.0, "norm")  # noqa: E731
        per_opt = tuple((g["lr"], g["eps"], g["weight_decay"], tuple(g["betas"]), o.reducer is not None, o.grad_scale)
                        for o in self.model.optimizers() for g in o.param_groups[:1])
        return (per_opt, clip(ac.vae_gradient_clip), clip(ac.expert_gradient_clip), clip(ac.adversarial_gradient_clip),
                float(self.model.adv_weight), mdist.world_size(), mdist.collectives_active())

    def _drop_train_plans(self) -> None:
        self.flush()
        torch.cuda.synchronize(self.device)
        for k, p in self._plans.items():
            if str(k[0]).startswith("train"):
                p.release()
        self._plans = {k: p for k, p in self._plans.items() if not str(k[0]).startswith("train")}
        self._ptr_seen.clear()
        self._prefetched = None

    DP_TUNE_WARM, DP_TUNE_STEPS = 12, 12

    def _dp_autotune(self, plan: "_Plan") -> None:
        """MMVAE_DP_KERNELS=auto under a real exchange: time DP_TUNE_STEPS training steps with the 2 x 4-wave (hardware-
        scheduled) GEMM kernels, then with the persistent ones, take the maximum over the ranks of each and keep the
        faster.  The steps are ordinary training steps; a switch drops the captured programs (they bake the kernel in).
        The schedule is a function of the number of training steps since the tune began and of nothing else (ADVICE r4:
        keyed to how often this rank's plan had run, ranks whose allocators handed out different batch pointers could
        leave a phase at different steps and meet the closing reduction at different places of their collective
        streams): DP_TUNE_WARM steps untimed -- eager run, capture and the first replays of every expert's program --
        then DP_TUNE_STEPS timed ones, per kernel family; the closing MAX over the ranks is a HOST collective (gloo) at
        a step index every rank reaches, behind a device synchronisation: no stream collective is outstanding there."""
        t = self._tune
        st = torch.cuda.current_stream()
        t["step"] += 1
        if t["step"] < self.DP_TUNE_WARM:
            return
        if t["step"] == self.DP_TUNE_WARM:
            t["ev"] = torch.cuda.Event(enable_timing=True)
            t["ev"].record(st)
            return
        if t["step"] < self.DP_TUNE_WARM + self.DP_TUNE_STEPS:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record(st)
        end.synchronize()
        self.dp_tuned[t["kind"]] = t["ev"].elapsed_time(end) / self.DP_TUNE_STEPS
        if t["kind"] == "dynamic":
            self.lib.mmvae_gemm_set_x3w(1)
            self._drop_train_plans()
            t.update(kind="persistent", step=0, ev=None)
            return
        import numpy as np

        both = np.array([int(self.dp_tuned["dynamic"] * 1e3), int(self.dp_tuned["persistent"] * 1e3)], dtype=np.int32)
        if mdist.world_size() > 1:
            self.flush()
            torch.cuda.synchronize(self.device)
            mdist.host_all_reduce_max(both)  # microseconds per step, the slowest rank's
        d, p = float(both[0]) / 1e3, float(both[1]) / 1e3
        self.dp_tuned.update(dynamic=d, persistent=p, choice="persistent" if p <= d else "dynamic")
        if self.dp_tuned["choice"] == "dynamic":
            self.lib.mmvae_gemm_set_x3w(0)
            self._drop_train_plans()
        self._tune = None

    def _check_signature(self) -> None:
        sig = self._signature()
        if sig != self._sig:
            # A captured program freezes the optimiser's hyper-parameters: a per-step schedule (learning-rate warm-up)
            # rebuilds and re-captures every step -- correct, but milliseconds instead of one replay.  Say so once.
            self._sig_changes = getattr(self, "_sig_changes", 0) + 1
            if self._sig_changes == 3 and not getattr(self, "_sig_warned", False):
                import warnings

                self._sig_warned = True
                warnings.warn("mmvae_amd.engine: optimiser settings changed on several training steps; every change drops "
                              "the captured step programs and re-captures them (a learning-rate schedule stepping per "
                              "batch costs milliseconds per step) -- change them per epoch, or set use_engine=False")
            self._drop_train_plans()
            self._configure_parallel()
            self._sig = sig

    # ------------------------------------------------------------------------------------------------ buffers
    def buf(self, name: str, shape, dtype=torch.float32) -> torch.Tensor:
        """Engine-level scratch pool: every plan of the same geometry shares its transient buffers (plans never
        overlap in time on the stream)."""
        key = (name, tuple(int(s) for s in shape), dtype)
        t = self._pool.get(key)
        if t is None:
            # Slack behind every buffer, zero and never written: 16 elements for a rows-contiguous GEMM operand whose
            # extent is not a multiple of 4 (60 530 genes: 16-byte groups reach 12 bytes past the last row,
            # MMVAE_GEMM_OPERAND_SLACK), and 32 more ROWS behind a matrix, so that a weight-gradient GEMM whose K is the
            # batch can run K up to the next multiple of 32 over zero rows instead of taking a K-tail path (kpad)
            n = 1
            for d in key[1]:
                n *= d
            extra = 16 + (32 * key[1][-1] if len(key[1]) >= 2 else 0)
            t = torch.zeros(n + extra, dtype=dtype, device=self.device)[:n].view(key[1])
            self._pool[key] = t
        return t

    def grad_of(self, p: torch.Tensor) -> torch.Tensor:
        return self._grad_of[id(p)]

    def locate_grad(self, t: torch.Tensor):
        """(optimiser, element offset) of a gradient-arena view, or None."""
        for opt in self.model.optimizers():
            g = opt.arena.grad
            d = t.data_ptr() - g.data_ptr()
            if 0 <= d < 4 * g.numel():
                return opt, d // 4
        return None

    def sq_buffer(self, opt) -> torch.Tensor:
        """Norm-partial slots of one optimiser when GEMM epilogues contribute (fused partials first, then the norm
        pass's chunk partials of the uncovered ranges)."""
        n = int(self.lib.mmvae_sqnorm_partials(opt.arena.numel)) + SQ_FUSED_SLOTS + 64
        return self.buf(f"sqparts.{id(opt)}", (n,))

    def close(self) -> None:
        """Release every captured program now (see _Plan.release) instead of when the collector finds the cycles."""
        self.flush()
        torch.cuda.synchronize(self.device)
        for p in self._plans.values():
            p.release()
        self._plans = {}
        self._ptr_seen.clear()
        # sharded expert updates leave (world - 1) / world of the Adam moments on other ranks: gather them while the
        # process group is alive, so that a later state_dict() -- on one rank only, or after destroy_process_group() -- is
        # local (COLLECTIVE under data parallelism: every rank closes its engine, like every rank takes every step)
        if mdist.collectives_active():
            for opt in self.model.optimizers():
                opt.sync_sharded_state()

    def flush(self) -> None:
        """Make the current stream wait for every deferred expert update (before parameters are read elsewhere)."""
        cur = torch.cuda.current_stream()
        if self.comm_stream is not None:
            cur.wait_stream(self.comm_stream)
            cur.wait_stream(self.small_stream)
        for ev in self._pending.values():
            cur.wait_event(ev)
        self._pending.clear()

    # ------------------------------------------------------------------------------------------------- inputs
    def _enc_planes(self, l0_in: int, l0_out: int, has_bn: bool, B: int, K: int, train: bool, iwae: bool) -> bool:
        """Does a program of this geometry read the operands of its first-layer weight gradient from bf16 planes?  (Then
        nothing reads the batch in 16-byte groups across row ends or beyond its last row: _select_input.)"""
        # (K-sample programs run the encoder once over the B cells: the same product as at K = 1)
        return bool(self.planes and train and not iwae and self.lib.mmvae_gemm_get_precision() == 1
                    and l0_out % 8 == 0 and has_bn
                    and self.lib.mmvae_gemm_planes_supported(TN, l0_out, l0_in, (B + 31) // 32 * 32, 1,
                                                             int(self._x_planes(l0_in, B)), 1))

    def _x_planes(self, l0_in: int, B: int) -> bool:
        """Is the BATCH pre-split for the first layer's weight gradient dW1 = dY^T x (True), or does that GEMM read it as
        fp32 and split it in its stagers, with only dY pre-split (False)?  Late r5: the split pass over x -- 41 MB in, 61 MB
        out, riding on the forward chain's tail launches -- costs that chain 27 us since the pipelined product runs beside
        it, and buys the GEMM 6 us (145 -> 151 us): fp32 wins at every configuration (C2 0.900 -> 0.888 ms, C3 3.452 ->
        3.405, C4 1.078 -> 1.063, C5 5.096 -> 5.039, CSR-fed 0.951 -> 0.934; HISTORY.md).  Planes stay where the fp32 batch
        could not be read in place: a gene count off a multiple of 4 or a batch off a multiple of 32 (rows-contiguous
        16-byte groups / whole k-tiles: the batch would have to be staged with slack every step -- the reference's
        60 530 / 52 437 genes)."""
        return bool(self.settings.x_planes or l0_in % 4 != 0 or B % 32 != 0)

    def _select_input(self, x: torch.Tensor, base_key: tuple, needs_slack: bool = True):
        """Plan selection: graphs are keyed by the input pointer once a pointer has been seen twice (resident
        batches); otherwise the batch is copied into a static buffer.  Returns (plan key, the tensor the plan reads).
        needs_slack: some kernel of the program reads the batch as a rows-contiguous fp32 GEMM operand (the first layer's
        weight gradient without planes): 16-byte groups that reach past a row's end when the gene count is not a multiple
        of 4, and zero rows behind a batch that is not a multiple of 32 (kpad)."""
        B = x.shape[0]
        if x.layout == torch.sparse_csr:  # CSR batch: densified by one HIP pass straight into the static input buffer
            from . import ops

            x_in = self.buf(f"x_static.{base_key[1]}", (B, x.shape[1]))
            if self._staged.pop(base_key[1], None) != self._batch_sig(x):  # (else: densified when it was announced)
                ops.csr_to_dense(x, out=x_in)
            return base_key + (0, 0), x_in
        pkey = base_key + (x.data_ptr(), x.stride(0))
        seen = self._ptr_seen.get(pkey, 0)
        self._ptr_seen[pkey] = seen + 1
        n_ptr_plans = sum(1 for k in self._plans if k[-2] != 0)
        # a caller's tensor has no slack behind it: with a gene count that is not a multiple of 4, or a batch that is
        # not a multiple of 32 (kpad reads zero rows behind the batch), it is always staged
        direct_ok = (x.shape[1] % 4 == 0 and B % 32 == 0) or not needs_slack
        if direct_ok and (pkey in self._plans or (seen >= 1 and n_ptr_plans < MAX_POINTER_PLANS)):
            return pkey, x
        x_in = self.buf(f"x_static.{base_key[1]}", (B, x.shape[1]))
        if self._staged.pop(base_key[1], None) == self._batch_sig(x):
            pass  # the previous step staged this very batch when it was announced (_stage_next)
        elif x.is_contiguous():  # own 16-byte copy kernel: the runtime's blit kernel reaches < 1 TB/s here
            from . import ops

            ops.axpby(1.0, x, 0.0, x_in)
        else:
            x_in.copy_(x)
        if len(self._ptr_seen) > 4096:
            self._ptr_seen.clear()
        return base_key + (0, 0), x_in

    @staticmethod
    def _dense_f32(x: torch.Tensor) -> torch.Tensor:
        if x.layout == torch.sparse_csr:
            return x  # densified in _select_input
        if x.dtype != torch.float32 or x.dim() != 2 or (x.shape[1] > 1 and x.stride(1) != 1):
            x = x.float().contiguous()
        return x

    def _set_kl_weight(self):
        """Per-step host values -> device scalars (the captured graphs read the device word)."""
        klw = float(self.model.kl_annealing_fn.kl_weight)
        if klw != self._klw_host:
            self.klw_dev.fill_(klw)
            self._klw_host = klw

    # ------------------------------------------------------------------------------------- eval / predict (f3)
    def _forward_only(self, mode: str, x: torch.Tensor, expert_id: str, metadata=None):
        """Forward-only plan (no autograd, no gradients, no optimiser): eval-mode BatchNorm (running statistics), no
        dropout, one rsample.  mode "validate": + fused reconstruction / ELBO; mode "embed": stops at z."""
        x = self._dense_f32(x)
        ev = self._pending.pop(expert_id, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        enc_mod = self.model.module.vae.encoder
        explicit = enc_mod.explicit_eps is not None
        B = x.shape[0]
        # (forward-only programs read the batch through the K-contiguous forward GEMM and the reconstruction epilogue's
        # guarded loads only: any caller's tensor serves as it is)
        key, x_in = self._select_input(x, (mode, expert_id, B, 1, explicit), needs_slack=False)
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(self, expert_id, B, 1, explicit, x_in, mode=mode)
            self._plans[key] = plan
        self._set_kl_weight()
        if explicit:
            plan.eps.copy_(enc_mod.explicit_eps.reshape(plan.eps.shape))
        if plan.cond is not None:
            plan.cond.load(metadata)
        plan.run()
        if plan.cond is not None:
            plan.cond.commit()
        return plan

    def validation_step(self, x: torch.Tensor, metadata, expert_id: str) -> dict:
        """`CMMVAEModel.validation_step` (models/cmmvae_model.py:219-248) as one captured forward program.  Returns the
        loss dict of `BaseVAE.elbo` (device scalars)."""
        plan = self._forward_only("validate", x, expert_id, metadata)
        m = plan.metrics.clone()
        return {RK.LOSS: m[0], RK.RECON_LOSS: m[1], RK.KL_LOSS: m[2], RK.KL_WEIGHT: m[3]}

    def latent_embeddings(self, x: torch.Tensor, metadata, expert_id: str) -> torch.Tensor:
        """z of `CMMVAE.get_latent_embeddings` (modules/cmmvae.py:115-142) as one captured forward program."""
        plan = self._forward_only("embed", x, expert_id)
        return plan.z[0].clone()

    # --------------------------------------------------------------------------------------------------- step
    # ------------------------------------------------------------------- software pipelining across steps (r5)
    # The first forward GEMM of a step -- x W1^T of the active expert, 21 GFLOP at C2 -- depends on that step's batch and
    # on weights the expert's LAST update left (modalities alternate: another expert trains in between).  When the caller
    # says which batch comes next (CMMVAEModel.hint_next_batch; mmvae_amd.trainer and bench.py look one batch ahead), this
    # step computes the next step's product as filler work: on the branch stream, capped to `prefetch` workgroups, beside
    # its own forward chain -- ~125 us of latency-bound launches that leave most of the chip idle -- into split-K slabs of
    # the next expert's own.  The next step's program then starts at its first layer's tail.  The same kernel on the same
    # operands: results are bit-identical to the unpipelined program (tested).  Conditions: single-rank in-order program
    # of the measured geometry (the forked C2 program), a DIFFERENT expert next (the same one's weights are about to be
    # updated), the hinted tensor unchanged when its step comes (pointer, shape, torch version counter) and the next
    # expert's weights not rewritten from the torch side in between (version counter: load_state_dict) -- otherwise the
    # next step computes the product itself and the slabs are dropped.
    def _prefetch_target(self, plan_eid: str, B: int, K: int, iwae: bool, next_batch):
        """(expert id, tensor) this step may compute ahead for, or None."""
        st = self.settings
        if (next_batch is None or not st.prefetch or not self.side_dw or self.overlap or self.world != 1
                or self.side_stream is None or K != 1 or iwae):
            return None
        x_n, eid_n = next_batch
        m = self.model.module
        if eid_n == plan_eid or eid_n not in m.experts:
            return None
        if getattr(m.vae, "conditionals", None) is not None:
            # (measured twice, r5: nothing to gain -- the conditional programs' device period is 1.30 ms under the tracer with
            # and without it: the chain's gather-bound conditional kernels stretch beside the capped GEMM by what the
            # product saves; profiles/HISTORY.md)
            return None
        if len(m.adversarials) > 0 and not (st.prefetch_adv and self.side_stream2 is not None):
            return None  # (adversarial programs: the product takes the second branch stream, free until the late branch)
        if not torch.is_tensor(x_n) or not x_n.is_cuda or x_n.dtype != torch.float32 or x_n.dim() != 2:
            return None
        if x_n.layout != torch.sparse_csr and (x_n.layout != torch.strided or x_n.stride(1) != 1):
            return None
        l0 = m.experts[eid_n].encoder.fc_layers[0]
        if not hasattr(l0, "bn") or x_n.shape[1] != l0.lin.in_features:
            return None
        return eid_n, x_n

    @staticmethod
    def _batch_sig(x: torch.Tensor) -> tuple:
        """What a batch tensor is recognised by one step later: storage pointer(s), shape, stride, torch version counter."""
        if x.layout == torch.sparse_csr:
            v, c = x.values(), x.crow_indices()
            return ("csr", v.data_ptr(), c.data_ptr(), x.col_indices().data_ptr(), tuple(x.shape), v._version, c._version)
        return (x.data_ptr(), tuple(x.shape), x.stride(0), x._version)

    def _stage_next(self, eid_n: str, x_n: torch.Tensor) -> torch.Tensor:
        """The tensor this step's program reads the announced batch from.  A resident batch (its pointer has been announced
        before) is read in place.  A streamed one -- a new tensor every step -- is copied into the next expert's static
        input buffer NOW: that is the staging copy its own step would make at its start (_select_input), made one step
        earlier, so the program's pointers stay the same from step to step (no plan per batch) and the next step finds its
        input in place (`_staged`)."""
        from . import ops

        buf = self.buf(f"x_static.{eid_n}", tuple(x_n.shape))
        if x_n.layout == torch.sparse_csr:  # a CSR batch is densified into that buffer by its own step anyway: one step early
            ops.csr_to_dense(x_n, out=buf)
        else:
            k = (eid_n, x_n.data_ptr(), x_n.stride(0), tuple(x_n.shape))
            seen = self._next_seen.get(k, 0)
            if len(self._next_seen) > 4096:
                self._next_seen.clear()
            self._next_seen[k] = seen + 1
            if seen >= 2:
                return x_n
            if x_n.is_contiguous():
                ops.axpby(1.0, x_n, 0.0, buf)
            else:
                buf.copy_(x_n)
        self._staged[eid_n] = self._batch_sig(x_n)
        self.prefetch_stats["staged_ahead"] += 1
        return buf

    def _take_prefetched(self, eid: str, x: torch.Tensor):
        """The slabs computed ahead for this step, when they are (still) the product of this batch and these weights."""
        pf, self._prefetched = self._prefetched, None
        if pf is None:
            return None
        w = self.model.module.experts[eid].encoder.fc_layers[0].lin.weight if eid in self.model.module.experts else None
        ok = pf["eid"] == eid and w is not None and pf["sig"] == self._batch_sig(x) and pf["w_version"] == w._version
        self.prefetch_stats["consumed" if ok else "discarded"] += 1
        return pf["slabs"] if ok else None

    def training_step(self, x: torch.Tensor, metadata, expert_id: str, next_batch=None) -> None:
        """next_batch: (x of the next training step, its expert id) or None -- see "software pipelining" above."""
        model = self.model
        self._check_signature()
        x_arg = x
        x = self._dense_f32(x)
        enc_mod = model.module.vae.encoder
        expert = model.module.experts[expert_id]
        explicit = enc_mod.explicit_eps is not None or expert.encoder.explicit_masks is not None
        K = int(enc_mod.n_samples)
        if enc_mod.explicit_eps is not None and enc_mod.explicit_eps.dim() == 3:
            K = enc_mod.explicit_eps.shape[0]
        B = x.shape[0]
        iwae = getattr(enc_mod, "elbo_mode", "analytic") == "iwae"
        l0 = expert.encoder.fc_layers[0]
        planes = self._enc_planes(l0.lin.in_features, l0.lin.out_features, hasattr(l0, "bn"), B, K, True, iwae)
        ahead = self._take_prefetched(expert_id, x) if x is x_arg else None  # slabs the previous step left for this one
        if x is not x_arg:
            self._prefetched = None
        target = self._prefetch_target(expert_id, B, K, iwae, next_batch)
        hinted = target[1] if target else None  # the caller's tensor: what the next step will be recognised by
        if target is not None:
            target = (target[0], self._stage_next(*target))
        tsig = (target[0], target[1].data_ptr(), tuple(target[1].shape), target[1].stride(0)) if target else None
        key, x_in = self._select_input(x, ("train-iwae" if iwae else "train", expert_id, B, K, explicit,
                                           ahead.data_ptr() if ahead is not None else 0, tsig),
                                       needs_slack=not planes)
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(self, expert_id, B, K, explicit, x_in, iwae=iwae, slabs_ahead=ahead, prefetch=target)
            self._plans[key] = plan
        if target is not None and plan.prefetch_slabs is None:
            target = None  # (the plan's geometry does not pipeline: _Plan._build)
        self._set_kl_weight()
        if explicit:
            plan.load_explicit_noise(enc_mod, expert)
        if plan.has_adv:
            plan.load_labels(metadata)
        if plan.cond is not None:
            plan.cond.load(metadata)
        ev = self._pending.pop(expert_id, None)
        if ev is not None:  # this expert's previous (deferred) update must land before its parameters are read
            torch.cuda.current_stream().wait_event(ev)
        self.last_plan = plan  # introspection (tests read the activations the step left in its buffers)
        ev = plan.run()
        if ev is not None:
            self._pending[expert_id] = ev
        if target is not None:  # this program has computed the next step's first product
            x_n = hinted
            w_n = model.module.experts[target[0]].encoder.fc_layers[0].lin.weight
            self._prefetched = dict(eid=target[0], sig=self._batch_sig(x_n), w_version=w_n._version, slabs=plan.prefetch_slabs)
            self.prefetch_stats["issued"] += 1
        if plan.cond is not None:
            plan.cond.commit()
        model.kl_annealing_fn.step()
        plan.log(model, expert_id)
        if self._tune is not None:
            self._dp_autotune(plan)


class _Plan(PlanEmit, PlanAdversaries, PlanRun):
    def __init__(self, eng: StepEngine, eid: str, B: int, K: int, explicit: bool, x: torch.Tensor, mode: str = "train",
                 iwae: bool = False, slabs_ahead: Optional[torch.Tensor] = None, prefetch=None):
        self.eng, self.eid, self.B, self.K, self.explicit = eng, eid, B, K, explicit
        self.mode = mode
        # software pipelining across steps (StepEngine.training_step): slabs of this step's first forward product that
        # the previous step computed, and the (expert id, batch) this step computes them for
        self.slabs_ahead, self.prefetch, self.prefetch_slabs = slabs_ahead, prefetch, None
        self.iwae = bool(iwae) and mode == "train"  # opt-in full-IWAE objective (training programs only)
        self.x = x
        self.R = B * K
        model = eng.model
        m = model.module
        self.lib = eng.lib
        g = eng.grad_of
        exp = m.experts[eid]

        def refs(block: FCBlock):
            return [_LayerRef(seq, g, block.config.return_hidden[i], block, i) for i, seq in enumerate(block.fc_layers)]

        self.enc_layers = refs(exp.encoder) + refs(m.vae.encoder.fc)
        self.n_expert_enc = len(exp.encoder.fc_layers)
        self.dec_layers = refs(m.vae.decoder) + refs(exp.decoder)
        self.G = exp.decoder.config.layers[-1]
        self.Z = m.vae.encoder.mean_encoder.out_features
        self.var_eps = float(m.vae.encoder.var_eps)
        self.hidden_z = bool(m.vae.encoder.hidden_z)
        self.mean_enc, self.var_enc = m.vae.encoder.mean_encoder, m.vae.encoder.var_encoder
        self.advs = list(m.adversarials)
        self.has_adv = len(self.advs) > 0
        self.adv_weight = float(model.adv_weight)
        self.opt_vae = eng.opts["vae"]
        self.opt_exp = eng.opts["experts"][eid]
        self.opt_adv = list(eng.opts.get("adversarials", {}).values()) if self.has_adv else []
        ac = model.autograd_config
        def clip_rule(c):
            """GradientClipConfig -> (max_norm for the fused clip, clamp bound for clip-by-value); config.py:4-26"""
            val = float(c.val) if (c and c.val) else 0.0
            if val and (c.algorithm or "norm") == "value":
                return 0.0, val
            if val and (c.algorithm or "norm") != "norm":
                raise _lib.HipLibraryError(f"engine: gradient_clip_algorithm {c.algorithm!r} (config.py:8 allows 'norm' / 'value')")
            return val, 0.0

        (self.clip_vae, cv_vae), (self.clip_exp, cv_exp), (self.clip_adv, cv_adv) = (
            clip_rule(ac.vae_gradient_clip), clip_rule(ac.expert_gradient_clip), clip_rule(ac.adversarial_gradient_clip))
        # clip-by-value: the bound is state word 5 of each optimiser (read by the Adam kernels; 0 = off)
        for opt, cv in [(self.opt_vae, cv_vae), (self.opt_exp, cv_exp)] + [(o, cv_adv) for o in self.opt_adv]:
            opt.set_clip_value(cv or None)
            if not cv:
                opt.state_dev[5] = 0.0
        self.conditions = list(Adversarial.labels.keys()) if self.has_adv else []
        self.metric_slots: Dict[str, int] = {}
        self.segments: List = []  # list of closure lists, separated by ("allreduce", opt) markers
        self._cur: List = []
        self.stamp_names: List[str] = []
        self._dirty: List = []          # branch streams with work the main stream has not joined yet
        self._sq_used: Dict[int, int] = {}    # per optimiser: norm-partial slots taken by fused GEMM epilogues
        self._sq_cover: Dict[int, list] = {}  # per optimiser: (offset, length) of the arena ranges they cover
        self._mask_layers: List = []   # (layer, Philox stream id) of every dropout keep mask of the program
        self._gemm_jobs: List = []     # small weight-gradient GEMMs queued for the next mmvae_gemm_batch_f32 launch
        self._sum_jobs: List = []      # reductions queued for the next mmvae_sum_parts_batch launch
        self._sum_keep: List = []
        self._job_tables: List = []    # device job tables of the launches already emitted
        self._ws_bytes = 0
        self._slab_floats = 0
        self._graphs: Optional[list] = None
        self._probe_next = None         # tag for the next emitted GEMM (measurement hook, see _emit_gemm)
        self.probe = None               # dict tag -> [(event, event, flops)] while an eager run is being measured
        self.probe_meta: Dict[str, dict] = {}  # tag -> {work, kernel, cus, bound} of the probed launches (_probed)
        self._forked = False            # the program has branches on other streams
        self._events: List = []         # the fork / join events of the program (kept alive: see _edge)
        self._runs = 0
        self.exp_norm_log = None  # overlapped mode: the expert's pre-clip gradient norm, copied on the comm stream
        self.metrics = eng.buf("metrics", (256,))
        self.rng_state = rng.state(eng.device)
        cl = getattr(m.vae, "conditionals", None)
        self.cond = None
        if cl is not None and mode != "embed":  # the embedding is z BEFORE the conditional layers (cmmvae.py:115-142)
            if K != 1:
                raise _lib.HipLibraryError("engine: conditional layers with the K-sample extension are not supported")
            self.cond = CondProgram(self, cl, eid, train=(mode == "train"))
        self._build()

    # ------------------------------------------------------------------------------------------- program building
    def _build(self):
        eng, lib = self.eng, self.lib
        B, K, R, Z, G = self.B, self.K, self.R, self.Z, self.G
        x, ldx = self.x, self.x.stride(0) if self.x.shape[0] > 1 else self.x.shape[1]
        self.eps = eng.buf("eps", (K, B, Z))
        self._mark("step begins (behind the noise fill)")

        train = self.mode == "train"
        # branches beside the latency-bound sections (in-order single-rank program only; see StepEngine.side_dw)
        # The caps are tuned for the shape where the two weight gradients are SHORTER than the chains they hide behind
        # (C2: B = 512, K = 1: 105 us of GEMM beside a 190 us chain).  With K samples or bigger batches the GEMMs grow
        # with the rows while the chains barely do, and a capped GEMM becomes the critical path (measured with the
        # branches on: C3 4.36 against 4.00 ms, C5 7.03 against 5.93 ms; with adversaries, C4: 1.64 against 1.60 ms).
        # Small models stay on ONE stream: when the kernels ahead of a fork finish while the host is still enqueuing
        # the rest of a multi-stream graph, hipGraphLaunch crashes now and then on this runtime (null dereference, box
        # dependent; only ever seen with the tests' toy shapes, whose whole step is ~100 us -- tools/debug/seg_hunt.sh).
        # A G-wide weight gradient of >= 5 GFLOP puts the first fork hundreds of microseconds into the replay.
        big = 2.0 * G * self.dec_layers[-1].n_in * R >= 5e9
        # The forked program is the default only for the geometry its caps were measured on (C2: a G-wide weight
        # gradient of ~21 GFLOP, 512 rows: within +-25 %), where it has run > 10^4 replays without a fault; any other
        # shape stays on one stream unless MMVAE_SIDE_DW_ANY=1 asks for it (the hipGraphLaunch hazard has no
        # root cause yet, and a crashed replay cannot be recovered in-process).
        dw_flops = 2.0 * G * self.dec_layers[-1].n_in * R
        measured = 0.75 * 2.1e10 <= dw_flops <= 1.25 * 2.1e10
        side_dw = eng.side_dw if (train and eng.side_stream is not None and not eng.overlap and eng.world == 1
                                  and K == 1 and R <= SIDE_MAX_ROWS and not self.has_adv and big
                                  and (measured or eng.side_dw_any)) else 0
        early_branch = bool(side_dw and not self.iwae)  # the small branches: loss words, bias sums, the VAE's optimiser
        # the late branch alone (the shared VAE's optimiser beside the encoder's capped weight gradient) also serves the
        # adversarial programs: their branch stream is free again once the adversaries' passes have been joined
        side_late = bool(side_dw or (eng.side_dw2 and eng.side_stream is not None and train and self.has_adv
                                     and not eng.overlap and eng.world == 1 and K == 1 and R <= SIDE_MAX_ROWS and big
                                     and (measured or eng.side_dw_any) and eng.settings.adv_aside))
        loss_aside, early_calls = False, []
        # ---- pre-split operands of the G-wide weight gradients (K = 1 training programs on the wave-specialised
        # kernels): x (split beside the forward chain) and the gradient at the first layer (from its column kernel) feed
        # dW1 = dY^T x; the last hidden activations (3 MB of planes from their layer tail) are B of dW4 = dP^T h, the wider
        # operand of that product's tile -- its stagers then split dP only.  (Measured and removed: dP planes from the
        # reconstruction epilogue -- epilogue +6 us, chain +12 us, no gain; planes of x for the first forward GEMM --
        # the split pass in front of it costs 17 us for 8 us gained.)
        l0, lastl = self.enc_layers[0], self.dec_layers[-1]
        pl_on = bool(eng.planes and train and not self.iwae and lib.mmvae_gemm_get_precision() == 1)
        # (any gene count: the planes of x get a leading dimension rounded up to 8, zero columns in between)
        self.pl_enc = eng._enc_planes(l0.n_in, l0.n_out, l0.bn is not None, B, K, train, self.iwae)
        self.xp = _PlaneBuf(eng, f"xp.{l0.n_in}", B, l0.n_in) if self.pl_enc else None
        self.dYp = _PlaneBuf(eng, f"dYp.{l0.n_out}", B, l0.n_out) if self.pl_enc else None
        self.pl_dec_h = bool(pl_on and lastl.n_in % 8 == 0 and len(self.dec_layers) >= 2
                             and lib.mmvae_gemm_planes_supported(TN, G, lastl.n_in, self.kpad(R), 1, 0, 1))
        self.hp = _PlaneBuf(eng, f"hp.{lastl.n_in}", R, lastl.n_in) if self.pl_dec_h else None
        # (r5) K-sample programs (R = K B rows, 20 row tiles of 256 at C3): the last layer's WEIGHTS pre-split once per step
        # for the input-gradient product dP . W, whose tiles would otherwise split each weight tile R / 256 times
        # (tools/ab_planes.py at 5120 rows: 944 -> 873 us for a 31 us split pass).  Not the reconstruction launch: with
        # both operands DMA'd it is no faster (1 007 -> 1 034 us: its stagers are not what it waits for), and not at
        # K = 1 (512 rows: -5 us against the 31 us pass).
        self.pl_dec_w = bool(pl_on and R >= 2048 and lastl.n_in % 8 == 0
                             and lib.mmvae_gemm_planes_supported(NN, R, lastl.n_in, (G + 31) // 32 * 32, 0, 0, 1))
        self.wp = _PlaneBuf(eng, f"wp.{G}.{lastl.n_in}", G, lastl.n_in) if self.pl_dec_w else None
        # (late r5) the batch itself stays fp32 wherever the weight-gradient GEMM can read it in place: StepEngine._x_planes
        self.x_fp32_dw1 = bool(self.pl_enc and not eng._x_planes(l0.n_in, B))
        if self.x_fp32_dw1:
            self.xp = None
        if self.pl_enc and not self.x_fp32_dw1:
            # The split of x is piggy-backed on the tail launches of the forward chain (extra workgroups of
            # fc_fwd_apply), a third of the rows each: those launches are latency-bound (5-13 us with the memory system
            # idle), 20 MB of streaming beside each is nearly free -- as one pass beside the first tail it cost 12 us, as a
            # launch of its own 17-22 us on the critical path, forked onto a second stream ~80 us (graph executor).
            per = (B + 2) // 3
            xpp, xld, xps = self.xp.args()
            self._x_split_jobs = [(min(per, B - r0), l0.n_in, _p(x) + 4 * r0 * ldx, ldx, xpp + 2 * r0 * xld, xld, xps)
                                  for r0 in range(0, B, per)]
        if self.pl_dec_w:  # the weights' split takes the tails first (82 MB at C3 against x's 41): three row ranges
            per = (G + 2) // 3
            wpp, wld, wps = self.wp.args()
            self._x_split_jobs = [(min(per, G - r0), lastl.n_in, _p(lastl.W) + 4 * r0 * lastl.n_in, lastl.n_in,
                                   wpp + 2 * r0 * wld, wld, wps) for r0 in range(0, G, per)] + getattr(self, "_x_split_jobs", [])
        # ---- forward, encoder side
        # (adversarial programs of the measured geometry fork too: the same test as `adv_aside` below)
        adv_forks = bool(self.has_adv and train and eng.adv_fused and eng.settings.adv_aside >= 2 and eng.side_stream2 is not None
                         and not eng.overlap and eng.world == 1 and K == 1 and big and (measured or eng.side_dw_any)
                         and self.cond is None and not any(o.reducer is not None for o in self.opt_adv))

        def emit_prefetch():
            if self.prefetch is None or not train or not (side_dw or adv_forks):
                return
            # the NEXT step's first forward product, beside this step's forward chain (no join of its own: the branch
            # stream is in order and joined ahead of the expert's optimiser -- see the note ahead of the reconstruction launch)
            eid_n, x_n = self.prefetch
            ln = _LayerRef(eng.model.module.experts[eid_n].encoder.fc_layers[0], eng.grad_of, False)
            Bn = x_n.shape[0]
            sk_n = self._plan_gemm(NT, Bn, ln.n_out, ln.n_in)
            if ln.bn is None or 2.0 * Bn * ln.n_out * ln.n_in < 5e9:
                return
            self.prefetch_slabs = eng.buf(f"prefetch.slabs.{eid_n}", (sk_n, Bn, ln.n_out))
            self._probe_next = "enc_l1_fwd"
            self._side_capped_gemm(NT, Bn, ln.n_out, ln.n_in, x_n, x_n.stride(0), ln.W, ln.n_in, self.prefetch_slabs,
                                   ln.n_out, eng.settings.prefetch_adv if adv_forks else eng.settings.prefetch,
                                   flags=_lib.GEMM_RAW_SLABS, sk=sk_n, stream=eng.side_stream2 if adv_forks else None)
            self._probe_next = None
            self._prefetch_join = True

        cur, ld = x, ldx
        if self.slabs_ahead is not None:
            # this step's own first product exists already (the previous step computed it): the next step's starts at
            # once and has the whole forward chain beside it; otherwise it follows this step's own first GEMM
            emit_prefetch()
        for i, l in enumerate(self.enc_layers):
            ahead = self.slabs_ahead if i == 0 else None
            if ahead is not None and tuple(ahead.shape) != (self._plan_gemm(NT, B, l.n_out, l.n_in), B, l.n_out):
                raise _lib.HipLibraryError("engine: slabs computed ahead do not fit this step's first layer")
            self._probe_next = "enc_l1_fwd" if (i == 0 and ahead is None) else None
            cur = self.fwd_layer(f"{self.eid}.enc{i}" if i < self.n_expert_enc else f"vae.enc{i}", l, cur, ld, B,
                                 training=train, mask_stream=i, split_job=self._next_x_split_job(l, B), slabs_from=ahead)
            self._probe_next = None
            ld = l.n_out
            if i == 0:
                self._mark("enc L1 forward done")
                if self.slabs_ahead is None:
                    emit_prefetch()
        q, HV = cur, self.enc_layers[-1].n_out
        # ---- heads + reparameterisation
        self.mu = eng.buf("mu", (B, Z))
        self.a_raw = eng.buf("a_raw", (B, Z))
        self.std = eng.buf("std", (B, Z))
        self.z = eng.buf("z", (K, B, Z))
        self.kl_row = eng.buf("kl_row", (B,))
        self.stat = eng.buf("stat", (2, B))
        if not self._gemm_group([
                (NT, B, Z, HV, q, HV, self.mean_enc.weight, HV, self.mu, Z, self.mean_enc.bias, 0, 1.0),
                (NT, B, Z, HV, q, HV, self.var_enc.weight, HV, self.a_raw, Z, self.var_enc.bias, 0, 1.0)]):
            self.gemm(NT, B, Z, HV, q, HV, self.mean_enc.weight, HV, self.mu, Z, bias=self.mean_enc.bias)
            self.gemm(NT, B, Z, HV, q, HV, self.var_enc.weight, HV, self.a_raw, Z, bias=self.var_enc.bias)
        self._emit(lib.mmvae_reparam_kl_fwd, B, Z, K, _p(self.mu), _p(self.a_raw), _p(self.eps), self.var_eps,
                   _p(self.std), _p(self.z), _p(self.kl_row), _p(self.stat))
        self._mark("reparameterised (fork point)")
        if self.mode == "embed":  # predict path: the program ends at z
            return self._finish_forward_only()
        if self.iwae:  # sampled log q(z) - log p(z) per (sample, cell)
            self.logratio = eng.buf("iwae.logratio", (K, B))
            self._emit(lib.mmvae_iwae_logratio, B, Z, K, _p(self.std), _p(self.eps), _p(self.z), _p(self.logratio))
        # ---- adversaries (fused passes): they read the hidden representations only -- h1 and z exist from here on.  In
        # the single-rank program of the measured geometry the seven launches of both phases run on the branch stream
        # beside the decoder's forward and the reconstruction GEMM, and are joined where the backward pass first reads
        # the reversed gradients; elsewhere they keep their place behind the ELBO.
        hidden = [l.a if l.a is not None else l.d for l in self.enc_layers if l.return_hidden]
        if self.hidden_z:
            hidden.append(self.z)  # first sample (rows 0..B-1)
        self.adv_grad_into: Dict[int, torch.Tensor] = {}
        adv_calls, adv_aside = None, False
        # Under a gradient exchange the adversaries' optimisers cut the program (an all-reduce of each adversary's gradients
        # in both phases).  Their whole section -- captured segments and exchange points -- then runs as a LANE of its own
        # on the branch stream, started by the host at this point of the program and joined where the chain reads the
        # reversed gradients (PlanRun._run_program): its launches and the latency of its four small all-reduces leave the
        # main stream.  (C4 on one rank with the slice of a world of 8: 1.33 -> see DESIGN section 8.)
        adv_dp = eng.overlap or any(o.reducer is not None for o in self.opt_adv)
        adv_lane = False
        if (self.has_adv and train and eng.adv_fused and adv_dp and eng.settings.adv_aside and eng.lane_stream is not None
                and K == 1 and self.cond is None):
            main_segments, main_cur = self.segments, self._cur
            self.segments, self._cur = [], []
            slot_before = dict(self.metric_slots)
            self._mark("adversaries' lane: first launch")
            built = self._build_adversaries_fused(hidden)
            self._mark("adversaries' lane: done")
            if built:
                lane = [it for it in self.segments + [self._cur] if isinstance(it, tuple) or it]
                self.segments, self._cur = main_segments, main_cur
                self._host_marker(("lane", eng.lane_stream, lane))
                adv_lane = True
            else:
                self.segments, self._cur = main_segments, main_cur
                self.metric_slots = slot_before
        if self.has_adv and train and eng.adv_fused and not adv_dp:
            start = len(self._cur)
            if self._build_adversaries_fused(hidden):
                adv_calls = self._take(start)
                adv_calls = [c for c in [self._mark_call("adversaries: first launch")] if c] + adv_calls + \
                            [c for c in [self._mark_call("adversaries: done")] if c]
                adv_aside = bool(eng.settings.adv_aside and eng.side_stream is not None and not eng.overlap
                                 and eng.world == 1 and K == 1 and big and (measured or eng.side_dw_any)
                                 and self.cond is None)
                if adv_aside:
                    self._fork()
        # ---- forward, decoder side (rows R = K*B)
        cur, ld = self.z, Z
        if self.cond is not None:  # CLVAE.after_reparameterize: the sample passes through the conditional layers
            cur, ld = self.cond.emit_forward(self.z)
        for i, l in enumerate(self.dec_layers[:-1]):
            cur = self.fwd_layer(f"{self.eid}.dec{i}.K{K}", l, cur, ld, R, training=train,
                                 mask_stream=len(self.enc_layers) + i,
                                 planes_out=self.hp if (self.hp is not None and K == 1 and i == len(self.dec_layers) - 2) else None,
                                 split_job=None if (self.hp is not None and K == 1 and i == len(self.dec_layers) - 2)
                                 else self._next_x_split_job(l, R))
            ld = l.n_out
        for job in getattr(self, "_x_split_jobs", []):  # tails the chain did not have: passes of their own
            self._emit(lib.mmvae_split_planes_f32, *job)
        self._x_split_jobs = []
        if getattr(self, "_prefetch_join", False) and eng.settings.prefetch_join:
            # (diagnostics) join the next step's product ahead of the reconstruction launch.  Default: no join here -- a
            # cross-stream join inside the captured program costs ~30 us on this runtime; the branch stream is in order,
            # so the decoder's weight gradient queues behind the product anyway, and the join ahead of the expert's
            # optimiser covers it.  A product that outlasts the forward chain delays the reconstruction launch's
            # workgroups on the CUs it still holds by its remainder, no more.
            self._mark("forward chain done (prefetch joined)")
            self._join()
        self._prefetch_join = False
        last = self.dec_layers[-1]
        fused_last = last.relu and last.bn is None and last.p == 0
        if not fused_last:
            raise _lib.HipLibraryError("engine: the last decoder layer must be Linear+ReLU (fused recon epilogue)")
        last.inp, last.ld_inp, last.rows = cur, ld, R
        T = lib.mmvae_recon_tiles(G)
        # dP [R, G]: with a gene count off the 32-wide k-tile (60 530, 52 437, 30 000) its leading dimension is rounded up
        # to 32 -- the columns in between are zeros nobody writes -- so that the input-gradient product dP . W runs its K
        # over whole k-tiles instead of a tail slab on the element-guarded kernel (12 us + a launch boundary per step).
        # The weight rows "beyond" W that those zero columns meet are the decoder bias behind it in the arena (finite).
        Gp = (G + 31) // 32 * 32
        if not (train and Gp != G and big and lib.mmvae_gemm_get_precision() == 1
                and (Gp - G) * last.n_in <= last.b.numel() + 32
                and last.b.data_ptr() == last.W.data_ptr() + 4 * last.W.numel()):
            Gp = G
        self.ldp = Gp
        self.dP = eng.buf(f"dP.{G}", (R, Gp))[:, :G] if train else None
        self.se_part = eng.buf(f"se_part.{G}", (T, R))
        self.w = eng.buf("w", (R,))
        # K = 1: the decoder bias's gradient is the column sum of dP; the epilogue that stores dP leaves its per-row-tile
        # partials (one reduction job instead of a 41 MB pass).  K > 1 re-weights the rows of dP first: separate pass.
        self.dp_colpart = None
        if train and K == 1:
            nrt = lib.mmvae_recon_row_tiles(R)
            self.dp_colpart = eng.buf(f"dP.colpart.{G}", (nrt, G))
            self._defer_sum(self.dp_colpart, nrt, G, 1, G, G, last.gb, G)
        h_in, ld_h, kpad = cur, ld, False
        if last.n_in % 32 != 0 and lib.mmvae_gemm_get_precision() == 1:
            # a hidden width off the 32-wide k-tile (1000): the fused kernel's pipelined loop needs whole k-tiles --
            # it gets a copy of h in a buffer padded with zero columns (a 2 MB pass: ~4 us) and runs over the padded
            # K; the weights' rows are read on into the next row / the arena's slack, against those zeros
            # (mmvae_recon_set_h_kpad; the guarded loop it replaces: 242 against 123 us at C2's sizes)
            Kp = (last.n_in + 31) // 32 * 32
            hpad = eng.buf(f"hpad.{R}.{last.n_in}", (R, Kp))
            # (reductions / grouped GEMMs queued so far wait for their own flush)
            pending, self._sum_jobs, pending_g, self._gemm_jobs = self._sum_jobs, [], self._gemm_jobs, []
            self._defer_sum(cur, 1, 0, R, last.n_in, ld, hpad, Kp)
            self._flush_sums()
            self._sum_jobs, self._gemm_jobs = pending, pending_g
            h_in, ld_h, kpad = hpad, Kp, True
        self._emit(lib.mmvae_decoder_recon_rows_colsum_f32, R, B, G, last.n_in, _p(h_in), ld_h, _p(last.W), last.n_in,
                   _p(last.b), _p(x), ldx, None, 0, _p(self.dP), self.ldp, _p(self.se_part), _p(self.dp_colpart),
                   probe=("dec_l2_recon", 2.0 * R * G * last.n_in))
        if kpad:  # the launch state brackets the launch
            launch = self._cur.pop()

            def recon_kpad(launch=launch):
                lib.mmvae_recon_set_h_kpad(1)
                try:
                    launch()
                finally:
                    lib.mmvae_recon_set_h_kpad(0)

            self._cur.append(recon_kpad)
        self.probe_meta["dec_l2_recon"].update(bound="mfma", cus=0, planes="",
                                               shape=f"NT {R}x{G}x{last.n_in} + reconstruction epilogue")
        self._mark("reconstruction done")
        self.recon_row = eng.buf("recon_row", (B,))
        if self.iwae:
            self.rows3 = eng.buf("iwae.rows3", (3, B))
            self._emit(lib.mmvae_elbo_finalize_iwae, B, K, T, _p(self.se_part), _p(self.logratio), _p(self.stat), Z,
                       _p(eng.klw_dev), 1.0, _p(self.metrics), _p(self.w), _p(self.rows3))
        else:
            if early_branch and not self.has_adv:
                # K = 1: the backward pass starts from dP, which the reconstruction epilogue has already written; the
                # loss words are only logged -- their two launches leave the critical path
                loss_aside = True
            start = len(self._cur)
            self._emit(lib.mmvae_elbo_finalize, B, K, T, _p(self.se_part), _p(self.kl_row), _p(self.stat), Z,
                       _p(eng.klw_dev), 1.0, _p(self.metrics), _p(self.w), _p(self.recon_row))
            if loss_aside:
                early_calls = self._take(start)
        if not train:  # validation: the program ends with the ELBO terms in the metrics buffer
            return self._finish_forward_only()
        # ---- adversarial phases: total loss = ELBO loss + adv_weight * sum of the generator-phase losses
        # (cmmvae_model.py:182-184; without adversaries it IS the ELBO loss word: no launch)
        self.dz_lat = eng.buf("dz_lat", (R, Z))
        if not self.has_adv:
            self.metric_slots["total_loss"] = 0
        elif adv_lane:
            pass  # (joined below, where the chain first reads the reversed gradients)
        elif adv_calls is None and eng.adv_fused and adv_dp and self._build_adversaries_fused(hidden):
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 1.0, self.mptr("total_loss"))
        elif adv_calls is None:  # per-layer program: the slot starts as the ELBO loss, every generator phase adds to it
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 0.0, self.mptr("total_loss"))
            self._build_adversaries(hidden)
        elif adv_aside:  # (emitted behind the main-stream work it runs beside: the executor enqueues in emission order)
            self._branch(eng.side_stream, adv_calls)
        else:
            self._cur.extend(adv_calls)
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 1.0, self.mptr("total_loss"))

        # ---- backward, decoder side
        dw_inp, dw_ld, w_rows = last.inp, last.ld_inp, None
        if K > 1:
            # The gradient at the decoder's output is diag(w) dP (w = softmax weights of the K-sample bound, known only
            # once every sample's loss is).  dP [K B, G] is not rewritten for it (that pass read and wrote 2 x 410 MB
            # at C3: 350 us of a 4.0 ms step): the bias gradient w^T dP is one read-only pass, the weight gradient
            # dP^T diag(w) h takes h scaled by w (a [K B, 1024] pass), and the input gradient diag(w) (dP W) gets its
            # rows scaled where the next layer's column kernel sums the split-K slabs.
            w_rows = self.w
            nch = lib.mmvae_weighted_colsum_chunks(R)
            parts = eng.buf(f"wcolsum.{G}", (nch, G))
            self._emit(lib.mmvae_weighted_colsum_f32, R, G, _p(self.dP), self.ldp, _p(self.w), _p(parts))
            self._defer_sum(parts, nch, G, 1, G, G, last.gb, G)
            dw_inp = eng.buf(f"h_weighted.{last.n_in}", (R, last.n_in))
            dw_ld = last.n_in
            self._emit(lib.mmvae_scale_rows, R, last.n_in, _p(last.inp), last.ld_inp, _p(self.w), _p(dw_inp), dw_ld)
            if self.hp is not None:  # B of the weight gradient as planes: of the SCALED activations (a 21 + 31 MB pass at C3)
                self._emit(lib.mmvae_split_planes_f32, R, last.n_in, _p(dw_inp), dw_ld, *self.hp.args())
        dw_pl = (None, self.hp) if self.pl_dec_h else None
        # (adversaries on the first branch stream: the capped weight gradient beside the chain takes the second one)
        dw_stream = eng.side_stream2 if adv_aside else None
        dw4_late = None
        if adv_aside and dw_stream is not None and eng.settings.adv_aside >= 2:
            # The branch stream carries the adversaries until the chain needs their reversed gradients; the weight
            # gradient is forked THERE, on a second branch beside the rest of the chain (reparameterisation, heads, the
            # encoder's BatchNorm layers: ~105 us), capped to the workgroup count that keeps its 500 work items at
            # three rounds.  C4 on one box: 1.183 -> 1.135 ms at 185, 1.132 at 170; 1.17 at 125, 1.20 at 150, 1.16 at
            # 200.  Forked right behind the input gradient instead -- beside the adversaries' generator phase -- the
            # step is 8 % SLOWER (1.25 ms): the three lanes fight over the CUs (profiles/r4_c4_dw_aside.txt).
            self._probe_next = "dec_l2_dx"
            S = self.gemm_raw(NN, R, last.n_in, self.ldp, self.dP, self.ldp, last.W, last.n_in)
            self._probe_next = None
            self._mark("decoder dX done")

            def dw4_late():
                self._probe_next = "dec_l2_dw"
                if not self._fuse_sqnorm(TN, G, last.n_in, self.kpad(R), 1.0, self.dP, self.ldp, last.inp, last.ld_inp, last.gW,
                                         last.n_in, None, 0, side_cap=ADV_DW_CAP, planes=dw_pl, stream=dw_stream):
                    self.gemm(TN, G, last.n_in, self.kpad(R), self.dP, self.ldp, last.inp, last.ld_inp, last.gW, last.n_in,
                              side=True, planes=dw_pl)
                self._probe_next = None
        elif side_dw:  # input gradient first (the chain waits for it), then the weight gradient on the side branch
            self._probe_next = "dec_l2_dx"
            S = self.gemm_raw(NN, R, last.n_in, self.ldp, self.dP, self.ldp, last.W, last.n_in)
            self._probe_next = "dec_l2_dw"
            if not self._fuse_sqnorm(TN, G, last.n_in, self.kpad(R), 1.0, self.dP, self.ldp, last.inp, last.ld_inp, last.gW,
                                     last.n_in, None, 0, side_cap=side_dw, planes=dw_pl):
                self.gemm(TN, G, last.n_in, self.kpad(R), self.dP, self.ldp, last.inp, last.ld_inp, last.gW, last.n_in, side=True,
                          planes=dw_pl)
            self._probe_next = None
            self._mark("decoder dX done")
            if early_branch:  # behind the weight gradient on its stream: one branch, in order (probe: DESIGN.md 5)
                self._fork()  # (the weight gradient may have stayed on the main stream)
                self._branch(eng.side_stream, early_calls + [c for c in [self._mark_call("decoder dW + loss words done (branch)")] if c])
        elif (eng.side_dw_dp and eng.overlap and eng.side_stream is not None and train and K == 1 and not self.has_adv
              and big and (measured or eng.side_dw_any) and self.cond is None
              and self._plan_gemm(TN, G, last.n_in, self.kpad(R)) == 1):
            # exchange program: input gradient first, the weight gradient capped on the side stream beside the chain up
            # to the VAE's exchange point (the cut there joins it)
            self._probe_next = "dec_l2_dx"
            S = self.gemm_raw(NN, R, last.n_in, self.ldp, self.dP, self.ldp, last.W, last.n_in)
            self._probe_next = "dec_l2_dw"
            self._side_capped_gemm(TN, G, last.n_in, self.kpad(R), self.dP, self.ldp, last.inp, last.ld_inp, last.gW, last.n_in,
                                   eng.side_dw_dp, planes=dw_pl)
            self._probe_next = None
        else:
            self._probe_next = "dec_l2_dw" if big else None
            self.gemm(TN, G, last.n_in, self.kpad(R), self.dP, self.ldp, dw_inp, dw_ld, last.gW, last.n_in, side=True,
                      planes=dw_pl)
            self._probe_next = "dec_l2_dx" if big else None
            S = self.gemm_raw(NN, R, last.n_in, self.ldp, self.dP, self.ldp, last.W, last.n_in,
                              planes=(None, self.wp) if self.pl_dec_w else None)
            self._probe_next = None
        rest = self.dec_layers[:-1]
        for j in range(len(rest) - 1, -1, -1):
            l = rest[j]
            rs = w_rows if j == len(rest) - 1 else None  # (the rows of the slabs coming out of the last layer)
            if j > 0:
                S = self.bwd_layer(l, None, S, need_dx="raw", row_scale=rs)
            else:
                self.bwd_layer(l, None, S, need_dx="full", row_scale=rs,
                               dx_out=self.dz_lat if self.cond is None else self.cond.d_out)
        if self.cond is not None:
            self.cond.emit_backward(self.dz_lat)
            if mdist.collectives_active() and train:
                # blocks another rank saw and this one did not: zeros into the exchange (job table of this step)
                self._emit(lib.mmvae_grad_zero_flagged_jobs, self.cond.max_jobs, self.cond.jobs_ptr,
                           _p(self.cond.opt.arena.grad))
        if not rest:
            raise _lib.HipLibraryError("engine: decoder needs at least two layers")
        if adv_lane:
            self._mark("chain reaches the adversaries' join")
            self._host_marker(("lane_join", eng.lane_stream))
            self._mark("adversaries joined")
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 1.0, self.mptr("total_loss"))
        if adv_aside:  # the adversaries' branch: its reversed gradients are read from here on
            self._mark("chain reaches the adversaries' join")
            self._join(only=eng.side_stream)
            self._mark("adversaries joined")
            if dw4_late is not None:
                dw4_late()
                c = self._mark_call("decoder dW done (second branch)")
                if c is not None:
                    self._branch(dw_stream, [c])
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 1.0, self.mptr("total_loss"))
        # gradient-reversed adversary gradient on z (first sample) joins here
        zi = self.adv_grad_into.get(id(self.z))
        if zi is not None:
            self._emit(lib.mmvae_axpby, B * Z, 1.0, _p(zi), 1.0, _p(self.dz_lat))
        # ---- reparameterisation + heads backward
        dm = eng.buf("dmu_da", (2, B, Z))  # one buffer: the two heads' bias column sums are one pass over [2B, Z]
        self.dmu, self.da = dm[0], dm[1]
        dq2 = eng.buf("dq", (2, B, HV))    # one slab per head: the next layer's tail sums them on the fly
        self.dq = dq2[0]
        if self.iwae:
            # d/dz of the log-ratio joins the decoder's gradient, its direct dependence on the variance arrives as
            # dstd_extra; the analytic-KL terms of the kernel are switched off (kl_scale_host = 0)
            self.dstd_extra = eng.buf("iwae.dstd", (B, Z))
            self._emit(lib.mmvae_iwae_bwd_terms, B, Z, K, _p(eng.klw_dev), 1.0, _p(self.w), _p(self.z), _p(self.std),
                       _p(self.dz_lat), _p(self.dstd_extra))
            self._emit(lib.mmvae_reparam_kl_bwd, B, Z, K, _p(self.mu), _p(self.std), _p(self.eps), _p(self.dz_lat), None,
                       _p(self.dstd_extra), None, _p(eng.klw_dev), 0.0, self.var_eps, _p(self.dmu), _p(self.da))
        else:
            self._emit(lib.mmvae_reparam_kl_bwd, B, Z, K, _p(self.mu), _p(self.std), _p(self.eps), _p(self.dz_lat), None,
                       None, None, _p(eng.klw_dev), 1.0 / B, self.var_eps, _p(self.dmu), _p(self.da))
        if B % 32 == 0:
            self._emit_colsum_pair(B, Z, dm, eng.grad_of(self.mean_enc.bias), eng.grad_of(self.var_enc.bias))
        else:
            for dy, lin in ((self.dmu, self.mean_enc), (self.da, self.var_enc)):
                self._emit_fc_bwd(B, Z, dy, None, None, None, eng.grad_of(lin.bias))
        for dy, lin in ((self.dmu, self.mean_enc), (self.da, self.var_enc)):
            self.gemm(TN, Z, HV, self.kpad(B), dy, Z, q, HV, eng.grad_of(lin.weight), HV, side=True)
        # ---- backward, encoder side
        if self._gemm_group([(NN, B, HV, Z, self.dmu, Z, self.mean_enc.weight, HV, dq2[0], HV, None, 0, 1.0),
                             (NN, B, HV, Z, self.da, Z, self.var_enc.weight, HV, dq2[1], HV, None, 0, 1.0)]):
            din, S = dq2, 2
        else:
            self.gemm(NN, B, HV, Z, self.dmu, Z, self.mean_enc.weight, HV, self.dq, HV)
            self.gemm(NN, B, HV, Z, self.da, Z, self.var_enc.weight, HV, self.dq, HV, flags=ACC)
            din, S = self.dq, 1
        early = eng.overlap
        if early and len(self.enc_layers) == self.n_expert_enc:  # no VAE-encoder layers: VAE gradients are final
            self._begin_exchange(self.opt_vae)
        for j in range(len(self.enc_layers) - 1, -1, -1):
            l = self.enc_layers[j]
            hid = l.a if l.a is not None else l.d
            addend = self.adv_grad_into.get(id(hid)) if l.return_hidden else None
            if j == 0 and side_late:
                self._defer_next_dw = True
            S_next = self.bwd_layer(l, din, S, addend=addend, need_dx="raw" if j > 0 else "none",
                                    dz_planes=self.dYp if (j == 0 and self.pl_enc) else None,
                                    inp_planes=self.xp if (j == 0 and self.pl_enc) else None)
            din, S = None, S_next
            if early and j == self.n_expert_enc:  # the last VAE layer is done: what remains is the expert's encoder
                self._begin_exchange(self.opt_vae)
        self._mark("backward chain done")
        # ---- clip + Adam (reference order: clip vae, clip expert, step vae, step expert)
        # Logged scalars: the step's metrics words (and the pre-clip gradient norms) are copied into a buffer of this
        # plan's own as the last node(s) of the captured program -- the logged tensors are views of it, valid until
        # this plan's next step -- instead of three D2D copies issued by the host behind every replay (~24 us).
        self.log_buf = torch.zeros(256, dtype=torch.float32, device=eng.device)

        def emit_log_copy():
            self._emit(lib.mmvae_axpby, 256, 1.0, _p(self.metrics), 0.0, _p(self.log_buf))

        dw = getattr(self, "_deferred_dw", None)
        self._deferred_dw = None
        late_branch = bool(side_late and dw is not None and self.cond is None)
        start = len(self._cur)
        self.optimizer(self.opt_vae, self.clip_vae, exchange="wait" if early else "inline",
                       join=not late_branch)
        self.log_norm(self.opt_vae, "grad_norms/vae")
        if late_branch:
            # the shared VAE's clip + Adam (a chain of small launches) beside the expert encoder's G-wide weight
            # gradient, whose persistent grid is capped to the workgroup count that keeps its number of rounds
            # Emission order matters to the graph executor: the weight gradient is enqueued first and the branch
            # forks from an event recorded ahead of it (a branch enqueued first made the GEMM wait for the branch's
            # last node; uncapped, the branch starves behind the GEMM's one-workgroup-per-CU grid -- timelines in
            # profiles/r2_branch_order.txt).
            calls = self._take(start)
            self._fork()  # the branch depends on the chain up to here; its kernels are enqueued behind the GEMM
            layout, M, N, Kk, A, lda, Bm, ldb, Cm, ldc = dw
            dwp = getattr(self, "_deferred_dw_planes", None)
            self._probe_next = "enc_l1_dw"
            if not self._fuse_sqnorm(layout, M, N, Kk, 1.0, A, lda, Bm, ldb, Cm, ldc, None, 0, side_cap=eng.side_dw2,
                                     on_side=False, planes=dwp):
                self.gemm(*dw, side=True, planes=dwp)
            self._probe_next = None
            self._branch(eng.side_stream, calls + [c for c in [self._mark_call("VAE optimiser done (branch)")] if c])
        elif dw is not None:
            dwp = getattr(self, "_deferred_dw_planes", None)
            self._probe_next = "enc_l1_dw" if big else None
            self.gemm(*dw, side=True, planes=dwp)
            self._probe_next = None
        if early:  # the expert's exchange + update leave the main stream: its norm is logged from the comm stream
            emit_log_copy()
        # in-order program: the log copy rides on the expert's Adam launch (its words -- losses, both norms -- are final
        # once adam_prepare has run) when the expert's norm is a state word inside the metrics buffer
        self._mark("encoder dW done / VAE optimiser branch emitted")
        ride = (not early and self.cond is None
                and eng._state_slot.get(id(self.opt_exp)) is not None
                and not (eng.shard and self.opt_exp.reducer is not None))
        self.optimizer(self.opt_exp, self.clip_exp, exchange="deferred" if early else "inline",
                       tail_copy=(256, self.metrics, self.log_buf) if ride else None)
        self._mark("expert optimiser done")
        if early:
            self.exp_norm_log = torch.zeros(1, dtype=torch.float32, device=eng.device)
        else:
            self.log_norm(self.opt_exp, "grad_norms/expert")
            if not ride:
                emit_log_copy()
        self.segments.append(self._cur)
        self._cur = []
        # noise: Philox fills (production) or explicit buffers (parity mode), at the head of the program
        if not self.explicit:
            n_max = K * B * Z
            fills = []  # every keep-mask and the rsample noise of the step: one launch (same numbers as one fill each)
            for l, stream in self._mask_layers:  # expert / VAE layers and both phases of every adversary
                n_max = max(n_max, l.mask.numel())
                fills.append(_lib.PhiloxJob(_p(l.mask), l.mask.numel(), rng.STREAM_DROPOUT + stream, l.p, 0))
            fills.append(_lib.PhiloxJob(_p(self.eps), K * B * Z, rng.STREAM_NORMAL, 0.0, 1))
            arr = (_lib.PhiloxJob * len(fills))(*fills)
            jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(eng.device)
            self._job_tables.append(jobs_dev)
            # (one launch: the fill's last workgroup -- a ticket -- advances the Philox counter)
            ticket = eng.buf("philox.ticket", (1,), torch.int32)
            self._emit(lib.mmvae_philox_fill_jobs_advance, len(fills), jobs_dev.data_ptr(), n_max, _p(self.rng_state),
                       (n_max + 3) // 4, _p(ticket))
            self.segments[0] = self._cur + self.segments[0]
            self._cur = []
        self._size_workspaces()

    def _finish_forward_only(self):
        """Close a forward-only program: rsample noise at its head, shared workspaces sized."""
        eng, lib = self.eng, self.lib
        self.exp_norm_log = None
        self.has_adv = False
        self.segments.append(self._cur)
        self._cur = []
        if not self.explicit:
            n = self.K * self.B * self.Z
            # (one launch: the fill's last workgroup advances the counter)
            arr = (_lib.PhiloxJob * 1)(_lib.PhiloxJob(_p(self.eps), n, rng.STREAM_NORMAL, 0.0, 1))
            jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(eng.device)
            self._job_tables.append(jobs_dev)
            self._emit(lib.mmvae_philox_fill_jobs_advance, 1, jobs_dev.data_ptr(), n, _p(self.rng_state),
                       (n + 3) // 4, _p(eng.buf("philox.ticket", (1,), torch.int32)))
            self.segments[0] = self._cur + self.segments[0]
            self._cur = []
        self._size_workspaces()

    def _size_workspaces(self):
        eng = self.eng
        self.fcws = eng.buf("fc_ws", (max(getattr(self, "_fcws_bytes", 0) // 4, 1),))
        for key, t in eng._pool.items():
            if key[0] == "fc_ws" and t.numel() > self.fcws.numel():
                self.fcws = t
        self.ws = eng.buf("gemm_ws", (max(self._ws_bytes // 4, 1),))
        self.slab = eng.buf("gemm_slabs", (max(self._slab_floats, 1),))
        # a shared buffer may have been re-allocated larger by a later plan: always take the biggest one
        for key, t in eng._pool.items():
            if key[0] == "gemm_ws" and t.numel() > self.ws.numel():
                self.ws = t
            if key[0] == "gemm_slabs" and t.numel() > self.slab.numel():
                self.slab = t
