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
per-step HOST values, uploaded into static device tables before the replay (_CondProgram).  Configurations outside this
shape (LayerNorm / non-ReLU activations in the FC blocks, conditional blocks that are not one Linear, conditional
layers under data parallelism) use the module path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib, cond_tables, dist as mdist, rng
from .constants import REGISTRY_KEYS as RK
from .modules.base.components import Adversarial, ConditionalLayer, FCBlock, _identity
from .optim import HipAdam, arena_of

NT, NN, TN = _lib.GEMM_NT, _lib.GEMM_NN, _lib.GEMM_TN
RAW = _lib.GEMM_RAW_SLABS
ACC = _lib.GEMM_ACCUMULATE
RELU = _lib.GEMM_RELU
SLACK = _lib.GEMM_OPERAND_SLACK  # every operand the engine hands to a GEMM has 16 readable bytes behind it
MAX_POINTER_PLANS = 8


def _p(t):
    return None if t is None else t.data_ptr()


def _s():
    return torch.cuda.current_stream().cuda_stream


class _PinnedRing:
    """Page-locked staging slots for per-step host tables.  Pinning a fresh tensor per step costs 0.2-0.8 ms on this
    runtime (measured; the copy itself is ~4 us to enqueue), so the slots are allocated once and reused round-robin;
    a slot is rewritten only after the copy that last read it has completed (event)."""

    def __init__(self, numel: int, dtype=torch.int32, slots: int = 4):
        self.slots = [torch.zeros(numel, dtype=dtype).pin_memory() for _ in range(slots)]
        self.views = [t.numpy() for t in self.slots]
        self.events = [None] * slots
        self.i = 0

    def take(self):
        """The next slot as a numpy array (its previous upload has completed)."""
        self.i = (self.i + 1) % len(self.slots)
        ev = self.events[self.i]
        if ev is not None:
            ev.synchronize()
        return self.views[self.i]

    def upload(self, dst: torch.Tensor) -> None:
        dst.copy_(self.slots[self.i], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.i] = ev


class _LayerRef:
    """One FCBlock layer bound to its parameter / gradient-arena tensors."""

    def __init__(self, seq: nn.Sequential, grad_of, return_hidden: bool, block=None, index: int = 0):
        self.block, self.index = block, index  # the FCBlock it belongs to: explicit keep masks are looked up there
        lin = seq.lin
        self.n_in, self.n_out = lin.in_features, lin.out_features
        self.W, self.b = lin.weight, lin.bias
        self.gW, self.gb = grad_of(lin.weight), grad_of(lin.bias)
        bn = getattr(seq, "bn", None)
        self.bn = bn
        if bn is not None:
            self.ggamma, self.gbeta = grad_of(bn.weight), grad_of(bn.bias)
        self.relu = isinstance(getattr(seq, "af", None), nn.ReLU)
        dr = getattr(seq, "dr", None)
        self.p = float(dr.p) if dr is not None else 0.0
        self.return_hidden = return_hidden and hasattr(seq, "af")


def _supported_block(block: FCBlock) -> bool:
    for seq in block.fc_layers:
        if hasattr(seq, "ln"):
            return False
        af = getattr(seq, "af", None)
        if af is not None and not isinstance(af, nn.ReLU):
            return False
    return True


class _PlaneBuf:
    """Three bf16 planes of an engine buffer [rows, cols] (int16 [3, rows + 32, cols]); the 32 slack rows of every plane
    stay zero: a weight-gradient GEMM runs its K over them (kpad)."""

    def __init__(self, eng: "StepEngine", name: str, rows: int, cols: int):
        self.rows, self.cols, self.ld = rows, cols, cols
        self.data = eng.buf(name, (3, rows + 32, cols), torch.int16)
        self.pstride = (rows + 32) * cols

    def ptr(self) -> int:
        return self.data.data_ptr()

    def args(self):
        """(planes pointer, leading dimension, plane stride) as the C-ABI takes them"""
        return self.data.data_ptr(), self.ld, self.pstride


_NOPL = (None, 0, 0)


class StepEngine:
    @staticmethod
    def try_build(model) -> Optional["StepEngine"]:
        m = model.module
        cl = getattr(m.vae, "conditionals", None)
        if cl is not None and os.environ.get("MMVAE_ENGINE_CONDITIONALS", "1") == "0":
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
            # ranks see different conditions: the blocks that step are the UNION over the ranks (_CondProgram.load);
            # MMVAE_ENGINE_CONDITIONALS_DP=0 sends such a model back to the module path under data parallelism
            if mdist.collectives_active() and os.environ.get("MMVAE_ENGINE_CONDITIONALS_DP", "1") == "0":
                return None
            if not _CondProgram.supported(cl, model.get_optimizers()["vae"], m.vae.encoder.mean_encoder.out_features):
                return None
        return StepEngine(model)

    def __init__(self, model, side_stream: bool = False, defer_expert_adam: Optional[bool] = None):
        self.model = model
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
        self.merge_launches = os.environ.get("MMVAE_MERGE_LAUNCHES", "1") != "0"  # diagnostics: =0 one launch per item
        if len(opts_list) <= 8 and self.merge_launches:
            metrics = self.buf("metrics", (256,))
            for i, opt in enumerate(opts_list):
                view = metrics[192 + 8 * i:200 + 8 * i]
                if opt.state_dev.data_ptr() != view.data_ptr():
                    view.copy_(opt.state_dev)
                    opt.state_dev = view
                self._state_slot[id(opt)] = 192 + 8 * i
        self._ptr_seen: Dict[tuple, int] = {}
        self.klw_dev = torch.ones(1, dtype=torch.float32, device=self.device)
        self._klw_host = None
        # weight-gradient GEMMs run on a side stream inside the captured graph (fork/join edges)
        self.batch_finish = os.environ.get("MMVAE_BATCH_FINISH", "1") != "0"
        self.batch_gemms = os.environ.get("MMVAE_BATCH_GEMMS", "1") != "0"
        self.fuse_sqnorm = os.environ.get("MMVAE_FUSE_SQNORM", "1") != "0"
        self.cond_packed_exchange = os.environ.get("MMVAE_COND_PACKED_EXCHANGE", "1") != "0"  # 0: dense VAE arena
        self.fuse_norm_prepare = os.environ.get("MMVAE_FUSE_NORM_PREPARE", "1") != "0"  # norm pass + adam_prepare: one launch
        self.fuse_dp_colsum = os.environ.get("MMVAE_FUSE_DP_COLSUM", "1") != "0"  # decoder-bias gradient from the recon epilogue
        # Operands of the G-wide GEMMs written once as bf16 planes by their producers (include/mmvae_hip.h, "Pre-split
        # operands") instead of being split inside every GEMM tile that reads them; K = 1 training programs
        self.planes = os.environ.get("MMVAE_PLANES", "1") != "0"
        self.planes_enc = os.environ.get("MMVAE_PLANES_ENC", "1") != "0"  # x, dY -> the first layer's weight gradient
        self.planes_dec = os.environ.get("MMVAE_PLANES_DEC", "0") != "0"  # dP, h -> the last layer's dW and dX
        self.planes_dec_h = os.environ.get("MMVAE_PLANES_DEC_H", "1") != "0"  # h alone -> B of the last layer's dW
        if os.environ.get("MMVAE_SIDE_STREAM", "0") != "0":
            side_stream = True
        self.side_stream_asked = bool(side_stream)  # the caller / environment asked for it (not only the dW branch)
        # The decoder's G-wide weight gradient (dW = dP^T h, ~105 us at C2, needed by the optimiser only) on a second
        # stream beside the backward chain of the core layers (~150 us of latency-bound launches that leave most CUs
        # idle): the persistent GEMM kernel is launched with its grid capped to `side_dw` workgroups = CUs, the chain
        # gets the rest.  Inside the captured graph (a forked branch joined ahead of the expert's optimiser).  Used by
        # the in-order single-rank program only.  Measured at C2 (profiles/r2_side_dw_sweep.txt): cap 125 -> -2.4 %,
        # 140/167 -> -0.8 %, 200 -> +2.6 %; bit-identical results.  0 = off.
        self.side_dw = int(os.environ.get("MMVAE_SIDE_DW", "125"))
        # cap of the expert encoder's weight gradient while the shared VAE's optimiser runs beside it (553 items at C2:
        # 3 rounds on 185 workgroups as on 256), and the switch for the small branches (loss words, bias column sums,
        # VAE optimiser) on a second branch stream
        self.side_dw2 = int(os.environ.get("MMVAE_SIDE_DW2", "185"))
        self.side_branches = os.environ.get("MMVAE_SIDE_BRANCHES", "1") != "0"
        # the same branch inside the exchange program (data parallelism): the decoder's weight gradient beside the part
        # of the backward chain that lies ahead of the shared VAE's exchange point (the cut joins it); 0 = in order
        self.side_dw_dp = int(os.environ.get("MMVAE_SIDE_DW_DP", "125"))
        # adversarial programs (C4): the decoder's last layer backward -- weight gradient AND input gradient -- needs only
        # dP, which the reconstruction epilogue has written before the adversaries' phases start: both GEMMs run capped
        # on the side stream beside those phases (~700 us of latency-bound launches), joined ahead of the backward chain
        self.side_dw_adv = int(os.environ.get("MMVAE_SIDE_DW_ADV", "0"))
        self.side_dw_any = os.environ.get("MMVAE_SIDE_DW_ANY", "0") != "0"  # fork outside the measured geometry too
        # adversaries without BatchNorm: both phases of all of them as five launches (_Plan._build_adversaries_fused);
        # 0 = the per-layer program (the path of adversaries with BatchNorm)
        self.adv_fused = os.environ.get("MMVAE_ADV_FUSED", "1") != "0"
        self.side_max_rows = int(os.environ.get("MMVAE_SIDE_MAX_ROWS", "640"))  # see _Plan._build
        if self.side_dw:
            side_stream = True
        self.side_stream = torch.cuda.Stream(device=self.device) if side_stream else None
        # only the small (latency-bound) weight-gradient GEMMs go aside; chip-filling ones stay in order on the main stream
        self.side_max_elems = int(os.environ.get("MMVAE_SIDE_MAX_ELEMS", 2 * 1024 * 1024))
        self._defer_arg = defer_expert_adam
        self.comm_stream = self.small_stream = None
        self._configure_parallel()
        self._sig = self._signature()
        self._pending: Dict[str, torch.cuda.Event] = {}
        # Lazy expert update (MMVAE_DP_LAZY_ADAM=1, off by default): only the expert's all-reduce runs beside the next
        # step; its clip + Adam (HBM-bound, 1.15 GB of traffic at C2) is run on the MAIN stream at the start of that
        # expert's next step, once the reduced gradients are there -- instead of on the communication stream beside
        # the next step.  Measured with single-rank RCCL: 1.319 ms against 1.315 (what the concurrent update hides, it
        # costs the next step's GEMMs in HBM contention), so the default stays the concurrent update, whose logged
        # gradient norm and parameters are final after a device synchronisation rather than after engine.flush().
        self.lazy_adam = os.environ.get("MMVAE_DP_LAZY_ADAM", "0") != "0"
        self._lazy: Dict[str, tuple] = {}

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
        defer = self._defer_arg
        if defer is None:
            ov = os.environ.get("MMVAE_DP_OVERLAP", os.environ.get("MMVAE_DEFER_EXPERT_ADAM", ""))
            defer = (ov != "0") if ov != "" else mdist.collectives_active()
        self.overlap = bool(defer)
        # sharded expert update under data parallelism: reduce-scatter of the gradient arena, clip + Adam on this rank's
        # 1 / world of it, all-gather of the parameters -- the same bytes on the wire as the all-reduce, the 1.2 GB
        # Adam pass world times shorter (MMVAE_DP_SHARD=0: all-reduce + the full update on every rank)
        self.shard = mdist.collectives_active() and os.environ.get("MMVAE_DP_SHARD", "1") != "0"
        # diagnostics (timing only, wrong numbers): on ONE rank, update the slice a rank of a world of N would own and skip
        # the collectives -- what the compute side of the N-rank program costs (bench.py --sim-world)
        self.shard_sim_world = int(os.environ.get("MMVAE_DP_SIM_WORLD", "0")) if self.world == 1 else 0
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
        self.dp_kernels = os.environ.get("MMVAE_DP_KERNELS", "auto")
        if self.dp_kernels not in ("auto", "dynamic", "persistent"):
            raise _lib.HipLibraryError(f"MMVAE_DP_KERNELS={self.dp_kernels!r}: dynamic | persistent | auto")
        if not hasattr(self, "_tune"):
            self._tune = None
            self.dp_tuned: Dict[str, float] = {}
        if "MMVAE_X3W" not in os.environ:
            if not mdist.collectives_active():
                self.lib.mmvae_gemm_set_x3w(-1)
            else:
                choice = self.dp_kernels if self.dp_kernels != "auto" else (self.dp_tuned.get("choice") or "dynamic")
                self.lib.mmvae_gemm_set_x3w(1 if choice == "persistent" else 0)
                tune = (self.dp_kernels == "auto" and "choice" not in self.dp_tuned
                        and (mdist.world_size() > 1 or os.environ.get("MMVAE_DP_AUTOTUNE_FORCE", "0") != "0"))
                if tune and self._tune is None:
                    self._tune = dict(phase="warm", kind="dynamic", steady=0, count=0, ev=None)
        if self.overlap and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.device)
            self.small_stream = torch.cuda.Stream(device=self.device)

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

    DP_TUNE_STEADY, DP_TUNE_STEPS = 4, 12

    def _dp_autotune(self, plan: "_Plan") -> None:
        """MMVAE_DP_KERNELS=auto under a real exchange: time DP_TUNE_STEPS replayed steps with the 2 x 4-wave (hardware-
        scheduled) GEMM kernels, then with the persistent ones, take the maximum over the ranks of each and keep the
        faster.  The steps are ordinary training steps; a switch drops the captured programs (they bake the kernel in)."""
        t = self._tune
        st = torch.cuda.current_stream()
        if t["phase"] == "warm":
            t["steady"] = t["steady"] + 1 if plan._runs >= 3 else 0  # (a third run is a replay of a captured program)
            if t["steady"] >= self.DP_TUNE_STEADY:
                t["ev"] = torch.cuda.Event(enable_timing=True)
                t["ev"].record(st)
                t.update(phase="measure", count=0)
            return
        t["count"] += 1
        if t["count"] < self.DP_TUNE_STEPS:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record(st)
        end.synchronize()
        self.dp_tuned[t["kind"]] = t["ev"].elapsed_time(end) / self.DP_TUNE_STEPS
        if t["kind"] == "dynamic":
            self.lib.mmvae_gemm_set_x3w(1)
            self._drop_train_plans()
            t.update(phase="warm", kind="persistent", steady=0, count=0, ev=None)
            return
        both = torch.tensor([self.dp_tuned["dynamic"], self.dp_tuned["persistent"]], dtype=torch.float64, device=self.device)
        if mdist.world_size() > 1:
            import torch.distributed as tdist

            tdist.all_reduce(both, op=tdist.ReduceOp.MAX)
        d, p = (float(v) for v in both.cpu())
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
        n = int(self.lib.mmvae_sqnorm_partials(opt.arena.numel)) + 4096 + 64
        return self.buf(f"sqparts.{id(opt)}", (n,))

    def close(self) -> None:
        """Release every captured program now (see _Plan.release) instead of when the collector finds the cycles."""
        self.flush()
        torch.cuda.synchronize(self.device)
        for p in self._plans.values():
            p.release()
        self._plans = {}
        self._ptr_seen.clear()

    def flush(self) -> None:
        """Make the current stream wait for every deferred expert update (before parameters are read elsewhere)."""
        for eid in list(self._lazy):
            self._finish_lazy(eid)
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            torch.cuda.current_stream().wait_stream(self.small_stream)
            self._pending.clear()

    def _finish_lazy(self, expert_id: str) -> None:
        """Run the stashed tail of `expert_id`'s last step (clip + Adam over the reduced gradients) on this stream."""
        item = self._lazy.pop(expert_id, None)
        if item is None:
            return
        plan, rest, launch, ev = item
        torch.cuda.current_stream().wait_event(ev)
        for it in rest:
            if isinstance(it, tuple):
                raise _lib.HipLibraryError("engine: exchange marker behind the deferred expert exchange")
            launch(it)
        plan.exp_norm_log.copy_(plan.opt_exp.state_dev[1:2])

    # ------------------------------------------------------------------------------------------------- inputs
    def _select_input(self, x: torch.Tensor, base_key: tuple):
        """Plan selection: graphs are keyed by the input pointer once a pointer has been seen twice (resident
        batches); otherwise the batch is copied into a static buffer.  Returns (plan key, the tensor the plan reads)."""
        B = x.shape[0]
        if x.layout == torch.sparse_csr:  # CSR batch: densified by one HIP pass straight into the static input buffer
            from . import ops

            x_in = self.buf(f"x_static.{base_key[1]}", (B, x.shape[1]))
            ops.csr_to_dense(x, out=x_in)
            return base_key + (0, 0), x_in
        pkey = base_key + (x.data_ptr(), x.stride(0))
        seen = self._ptr_seen.get(pkey, 0)
        self._ptr_seen[pkey] = seen + 1
        n_ptr_plans = sum(1 for k in self._plans if k[-2] != 0)
        # a caller's tensor has no slack behind it: with a gene count that is not a multiple of 4, or a batch that is
        # not a multiple of 32 (kpad reads zero rows behind the batch), it is always staged
        if x.shape[1] % 4 == 0 and B % 32 == 0 and (pkey in self._plans or (seen >= 1 and n_ptr_plans < MAX_POINTER_PLANS)):
            return pkey, x
        x_in = self.buf(f"x_static.{base_key[1]}", (B, x.shape[1]))
        if x.is_contiguous():  # own 16-byte copy kernel: the runtime's blit kernel reaches < 1 TB/s here
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
        self._finish_lazy(expert_id)
        ev = self._pending.pop(expert_id, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        enc_mod = self.model.module.vae.encoder
        explicit = enc_mod.explicit_eps is not None
        B = x.shape[0]
        key, x_in = self._select_input(x, (mode, expert_id, B, 1, explicit))
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
    def training_step(self, x: torch.Tensor, metadata, expert_id: str) -> None:
        model = self.model
        self._check_signature()
        x = self._dense_f32(x)
        enc_mod = model.module.vae.encoder
        expert = model.module.experts[expert_id]
        explicit = enc_mod.explicit_eps is not None or expert.encoder.explicit_masks is not None
        K = int(enc_mod.n_samples)
        if enc_mod.explicit_eps is not None and enc_mod.explicit_eps.dim() == 3:
            K = enc_mod.explicit_eps.shape[0]
        B = x.shape[0]
        iwae = getattr(enc_mod, "elbo_mode", "analytic") == "iwae"
        key, x_in = self._select_input(x, ("train-iwae" if iwae else "train", expert_id, B, K, explicit))
        plan = self._plans.get(key)
        if plan is None:
            plan = _Plan(self, expert_id, B, K, explicit, x_in, iwae=iwae)
            self._plans[key] = plan
        self._set_kl_weight()
        if explicit:
            plan.load_explicit_noise(enc_mod, expert)
        if plan.has_adv:
            plan.load_labels(metadata)
        if plan.cond is not None:
            plan.cond.load(metadata)
        self._finish_lazy(expert_id)  # this expert's previous update (lazy: run here, on the reduced gradients)
        ev = self._pending.pop(expert_id, None)
        if ev is not None:  # this expert's previous (deferred) update must land before its parameters are read
            torch.cuda.current_stream().wait_event(ev)
        self.last_plan = plan  # introspection (tests read the activations the step left in its buffers)
        ev = plan.run()
        if ev is not None:
            self._pending[expert_id] = ev
        if plan.cond is not None:
            plan.cond.commit()
        model.kl_annealing_fn.step()
        plan.log(model, expert_id)
        if self._tune is not None:
            self._dp_autotune(plan)


def _planes_desc(planes) -> str:
    """Which operands of a launch are pre-split planes ("", "A", "B", "A+B"): probe metadata."""
    if not planes:
        return ""
    return "+".join(n for n, p in zip("AB", planes) if p is not None)


class _Plan:
    def __init__(self, eng: StepEngine, eid: str, B: int, K: int, explicit: bool, x: torch.Tensor, mode: str = "train",
                 iwae: bool = False):
        self.eng, self.eid, self.B, self.K, self.explicit = eng, eid, B, K, explicit
        self.mode = mode
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
        self._dirty: List = []          # branch streams with work the main stream has not joined yet
        self._side_foreign = False      # a gradient that is not the active expert's was computed on the side stream
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
        self.use_side = False           # this plan forks work onto the engine's side stream (set in _build)
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
            self.cond = _CondProgram(self, cl, eid, train=(mode == "train"))
        self._build()

    # ------------------------------------------------------------------------------------------- program building
    def _emit(self, fn, *args, probe=None):
        """probe: (tag, work) -- see _probed."""
        lib_fn = fn

        def call():
            rc = lib_fn(*args, _s())
            if rc != 0:
                raise _lib.HipLibraryError(f"{lib_fn.__name__} failed with code {rc}")

        self._cur.append(self._probed(probe[0], probe[1], call) if probe else call)

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
                     on_side: bool = True, planes=None, fork: bool = True) -> bool:
        """Unsplit weight-gradient GEMM straight into a gradient arena: let its epilogue also leave the partial sums of
        squares of what it stores (mmvae_gemm_f32_sq), so that the clip's norm pass does not read the 82 MB back.  Only
        without a gradient exchange: under data parallelism the norm is that of the REDUCED gradients."""
        eng = self.eng
        if not eng.fuse_sqnorm or eng.overlap or eng.world > 1 or ldc != N or (flags & ~ACC):
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
        if base + n_part > 4096:
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
            side = eng.side_stream if on_side else None
            if on_side and fork:
                self._fork()

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
                          sk: int = 1, fork: bool = True) -> None:
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
        side = self.eng.side_stream
        if fork:
            self._fork()

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
        self._gemm_jobs.append(job)
        self._sum_keep.append((A, Bm, Cm, bias))
        return True

    def _gemm_group(self, jobs) -> bool:
        """Independent GEMMs of the planner's 64x64-tile class in ONE launch, in place (not deferred): the two heads of
        the encoder forward and backward.  jobs: (layout, M, N, K, A, lda, B, ldb, C, ldc, bias, flags, alpha).  False
        (nothing emitted) when a job is not of that class."""
        if not self.eng.batch_gemms:
            return False
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
        if side and self.eng.batch_gemms and self._queue_gemm(layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags):
            return
        sk = self._plan_gemm(layout, M, N, K)
        if side and sk > 1 and self.eng.batch_finish and not (flags & ~ACC) and bias is None:
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
        if side and self.use_side and M * N <= self.eng.side_max_elems:
            self._ws_side_bytes = max(getattr(self, "_ws_side_bytes", 0), nbytes)
            hit = self.eng.locate_grad(Cm)
            if hit is None or hit[0] is not self.opt_exp:
                self._side_foreign = True
            self._fork()
            self._emit_gemm(layout, M, N, K, alpha, A, lda, Bm, ldb, Cm, ldc, bias, flags, sk, "side", planes=planes)
            return
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

    def _record_event(self, stream):
        """Event recorded on `stream` (None = the main stream at run time) at this point of the program."""
        ev = torch.cuda.Event()
        self._events.append(ev)

        def call():
            ev.record(stream if stream is not None else torch.cuda.current_stream())

        self._cur.append(call)
        return ev

    def _wait_event(self, ev, stream=None):
        """`stream` (None = main) waits for an event of _record_event: one graph edge from that node only."""
        def call():
            (stream if stream is not None else torch.cuda.current_stream()).wait_event(ev)

        self._cur.append(call)

    def _fork(self, stream=None):
        """A branch stream (default: the side stream) waits for everything enqueued so far on the main stream."""
        side = stream if stream is not None else self.eng.side_stream
        self._edge(None, side)
        self._forked = True
        if side not in self._dirty:
            self._dirty.append(side)

    def _join(self):
        """Main stream waits for every branch with outstanding work (before the optimiser reads the gradient arenas)."""
        for side in self._dirty:
            self._edge(side, None)
        self._dirty = []

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
            ws = plan.ws_side if use_ws == "side" else plan.ws
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

        if use_ws == "side":
            side = self.eng.side_stream

            def call():
                with torch.cuda.stream(side):
                    launch()
        else:
            call = launch
        self._cur.append(call)

    # ---- one FCBlock layer forward: cur [rows, n_in] -> l.d
    def fwd_layer(self, tag: str, l: _LayerRef, cur: torch.Tensor, ld_cur: int, rows: int, training: bool = True,
                  mask_tag: Optional[str] = None, mask_stream: Optional[int] = None, planes_out: Optional[_PlaneBuf] = None,
                  after_gemm=None, inp_planes: Optional[_PlaneBuf] = None, split_job=None):
        """`mask_tag`: name of the keep-mask buffer when it must differ from the layer's other buffers (the two phases
        of an adversary share activations but draw fresh masks); `mask_stream`: its Philox stream id."""
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
        S = self.gemm_raw(NT, rows, l.n_out, l.n_in, cur, ld_cur, l.W, l.n_in, planes=(inp_planes, None) if inp_planes else None)
        if after_gemm is not None:
            after_gemm()
        bnp = None
        if l.bn is not None:
            bn = l.bn
            bnp = _lib.BnParams(_p(bn.weight), _p(bn.bias), _p(bn.running_mean), _p(bn.running_var),
                                _p(bn.num_batches_tracked), float(bn.momentum), float(bn.eps))
            l._bnp = bnp  # keep the struct alive for the lifetime of the plan
        plan = self

        def call():
            args = (rows, l.n_out, plan.slab.data_ptr(), l.n_out, S, _p(l.b),
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
                  dx_alpha=1.0, dz_planes: Optional[_PlaneBuf] = None, inp_planes: Optional[_PlaneBuf] = None):
        """dz_planes / inp_planes: pre-split forms of this layer's output gradient (written by its column kernel) and of
        its input -- both operands of its weight-gradient GEMM."""
        rows = l.rows
        dw_planes = (dz_planes, inp_planes) if (dz_planes is not None and inp_planes is not None) else None
        plan = self
        relu_src = l.a if l.a is not None else l.d
        has_bn = l.bn is not None

        own_ws = None
        if not has_bn and l.gb is not None and self.eng.batch_finish:
            own_ws = self._bias_partials(rows, l.n_out, l.gb)

        def call():
            din_ptr = _p(din) if din is not None else plan.slab.data_ptr()
            ws = own_ws if own_ws is not None else plan.fcws
            # `addend` is a gradient on the hidden representation = the activation BEFORE dropout: it bypasses the mask
            args = (rows, l.n_out, din_ptr, l.n_out, S_in, None, _p(addend), None, _p(l.mask), l.p, int(l.relu),
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
        own_ws = self._bias_partials(rows, N, dbias) if (dbias is not None and self.eng.batch_finish) else None

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
        if opt.reducer is not None or self.eng.overlap:
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
        if self.eng.fuse_norm_prepare and 1 <= len(ranges) <= 4 and opt.reducer is None and not self.eng.overlap:
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
        opt.sharded = True
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
    def _build(self):
        eng, lib = self.eng, self.lib
        B, K, R, Z, G = self.B, self.K, self.R, self.Z, self.G
        x, ldx = self.x, self.x.stride(0) if self.x.shape[0] > 1 else self.x.shape[1]
        self.eps = eng.buf("eps", (K, B, Z))

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
                                  and K == 1 and R <= eng.side_max_rows and not self.has_adv and big
                                  and (measured or eng.side_dw_any)) else 0
        early_branch = bool(side_dw and K == 1 and not self.iwae and eng.batch_finish and eng.merge_launches
                            and eng.side_branches)
        # (MMVAE_SIDE_STREAM=1, the diagnostic that forks the small weight-gradient GEMMs: big shapes only -- toy-sized
        # forked graphs are where hipGraphLaunch faulted; MMVAE_SIDE_STREAM=force lifts that for the crash hunt)
        self.use_side = bool(side_dw) or (eng.side_stream_asked and eng.side_stream is not None
                                          and (big or os.environ.get("MMVAE_SIDE_STREAM") == "force"))
        loss_aside, early_calls = False, []
        # ---- pre-split operands of the G-wide GEMMs (K = 1 training programs on the wave-specialised kernels): the
        # decoder side (dP from the reconstruction epilogue, the last hidden activations from their layer tail) feeds
        # dW = dP^T h and dX = dP W; the encoder side (the gradient at the first layer from its column kernel, x from a
        # split pass) feeds dW = dY^T x.  The forward GEMM of the first layer keeps reading x as fp32: the split pass
        # would sit in front of it (17 us for 8 us gained), while beside the forward chain it is free.
        l0, lastl = self.enc_layers[0], self.dec_layers[-1]
        pl_on = bool(eng.planes and train and K == 1 and not self.iwae and lib.mmvae_gemm_get_precision() == 1)
        self.pl_dec = bool(pl_on and eng.planes_dec and G % 8 == 0 and lastl.n_in % 8 == 0 and len(self.dec_layers) >= 2
                           and lib.mmvae_gemm_planes_supported(TN, G, lastl.n_in, self.kpad(R), 1, 1, 1)
                           and lib.mmvae_gemm_planes_supported(NN, R, lastl.n_in, G, 0, 1, 0))
        self.pl_enc = bool(pl_on and eng.planes_enc and l0.n_in % 8 == 0 and l0.n_out % 8 == 0 and l0.bn is not None
                           and lib.mmvae_gemm_planes_supported(TN, l0.n_out, l0.n_in, self.kpad(B), 1, 1, 1))
        self.xp = _PlaneBuf(eng, f"xp.{l0.n_in}", B, l0.n_in) if self.pl_enc else None
        self.dYp = _PlaneBuf(eng, f"dYp.{l0.n_out}", B, l0.n_out) if self.pl_enc else None
        # the last hidden activations alone (3 MB of planes from their layer tail): B of dW = dP^T h, the wider operand of
        # that product's tile -- its stagers then split dP only
        self.pl_dec_h = bool(pl_on and eng.planes_dec_h and not self.pl_dec and lastl.n_in % 8 == 0
                             and len(self.dec_layers) >= 2
                             and lib.mmvae_gemm_planes_supported(TN, G, lastl.n_in, self.kpad(R), 1, 0, 1))
        self.hp = _PlaneBuf(eng, f"hp.{lastl.n_in}", R, lastl.n_in) if (self.pl_dec or self.pl_dec_h) else None
        self.dPp = _PlaneBuf(eng, f"dPp.{G}", R, G) if self.pl_dec else None
        self._x_split_ev = None
        x_split_late = False
        x_split_hook = None
        if self.pl_enc:
            split_x = len(self._cur)
            self._emit(lib.mmvae_split_planes_f32, B, l0.n_in, _p(x), ldx, *self.xp.args())
            x_calls = self._take(split_x)
            xs_mode = os.environ.get("MMVAE_PLANES_XSPLIT", "tail")
            if xs_mode == "fwd" and early_branch:
                # on the side stream, forked BEHIND the first layer's GEMM: beside the latency-bound forward chain.
                # (Forked at the head of the program it ran beside that GEMM and took its CUs: 57 us instead of 17.
                # Measured un-profiled: any fork this early costs the captured program ~80 us -- the graph executor
                # serialises the main chain behind it -- so this is not the default.)
                def x_split_hook():
                    self._fork()
                    self._branch(eng.side_stream, x_calls)
                    self._x_split_ev = self._record_event(eng.side_stream)
            elif xs_mode == "branch" and side_dw:
                # at the head of the existing side branch, ahead of the decoder's weight gradient (beside the start of
                # the backward chain): no fork of its own -- but it delays that chain by ~30 us
                self._x_split_side = x_calls
            elif xs_mode == "tail":
                # piggy-backed on the tail launches of the forward chain (extra workgroups of fc_fwd_apply), a fifth of
                # the rows each: those launches are latency-bound (5-13 us with the memory system idle), 20 MB of
                # streaming beside each is nearly free -- as one pass beside the first tail it cost 12 us, as a launch
                # of its own 17-22 us on the critical path, forked onto a second stream ~80 us (graph executor)
                n_jobs = int(os.environ.get("MMVAE_PLANES_XSPLIT_JOBS", "3"))
                per = (B + n_jobs - 1) // n_jobs
                xpp, xld, xps = self.xp.args()
                self._x_split_jobs = [(min(per, B - r0), l0.n_in, _p(x) + 4 * r0 * ldx, ldx, xpp + 2 * r0 * xld, xld, xps)
                                      for r0 in range(0, B, per)]
            elif xs_mode == "head":
                # at the head of the program, on the main stream: 17 us, and the first layer's forward GEMM reads the
                # planes too (8 us back)
                self._cur.extend(x_calls)
                self._x_head = True
            else:  # one stream: just ahead of its consumer (emitted there)
                self._x_split_call = x_calls
                x_split_late = True
        # ---- forward, encoder side
        cur, ld = x, ldx
        for i, l in enumerate(self.enc_layers):
            self._probe_next = "enc_l1_fwd" if i == 0 else None
            cur = self.fwd_layer(f"{self.eid}.enc{i}" if i < self.n_expert_enc else f"vae.enc{i}", l, cur, ld, B,
                                 training=train, mask_stream=i, after_gemm=x_split_hook if i == 0 else None,
                                 inp_planes=self.xp if (i == 0 and getattr(self, "_x_head", False)) else None,
                                 split_job=self._next_x_split_job(l, B))
            ld = l.n_out
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
        if self.mode == "embed":  # predict path: the program ends at z
            return self._finish_forward_only()
        if self.iwae:  # sampled log q(z) - log p(z) per (sample, cell)
            self.logratio = eng.buf("iwae.logratio", (K, B))
            self._emit(lib.mmvae_iwae_logratio, B, Z, K, _p(self.std), _p(self.eps), _p(self.z), _p(self.logratio))
        # ---- forward, decoder side (rows R = K*B)
        cur, ld = self.z, Z
        if self.cond is not None:  # CLVAE.after_reparameterize: the sample passes through the conditional layers
            cur, ld = self.cond.emit_forward(self.z)
        for i, l in enumerate(self.dec_layers[:-1]):
            cur = self.fwd_layer(f"{self.eid}.dec{i}.K{K}", l, cur, ld, R, training=train,
                                 mask_stream=len(self.enc_layers) + i,
                                 planes_out=self.hp if (self.hp is not None and i == len(self.dec_layers) - 2) else None,
                                 split_job=None if (self.hp is not None and i == len(self.dec_layers) - 2)
                                 else self._next_x_split_job(l, R))
            ld = l.n_out
        for job in getattr(self, "_x_split_jobs", []):  # tails the chain did not have: passes of their own
            self._emit(lib.mmvae_split_planes_f32, *job)
        self._x_split_jobs = []
        last = self.dec_layers[-1]
        fused_last = last.relu and last.bn is None and last.p == 0
        if not fused_last:
            raise _lib.HipLibraryError("engine: the last decoder layer must be Linear+ReLU (fused recon epilogue)")
        last.inp, last.ld_inp, last.rows = cur, ld, R
        T = lib.mmvae_recon_tiles(G)
        self.dP = eng.buf(f"dP.{G}", (R, G)) if train else None
        self.se_part = eng.buf(f"se_part.{G}", (T, R))
        self.w = eng.buf("w", (R,))
        # K = 1: the decoder bias's gradient is the column sum of dP; the epilogue that stores dP leaves its per-row-tile
        # partials (one reduction job instead of a 41 MB pass).  K > 1 re-weights the rows of dP first: separate pass.
        self.dp_colpart = None
        if train and K == 1 and eng.batch_finish and eng.fuse_dp_colsum:
            nrt = lib.mmvae_recon_row_tiles(R)
            self.dp_colpart = eng.buf(f"dP.colpart.{G}", (nrt, G))
            self._defer_sum(self.dp_colpart, nrt, G, 1, G, G, last.gb, G)
        if self.pl_dec:  # the epilogue that produces dP also leaves its bf16 planes (MMVAE_PLANES_KEEP_DP=1: and dP)
            keep_dp = os.environ.get("MMVAE_PLANES_KEEP_DP", "0") != "0"
            self._emit(lib.mmvae_decoder_recon_planes_f32, R, B, G, last.n_in, _p(cur), ld, None, 0, 0, _p(last.W),
                       last.n_in, _p(last.b), _p(x), ldx, None, 0, _p(self.dP) if keep_dp else None, G, *self.dPp.args(),
                       _p(self.se_part), _p(self.dp_colpart), probe=("dec_l2_recon", 2.0 * R * G * last.n_in))
        else:
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
                       _p(last.b), _p(x), ldx, None, 0, _p(self.dP), G, _p(self.se_part), _p(self.dp_colpart),
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
        # total loss slot starts as the ELBO loss (without adversaries it IS the ELBO loss word: no launch)
        if self.has_adv or not eng.merge_launches:
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 0.0, self.mptr("total_loss"))
        else:
            self.metric_slots["total_loss"] = 0
        for _ in range(int(os.environ.get("MMVAE_EXTRA_LAUNCHES", "0"))):  # diagnostics: price of one trivial launch
            self._emit(lib.mmvae_axpby, 1, 1.0, _p(self.metrics), 0.0, self.mptr("total_loss"))

        # ---- adversarial phases
        hidden = [l.a if l.a is not None else l.d for l in self.enc_layers if l.return_hidden]
        if self.hidden_z:
            hidden.append(self.z)  # first sample (rows 0..B-1)
        self.adv_grad_into: Dict[int, torch.Tensor] = {}
        self.dz_lat = eng.buf("dz_lat", (R, Z))
        last_ = self.dec_layers[-1]
        adv_side = bool(self.has_adv and eng.side_dw_adv and train and K == 1 and eng.side_stream is not None
                        and not eng.overlap and eng.world == 1 and big and (measured or eng.side_dw_any)
                        and self.cond is None and R <= eng.side_max_rows and self.dp_colpart is not None
                        and self._plan_gemm(TN, G, last_.n_in, self.kpad(R)) == 1)
        self._adv_dx = None
        if adv_side:
            # the branch is emitted ahead of the phases it runs beside (enqueued behind them it started behind them), and
            # the adversaries' optimisers do not join it (joined at their first clip it cost 1.55 -> 1.7-2.0 ms)
            self._fork()
            dw_pl_ = (self.dPp, self.hp) if self.pl_dec else ((None, self.hp) if self.pl_dec_h else None)
            self._probe_next = "dec_l2_dw"
            if not self._fuse_sqnorm(TN, G, last_.n_in, self.kpad(R), 1.0, self.dP, G, last_.inp, last_.ld_inp, last_.gW,
                                     last_.n_in, None, 0, side_cap=eng.side_dw_adv, planes=dw_pl_, fork=False):
                self._side_capped_gemm(TN, G, last_.n_in, self.kpad(R), self.dP, G, last_.inp, last_.ld_inp, last_.gW,
                                       last_.n_in, eng.side_dw_adv, planes=dw_pl_, fork=False)
            sk_dx = self._plan_gemm(NN, R, last_.n_in, G)
            dx_slab = eng.buf(f"dx_slab.{sk_dx}", (sk_dx, R, last_.n_in))
            self._probe_next = "dec_l2_dx"
            self._side_capped_gemm(NN, R, last_.n_in, G, self.dP, G, last_.W, last_.n_in, dx_slab, last_.n_in,
                                   eng.side_dw_adv, planes=(self.dPp, None) if self.pl_dec else None, flags=RAW, sk=sk_dx,
                                   fork=False)
            self._probe_next = None
            self._adv_dx = (dx_slab, sk_dx)
        if self.has_adv:
            held_dirty, self._dirty = self._dirty, []  # (the adversaries' optimisers must not join this branch)
            self._build_adversaries(hidden)
            self._dirty = held_dirty + [d for d in self._dirty if d not in held_dirty]

        # ---- backward, decoder side
        if K > 1:
            # dP <- diag(w) dP in place (w = softmax weights of the K-sample bound), dbias = column sums
            self._emit_fc_bwd(R, G, self.dP, None, self.w, self.dP, last.gb)
        elif self.dp_colpart is None:
            start = len(self._cur)
            self._emit_fc_bwd(R, G, self.dP, None, None, None, last.gb)
            if early_branch:  # the bias gradient (a pass over dP) is needed by the optimiser only
                early_calls += self._take(start)
        dx_pl = (self.dPp, None) if self.pl_dec else None
        dw_pl = (self.dPp, self.hp) if self.pl_dec else ((None, self.hp) if self.pl_dec_h else None)
        din_first = None
        if self._adv_dx is not None:  # both GEMMs ran beside the adversaries' phases
            self._join()
            din_first, S = self._adv_dx
        elif side_dw:  # input gradient first (the chain waits for it), then the weight gradient on the side branch
            self._probe_next = "dec_l2_dx"
            S = self.gemm_raw(NN, R, last.n_in, G, self.dP, G, last.W, last.n_in, planes=dx_pl)
            xs = getattr(self, "_x_split_side", None)
            if xs:
                self._fork()
                self._branch(eng.side_stream, xs)
                self._x_split_ev = self._record_event(eng.side_stream)
                self._x_split_side = None
            self._probe_next = "dec_l2_dw"
            if not self._fuse_sqnorm(TN, G, last.n_in, self.kpad(R), 1.0, self.dP, G, last.inp, last.ld_inp, last.gW,
                                     last.n_in, None, 0, side_cap=side_dw, planes=dw_pl):
                self.gemm(TN, G, last.n_in, self.kpad(R), self.dP, G, last.inp, last.ld_inp, last.gW, last.n_in, side=True,
                          planes=dw_pl)
            self._probe_next = None
            if early_branch:  # behind the weight gradient on its stream: one branch, in order (probe: DESIGN.md 5)
                self._fork()  # (the weight gradient may have stayed on the main stream)
                self._branch(eng.side_stream, early_calls)
        elif (eng.side_dw_dp and eng.overlap and eng.side_stream is not None and train and K == 1 and not self.has_adv
              and big and (measured or eng.side_dw_any) and self.cond is None
              and self._plan_gemm(TN, G, last.n_in, self.kpad(R)) == 1):
            # exchange program: input gradient first, the weight gradient capped on the side stream beside the chain up
            # to the VAE's exchange point (the cut there joins it)
            self._probe_next = "dec_l2_dx"
            S = self.gemm_raw(NN, R, last.n_in, G, self.dP, G, last.W, last.n_in, planes=dx_pl)
            self._probe_next = "dec_l2_dw"
            self._side_capped_gemm(TN, G, last.n_in, self.kpad(R), self.dP, G, last.inp, last.ld_inp, last.gW, last.n_in,
                                   eng.side_dw_dp, planes=dw_pl)
            self._probe_next = None
        else:
            self._probe_next = "dec_l2_dw" if big else None
            self.gemm(TN, G, last.n_in, self.kpad(R), self.dP, G, last.inp, last.ld_inp, last.gW, last.n_in, side=True,
                      planes=dw_pl)
            self._probe_next = "dec_l2_dx" if big else None
            S = self.gemm_raw(NN, R, last.n_in, G, self.dP, G, last.W, last.n_in, planes=dx_pl)
            self._probe_next = None
        rest = self.dec_layers[:-1]
        for j in range(len(rest) - 1, -1, -1):
            l = rest[j]
            d_in = din_first if j == len(rest) - 1 else None
            if j > 0:
                S = self.bwd_layer(l, d_in, S, need_dx="raw")
            else:
                self.bwd_layer(l, d_in, S, need_dx="full",
                               dx_out=self.dz_lat if self.cond is None else self.cond.d_out)
        if self.cond is not None:
            self.cond.emit_backward(self.dz_lat)
            if mdist.collectives_active() and train:
                # blocks another rank saw and this one did not: zeros into the exchange (job table of this step)
                self._emit(lib.mmvae_grad_zero_flagged_jobs, self.cond.max_jobs, self.cond.jobs_ptr,
                           _p(self.cond.opt.arena.grad))
        if not rest:
            raise _lib.HipLibraryError("engine: decoder needs at least two layers")
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
        if B % 32 == 0 and eng.batch_finish:
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
            if j == 0 and side_dw:
                self._defer_next_dw = True
            if j == 0 and x_split_late:
                self._cur.extend(self._x_split_call)
            S_next = self.bwd_layer(l, din, S, addend=addend, need_dx="raw" if j > 0 else "none",
                                    dz_planes=self.dYp if (j == 0 and self.pl_enc) else None,
                                    inp_planes=self.xp if (j == 0 and self.pl_enc) else None)
            din, S = None, S_next
            if early and j == self.n_expert_enc:  # the last VAE layer is done: what remains is the expert's encoder
                self._begin_exchange(self.opt_vae)
        # ---- clip + Adam (reference order: clip vae, clip expert, step vae, step expert)
        # Logged scalars: the step's metrics words (and the pre-clip gradient norms) are copied into a buffer of this
        # plan's own as the last node(s) of the captured program -- the logged tensors are views of it, valid until
        # this plan's next step -- instead of three D2D copies issued by the host behind every replay (~24 us).
        self.log_buf = torch.zeros(256, dtype=torch.float32, device=eng.device)

        def emit_log_copy():
            self._emit(lib.mmvae_axpby, 256, 1.0, _p(self.metrics), 0.0, _p(self.log_buf))

        dw = getattr(self, "_deferred_dw", None)
        self._deferred_dw = None
        late_branch = bool(side_dw and eng.side_branches and dw is not None and not self._side_foreign
                           and self.cond is None)
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
            if dwp is not None and self._x_split_ev is not None:
                self._wait_event(self._x_split_ev)  # the planes of x come from the side stream
            self._probe_next = "enc_l1_dw"
            if not self._fuse_sqnorm(layout, M, N, Kk, 1.0, A, lda, Bm, ldb, Cm, ldc, None, 0, side_cap=eng.side_dw2,
                                     on_side=False, planes=dwp):
                self.gemm(*dw, side=True, planes=dwp)
            self._probe_next = None
            self._branch(eng.side_stream, calls)
        elif dw is not None:
            dwp = getattr(self, "_deferred_dw_planes", None)
            if dwp is not None and self._x_split_ev is not None:
                self._wait_event(self._x_split_ev)
            self._probe_next = "enc_l1_dw" if big else None
            self.gemm(*dw, side=True, planes=dwp)
            self._probe_next = None
        if early:  # the expert's exchange + update leave the main stream: its norm is logged from the comm stream
            emit_log_copy()
        # in-order program: the log copy rides on the expert's Adam launch (its words -- losses, both norms -- are final
        # once adam_prepare has run) when the expert's norm is a state word inside the metrics buffer
        ride = (not early and eng.fuse_norm_prepare and self.cond is None
                and eng._state_slot.get(id(self.opt_exp)) is not None
                and not (eng.shard and self.opt_exp.reducer is not None))
        self.optimizer(self.opt_exp, self.clip_exp, exchange="deferred" if early else "inline",
                       tail_copy=(256, self.metrics, self.log_buf) if ride else None)
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
            if eng.merge_launches:
                arr = (_lib.PhiloxJob * len(fills))(*fills)
                jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(eng.device)
                self._job_tables.append(jobs_dev)
                if eng.fuse_norm_prepare:  # (same switch: launches folded through a last-ticket workgroup)
                    ticket = eng.buf("philox.ticket", (1,), torch.int32)
                    self._emit(lib.mmvae_philox_fill_jobs_advance, len(fills), jobs_dev.data_ptr(), n_max,
                               _p(self.rng_state), (n_max + 3) // 4, _p(ticket))
                    self.segments[0] = self._cur + self.segments[0]
                    self._cur = []
                    self._size_workspaces()
                    return
                self._emit(lib.mmvae_philox_fill_jobs, len(fills), jobs_dev.data_ptr(), n_max, _p(self.rng_state))
            else:
                for f in fills:
                    if f.kind == 0:
                        self._emit(lib.mmvae_philox_keep_mask, f.n, f.p_drop, f.out, _p(self.rng_state), f.stream_id, 0)
                    else:
                        self._emit(lib.mmvae_philox_normal, f.n, f.out, _p(self.rng_state), f.stream_id, 0)
            self._emit(lib.mmvae_philox_advance, _p(self.rng_state), (n_max + 3) // 4)
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
            if eng.merge_launches and eng.fuse_norm_prepare:  # one launch: the fill's last workgroup advances the counter
                arr = (_lib.PhiloxJob * 1)(_lib.PhiloxJob(_p(self.eps), n, rng.STREAM_NORMAL, 0.0, 1))
                jobs_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(eng.device)
                self._job_tables.append(jobs_dev)
                self._emit(lib.mmvae_philox_fill_jobs_advance, 1, jobs_dev.data_ptr(), n, _p(self.rng_state),
                           (n + 3) // 4, _p(eng.buf("philox.ticket", (1,), torch.int32)))
            else:
                self._emit(lib.mmvae_philox_normal, n, _p(self.eps), _p(self.rng_state), rng.STREAM_NORMAL, 0)
                self._emit(lib.mmvae_philox_advance, _p(self.rng_state), (n + 3) // 4)
            self.segments[0] = self._cur + self.segments[0]
            self._cur = []
        self._size_workspaces()

    def _size_workspaces(self):
        eng = self.eng
        self.fcws = eng.buf("fc_ws", (max(getattr(self, "_fcws_bytes", 0) // 4, 1),))
        for key, t in eng._pool.items():
            if key[0] == "fc_ws" and t.numel() > self.fcws.numel():
                self.fcws = t
        self.ws_side = eng.buf("gemm_ws_side", (max(getattr(self, "_ws_side_bytes", 0) // 4, 1),))
        for key, t in eng._pool.items():
            if key[0] == "gemm_ws_side" and t.numel() > self.ws_side.numel():
                self.ws_side = t
        self.ws = eng.buf("gemm_ws", (max(self._ws_bytes // 4, 1),))
        self.slab = eng.buf("gemm_slabs", (max(self._slab_floats, 1),))
        # a shared buffer may have been re-allocated larger by a later plan: always take the biggest one
        for key, t in eng._pool.items():
            if key[0] == "gemm_ws" and t.numel() > self.ws.numel():
                self.ws = t
            if key[0] == "gemm_slabs" and t.numel() > self.slab.numel():
                self.slab = t

    def _build_adversaries_fused(self, hidden) -> bool:
        """Both phases of ALL adversaries as five launches (adv_program.py; kernels: csrc/adv_fused.hip): adversaries
        whose encoder has no BatchNorm -- every adversary of the reference's configurations -- are row-local up to the
        weight gradients.  False (nothing emitted): a shape outside those kernels; the per-layer program follows."""
        from .adv_program import AdvLayer, AdvNet, AdvProgram, supported

        eng, lib, B = self.eng, self.lib, self.B
        pairs = list(zip(hidden, self.advs))
        H = len(self.conditions)
        if not pairs or not (1 <= H <= _lib.ADV_MAX_HEADS):
            return False
        g = eng.grad_of
        pad4 = lambda c: (c + 3) // 4 * 4  # noqa: E731
        nets, mask_refs = [], []
        for i, (h, adv) in enumerate(pairs, start=1):
            opt = self.opt_adv[i - 1]
            lins = [adv.heads[c].fc_layers[0].lin for c in self.conditions]
            if any(l.bias is None for l in lins):
                return False
            n_e = lins[0].in_features
            a = opt.arena
            if H > 1:  # the heads as ONE matrix / bias vector of the arena (HipAdam pack=): rows padded to 4 per head
                ws, bs = [l.weight for l in lins], [l.bias for l in lins]
                rows_of = [pad4(l.out_features) for l in lins]
                chain = lambda ts, per_row: all(ts[k + 1].data_ptr() == ts[k].data_ptr() + 4 * rows_of[k] * per_row  # noqa: E731
                                                for k in range(H - 1))
                if not (chain(ws, n_e) and chain(bs, 1) and chain([g(w) for w in ws], n_e) and chain([g(b) for b in bs], 1)):
                    return False
                Ct = sum(rows_of)
                ow, ob = a.offsets[arena_of(ws[0])[1]], a.offsets[arena_of(bs[0])[1]]
                Wh, bh = a.data[ow:ow + Ct * n_e].view(Ct, n_e), a.data[ob:ob + Ct]
                gWh, gbh = a.grad[ow:ow + Ct * n_e].view(Ct, n_e), a.grad[ob:ob + Ct]
                col = [sum(rows_of[:k]) for k in range(H)]
            else:
                Wh, bh, gWh, gbh, col = lins[0].weight, lins[0].bias, g(lins[0].weight), g(lins[0].bias), [0]
            layers, covered = [], {id(l.weight) for l in lins} | {id(l.bias) for l in lins}
            for j, seq in enumerate(adv.encoder.fc_layers):
                if getattr(seq, "bn", None) is not None or getattr(seq.lin, "bias", None) is None:
                    return False
                refs = {ph: _LayerRef(seq, g, False, adv.encoder, j) for ph in ("discriminator", "generator")}
                r = refs["discriminator"]
                lay = AdvLayer(W=r.W, b=r.b, gW=r.gW, gb=r.gb, relu=r.relu, p_drop=r.p)
                covered |= {id(r.W), id(r.b)}
                if r.p > 0:
                    for ph, ref in refs.items():
                        ref.mask = eng.buf(f"adv{i}.{ph}.enc{j}.mask", (B, r.n_out), torch.uint8)
                        lay.masks[ph] = ref.mask
                        mask_refs.append((ref, 1000 + 64 * i + 32 * int(ph == "generator") + j))
                layers.append(lay)
            if not layers or {id(p) for p in a.params} != covered:  # the fused norm is the norm of what the jobs write
                return False
            net = AdvNet(x=h, ldx=layers[0].W.shape[1], layers=layers, Wh=Wh, bh=bh, gWh=gWh, gbh=gbh, col=col,
                         classes=[l.out_features for l in lins], opt=opt)
            if supported(lib, net, B) is None:
                return False
            nets.append(net)
        self._labels_all = eng.buf("labels.all", (H, B), torch.int64)
        self.labels_dev = {c: self._labels_all[k] for k, c in enumerate(self.conditions)}
        self.n_adv = len(nets)
        self._mask_layers += mask_refs
        prog = AdvProgram(lib, eng.buf, nets, B, self._labels_all, eng.device)
        self.adv_prog = prog
        # a gradient exchange (data parallelism) sits between the weight gradients and the norm: the optimiser launches
        # of the per-layer program then follow the fused passes
        dp = eng.overlap or any(o.reducer is not None for o in self.opt_adv[:len(nets)])
        gs = 1.0 / eng.world
        for phase, gen in (("discriminator", False), ("generator", True)):
            firsts, totals = [], []
            for i in range(1, len(nets) + 1):
                first = self.slot(f"{phase}_{i}/{self.conditions[0]}")
                for k, c in enumerate(self.conditions):
                    assert self.slot(f"{phase}_{i}/{c}") == first + k
                assert self.slot(f"{phase}_{i}/summed") == first + H
                firsts.append(self.metrics.data_ptr() + 4 * first)
                totals.append(self.metrics.data_ptr() + 4 * (first + H))
            opts = None
            if not dp:
                opts = [dict(flags=_lib.PREPARE_NORM | (0 if gen else _lib.PREPARE_ADVANCE),
                             max_norm=0.0 if gen else self.clip_adv,
                             norm_out=None if gen else self.mptr(f"grad_norms/discriminator_{i}"))
                        for i in range(1, len(nets) + 1)]
            prog.build_phase(phase, dict(gscale=self.adv_weight if gen else 1.0, reverse=gen, loss_each=firsts,
                                         loss_total=totals, total_loss=self.mptr("total_loss") if gen else None,
                                         total_scale=self.adv_weight, opts=opts, grad_scale=gs))
        prog.build_adam(gs)
        for phase, gen in (("discriminator", False), ("generator", True)):
            self._cur.append(lambda ph=phase: prog.launch_pass(ph))
            self._cur.append(lambda ph=phase: prog.launch_dw(ph))
            for i, net in enumerate(nets, start=1):
                if dp:
                    self.optimizer(net.opt, 0.0 if gen else self.clip_adv, step=not gen)
                    self.log_norm(net.opt, f"grad_norms/{phase}_{i}", final=gen)
                elif gen:
                    self.log_norm(net.opt, f"grad_norms/generator_{i}")
            if not gen and not dp:
                self._cur.append(prog.launch_adam)
        for (h, _), b in zip(pairs, prog.bufs):
            self.adv_grad_into[id(h)] = b["gx"]
        return True

    def _build_adversaries(self, hidden):
        eng, lib, B = self.eng, self.lib, self.B
        if eng.adv_fused and self._build_adversaries_fused(hidden):
            return
        self._labels_all = eng.buf("labels.all", (len(self.conditions), B), torch.int64)
        self.labels_dev = {c: self._labels_all[i] for i, c in enumerate(self.conditions)}  # one upload per step
        self.n_adv = min(len(hidden), len(self.advs))
        for i, (h, adv) in enumerate(zip(hidden, self.advs), start=1):
            g = eng.grad_of
            # one set of layer records per phase: the phases share activations and gradient buffers (same tags) but each
            # draws its own dropout keep masks, like two forward calls of the reference's nn.Dropout
            phase_layers = {ph: [_LayerRef(seq, g, False, adv.encoder, j) for j, seq in enumerate(adv.encoder.fc_layers)]
                            for ph in ("discriminator", "generator")}
            layers = phase_layers["discriminator"]
            n_e = layers[-1].n_out
            heads = {c: adv.heads[c].fc_layers[0].lin for c in self.conditions}
            logits = {c: eng.buf(f"adv{i}.logits.{c}", (B, heads[c].out_features)) for c in self.conditions}
            dlogits = {c: eng.buf(f"adv{i}.dlogits.{c}", (B, heads[c].out_features)) for c in self.conditions}
            H = len(self.conditions)
            rows = eng.buf(f"adv{i}.ce_rows", (max(H, 1), B))
            de = eng.buf(f"adv{i}.de", (B, n_e))
            gh = eng.buf(f"adv{i}.gh", (B, layers[0].n_in))
            opt = self.opt_adv[i - 1]
            # heads laid out back to back in the optimiser arena (HipAdam pack=): ONE matrix [sum of classes, n_e] and
            # one bias vector -> forward, bias gradient, weight gradient and input gradient of all heads are one launch
            # each instead of one per head (and the input gradient loses its accumulate chain)
            fused = None
            lins = [heads[c] for c in self.conditions]
            if H > 1 and os.environ.get("MMVAE_FUSE_HEADS", "1") != "0":
                ws, bs = [l.weight for l in lins], [l.bias for l in lins]
                pad4 = lambda c: (c + 3) // 4 * 4
                rows_of = [pad4(l.out_features) for l in lins]  # class counts padded to 4 (HipAdam pack alignment)
                chain = lambda ts, per_row: all(ts[k + 1].data_ptr() == ts[k].data_ptr() + 4 * rows_of[k] * per_row
                                                for k in range(H - 1))
                if (chain(ws, n_e) and chain(bs, 1) and chain([g(w) for w in ws], n_e) and chain([g(b) for b in bs], 1)):
                    Ct = sum(rows_of)
                    a, iw, ib = opt.arena, arena_of(ws[0])[1], arena_of(bs[0])[1]
                    ow, ob = a.offsets[iw], a.offsets[ib]
                    fused = dict(Ct=Ct, W=a.data[ow:ow + Ct * n_e].view(Ct, n_e), b=a.data[ob:ob + Ct],
                                 gW=a.grad[ow:ow + Ct * n_e].view(Ct, n_e), gb=a.grad[ob:ob + Ct],
                                 logits=eng.buf(f"adv{i}.logits_all", (B, Ct)), dlogits=eng.buf(f"adv{i}.dlogits_all", (B, Ct)))
            for phase in ("discriminator", "generator"):
                gen = phase == "generator"
                layers = phase_layers[phase]
                cur, ld = h, layers[0].n_in
                for j, l in enumerate(layers):
                    cur = self.fwd_layer(f"adv{i}.enc{j}", l, cur, ld, B, mask_tag=f"adv{i}.{phase}.enc{j}",
                                         mask_stream=1000 + 64 * i + 32 * int(gen) + j)
                    ld = l.n_out
                e = cur
                gscale = self.adv_weight if gen else 1.0
                if fused is not None:
                    Ct, col = fused["Ct"], 0
                    self.gemm(NT, B, Ct, n_e, e, n_e, fused["W"], n_e, fused["logits"], Ct, bias=fused["b"])
                    widths = [heads[c].out_features for c in self.conditions]
                    padded = [(w + 3) // 4 * 4 for w in widths]  # a head's columns start on a multiple of 4
                    if max(widths) <= 8192 and eng.merge_launches:  # every head's cross-entropy in one launch
                        if "cols" not in fused:
                            starts = [sum(padded[:k]) for k in range(H)]
                            fused["cols"] = torch.tensor(starts + widths, dtype=torch.int32, device=eng.device)
                            self._job_tables.append(fused["cols"])  # the captured program reads it on every replay
                        cw = fused["cols"]
                        self._emit(lib.mmvae_cross_entropy_heads, B, H, max(widths), _p(cw), cw.data_ptr() + 4 * H,
                                   _p(fused["logits"]), Ct, _p(self._labels_all), _p(rows), _p(fused["dlogits"]), Ct,
                                   gscale)
                    else:
                        for ci, c in enumerate(self.conditions):
                            Cn = heads[c].out_features
                            self._emit(lib.mmvae_cross_entropy_sum, B, Cn, fused["logits"].data_ptr() + 4 * col, Ct,
                                       _p(self.labels_dev[c]), _p(rows[ci]), fused["dlogits"].data_ptr() + 4 * col, Ct,
                                       None, gscale)
                            col += (Cn + 3) // 4 * 4
                    self._emit_fc_bwd(B, Ct, fused["dlogits"], None, None, None, fused["gb"])
                    self.gemm(TN, Ct, n_e, self.kpad(B), fused["dlogits"], Ct, e, n_e, fused["gW"], n_e, side=True)
                    self.gemm(NN, B, n_e, Ct, fused["dlogits"], Ct, fused["W"], n_e, de, n_e)
                for ci, c in enumerate(self.conditions if fused is None else []):
                    lin = heads[c]
                    Cn = lin.out_features
                    self.gemm(NT, B, Cn, n_e, e, n_e, lin.weight, n_e, logits[c], Cn, bias=lin.bias)
                    self._emit(lib.mmvae_cross_entropy_sum, B, Cn, _p(logits[c]), Cn, _p(self.labels_dev[c]), _p(rows[ci]),
                               _p(dlogits[c]), Cn, None, gscale)
                    # head backward
                    self._emit_fc_bwd(B, Cn, dlogits[c], None, None, None, g(lin.bias))
                    self.gemm(TN, Cn, n_e, self.kpad(B), dlogits[c], Cn, e, n_e, g(lin.weight), n_e, side=True)
                    self.gemm(NN, B, n_e, Cn, dlogits[c], Cn, lin.weight, n_e, de, n_e, flags=ACC if ci > 0 else 0)
                # the heads' losses and their sum: consecutive metrics words, one launch
                first = self.slot(f"{phase}_{i}/{self.conditions[0]}") if H else None
                for k, c in enumerate(self.conditions):
                    assert self.slot(f"{phase}_{i}/{c}") == first + k
                total_slot = self.slot(f"{phase}_{i}/summed")
                if H:
                    self._emit(lib.mmvae_sum_rows_f32, H, B, _p(rows), B, self.metrics.data_ptr() + 4 * first,
                               self.metrics.data_ptr() + 4 * total_slot)
                din, S = de, 1
                for j in range(len(layers) - 1, -1, -1):
                    l = layers[j]
                    if j > 0:
                        S = self.bwd_layer(l, din, S, need_dx="raw")
                        din = None
                    elif gen:
                        # gradient reversal (components.py:889-899): d h = -alpha * d(adv loss)/d h, alpha = 1
                        self.bwd_layer(l, din, S, need_dx="full", dx_out=gh, dx_alpha=-1.0)
                    else:
                        self.bwd_layer(l, din, S, need_dx="none")
                if gen:
                    self._emit(lib.mmvae_axpby, 1, self.adv_weight, self.mptr(f"generator_{i}/summed"), 1.0,
                               self.mptr("total_loss"))
                    self.optimizer(opt, 0.0, step=False)  # norm of the (never applied) generator-phase gradients
                    self.log_norm(opt, f"grad_norms/generator_{i}")
                    self.adv_grad_into[id(h)] = gh
                else:
                    self.optimizer(opt, self.clip_adv)
                    self.log_norm(opt, f"grad_norms/discriminator_{i}", final=False)

    # ------------------------------------------------------------------------------------------------ execution
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

    def load_labels(self, metadata):
        """The step's class indices: metadata columns -> int64 through the class-level Adversarial.labels tables
        (cmmvae_model.py:111-115), recomputed on every step like the reference does (nothing is cached on the DataFrame:
        pandas copies `attrs` into frames derived from it, and a column may be edited in place), written into a
        page-locked slot and uploaded with one copy."""
        import numpy as np

        n = len(metadata)
        if n != self.B:
            raise ValueError(f"engine: metadata has {n} rows, the batch has {self.B}")
        if getattr(self, "_label_ring", None) is None:
            self._label_ring = _PinnedRing(len(self.conditions) * self.B, torch.int64)
            self._labels_all = self.eng.buf("labels.all", (len(self.conditions), self.B), torch.int64)
        slot = self._label_ring.take()
        for i, c in enumerate(self.conditions):
            table = Adversarial.labels[c]
            slot[i * n:(i + 1) * n] = np.fromiter((table[v] for v in metadata[c].values), dtype=np.int64, count=n)
        self._label_ring.upload(self._labels_all.view(-1))

    def _exchange(self, marker, tail):
        """One data-parallel exchange point between two captured segments.  Returns the stream the rest of the
        program runs on (None = stay on the main stream)."""
        kind, opt = marker
        red = opt.reducer
        eng = self.eng
        main = torch.cuda.current_stream()

        def reduce_small():
            c = self.cond
            if c is not None and opt is self.opt_vae and c.n_exchange and eng.cond_packed_exchange:
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
                    else:  # (no peers to hear from: the slice's own sum)
                        si["allsq"][:1].copy_(si["mine"])
                elif live:
                    full = a.data_full[:si["per"] * W]
                    tdist.all_gather_into_tensor(full, full[si["lo"]:si["lo"] + si["per"]], group=red.group)
        else:
            raise _lib.HipLibraryError(f"engine: unknown exchange marker {kind}")
        return tail

    def _run_program(self, items, launch):
        tail = None  # once set, the rest of the program (the deferred update) runs on the communication stream
        for idx, it in enumerate(items):
            if isinstance(it, tuple) and it[0] == "ar_deferred" and self.eng.lazy_adam:
                # only the collective leaves the main stream; the update behind it is stashed for this expert's next step
                eng, opt = self.eng, it[1]
                eng.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(eng.comm_stream):
                    if opt.reducer is not None:
                        opt.reducer.reduce_here(opt.arena.grad)
                    ev = torch.cuda.Event()
                    ev.record(eng.comm_stream)
                eng._lazy[self.eid] = (self, list(items[idx + 1:]), launch, ev)
                return None
            if isinstance(it, tuple):
                tail = self._exchange(it, tail)
            elif tail is None:
                launch(it)
            else:
                with torch.cuda.stream(tail):
                    launch(it)
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
        for g in self._graphs or []:
            if not isinstance(g, tuple):
                g.reset()
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
        if self._runs == 1 or os.environ.get("MMVAE_NO_GRAPH", "0") != "0":
            # a real step; the first run also loads every code object before capture
            return self._run_program(self.segments, self._launch_eager)
        if self._graphs is None:
            torch.cuda.synchronize()
            graphs = []
            for seg in self.segments:
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
            self._graphs = graphs
        return self._run_program(self._graphs, lambda g: g.replay())

    def log(self, model, eid: str):
        m = self.log_buf  # filled by the captured program itself; logged scalars are views of it (no copy, no host sync)
        if os.environ.get("MMVAE_LOG_COPY", "") == "host":  # diagnostics: the former host-issued copy
            m = self.log_buf.clone()
        stage = model.stage_name
        main = {RK.LOSS: m[self.slot("total_loss")], RK.RECON_LOSS: m[1], RK.KL_LOSS: m[2], RK.KL_WEIGHT: m[3],
                "Mean": m[4], "Variance": m[5]}
        for i in range(1, getattr(self, "n_adv", 0) + 1):
            for phase in ("discriminator", "generator"):
                tags = [f"{phase}_{i}", stage, eid, RK.ADV_LOSS]
                for c in self.conditions + ["summed"]:
                    model.auto_log({c: m[self.slot(f"{phase}_{i}/{c}")]}, tags=tags, key_pos="last")
                model.log(f"grad_norms/{phase}_{i}", m[self.slot(f"grad_norms/{phase}_{i}")])
        model.log("grad_norms/vae", m[self.slot("grad_norms/vae")])
        # overlapped mode: the norm is produced on the communication stream; the logged tensor is filled when that
        # stream gets there (read it after engine.flush() / a device synchronisation)
        model.log(f"grad_norms/expert_{eid}", self.exp_norm_log[0] if self.exp_norm_log is not None
                  else m[self.slot("grad_norms/expert")])
        model.auto_log(main, tags=[stage, eid])


class _CondProgram:
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
                yield from _CondProgram._blocks(sub)

    @staticmethod
    def supported(cl, opt_vae, Z: int) -> bool:
        has_ln = None
        for blk in _CondProgram._all_blocks(cl):
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

    def __init__(self, plan: "_Plan", cl, eid: str, train: bool):
        import numpy as np

        self.np = np
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
            ent = dict(base=len(w_off), w_idx=w_idx, b_idx=b_idx, layer=layer if isinstance(layer, ConditionalLayer) else None,
                       raw_index={})
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
        self.dw_partials = eng.buf("cond.dw_partials", (cond_tables.partial_slots(R) * (Z * Z + Z),)) if train else None
        self.idx_words = (self.n_pos * self.P + 1) // 2 * 2
        words = self.idx_words + 6 * self.max_jobs
        self.pack_dev = eng.buf(f"cond.pack.{eid}.{int(train)}", (words,), torch.int32)
        self.ring = _PinnedRing(words)
        self._scratch = np.zeros(words, dtype=np.int32)
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

    def _ptr(self, j: int, name: str) -> int:
        """Device address of array `name` (cond_tables.layout) of position j."""
        return self.pack_dev.data_ptr() + 4 * (j * self.P + self.lay[name])

    # ------------------------------------------------------------------------------------------ program emission
    def emit_forward(self, z: torch.Tensor):
        plan, lib, R, Z = self.plan, self.plan.lib, self.R, self.Z
        params = self.opt.arena.data
        cur = z
        for j in range(self.n_pos):
            x = z if self.parallel else cur
            self.x_in[j] = x
            if self.parallel:
                y, ldy = self.out[:, j * Z:(j + 1) * Z], self.out.shape[1]
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
        for j in range(self.n_pos - 1, -1, -1):
            y, ldy = self.y[j]
            gj = g[:, j * Z:(j + 1) * Z] if self.parallel else g
            if self.ln_eps is not None:
                plan._emit(lib.mmvae_layernorm_bwd, R, Z, _p(gj), ldg, _p(y), ldy, _p(self.invstd[j]), _p(self.gl), Z)
                gl, ldgl = self.gl, Z
            else:
                gl, ldgl = gj, ldg
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

    # ------------------------------------------------------------------------------------------------ per step
    def _local_indices(self, ent, key, metadata):
        np = self.np
        B = len(metadata)
        if ent["layer"] is None:  # the species block: every cell goes through the one block of this expert
            return np.zeros(B, dtype=np.int32)
        raw_index, layer = ent["raw_index"], ent["layer"]
        values = metadata[layer.batch_key].tolist()
        try:
            return np.fromiter((raw_index[v] for v in values), dtype=np.int32, count=B)
        except KeyError:
            for v in set(values) - raw_index.keys():
                raw_index[v] = ent["index"][layer.format_condition_key(str(v))]  # KeyError: unknown condition
            return np.fromiter((raw_index[v] for v in values), dtype=np.int32, count=B)

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
        for j, key in enumerate(order):
            ent = self.entries[key]
            local = self._local_indices(ent, key, metadata)
            if len(local) != R:
                raise _lib.HipLibraryError(f"engine: metadata has {len(local)} rows, the batch {R}")
            t = cond_tables.group_tables(local, ent["base"])
            cond_tables.fill_padded(pack[j * P:(j + 1) * P], t, R)
            active.append(ent["w_idx"][t["present"]])
            active.append(ent["b_idx"][t["present"]])
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
        steps = self.opt.host_steps()
        steps[self._active] += 1
