"""Flat-arena Adam with fused global-norm clipping (k12 + k13).

Replaces `torch.optim.Adam(module.parameters(), lr=5e-3, weight_decay=1e-6)` + Lightning `clip_gradients(opt, 10,
"norm")` of the reference (models/cmmvae_model.py:299-324, :126-131, :203-213).

MI355X-first layout: all parameters of one optimiser live contiguously in ONE fp32 arena in HBM (each tensor padded to
16 B), with parallel arenas for gradients, exp_avg and exp_avg_sq.  `nn.Parameter.data` of every parameter is re-pointed
at its slice, so modules, checkpoints and kernels all see the same memory, and:
  * clip + Adam over a whole optimiser = 3 launches over one contiguous range (norm partials, prepare, update), instead
    of ~10 tensors x several foreach launches;
  * the DDP gradient all-reduce is a handful of large contiguous buckets of the gradient arena (mmvae_amd.dist).
Gradients written by autograd into separate tensors (module path) are gathered into the arena with D2D copies at step();
the graph-captured engine writes its GEMM outputs straight into the arena.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

from . import backend, ops
from . import dist as mdist


def _pad4(n: int) -> int:
    return (n + 3) // 4 * 4


class ParamArena:
    """Contiguous storage for a list of parameters (+ grads and Adam moments).  Device (HIP) or CPU plumbing."""

    def __init__(self, params: Iterable[torch.nn.Parameter], pack: Optional[list] = None):
        """`pack`: groups of parameters to be laid out back to back in the order given -- a list of parameters, or
        (list, alignment in elements) to start every member on a multiple of the alignment inside the group (the heads
        of an adversary: their weights form one [sum of padded class counts, width] matrix and their biases one vector,
        so that all heads run as one GEMM; rows padded to multiples of 4 keep that GEMM on the 16-byte loaders; the
        holes stay zero).  Placement only: the parameter order -- and the optimiser's state_dict -- is that of `params`."""
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("empty parameter list")
        dev = self.params[0].device
        if any(p.device != dev or p.dtype != torch.float32 for p in self.params):
            raise ValueError("all parameters of an arena must be fp32 on one device")
        self.device = dev
        groups = [(g if isinstance(g, tuple) else (g, 1)) for g in (pack or [])]  # (members, member alignment)
        group_of = {id(q): gi for gi, (grp, _) in enumerate(groups) for q in grp}
        placed: Dict[int, int] = {}
        n = 0
        for p in self.params:
            if id(p) in placed:
                continue
            gi = group_of.get(id(p))
            if gi is None:
                placed[id(p)] = n
                n += _pad4(p.numel())
            else:  # the whole group here: members back to back, each starting on a multiple of the group's alignment
                members, align = groups[gi]
                start = n
                for q in members:
                    n = start + (n - start + align - 1) // align * align  # (holes stay zero: data, grads, moments)
                    placed[id(q)] = n
                    n += q.numel()
                n = start + (n - start + align - 1) // align * align
                n = _pad4(n)
        self.offsets: List[int] = [placed[id(p)] for p in self.params]
        self.numel = n
        # 32 spare elements behind each arena: its tensors may be read as rows-contiguous GEMM operands in 16-byte groups
        # up to 12 bytes past their end (MMVAE_GEMM_OPERAND_SLACK), a weight matrix row up to the next multiple of 32
        # columns (mmvae_recon_set_h_kpad)
        # (the padded tensors themselves: a sharded exchange works on the arena rounded up to 4 x world elements)
        self.data_full = torch.zeros(n + 32, dtype=torch.float32, device=dev)
        self.grad_full = torch.zeros(n + 32, dtype=torch.float32, device=dev)
        self.data, self.grad = self.data_full[:n], self.grad_full[:n]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            view = self.data[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view

    def shard(self, world: int, rank: int):
        """(elements per shard, first element, valid elements) of rank `rank`'s contiguous slice when the arena, rounded up
        to world equal slices of a multiple of 4 elements, is reduce-scattered / all-gathered; None when the rounding
        does not fit in the 32 spare elements behind the arena."""
        per = (self.numel + 4 * world - 1) // (4 * world) * 4
        if per * world > self.numel + 32:
            return None
        lo = rank * per
        return per, lo, max(0, min(per, self.numel - lo))

    def view(self, arena: torch.Tensor, i: int) -> torch.Tensor:
        p, off = self.params[i], self.offsets[i]
        return arena[off:off + p.numel()].view(p.shape)

    def grad_view(self, i: int) -> torch.Tensor:
        return self.view(self.grad, i)

    def gather_grads(self, direct=(), zeroed: bool = False, only=None) -> List[int]:
        """Bring autograd-produced .grad tensors into the gradient arena (D2D copies; no-ops for arena views).
        Returns the indices of the parameters WITHOUT a gradient (their arena slice is zeroed, so the global norm is
        right, and the optimiser skips them like torch.optim.Adam skips `p.grad is None`)."""
        inactive = []
        for i in (range(len(self.params)) if only is None else only):
            p = self.params[i]
            if i in direct:  # written in place by a kernel
                continue
            if p.grad is None:
                if not zeroed:  # `zeroed`: the whole gradient arena was cleared by zero_grad()
                    self.grad_view(i).zero_()
                inactive.append(i)
                continue
            gv = self.grad_view(i)
            if p.grad.data_ptr() != gv.data_ptr():
                gv.copy_(p.grad)
        return inactive


# parameter -> (optimiser, index in its arena): lets kernels that address parameters through arena offsets (the grouped
# conditional layers) find the arena a parameter lives in.  Keyed by id(); entries are validated against the live
# object on lookup.
_ARENA_OF: Dict[int, tuple] = {}


def arena_of(p: torch.nn.Parameter):
    """(HipAdam, index) of a parameter that lives in a flat arena, else None."""
    hit = _ARENA_OF.get(id(p))
    if hit is None or hit[0].arena.params[hit[1]] is not p:
        return None
    return hit


class HipAdam(torch.optim.Optimizer):
    """Adam (coupled L2 weight decay, amsgrad=False; identical update rule to torch.optim.Adam) over a ParamArena,
    with the gradient-norm clip fused in.  State (`step`, last pre-clip grad norm, clip coefficient) stays on device."""

    def __init__(self, params, lr: float = 5e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-6,
                 max_grad_norm: Optional[float] = None, pack: Optional[list] = None):
        params = list(params)
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.arena = ParamArena(params, pack=pack)
        for i, p in enumerate(self.arena.params):
            _ARENA_OF[id(p)] = (self, i)
        self._direct: set = set()  # parameters whose gradient a kernel wrote straight into the arena this step
        self.managed: set = set()  # parameters only ever written that way (condition blocks of the grouped kernels)
        self.sparse_presence = False  # some parameters get no gradient in some steps (condition blocks): set by the model
        self._walk = None
        self.max_grad_norm = max_grad_norm
        self.grad_scale = 1.0  # 1 / world_size under DDP (gradient averaging)
        self.reducer = None  # mmvae_amd.dist.GradAllReducer under DDP
        # The step engine's data-parallel program updates only this rank's slice of the arena (reduce-scatter -> clip + Adam
        # on the slice -> all-gather of the parameters): the Adam moments of the other slices are then stale here until
        # sync_sharded_state() gathers them (checkpoints, a switch to the module path).
        self.sharded = False
        # A step engine that leaves part of an update running on another stream (deferred expert updates) points this at
        # its flush(): called before the moments / parameters are read or rewritten from the torch side.
        self.settle = None
        self._hip = self.arena.device.type == "cuda"
        if not self._hip and not backend.cpu_plumbing_enabled():
            raise RuntimeError("HipAdam needs device parameters (or caller-enabled backend.cpu_plumbing())")
        dev = self.arena.device
        self.state_dev = torch.zeros(8, dtype=torch.float32, device=dev)  # step, norm, clip, bc1, bc2
        # torch.optim.Adam counts steps per parameter and skips parameters without a gradient (conditional layers:
        # the conditions absent from a batch).  While every parameter has taken part in every step the counts are all
        # equal to state_dev[0] and live on the device only; the first partial step materialises them here.
        self._steps: Optional[List[int]] = None
        self._numel = None
        self._inactive: List[int] = []
        if self._hip:
            self.partials = torch.empty(max(ops.sqnorm_partials(self.arena.numel), 1), dtype=torch.float32, device=dev)

    def sync_sharded_state(self) -> None:
        """All-gather the Adam moments after sharded steps.  COLLECTIVE: every rank of the reducer's group must call it at
        the same point of its program -- and so must state_dict() / step() of an optimiser that has taken sharded steps
        (a rank-0-only `torch.save(opt.state_dict())` under data parallelism hangs: gather on every rank first, e.g.
        through StepEngine.close() or CMMVAEModel.gather_optimizer_state(), then save on one).  Raises when the moments
        are sharded and the gather cannot run any more (process group destroyed): a checkpoint written then would hold
        stale moments for (world - 1) / world of the arena."""
        if not self.sharded:
            return
        if not mdist.collectives_active():
            raise RuntimeError("HipAdam.sync_sharded_state: this optimiser took sharded data-parallel steps and the process "
                               "group is gone -- the Adam moments of the other ranks' slices are stale here.  Gather them "
                               "(engine.close() / sync_sharded_state() on every rank) BEFORE destroy_process_group().")
        group = self.reducer.group if self.reducer is not None else None
        world = torch.distributed.get_world_size(group)
        rank = torch.distributed.get_rank(group)
        a = self.arena
        per, lo, n_loc = a.shard(world, rank)
        dev = a.data.device
        for t in (a.exp_avg, a.exp_avg_sq):
            full = torch.zeros(per * world, dtype=torch.float32, device=dev)
            mine = torch.zeros(per, dtype=torch.float32, device=dev)
            mine[:n_loc] = t[lo:lo + n_loc]
            torch.distributed.all_gather_into_tensor(full, mine, group=group)
            t.copy_(full[:a.numel])
        self.sharded = False

    # ---- Lightning-style clipping hook: remembered, applied inside step()
    def set_clip(self, max_norm: Optional[float]) -> None:
        self.max_grad_norm = max_norm
        if getattr(self, "clip_value", None):
            self.set_clip_value(None)

    def set_clip_value(self, bound: Optional[float]) -> None:
        """gradient_clip_algorithm "value" (config.py:8): every gradient element clamped to [-bound, bound] inside the
        fused update (state word 5; include/mmvae_hip.h); None / 0 switches it off.  Replaces the norm clip."""
        self.clip_value = float(bound) if bound else None
        if self.clip_value:
            self.max_grad_norm = None
        self.state_dev[5] = self.clip_value or 0.0

    @property
    def grad_norm(self) -> torch.Tensor:
        """Pre-clip global gradient norm of the last compute_norm()/step() (device scalar; no sync)."""
        return self.state_dev[1]

    @torch.no_grad()
    def compute_grad_norm(self) -> torch.Tensor:
        """Gather grads into the arena and compute their global L2 norm (device scalar).  step() reuses it."""
        a = self.arena
        self._inactive = self._gather()
        self._allreduce()
        if self._hip:
            b1, b2 = self.param_groups[0]["betas"]
            ops.clip_adam_step(a.data, a.grad, a.exp_avg, a.exp_avg_sq, self.state_dev, self.partials, beta1=b1,
                               beta2=b2, max_norm=self.max_grad_norm or 0.0, grad_scale=self.grad_scale,
                               do_norm=True, do_step=False)
        else:
            self.state_dev[1] = (a.grad * self.grad_scale).norm()
        self._norm_valid = True
        return self.state_dev[1].clone()  # detached from the state word, which the next pass overwrites

    def _allreduce(self) -> None:
        """Sum the gradient arena over the ranks.  A parameter takes part in the step when ANY rank produced a gradient
        for it (DDP with unused parameters: the ranks that did not use it contribute zeros, every rank then steps it
        identically) -- with condition blocks, which blocks a rank saw is a per-rank fact, so the local "no gradient"
        list is replaced by the intersection over the ranks (one small MAX all-reduce of presence flags; only for
        optimisers declared to hold such blocks: the call sequence must be the same on every rank)."""
        if self.reducer is None:
            return
        self.reducer.launch(self.arena.grad)
        if (self.managed or self.sparse_presence) and mdist.collectives_active():
            n = len(self.arena.params)
            present = torch.ones(n, dtype=torch.int32, device=self.arena.grad.device)
            if self._inactive:
                present[torch.as_tensor(self._inactive, dtype=torch.long, device=present.device)] = 0
            torch.distributed.all_reduce(present, op=torch.distributed.ReduceOp.MAX, group=self.reducer.small_group)
            self._inactive = torch.nonzero(present == 0).flatten().tolist()
        self.reducer.wait()

    def _gather(self) -> List[int]:
        """Autograd gradients into the arena; returns the parameters without a gradient this step.  Parameters managed
        by the grouped kernels are not walked one by one (thousands of condition blocks): those written this step are
        present, the rest absent."""
        zeroed = getattr(self, "_zeroed", False)
        if not self.managed:
            return self.arena.gather_grads(self._direct, zeroed)
        if self._walk is None or len(self._walk) + len(self.managed) != len(self.arena.params):
            self._walk = [i for i in range(len(self.arena.params)) if i not in self.managed]
        inactive = self.arena.gather_grads(self._direct, zeroed, only=self._walk)
        if not zeroed:
            for i in self.managed - self._direct:
                self.arena.grad_view(i).zero_()
        return inactive + sorted(self.managed - self._direct)

    def note_direct_grads(self, indices) -> None:
        """The gradients of these parameters (arena indices) were written straight into the gradient arena by a kernel
        (no autograd .grad tensor exists): they count as present in this step."""
        self._direct.update(int(i) for i in indices)

    def zero_grad(self, set_to_none: bool = True) -> None:
        self._norm_valid = False
        self._direct.clear()
        for p in self.arena.params:
            p.grad = None
        # Arenas of conditional-layer models hold thousands of tensors of which a step touches a few: one memset of the
        # gradient arena here replaces a zeroing launch per untouched tensor in gather_grads.  Plain arenas need no
        # clearing: every live gradient is overwritten before it is read.
        self._zeroed = len(self.arena.params) > 64
        if self._zeroed:
            self.arena.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("closures are not used by the MMVAE trainer")
        a = self.arena
        self._settle()
        self.sync_sharded_state()
        reuse = getattr(self, "_norm_valid", False)
        if not reuse:
            self._inactive = self._gather()
            self._allreduce()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        if self._inactive or self._steps is not None:
            self._step_partial(g, b1, b2, reuse)
        elif self._hip:
            ops.clip_adam_step(a.data, a.grad, a.exp_avg, a.exp_avg_sq, self.state_dev, self.partials, lr=g["lr"],
                               beta1=b1, beta2=b2, eps=g["eps"], weight_decay=g["weight_decay"],
                               max_norm=self.max_grad_norm or 0.0, grad_scale=self.grad_scale, do_norm=not reuse)
        else:
            self._step_cpu_plumbing(g, b1, b2)
        self._norm_valid = False

    JOB_DTYPE = [("offset", "<i8"), ("len", "<i4"), ("bc1", "<f4"), ("bc2", "<f4"), ("reserved", "<i4")]

    def host_steps(self):
        """Per-parameter step counts kept on the host (created from the common device count on first use)."""
        import numpy as np

        if self._steps is None:
            self._steps = np.full(len(self.arena.params), int(round(float(self.state_dev[0]))), dtype=np.int64)
        return self._steps

    def job_table(self, active, b1, b2, with_owner: bool = False):
        """`mmvae_adam_job` records (numpy structured array) of the parameters `active` (indices into the arena) for
        their NEXT step: a job per <= 16384-element chunk carrying the tensor's own bias corrections.  `with_owner`:
        also the position in `active` of the parameter every job belongs to."""
        import numpy as np

        a = self.arena
        if self._numel is None:
            self._numel = np.fromiter((p.numel() for p in a.params), dtype=np.int64, count=len(a.params))
            self._offsets = np.asarray(a.offsets, dtype=np.int64)
        t_act = np.asarray(self.host_steps(), dtype=np.int64)[active] + 1
        bc1 = (np.float32(1.0) - np.power(np.float32(b1), t_act.astype(np.float32))).astype(np.float32)
        bc2 = (np.float32(1.0) - np.power(np.float32(b2), t_act.astype(np.float32))).astype(np.float32)
        off, num = self._offsets[active], self._numel[active]
        J = 16384
        nchunk = (num + J - 1) // J
        owner = np.repeat(np.arange(len(active)), nchunk)
        first = np.cumsum(nchunk) - nchunk
        k = np.arange(int(nchunk.sum())) - first[owner]
        jobs = np.zeros(len(owner), dtype=np.dtype(self.JOB_DTYPE))
        jobs["offset"] = off[owner] + k * J
        jobs["len"] = np.minimum(num[owner] - k * J, J)
        jobs["bc1"], jobs["bc2"] = bc1[owner], bc2[owner]
        return (jobs, owner) if with_owner else jobs

    def _step_partial(self, g, b1, b2, norm_valid: bool):
        """A step in which some parameters have no gradient, or after such a step: torch.optim.Adam semantics --
        parameters without a gradient are left untouched (no moment decay, no weight decay) and every parameter uses
        the bias corrections of its OWN step count.  The global norm / clip coefficient come from the whole arena (the
        skipped slices are zero); the update runs over a job table of the tensors that took part, in one launch
        (mmvae_adam_step_jobs)."""
        import numpy as np

        a = self.arena
        n = len(a.params)
        if self._steps is None:
            self._steps = np.full(n, int(round(float(self.state_dev[0]))), dtype=np.int64)
        skip = set(self._inactive)
        if self._hip:
            # norm (unless compute_grad_norm() just produced it) and the clip coefficient for the CURRENT max_grad_norm
            # (Lightning sets the clip value between the norm logging and the step)
            ops.clip_adam_step(a.data, a.grad, a.exp_avg, a.exp_avg_sq, self.state_dev, self.partials, beta1=b1,
                               beta2=b2, max_norm=self.max_grad_norm or 0.0, grad_scale=self.grad_scale,
                               do_norm=not norm_valid, do_step=False)
            norm, clip = self.state_dev[1], self.state_dev[2]
        else:
            grad = a.grad * self.grad_scale
            norm = grad.norm()
            clip = 1.0
            if self.max_grad_norm:
                clip = min(1.0, float(self.max_grad_norm) / (float(norm) + 1e-6))
            self.state_dev[1], self.state_dev[2] = norm, clip
        active = np.setdiff1d(np.arange(n), np.fromiter(skip, dtype=np.int64, count=len(skip)), assume_unique=True)
        steps = np.asarray(self._steps, dtype=np.int64)
        t_act = steps[active] + 1
        bc1 = (np.float32(1.0) - np.power(np.float32(b1), t_act.astype(np.float32))).astype(np.float32)
        bc2 = (np.float32(1.0) - np.power(np.float32(b2), t_act.astype(np.float32))).astype(np.float32)
        if self._hip:
            # one launch for every tensor that took part: a job per <= 16384-element chunk, carrying the tensor's own
            # bias corrections (mmvae_adam_step_jobs)
            from . import _lib

            jobs = self.job_table(active, b1, b2)
            jobs_dev = torch.from_numpy(jobs.view(np.uint8)).to(a.device, non_blocking=False)
            lib = _lib.load()
            rc = lib.mmvae_adam_step_jobs(len(jobs), jobs_dev.data_ptr(), a.data.data_ptr(), a.grad.data_ptr(),
                                          a.exp_avg.data_ptr(), a.exp_avg_sq.data_ptr(), self.state_dev.data_ptr(),
                                          g["lr"], b1, b2, g["eps"], g["weight_decay"], float(self.grad_scale),
                                          torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "mmvae_adam_step_jobs")
            self._keep_jobs = jobs_dev  # outlives the launch
        else:
            for j, i in enumerate(active):
                p, o = a.params[i], a.offsets[i]
                sl = slice(o, o + p.numel())
                gr = a.grad[sl] * self.grad_scale * float(clip)
                if getattr(self, "clip_value", None):  # gradient_clip_algorithm "value" (config.py:8)
                    gr = gr.clamp(-self.clip_value, self.clip_value)
                gr = gr + g["weight_decay"] * a.data[sl]
                a.exp_avg[sl].lerp_(gr, 1 - b1)
                a.exp_avg_sq[sl].mul_(b2).addcmul_(gr, gr, value=1 - b2)
                denom = a.exp_avg_sq[sl].sqrt() / (float(bc2[j]) ** 0.5) + g["eps"]
                a.data[sl].addcdiv_(a.exp_avg[sl], denom, value=-g["lr"] / float(bc1[j]))
        steps[active] = t_act
        self._steps = steps
        self.state_dev[0] = float(self._steps.max())
        self._norm_valid = False

    def _step_cpu_plumbing(self, g, b1, b2):
        a = self.arena
        grad = a.grad * self.grad_scale
        norm = grad.norm()
        clip = 1.0
        if self.max_grad_norm:
            clip = min(1.0, float(self.max_grad_norm) / (float(norm) + 1e-6))
        self.state_dev[0] += 1
        t = float(self.state_dev[0])
        self.state_dev[1], self.state_dev[2] = norm, clip
        grad = grad * clip
        if getattr(self, "clip_value", None):
            grad = grad.clamp(-self.clip_value, self.clip_value)
        grad = grad + g["weight_decay"] * a.data
        a.exp_avg.lerp_(grad, 1 - b1)
        a.exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        denom = a.exp_avg_sq.sqrt() / (bc2 ** 0.5) + g["eps"]
        a.data.addcdiv_(a.exp_avg, denom, value=-g["lr"] / bc1)

    # ---- checkpoint surface compatible with torch.optim.Adam's per-parameter state
    def _settle(self) -> None:
        if self.settle is not None:
            self.settle()

    def state_dict(self):
        a = self.arena
        self._settle()
        self.sync_sharded_state()
        step_of = (lambda i: torch.tensor(float(self._steps[i]))) if self._steps is not None else (
            lambda i: self.state_dev[0].detach().clone().cpu())
        st = {i: {"step": step_of(i), "exp_avg": a.view(a.exp_avg, i).clone(),
                  "exp_avg_sq": a.view(a.exp_avg_sq, i).clone()} for i in range(len(a.params))}
        return {"state": st, "param_groups": [{**{k: v for k, v in self.param_groups[0].items() if k != "params"},
                                               "params": list(range(len(a.params)))}]}

    def load_state_dict(self, sd):
        a = self.arena
        self._settle()
        steps = {}
        for i, s in sd["state"].items():
            i = int(i)
            a.view(a.exp_avg, i).copy_(s["exp_avg"])
            a.view(a.exp_avg_sq, i).copy_(s["exp_avg_sq"])
            steps[i] = int(round(float(s["step"])))
        if steps:
            self.state_dev[0] = float(max(steps.values()))
            uniform = len(set(steps.values())) == 1 and len(steps) == len(a.params)
            self._steps = None if uniform else np.array([steps.get(i, 0) for i in range(len(a.params))], dtype=np.int64)
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
