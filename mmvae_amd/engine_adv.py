"""The adversaries of a step plan (mixin of mmvae_amd.engine._Plan).  Reference: CMMVAEModel.grf /
gradient_reversal_domain_classifier (models/cmmvae_model.py:59-136), Adversarial (modules/base/components.py:638-674).
Two programs: the fused row-owner passes (adversaries without BatchNorm: every configuration of the reference) and the
per-layer program (any FCBlock the engine supports)."""
from __future__ import annotations

import torch

from . import _lib, cond_tables
from .engine_common import ACC, NN, NT, TN, _LayerRef, _PinnedRing, _p
from .modules.base.components import Adversarial
from .optim import arena_of


class PlanAdversaries:
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
            if dp:  # ONE exchange point per phase: every adversary's gradient arena, then their optimisers' launches
                self._flush_sums()
                self._cut(("ar_many", [net.opt for net in nets]))
            for i, net in enumerate(nets, start=1):
                if dp:
                    self.optimizer(net.opt, 0.0 if gen else self.clip_adv, step=not gen, exchange="done")
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
            if H > 1:
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
                    if max(widths) <= 8192:  # every head's cross-entropy in one launch
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
            self._labels_tmp = np.zeros(self.B, dtype=np.int32)
        slot = self._label_ring.take()
        tmp = self._labels_tmp
        for i, c in enumerate(self.conditions):
            table = Adversarial.labels[c]
            # (csrc/pylookup.c through the CPython API: 4 x 512 look-ups per step were a quarter of the host's share of a C4
            # step from the interpreter; an unknown label still raises KeyError)
            values = metadata[c].tolist()
            hit = cond_tables.lookup_i32(table, values, tmp)
            if hit != n:
                raise KeyError(values[hit])
            slot[i * n:(i + 1) * n] = tmp
        self._label_ring.upload(self._labels_all.view(-1))
