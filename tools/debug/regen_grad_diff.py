"""Debug aid: engine gradients of a regen case's first step against the oracle's (full tensors), per parameter."""
import sys, os, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mmvae_oracle as O
from tests import helpers as H, mirror_utils as MU

name = sys.argv[1] if len(sys.argv) > 1 else "mid_odd"
use_engine = (sys.argv[2] != "module") if len(sys.argv) > 2 else True
case, z, results = MU.replay_regen(name, "cuda", use_engine=use_engine, steps=1)
spec, hp = H.spec_from_case(case), H.hparams_from_case(case)
with tempfile.TemporaryDirectory() as d:
    from mmvae_amd import backend
    with backend.cpu_plumbing():
        twin = MU.build_mirror(case, "cpu", d, use_engine=False).module
        eid = case["schedule"][0]
        H.regen_state(case, 0, twin, eid)
        sd = {k: v.detach().clone() for k, v in twin.state_dict().items()}
x, eps, masks, labels = H.RegenStream(case).step(0, eid)
out, _ = O.train_step(spec, sd, {}, x, eid, eps, masks, labels or None, case["kl_weights"][0], hp)
for n, g in results[0]["grads"].items():
    ref = out["grads"].get(n)
    if ref is None:
        continue
    d = (g.double() - ref.double())
    rel = float(d.norm() / ref.double().norm())
    i = int(d.abs().argmax())
    print(f"{n:55s} rel-L2 {rel:.2e}  max|d| {float(d.abs().max()):.3e} at {np.unravel_index(i, tuple(g.shape))} ref {float(ref.flatten()[i]):.3e} |ref|max {float(ref.abs().max()):.3e}")
    if n.endswith("encoder.fc_layers.0.lin.weight"):
        col = d.abs().amax(0)
        top = torch.topk(col, 8)
        print("   worst columns", top.indices.tolist(), [f"{v:.2e}" for v in top.values.tolist()], "col norms ref", [f"{float(ref[:, c].norm()):.2e}" for c in top.indices.tolist()])
        row = d.abs().amax(1)
        top = torch.topk(row, 8)
        print("   worst rows", top.indices.tolist(), [f"{v:.2e}" for v in top.values.tolist()])
