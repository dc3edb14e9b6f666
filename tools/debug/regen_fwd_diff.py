"""Debug aid: module-path forward activations of a regen case's first step against the oracle's."""
import sys, os, tempfile
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import mmvae_oracle as O
from tests import helpers as H, mirror_utils as MU
from mmvae_amd import backend

name = sys.argv[1] if len(sys.argv) > 1 else "mid_odd"
case, z = H.load_case(name)
spec, hp = H.spec_from_case(case), H.hparams_from_case(case)
eid = case["schedule"][0]
with tempfile.TemporaryDirectory() as d:
    with backend.cpu_plumbing():
        twin = MU.build_mirror(case, "cpu", d, use_engine=False).module
        H.regen_state(case, 0, twin, eid)
        sd = {k: v.detach().clone() for k, v in twin.state_dict().items()}
    model = MU.build_mirror(case, "cpu", d + "/", use_engine=False).to("cuda")
    model.module.load_state_dict(sd)
    model.train()
    x, eps, masks, labels = H.RegenStream(case).step(0, eid)
    ref = O.model_forward(spec, sd, x, eid, eps, True, masks, hp, {})
    model.module.vae.encoder.explicit_eps = eps.cuda()
    model.module.experts[eid].encoder.explicit_masks = {int(k.split(".")[4]): m.cuda() for k, m in masks.items()}
    # layer by layer through the expert encoder
    enc = model.module.experts[eid].encoder
    h = x.cuda()
    href = x
    for i, layer in enumerate(enc.fc_layers):
        h, a = enc._hip_layer(i, layer, h)
        p = f"experts.{eid}.encoder.fc_layers.{i}"
        zz = torch.nn.functional.linear(href, sd[p + ".lin.weight"], sd[p + ".lin.bias"])
        z64 = torch.nn.functional.linear(href.double(), sd[p + ".lin.weight"].double(), sd[p + ".lin.bias"].double())
        print(f"enc layer {i}: cpu-fp32 linear vs fp64 rel-L2 {H.rel_l2(zz, z64):.2e}")
        mean, var = zz.mean(0), zz.var(0, unbiased=False)
        y = (zz - mean) / torch.sqrt(var + 1e-3) * sd[p + ".bn.weight"] + sd[p + ".bn.bias"]
        y = torch.relu(y) * masks[p + ".dr"].float() / 0.9
        print(f"enc layer {i}: out rel-L2 vs oracle-style fp32 {H.rel_l2(h, y):.2e}; min batch var {float(var.min()):.3e} max |mean|/std {float((mean.abs()/var.sqrt()).max()):.2e}")
        href = y
    qz, pz, zz, xhats, hidden = model.module(x.cuda(), pd.DataFrame({"dummy": [0] * x.shape[0]}), eid)
    print("mu   ", H.rel_l2(qz.loc, ref["mu"]))
    print("std  ", H.rel_l2(qz.scale, ref["std"]))
    print("z    ", H.rel_l2(zz, ref["z"]))
    print("xhat ", H.rel_l2(xhats[eid], ref["xhat"]))
    for i, (a, b) in enumerate(zip(hidden, ref["hidden"])):
        print("hidden", i, H.rel_l2(a, b))
    a, b = xhats[eid].detach().cpu(), ref["xhat"]
    flip = (a > 0) != (b > 0)
    idx = flip.nonzero()
    print("ReLU sign flips at the output:", int(flip.sum()), "of", a.numel())
    for r, c in idx[:20].tolist():
        print(f"   ({r},{c}) hip {float(a[r,c]):.3e} ref {float(b[r,c]):.3e} x {float(x[r,c]):.3f}")
    for i, (ha, hb) in enumerate(zip(hidden, ref["hidden"])):
        print("hidden", i, "flips", int(((ha.detach().cpu() > 0) != (hb > 0)).sum()))
