"""Diagnostic: where the row-owner adversary kernels (csrc/adv_fused.hip) spend their time at config C4's shapes.
Builds a stamped copy of the library on the GPU box (-DMMVAE_ADV_STAMPS=1 for adv_fused.hip only), runs both
adversaries as two jobs and prints, per phase, when the workgroups got there (us from the first workgroup's start)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "mmvae_amd", "csrc")
out = os.path.join(ROOT, "tools", "ubench", "libmmvae_hip_advstamps.so")
objs = [os.path.join(csrc, f) for f in ("gemm_f32.o", "gemm_planes.o", "fc_epilogue.o", "elbo_optim.o", "cond_layers.o")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DMMVAE_ADV_STAMPS=1",
                       "-c", os.path.join(csrc, "adv_fused.hip"), "-o", "/tmp/adv_stamps.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, "/tmp/adv_stamps.o"] + objs)
os.environ["MMVAE_LIB"] = out

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mmvae_amd import _lib  # noqa: E402
from mmvae_amd.adv_program import AdvProgram  # noqa: E402
from tests.test_adv_fused_gpu import _Pool, _build  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda")
B, classes = 512, [8, 2, 273, 4644]
H = len(classes)
g = torch.Generator().manual_seed(11)
labels = torch.stack([torch.randint(0, c, (B,), generator=g) for c in classes]).to(dev)
nets = [_build(w, classes, B, seed=s, p_drop=0.0, device=dev)[0] for w, s in (([256, 128, 64], 1), ([128, 64], 2))]
metrics = torch.zeros(256, device=dev)
prog = AdvProgram(lib, _Pool(dev), nets, B, labels, dev)
cfg = dict(gscale=25.0, reverse=True, loss_each=[metrics.data_ptr() + 32 * i for i in range(2)],
           loss_total=[metrics.data_ptr() + 4 * (8 * i + H) for i in range(2)], total_loss=metrics.data_ptr() + 400,
           total_scale=25.0, opts=[dict(flags=_lib.PREPARE_NORM, max_norm=10.0, norm_out=None)] * 2)
prog.build_phase("g", cfg)
for _ in range(5):
    prog.launch_pass("g")
    prog.launch_dw("g")
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
e0.record()
prog.launch_pass("g")
e1.record()
prog.launch_dw("g")
e2.record()
torch.cuda.synchronize()
print(f"events: pass {e0.elapsed_time(e1) * 1e3:.1f} us, dw {e1.elapsed_time(e2) * 1e3:.1f} us; splits {prog.splits}")
fn = lib.mmvae_debug_adv_trace
fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
names = (["start", "encoder fwd done", "heads done"], ["start", "merged: lse + losses", "d encoded done", "backward done"],
         ["setup done", "k loop done", "stored + ticket"])
n_rt = (B + 15) // 16
for which, kname, nb in ((0, "forward", n_rt * prog.splits * 2), (1, "backward", n_rt * 2),
                         (2, "dw", min(1024, prog.phase_tables["g"]["blocks"]))):
    buf = (ctypes.c_longlong * (8 * nb))()
    assert fn(which, buf, nb) == 0
    T = np.array(buf[:], dtype=np.int64).reshape(nb, 8).astype(np.float64)
    t0 = T[:, 0].min()
    print(f"{kname} kernel, {nb} workgroups")
    for i, nm in enumerate(names[which]):
        v = (T[:, i] - t0) / 100.0
        print(f"   {nm:28s} min {v.min():7.1f}  median {np.median(v):7.1f}  max {v.max():7.1f} us")
    if which == 0:
        n = np.maximum(T[:, 6], 1)
        for i, nm in enumerate(["product 1 -> logits", "softmax bookkeeping", "product 2"]):
            print(f"   wave 0, per tile: {nm:22s} median {np.median(T[:, 3 + i] / n) / 100.0:6.2f} us   (tiles per wave: {np.median(n):.0f})")
    if which == 2:
        for i, nm in enumerate(["stage (transform + LDS writes)", "barrier", "fetch issue + MFMAs", "barrier"]):
            print(f"   wave 0, per chunk: {nm:30s} median {np.median(T[:, 3 + i]) / 4 / 100.0:6.2f} us")
