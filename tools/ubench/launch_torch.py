"""Boundary cost of trivial libmmvae_hip.so kernels replayed from a torch-captured graph vs launched eagerly."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mmvae_amd import _lib
lib = _lib.load()
x = torch.zeros(1024, device="cuda"); y = torch.zeros(1024, device="cuda")
N = 400
def run():
    s = torch.cuda.current_stream().cuda_stream
    for i in range(N):
        lib.mmvae_axpby(1, 1.0, x.data_ptr(), 0.0, y.data_ptr(), s)
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); run(); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"eager (python ctypes loop): {(t1 - t0) / N * 1e6:.2f} us/kernel")
for mode in ("global", "thread_local"):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=mode):
        run()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"torch graph ({mode}): {(t1 - t0) / (5 * N) * 1e6:.2f} us/kernel")
# side stream replay
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"torch graph replayed on a side stream: {(t1 - t0) / (5 * N) * 1e6:.2f} us/kernel")
