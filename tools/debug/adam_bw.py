import sys, torch
sys.path.insert(0, "/root/repo")
from mmvae_amd import _lib
lib = _lib.load()
dev = "cuda"
for n in (10_000_000, 42_000_000, 84_000_000, 124_000_000, 200_000_000):
    p, g, m, v = (torch.randn(n, device=dev) * 0.01 for _ in range(4))
    v.abs_()
    state = torch.tensor([1.0, 1.0, 1.0, 0.1, 0.001, 0.0, 0.0, 0.0], device=dev)
    def run():
        lib.mmvae_adam_step(n, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), state.data_ptr(), 1e-3, 0.9, 0.999, 1e-8, 1e-6, 1.0,
                            torch.cuda.current_stream().cuda_stream)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"n = {n/1e6:6.0f} M: {us:8.1f} us  {28.0 * n / us / 1e6:6.2f} TB/s")
    del p, g, m, v
