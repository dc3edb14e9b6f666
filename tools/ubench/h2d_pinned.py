"""Host-to-device copy rate of a 24 MB batch from page-locked and from pageable memory (data feed): on this
pool the asynchronous copy out of page-locked memory runs on the SDMA path at ~21 GB/s, `HSA_ENABLE_SDMA=0` (shader
copies) reaches ~52 GB/s at the price of compute units."""
import time, torch, numpy as np
n = 24 * 1024 * 1024 // 4
dev = torch.zeros(n, device="cuda")
for kind in ("pinned", "pageable"):
    h = torch.zeros(n).pin_memory() if kind == "pinned" else torch.zeros(n)
    src = np.random.rand(n).astype(np.float32)
    hv = h.numpy()
    t0 = time.perf_counter()
    for _ in range(10): hv[:] = src
    w = (time.perf_counter() - t0) / 10
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): dev.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); c = (time.perf_counter() - t0) / 10
    print(f"{kind}: host write of 24 MB {w*1e3:.2f} ms ({24/w/1e3:.1f} GB/s); H2D {c*1e3:.2f} ms ({24/c/1e3:.1f} GB/s)")
