#!/usr/bin/env python3
"""The npz-CSR feed by itself (bench.py --input npz builds the same): batches per second of the Prefetcher with the
consumer doing nothing, and the producer's time per batch by piece (gather wait, tensors + H2D, metadata slice)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pandas as pd, scipy.sparse as sp, torch
from mmvae_amd import data as mdata, synthetic
from mmvae_amd.trainer import MultiModalBatches

B, G = 512, 20000
dev = torch.device("cuda", 0)
tmp = tempfile.mkdtemp(prefix="feed_probe_")
feeds = {}
for i, eid in enumerate(("human", "mouse")):
    rows = torch.cat([synthetic.synthetic_counts(B, G, seed=77 + 31 * i + j, device="cpu") for j in range(8)])
    meta = pd.concat([synthetic.synthetic_metadata(B, seed=9 + j) for j in range(8)], ignore_index=True)
    REP = int(os.environ.get("FEED_REP", "1"))  # chunk = 4 * REP batches
    mdata.write_chunks(os.path.join(tmp, eid), eid, sp.vstack([sp.csr_matrix(rows.numpy())] * REP, format="csr"),
                       pd.concat([meta] * REP, ignore_index=True), chunk_rows=4 * REP * B, compressed=False)
    feeds[eid] = mdata.SpeciesChunks(os.path.join(tmp, eid), f"{eid}_train_counts_*.npz", f"{eid}_train_metadata_*.pkl", B, eid,
                                     seed=i, device=dev, workers=int(os.environ.get("FEED_WORKERS", "3")),
                                     gather_threads=int(os.environ.get("FEED_GT", "2")))
T = {}
def timed(owner, name, tag):
    orig = getattr(owner, name)
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            T[tag] = T.get(tag, 0.0) + time.perf_counter() - t0
    setattr(owner, name, wrapper)
timed(mdata.SpeciesChunks, "_tensor", "tensor+h2d")
timed(mdata.SpeciesChunks, "_gather", "gather(worker)")
timed(mdata, "load_chunk", "load_chunk(thread)")
timed(torch, "sparse_csr_tensor", "sparse_csr_tensor")
import concurrent.futures as _cf
timed(_cf.Future, "result", "wait for gather")
timed(_cf.ThreadPoolExecutor, "submit", "submit")
timed(torch.cuda.Event, "synchronize", "slot event sync")
_iloc_cls = type(pd.DataFrame({"a": [1]}).iloc)
timed(_iloc_cls, "__getitem__", "iloc")
orig_iloc = pd.DataFrame.reset_index
def ri(self, *a, **k):
    t0 = time.perf_counter(); r = orig_iloc(self, *a, **k); T["reset_index"] = T.get("reset_index", 0) + time.perf_counter() - t0; return r
pd.DataFrame.reset_index = ri

def endless():
    while True:
        yield from MultiModalBatches(feeds, seed=0, round_robin=True)

for label, it in (("direct (no Prefetcher)", endless()), ("Prefetcher depth 3", iter(mdata.Prefetcher(endless(), depth=3, device=dev)))):
    for _ in range(20):
        next(it)
    torch.cuda.synchronize(); T.clear()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
        x, md, eid = next(it)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:26s}: {1e3 * (t1 - t0) / n:.3f} ms per batch; nnz {x.values().numel()}; pieces (ms per batch): "
          + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in T.items()))
