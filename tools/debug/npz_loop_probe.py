#!/usr/bin/env python3
"""The npz-fed C2 training loop of bench.py --input npz, with the main thread's time per step split into: waiting for the
feed (next), announcing the next batch, training_step (staging + replay).  FEED_REP=r: chunks of 4 r batches."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pandas as pd, scipy.sparse as sp, torch
from mmvae_amd import data as mdata, instantiate, synthetic
from mmvae_amd.trainer import MultiModalBatches

def main():
    global model, feed, pending, T, P
    if os.environ.get("SWITCH_US"):
        sys.setswitchinterval(float(os.environ["SWITCH_US"]) * 1e-6)
    dev = torch.device("cuda", 0)
    B, G = 512, 20000
    model = instantiate.load_yaml(os.path.join(os.path.dirname(__file__), "..", "..", "configs", "model", "c2_two_modality_20k.yaml")).to(dev)
    model.train(); model.trainer.set_stage("training"); model.optimizers()
    REP = int(os.environ.get("FEED_REP", "1"))
    tmp = tempfile.mkdtemp(prefix="npz_probe_")
    feeds = {}
    for i, eid in enumerate(("human", "mouse")):
        rows = torch.cat([synthetic.synthetic_counts(B, G, seed=77 + 31 * i + j, device="cpu") for j in range(8)])
        meta = pd.concat([synthetic.synthetic_metadata(B, seed=9 + j) for j in range(8)], ignore_index=True)
        mdata.write_chunks(os.path.join(tmp, eid), eid, sp.vstack([sp.csr_matrix(rows.numpy())] * REP, format="csr"),
                           pd.concat([meta] * REP, ignore_index=True), chunk_rows=4 * REP * B, compressed=False)
        feeds[eid] = mdata.SpeciesChunks(os.path.join(tmp, eid), f"{eid}_train_counts_*.npz", f"{eid}_train_metadata_*.pkl", B, eid,
                                         seed=i, device=dev, workers=int(os.environ.get("FEED_WORKERS", "3")))

    specs = {eid: dict(directory_path=os.path.join(tmp, eid), npz_masks=f"{eid}_train_counts_*.npz",
                       metadata_masks=f"{eid}_train_metadata_*.pkl", batch_size=B, name=eid, seed=i)
             for i, eid in enumerate(("human", "mouse"))}


    def endless():
        while True:
            yield from MultiModalBatches(feeds, seed=0, round_robin=True)

    P = {}
    def timed(owner, name, tag):
        orig = getattr(owner, name)
        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return orig(*a, **k)
            finally:
                P[tag] = P.get(tag, 0.0) + time.perf_counter() - t0
        setattr(owner, name, wrapper)
    timed(mdata.SpeciesChunks, "_tensor", "tensor+h2d")
    timed(mdata.SpeciesChunks, "_gather", "gather(worker)")
    timed(mdata, "load_chunk", "load_chunk(thread)")
    timed(mdata._Workers, "submit", "submit")
    timed(mdata._Workers._Job, "result", "wait gather")
    timed(pd.DataFrame, "take", "take")
    timed(torch.cuda.Event, "synchronize", "slot event sync")
    import multiprocessing.queues as _mq
    timed(_mq.Queue, "get", "mp queue get (incl. unpickling)")
    timed(torch.Tensor, "to", "tensor.to")
    import queue as _q
    timed(_q.Queue, "put", "queue put (producer blocked when full)")
    feed = iter(mdata.Prefetcher(endless(), depth=int(os.environ.get("FEED_DEPTH", "3")), device=dev))
    if os.environ.get("FEED_PREFILL", "0") == "1":
        # every batch of the run produced BEFORE the loop (events and all): the loop then shares the interpreter with nobody
        src = iter(mdata.Prefetcher(endless(), depth=3, device=dev))
        ready = [next(src) for _ in range(340)]
        torch.cuda.synchronize()
        del src
        import gc; gc.collect()
        time.sleep(0.5)
        feed = iter(ready)
    pending = [next(feed)]
    T = {"next": 0.0, "hint": 0.0, "step": 0.0}
    def step(i, timed):
        x, meta, eid = pending.pop()
        t0 = time.perf_counter()
        pending.append(next(feed))
        t1 = time.perf_counter()
        if os.environ.get("NO_HINT", "0") != "1":
            model.hint_next_batch(pending[0])
        t2 = time.perf_counter()
        model.training_step((x, meta, eid), i)
        t3 = time.perf_counter()
        if timed:
            T["next"] += t1 - t0; T["hint"] += t2 - t1; T["step"] += t3 - t2
    for i in range(30):
        step(i, False)
    torch.cuda.synchronize()
    P.clear()
    n = 300
    t0 = time.perf_counter()
    for i in range(n):
        step(30 + i, True)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"rep {REP}: {1e3 * el / n:.3f} ms / step; main thread per step: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in T.items()),
          "| producer side per batch: " + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in sorted(P.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":  # (ProcessFeed spawns: the child imports this file)
    main()
