"""the chunked npz-CSR / pickled-metadata feed against the on-disk format and batch semantics of the
reference's local datapipes (data/local/cellxgene_datapipe.py:31-193, cellxgene_manager.py:76-88).  Host logic: CPU."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp
import torch

from mmvae_amd import data as D
from mmvae_amd.trainer import MultiModalBatches


def _dataset(n, g, seed):
    rng = np.random.default_rng(seed)
    dense = rng.random((n, g), dtype=np.float32) * (rng.random((n, g)) < 0.1)
    dense[:, 0] = np.arange(n, dtype=np.float32) + 1.0  # column 0 identifies the row
    meta = pd.DataFrame({"row": np.arange(n), "assay": [f"a{i % 3}" for i in range(n)]})
    return sp.csr_matrix(dense), meta


def test_chunks_are_listed_sorted_and_paired(tmp_path):
    m, meta = _dataset(50, 16, 0)
    D.write_chunks(str(tmp_path), "human", m, meta, chunk_rows=20)
    D.write_chunks(str(tmp_path), "human", m[:10], meta.iloc[:10], chunk_rows=20, split="val")
    pairs = D.list_chunks(str(tmp_path), "human_train_counts_*.npz", "human_train_metadata_*.pkl")
    assert [p[0].rsplit("_", 1)[1] for p in pairs] == ["1.npz", "2.npz", "3.npz"]
    assert all(a.replace("counts", "metadata").replace(".npz", ".pkl") == b for a, b in pairs)
    with pytest.raises(RuntimeError, match="No files found"):
        D.list_chunks(str(tmp_path), "mouse_*.npz", "mouse_*.pkl")


@pytest.mark.parametrize("shuffle", [False, True])
@pytest.mark.parametrize("return_dense", [False, True])
def test_batches_keep_rows_and_metadata_index_matched(tmp_path, shuffle, return_dense):
    m, meta = _dataset(70, 24, 1)
    D.write_chunks(str(tmp_path), "human", m, meta, chunk_rows=32)  # chunks of 32, 32, 6 rows
    feed = D.SpeciesChunks(str(tmp_path), "human_train_counts_*.npz", ["human_train_metadata_*.pkl"], batch_size=8,
                           name="human", shuffle=shuffle, return_dense=return_dense, seed=3, prefetch=True)
    seen = []
    for x, md, eid in feed:
        assert eid == "human" and len(md) == 8 and list(md.index) == list(range(8))
        if return_dense:
            assert x.layout == torch.strided and x.shape == (8, 24)
            dense = x
        else:
            assert x.layout == torch.sparse_csr and x.dtype == torch.float32
            # int32 indices, as the reference's torch.sparse_csr_tensor over a scipy slice carries them
            assert x.crow_indices().dtype == torch.int32 and x.col_indices().dtype == torch.int32
            dense = x.to_dense()
        rows = (dense[:, 0] - 1).long().tolist()  # column 0 carries the original row number
        assert rows == md["row"].tolist(), "matrix rows and metadata rows must stay index-matched"
        assert torch.equal(dense, torch.from_numpy(m[rows].toarray()))
        seen += rows
    # partial batches are dropped per chunk: 32 -> 4 batches, 32 -> 4, 6 -> 0
    assert len(seen) == 64 and len(set(seen)) == 64
    if not shuffle:
        assert seen == list(range(32)) + list(range(32, 64))
    else:
        assert seen != sorted(seen)
        again = [r for x, md, _ in D.SpeciesChunks(str(tmp_path), "human_train_counts_*.npz",
                                                   "human_train_metadata_*.pkl", 8, "human", seed=3)
                 for r in md["row"].tolist()]
        assert again == seen, "same seed, same epoch -> same order"


def test_partial_batches_and_rank_sharding(tmp_path):
    m, meta = _dataset(45, 8, 2)
    D.write_chunks(str(tmp_path), "mouse", m, meta, chunk_rows=45)
    full = list(D.SpeciesChunks(str(tmp_path), "mouse_*counts*.npz", "mouse_*metadata*.pkl", 10, "mouse",
                                allow_partials=True, shuffle=False))
    assert [len(b[1]) for b in full] == [10, 10, 10, 10, 5]
    shards = [list(D.SpeciesChunks(str(tmp_path), "mouse_*counts*.npz", "mouse_*metadata*.pkl", 10, "mouse", seed=5,
                                   rank=r, world=2)) for r in range(2)]
    assert [len(s) for s in shards] == [2, 2]
    rows = [set(r for _, md, _ in s for r in md["row"].tolist()) for s in shards]
    assert not (rows[0] & rows[1]) and len(rows[0] | rows[1]) == 40


@pytest.mark.parametrize("rows,chunk_rows,world", [(50, 50, 2), (50, 50, 8), (130, 30, 4), (95, 25, 3)])
def test_every_rank_yields_the_same_number_of_batches(tmp_path, rows, chunk_rows, world):
    """Odd batch counts, and more ranks than batches per chunk: one counter over the epoch's kept batches deals rounds of
    `world`; every rank gets floor(n / world) batches, disjoint rows, and the ranks together cover whole rounds."""
    m, meta = _dataset(rows, 8, 4)
    D.write_chunks(str(tmp_path), "mouse", m, meta, chunk_rows=chunk_rows)
    feed = lambda **kw: list(D.SpeciesChunks(str(tmp_path), "mouse_*counts*.npz", "mouse_*metadata*.pkl", 10, "mouse",
                                             seed=5, **kw))
    whole = feed()
    shards = [feed(rank=r, world=world) for r in range(world)]
    assert [len(s) for s in shards] == [len(whole) // world] * world
    rows_of = [[tuple(md["row"].tolist()) for _, md, _ in s] for s in shards]
    flat = [r for s in rows_of for b in s for r in b]
    assert len(flat) == len(set(flat))
    # round k of the un-sharded stream is the ranks' k-th batches, in rank order
    for k in range(len(whole) // world):
        for r in range(world):
            assert rows_of[r][k] == tuple(whole[k * world + r][1]["row"].tolist())


def test_multi_modal_interleave_feeds_the_trainer(tmp_path):
    feeds = {}
    for k, (name, g) in enumerate((("human", 12), ("mouse", 9))):
        m, meta = _dataset(32, g, 10 + k)
        D.write_chunks(str(tmp_path / name), name, m, meta, chunk_rows=16)
        feeds[name] = D.SpeciesChunks(str(tmp_path / name), f"{name}_train_counts_*.npz", f"{name}_train_metadata_*.pkl",
                                      8, name, seed=k)
    batches = list(MultiModalBatches(feeds, seed=1))
    assert sorted(b[2] for b in batches) == ["human"] * 4 + ["mouse"] * 4
    assert all(b[0].shape[1] == (12 if b[2] == "human" else 9) for b in batches)


def test_native_row_gather_matches_scipy_and_validates():
    """libmmvae_feed.so (include/mmvae_feed.h): bit-exact row gather into torch's CSR layout for int32 and int64 chunk
    indices, single- and multi-threaded; bad row indices and short staging buffers are refused."""
    import ctypes as C
    import re

    lib = D.feed_lib()
    header = open(__import__("os").path.join(__import__("os").path.dirname(D.__file__), "..", "include", "mmvae_feed.h")).read()
    for sym in set(re.findall(r"\b(mmvae_feed_[a-z_]+)\s*\(", header)):
        assert hasattr(lib, sym), f"{sym} declared in include/mmvae_feed.h but not exported"
    m, _ = _dataset(300, 40, 7)
    m = m.tocsr()
    rng = np.random.default_rng(0)
    rows = rng.permutation(300)[:97].astype(np.int64)
    ref = m[rows]
    for idx_dtype in (np.int32, np.int64):
        indptr, indices = m.indptr.astype(idx_dtype), m.indices.astype(idx_dtype)
        for threads in (1, 4):
            crow = np.empty(len(rows) + 1, np.int64)
            col = np.full(ref.nnz + 5, -1, np.int64)
            val = np.full(ref.nnz + 5, -1, np.float32)
            got = C.c_int64(0)
            rc = lib.mmvae_feed_gather_rows(indptr.ctypes.data, indices.ctypes.data, indptr.itemsize, m.data.ctypes.data,
                                            m.shape[0], rows.ctypes.data, len(rows), crow.ctypes.data, col.ctypes.data,
                                            val.ctypes.data, len(col), threads, C.byref(got))
            assert rc == 0 and got.value == ref.nnz
            assert np.array_equal(crow, ref.indptr) and np.array_equal(col[:ref.nnz], ref.indices)
            assert np.array_equal(val[:ref.nnz], ref.data) and (col[ref.nnz:] == -1).all()
            assert lib.mmvae_feed_rows_nnz(indptr.ctypes.data, indptr.itemsize, m.shape[0], rows.ctypes.data, len(rows)) == ref.nnz
    bad = np.array([5, 300], np.int64)
    crow = np.empty(3, np.int64)
    assert lib.mmvae_feed_gather_rows(m.indptr.ctypes.data, m.indices.ctypes.data, 4, m.data.ctypes.data, 300,
                                      bad.ctypes.data, 2, crow.ctypes.data, col.ctypes.data, val.ctypes.data, len(col), 1,
                                      None) == 1
    assert lib.mmvae_feed_gather_rows(m.indptr.ctypes.data, m.indices.ctypes.data, 4, m.data.ctypes.data, 300,
                                      rows.ctypes.data, len(rows), np.empty(98, np.int64).ctypes.data, col.ctypes.data,
                                      val.ctypes.data, 3, 1, None) == 2


def test_prefetcher_keeps_order_and_reraises(tmp_path):
    """mmvae_amd.data.Prefetcher: same batches in the same order as the wrapped iterable; producer errors surface in
    the consumer; an abandoned consumer stops the producer."""
    import threading
    import time

    from mmvae_amd.data import Prefetcher

    items = [(torch.full((2, 3), float(i)), {"i": i}, "human") for i in range(17)]
    got = list(Prefetcher(iter(items), depth=2))
    assert [int(g[0][0, 0]) for g in got] == list(range(17)) and all(a is b for a, b in zip(got, items))

    def failing():
        yield items[0]
        raise KeyError("chunk 3 is missing")

    it = iter(Prefetcher(failing(), depth=2))
    assert next(it) is items[0]
    with pytest.raises(KeyError, match="chunk 3"):
        next(it)

    produced = []

    def endless():
        i = 0
        while True:
            produced.append(i)
            yield (torch.zeros(1), None, "human")
            i += 1

    before = threading.active_count()
    it = iter(Prefetcher(endless(), depth=2))
    next(it)
    it.close()  # consumer walks away: the producer must not spin or block forever
    time.sleep(0.5)
    n = len(produced)
    time.sleep(0.3)
    assert len(produced) == n and threading.active_count() <= before + 1


@pytest.mark.gpu
def test_prefetched_gpu_batches_equal_the_direct_feed(tmp_path):
    """Prefetcher on the GPU: batches staged by the background thread on its own copy stream are, once the consumer's
    stream has joined them, exactly the batches the feed yields directly -- also while other work keeps the device busy."""
    m, meta = _dataset(200, 96, 5)
    D.write_chunks(str(tmp_path), "human", m, meta, chunk_rows=64, compressed=False)

    def feed():
        return D.SpeciesChunks(str(tmp_path), "human_train_counts_*.npz", "human_train_metadata_*.pkl", 16, "human",
                               seed=7, device="cuda")

    direct = [(x.to_dense().cpu(), md["row"].tolist()) for x, md, _ in feed()]
    busy = torch.randn(2048, 2048, device="cuda")
    got = []
    for x, md, eid in D.Prefetcher(feed(), depth=3, device="cuda"):
        busy = busy @ busy * 1e-3  # the consumer's stream has work queued while the producer copies
        assert x.is_cuda and x.layout == torch.sparse_csr and eid == "human"
        got.append((x.to_dense().cpu(), md["row"].tolist()))
    assert len(got) == len(direct) == 12  # chunks of 64, 64, 64, 8 rows: 4 + 4 + 4 + 0 batches of 16
    for (a, ra), (b, rb) in zip(got, direct):
        assert ra == rb and torch.equal(a, b)


def test_index_dtype_option_and_int32_gather(tmp_path):
    """index_dtype=torch.int64 yields the same batches with int64 indices; the int32 gather entry point of the native
    helper is bit-exact against scipy."""
    import ctypes as C

    m, meta = _dataset(70, 24, 9)
    D.write_chunks(str(tmp_path), "human", m, meta, chunk_rows=32)
    kw = dict(directory_path=str(tmp_path), npz_masks="human_train_counts_*.npz", metadata_masks="human_train_metadata_*.pkl",
              batch_size=8, name="human", seed=3)
    a = list(D.SpeciesChunks(**kw))
    b = list(D.SpeciesChunks(index_dtype=torch.int64, **kw))
    assert len(a) == len(b) == 8
    for (xa, ma, _), (xb, mb, _) in zip(a, b):
        assert xa.col_indices().dtype == torch.int32 and xb.col_indices().dtype == torch.int64
        assert torch.equal(xa.to_dense(), xb.to_dense()) and ma.equals(mb)
    with pytest.raises(ValueError):
        D.SpeciesChunks(index_dtype=torch.int16, **kw)
    lib = D.feed_lib()
    rows = np.random.default_rng(1).permutation(70)[:33].astype(np.int64)
    ref = m[rows]
    crow, col = np.empty(34, np.int32), np.full(ref.nnz + 3, -1, np.int32)
    val, got = np.zeros(ref.nnz + 3, np.float32), C.c_int64(0)
    mi = m.copy()
    rc = lib.mmvae_feed_gather_rows_i32(mi.indptr.ctypes.data, mi.indices.ctypes.data, mi.indptr.dtype.itemsize,
                                        mi.data.ctypes.data, 70, rows.ctypes.data, 33, crow.ctypes.data, col.ctypes.data,
                                        val.ctypes.data, len(col), 2, C.byref(got))
    assert rc == 0 and got.value == ref.nnz
    assert np.array_equal(crow, ref.indptr) and np.array_equal(col[:ref.nnz], ref.indices)
    assert np.array_equal(val[:ref.nnz], ref.data) and (col[ref.nnz:] == -1).all()

