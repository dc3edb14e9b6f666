"""Data feed of the training step: chunked npz-CSR counts + pickled metadata -> `(x, metadata, expert_id)`
batches, the tuples `CMMVAEModel.training_step` consumes.

Mirrors the reference's local datapipes without torchdata (not installed here; data/local/cellxgene_datapipe.py):
  * chunk discovery  -- `FileLister(root, masks, non_deterministic=False)` zipped pairwise (:295-318): files of one
    directory matched by fnmatch masks, sorted, counts chunk i paired with metadata chunk i;
  * chunk loading    -- `scipy.sparse.load_npz` + `pickle.load` of a DataFrame (:60-86);
  * shuffling        -- chunk order and, per chunk, one row permutation applied to matrix and DataFrame alike (:110-122);
  * batching         -- consecutive `batch_size` rows as `torch.sparse_csr_tensor(indptr, indices, data)` (int32 indices by default,
    fp32 values) or, with `return_dense`, its dense form; partial batches dropped unless `allow_partials` (:169-193);
  * tagging          -- `(tensor, metadata, species_name)` (data/local/cellxgene_manager.py:76-88).
Differences, all on purpose: the shuffles draw from a seeded `numpy.random.Generator` (the reference uses the global
numpy state) so that an epoch is reproducible and data-parallel ranks can be given disjoint, rank-synchronous
streams; the next chunk is loaded and permuted by a background thread while the current one is consumed; batches can be
staged to the device as CSR components from pinned memory (12 B per stored element instead of 4 B per gene), where
`mmvae_csr_to_dense_f32` densifies them straight into the step's input buffer.
"""
from __future__ import annotations

import fnmatch
import atexit
import os
import pickle
import queue
import threading
from typing import Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd
import torch


def _wrap(v) -> List[str]:
    return [v] if isinstance(v, str) else list(v)


def list_chunks(directory_path: str, npz_masks: Union[str, Sequence[str]],
                metadata_masks: Union[str, Sequence[str]]) -> List[Tuple[str, str]]:
    """Sorted (counts.npz, metadata.pkl) pairs of one directory (non-recursive), zipped pairwise like the reference's
    two FileListers; raises when nothing matches (cellxgene_datapipe.py:312-318)."""
    names = sorted(os.listdir(directory_path))

    def pick(masks):
        return [os.path.join(os.path.abspath(directory_path), n) for n in names
                if os.path.isfile(os.path.join(directory_path, n)) and any(fnmatch.fnmatch(n, m) for m in _wrap(masks))]

    pairs = list(zip(pick(npz_masks), pick(metadata_masks)))
    if not pairs:
        raise RuntimeError("No files found for masks from file lister")
    return pairs


def _mapped_npz_csr(npz_path: str):
    """Memory-mapped CSR matrix of an UNCOMPRESSED scipy npz (`save_npz(..., compressed=False)`): the .npy members of
    a stored zip are plain byte ranges of the file, so indptr / indices / data become page-cache-backed views instead
    of being read and copied through zipfile (0.29 s -> ~1 ms per 2048-cell chunk).  Returns None when the archive is
    compressed or is not a CSR matrix: the caller falls back to scipy.sparse.load_npz."""
    import struct
    import zipfile

    import scipy.sparse as sp

    arrays = {}
    with zipfile.ZipFile(npz_path) as zf, open(npz_path, "rb") as f:
        for info in zf.infolist():
            if info.compress_type != zipfile.ZIP_STORED:
                return None
            f.seek(info.header_offset)
            hdr = f.read(30)
            n_name, n_extra = struct.unpack("<HH", hdr[26:30])
            start = info.header_offset + 30 + n_name + n_extra
            f.seek(start)
            version = np.lib.format.read_magic(f)
            shape, fortran, dtype = (np.lib.format.read_array_header_1_0(f) if version == (1, 0)
                                     else np.lib.format.read_array_header_2_0(f))
            name = info.filename[:-4] if info.filename.endswith(".npy") else info.filename
            if dtype.hasobject or fortran:
                return None
            if int(np.prod(shape)) * dtype.itemsize <= 4096:  # format / shape scalars: just read them
                arrays[name] = np.fromfile(f, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
            else:
                arrays[name] = np.memmap(npz_path, dtype=dtype, mode="r", offset=f.tell(), shape=shape)
    fmt = arrays.get("format")
    fmt = fmt.item() if fmt is not None else b""
    if (fmt.decode() if isinstance(fmt, bytes) else str(fmt)) != "csr":
        return None
    return sp.csr_matrix((arrays["data"], arrays["indices"], arrays["indptr"]), shape=tuple(arrays["shape"]), copy=False)


def load_chunk(npz_path: str, metadata_path: str):
    """One chunk: scipy CSR matrix + its index-matched metadata DataFrame (cellxgene_datapipe.py:60-86)."""
    import scipy.sparse as sp

    matrix = _mapped_npz_csr(npz_path)
    if matrix is None:
        with open(npz_path, "rb") as f:
            matrix = sp.load_npz(f).tocsr()
    with open(metadata_path, "rb") as f:
        metadata = pickle.load(f)
    if matrix.shape[0] != len(metadata):
        raise ValueError(f"{npz_path}: {matrix.shape[0]} rows but {len(metadata)} metadata rows in {metadata_path}")
    return matrix, metadata


_FEED = None


def feed_lib():
    """libmmvae_feed.so (include/mmvae_feed.h): the native row gather.  Built by `make -C mmvae_amd/csrc` together with
    the HIP library; built on demand here (g++, one file) when it is missing -- it is host code, not the HIP path."""
    global _FEED
    if _FEED is None:
        import ctypes as C
        import subprocess

        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
        path = os.path.join(here, "libmmvae_feed.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", here, "libmmvae_feed.so"], check=True, capture_output=True)
        lib = C.CDLL(path)
        lib.mmvae_feed_rows_nnz.restype = C.c_int64
        lib.mmvae_feed_rows_nnz.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int64]
        lib.mmvae_feed_gather_rows.restype = C.c_int
        lib.mmvae_feed_gather_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                               C.c_void_p]
        lib.mmvae_feed_gather_rows_i32.restype = C.c_int
        lib.mmvae_feed_gather_rows_i32.argtypes = lib.mmvae_feed_gather_rows.argtypes
        lib.mmvae_feed_cond_tables.restype = C.c_int
        lib.mmvae_feed_cond_tables.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p]
        if lib.mmvae_feed_abi_version() != 3:
            raise RuntimeError(f"{path} is a stale build (ABI {lib.mmvae_feed_abi_version()}, expected 3): "
                               "make -C mmvae_amd/csrc")
        _FEED = lib
    return _FEED


class _Slot:
    """Staging arrays of one batch in flight: row pointers / column indices (int32 or int64) and fp32 values in the layout
    torch.sparse_csr uses.  Page-locked when the batches go to a GPU; reused once the asynchronous H2D copies that
    read it have completed (`event`)."""

    def __init__(self, n_rows: int, pinned: bool, index_dtype=torch.int32):
        self.pinned = pinned
        self.index_dtype = index_dtype
        self.n_rows = n_rows
        # page-locked int32 batches live in ONE buffer -- row pointers | column indices | values, each part starting on a
        # 16-byte boundary -- and go to the device with ONE copy (three `.to()` calls were 0.15 ms of the producer's share
        # of an interpreter-bound loop: profiles/HISTORY.md r5)
        self.packed = pinned and index_dtype == torch.int32
        self.buf = None
        self.crow = self._alloc(n_rows + 1, index_dtype) if not self.packed else None
        self.col = self._alloc(16, index_dtype) if not self.packed else None
        self.val = self._alloc(16, torch.float32) if not self.packed else None
        self.event = None
        if self.packed:
            self.reserve(16)

    def _alloc(self, n: int, dtype) -> torch.Tensor:
        t = torch.empty(n, dtype=dtype)
        return t.pin_memory() if self.pinned else t

    @staticmethod
    def _pad4(n: int) -> int:
        return (n + 3) // 4 * 4

    def layout(self, nnz: int):
        """(offset of the column indices, offset of the values, words in use) inside the packed buffer."""
        o_col = self._pad4(self.n_rows + 1)
        o_val = o_col + self._pad4(nnz)
        return o_col, o_val, o_val + nnz

    def reserve(self, nnz: int) -> None:
        if self.packed:
            o_col, o_val, words = self.layout(nnz)
            if self.buf is None or self.buf.numel() < words:
                cap = int(nnz * 1.25) + 16  # reused: leave headroom
                self.buf = self._alloc(self.layout(cap)[2], torch.int32)
            self.crow = self.buf[:self.n_rows + 1]
            self.col = self.buf[o_col:o_col + nnz]
            self.val = self.buf[o_val:o_val + nnz].view(torch.float32)
            return
        if self.col.numel() < nnz:
            cap = (int(nnz * 1.25) + 16) if self.pinned else max(nnz, 1)  # pinned buffers are reused: leave headroom
            self.col, self.val = self._alloc(cap, self.index_dtype), self._alloc(cap, torch.float32)


class _Workers:
    """A few persistent threads behind a C-level queue: `submit(fn, *args)` returns an object whose `result()` waits for
    and returns fn's value (or re-raises its exception).  What the feed needs of ThreadPoolExecutor at a fifth of its cost
    per job (the executor's submit was 0.1 ms per batch on the GPU box, under the interpreter lock the training thread also
    needs)."""

    class _Job:
        __slots__ = ("done", "value", "error")

        def __init__(self):
            self.done, self.value, self.error = threading.Event(), None, None

        def result(self):
            self.done.wait()
            if self.error is not None:
                raise self.error
            return self.value

    def __init__(self, n: int):
        self.jobs: "queue.SimpleQueue" = queue.SimpleQueue()
        self.threads = [threading.Thread(target=self._work, daemon=True) for _ in range(max(1, n))]
        for t in self.threads:
            t.start()

    def _work(self):
        while True:
            item = self.jobs.get()
            if item is None:
                return
            job, fn, args = item
            try:
                job.value = fn(*args)
            except BaseException as e:  # noqa: BLE001 -- re-raised by result()
                job.error = e
            job.done.set()

    def submit(self, fn, *args):
        job = self._Job()
        self.jobs.put((job, fn, args))
        return job

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for _ in self.threads:
            self.jobs.put(None)
        for t in self.threads:
            t.join(timeout=10.0)
        return False


class SpeciesChunks:
    """Iterable over the batches of one modality's chunk files: `(x, metadata, name)`.

    A batch is gathered row by row out of the (memory-mapped) chunk by the native helper -- the row permutation is an
    index list, the chunk itself is never permuted or sliced in Python -- into a ring of staging slots, `workers`
    batches ahead of the consumer and off the interpreter lock.

    `rank` / `world`: data-parallel sharding -- every rank walks the same (seeded) chunk order and row permutations;
    the epoch's kept batches (across chunks, dropped partial batches not counted) are dealt out in rounds of `world`,
    rank r takes the r-th batch of every COMPLETE round and the incomplete last round is dropped, so every rank yields
    exactly floor(n_batches / world) batches of the same modality schedule -- a rank that ran short would leave the
    others blocked in the gradient all-reduce (the reference has no distributed sampler)."""

    def __init__(self, directory_path: str, npz_masks, metadata_masks, batch_size: int, name: str,
                 allow_partials: bool = False, shuffle: bool = True, return_dense: bool = False, seed: int = 0,
                 device: Optional[Union[str, torch.device]] = None, prefetch: bool = True, rank: int = 0, world: int = 1,
                 workers: int = 3, gather_threads: int = 2, index_dtype: torch.dtype = torch.int32):
        self.chunks = list_chunks(directory_path, npz_masks, metadata_masks)
        self.batch_size = int(batch_size)
        self.name = name
        self.allow_partials = allow_partials
        self.shuffle = shuffle
        self.return_dense = return_dense
        self.seed = seed
        self.device = torch.device(device) if device is not None else None
        self.prefetch = prefetch
        self.rank, self.world = rank, world
        self.workers = max(1, int(workers))
        self.gather_threads = max(1, int(gather_threads))
        # int32: what the reference's batches carry (torch.sparse_csr_tensor keeps the int32 indptr / indices of the scipy
        # slice, cellxgene_datapipe.py:178-183) and a third less data per batch on the wire; int64 on request
        if index_dtype not in (torch.int32, torch.int64):
            raise ValueError("index_dtype must be torch.int32 or torch.int64")
        self.index_dtype = index_dtype
        self.epoch = 0
        self._slots: List[_Slot] = []

    # ---- chunks: (scipy CSR over the mapped file, metadata, row order)
    def _prepared_chunks(self, rng: np.random.Generator) -> Iterator:
        order = list(range(len(self.chunks)))
        if self.shuffle:
            order = [int(i) for i in rng.permutation(len(order))]
        for ci in order:
            matrix, metadata = load_chunk(*self.chunks[ci])
            rows = rng.permutation(matrix.shape[0]) if self.shuffle else np.arange(matrix.shape[0])
            yield matrix, metadata, rows.astype(np.int64)

    def _background(self, gen: Iterator) -> Iterator:
        """Run `gen` in a thread, one item ahead (loading a chunk overlaps the training on the last one).  The thread
        stops when the consumer is closed or collected."""
        q: "queue.Queue" = queue.Queue(maxsize=1)
        done, stop = object(), threading.Event()

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                for item in gen:
                    if not put(item):
                        return
                put(done)
            except BaseException as e:  # noqa: BLE001 -- re-raised in the consumer
                put(e)

        thread = threading.Thread(target=work, daemon=True)
        thread.start()
        _LIVE_PRODUCERS.append((stop, thread))
        try:
            while True:
                item = q.get()
                if item is done:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            if _LIVE_PRODUCERS is not None and (stop, thread) in _LIVE_PRODUCERS:
                _LIVE_PRODUCERS.remove((stop, thread))

    # ---- one batch: native gather into a staging slot (worker thread), tensors + H2D (consumer thread)
    def _gather(self, slot: _Slot, matrix, rows: np.ndarray):
        lib = feed_lib()
        ib = matrix.indptr.dtype.itemsize
        if matrix.indices.dtype.itemsize != ib or ib not in (4, 8) or matrix.data.dtype != np.float32:
            matrix = matrix.astype(np.float32) if matrix.data.dtype != np.float32 else matrix
            matrix.indices = matrix.indices.astype(matrix.indptr.dtype)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        nnz = lib.mmvae_feed_rows_nnz(matrix.indptr.ctypes.data, ib, matrix.shape[0], rows.ctypes.data, len(rows))
        if nnz < 0:
            raise ValueError("row index outside the chunk")
        slot.reserve(nnz)
        import ctypes as C

        got = C.c_int64(0)
        if slot.index_dtype == torch.int32 and matrix.shape[1] > 0x7fffffff:
            raise ValueError("a chunk with more than 2^31 - 1 columns needs index_dtype=torch.int64")
        fn = lib.mmvae_feed_gather_rows_i32 if slot.index_dtype == torch.int32 else lib.mmvae_feed_gather_rows
        rc = fn(matrix.indptr.ctypes.data, matrix.indices.ctypes.data if nnz else None, ib,
                matrix.data.ctypes.data if nnz else None, matrix.shape[0], rows.ctypes.data, len(rows), slot.crow.data_ptr(),
                slot.col.data_ptr(), slot.val.data_ptr(), max(slot.col.numel(), nnz), self.gather_threads, C.byref(got))
        if rc != 0 or got.value != nnz:
            raise RuntimeError(f"mmvae_feed_gather_rows failed with code {rc}")
        return slot, int(nnz), len(rows)

    def _tensor(self, slot: _Slot, nnz: int, n_rows: int, n_cols: int) -> torch.Tensor:
        """Staging slot -> torch.sparse_csr (int32 / int64 indices, fp32 values), on the device when one
        was given (three non-blocking copies out of page-locked memory; the slot is reusable once they are done)."""
        crow, col, val = slot.crow[:n_rows + 1], slot.col[:nnz], slot.val[:nnz]
        if self.device is not None and self.device.type == "cuda" and slot.packed:
            o_col, o_val, words = slot.layout(nnz)
            dev = slot.buf[:words].to(self.device, non_blocking=True)  # one copy; the three arrays are views of it
            crow, col, val = dev[:n_rows + 1], dev[o_col:o_col + nnz], dev[o_val:o_val + nnz].view(torch.float32)
            slot.event = torch.cuda.Event()
            slot.event.record()
        elif self.device is not None and self.device.type == "cuda":
            crow, col, val = (t.to(self.device, non_blocking=True) for t in (crow, col, val))
            slot.event = torch.cuda.Event()
            slot.event.record()
        else:  # host batches own their arrays: the slot was allocated for this batch alone (see __iter__)
            slot.crow, slot.col, slot.val = slot.crow.new_empty(0), slot.col.new_empty(0), slot.val.new_empty(0)
        t = torch.sparse_csr_tensor(crow, col, val, size=(n_rows, n_cols))
        if self.return_dense:
            from . import backend

            t = backend.to_dense(t) if t.values().is_cuda else t.to_dense()
        return t

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, pd.DataFrame, str]]:
        from collections import deque

        rng = np.random.default_rng([self.seed, self.epoch])
        self.epoch += 1
        chunks = self._prepared_chunks(rng)
        if self.prefetch:
            chunks = self._background(chunks)
        pinned = self.device is not None and self.device.type == "cuda"
        depth = self.workers + 1
        if (len(self._slots) != depth or (self._slots and self._slots[0].pinned != pinned)
                or (self._slots and self._slots[0].index_dtype != self.index_dtype)):
            self._slots = [_Slot(self.batch_size, pinned, self.index_dtype) for _ in range(depth)]
        free = deque(self._slots)
        inflight: deque = deque()

        def jobs():
            dealt = []  # the current round of `world` consecutive kept batches of the epoch
            for matrix, metadata, order in chunks:
                n = matrix.shape[0]
                for i in range(0, n, self.batch_size):
                    if i + self.batch_size > n and not self.allow_partials:
                        continue
                    dealt.append((matrix, metadata, order[i:i + self.batch_size]))
                    if len(dealt) == self.world:
                        yield dealt[self.rank]
                        dealt = []

        with _Workers(self.workers) as pool:
            def finish():
                fut, matrix, metadata, rows = inflight.popleft()
                slot, nnz, n_rows = fut.result()
                x = self._tensor(slot, nnz, n_rows, matrix.shape[1])
                free.append(slot)
                # the batch's metadata rows under a fresh 0..n-1 index (cellxgene_datapipe.py:110-122 slices the permuted
                # frame): take + a new RangeIndex is half the cost of iloc + reset_index, same frame (tested)
                md = metadata.take(rows)
                md.index = pd.RangeIndex(len(md))
                return x, md, self.name

            for matrix, metadata, rows in jobs():
                if not free:
                    yield finish()
                slot = free.popleft()
                if not pinned:  # host batches keep the arrays they were gathered into
                    slot.crow = torch.empty(self.batch_size + 1, dtype=self.index_dtype)
                if slot.event is not None:  # the H2D copies that read this slot (depth batches ago) must be done
                    slot.event.synchronize()
                    slot.event = None
                inflight.append((pool.submit(self._gather, slot, matrix, rows), matrix, metadata, rows))
            while inflight:
                yield finish()

    def __len__(self) -> int:
        raise TypeError("SpeciesChunks streams chunk files: its length is not known without reading them")


_LIVE_PRODUCERS: list = []  # (stop event, thread) of running Prefetcher producers, stopped and joined at interpreter exit


def _stop_producers() -> None:
    """A daemon thread that is still inside torch when the interpreter finalises is torn down by a forced unwind through
    C++ frames (`terminate called without an active exception`): stop and join the producers first."""
    for stop, thread in list(_LIVE_PRODUCERS):
        stop.set()
    for stop, thread in list(_LIVE_PRODUCERS):
        thread.join(timeout=10.0)
    _LIVE_PRODUCERS.clear()


atexit.register(_stop_producers)
if hasattr(threading, "_register_atexit"):  # also before the interpreter starts joining threads
    threading._register_atexit(_stop_producers)


class Prefetcher:
    """Runs a batch iterable -- `SpeciesChunks`, `trainer.MultiModalBatches` over several of them -- in a background
    thread, `depth` batches ahead of the training loop, so that the consumer thread only launches steps: waiting for a
    gather, slicing the metadata and enqueueing the host-to-device copies no longer sit between two step launches.
    On a GPU the producer issues its copies on a stream of its own; the consumer's stream waits for the event recorded
    behind each batch, and the batch's device arrays are marked as used by the consumer's stream (allocator safety).
    Order and content of the batches are unchanged; an exception in the producer is re-raised in the consumer."""

    def __init__(self, batches, depth: int = 3, device: Optional[Union[str, torch.device]] = None):
        self.batches = batches
        self.depth = max(1, int(depth))
        self.device = torch.device(device) if device is not None else None

    def __iter__(self):
        from contextlib import nullcontext

        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        done, stop = object(), threading.Event()
        on_gpu = self.device is not None and self.device.type == "cuda"
        stream = torch.cuda.Stream(device=self.device) if on_gpu else None

        def put(item) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def work():
            try:
                with (torch.cuda.stream(stream) if on_gpu else nullcontext()):
                    source = iter(self.batches)
                    while True:
                        try:
                            item = next(source)
                        except StopIteration:
                            break
                        ev = None
                        if on_gpu:
                            ev = torch.cuda.Event()
                            ev.record(stream)
                        if not put((item, ev)):
                            return
                put(done)
            except BaseException as e:  # noqa: BLE001 -- re-raised in the consumer
                put(e)

        thread = threading.Thread(target=work, daemon=True)
        thread.start()
        entry = (stop, thread)
        _LIVE_PRODUCERS.append(entry)
        try:
            while True:
                got = q.get()
                if got is done:
                    return
                if isinstance(got, BaseException):
                    raise got
                item, ev = got
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    x = item[0] if isinstance(item, tuple) else item
                    if torch.is_tensor(x) and x.is_cuda:
                        parts = ((x.crow_indices(), x.col_indices(), x.values()) if x.layout == torch.sparse_csr else (x,))
                        for part in parts:
                            part.record_stream(cur)
                yield item
        finally:
            stop.set()
            if _LIVE_PRODUCERS is not None and entry in _LIVE_PRODUCERS:  # (None: the interpreter is shutting down)
                _LIVE_PRODUCERS.remove(entry)


def write_chunks(directory: str, name: str, matrix, metadata: pd.DataFrame, chunk_rows: int, split: str = "train",
                 compressed: bool = True) -> List[str]:
    """Write `{name}_{split}_counts_{i}.npz` / `{name}_{split}_metadata_{i}.pkl` chunk pairs (the layout the
    reference's preprocessing produces, scripts/data-preprocessing/: `scipy.sparse.save_npz` + pickled DataFrame).
    Used by the tests and by bench.py's npz leg to put synthetic data on disk."""
    import scipy.sparse as sp

    os.makedirs(directory, exist_ok=True)
    matrix = sp.csr_matrix(matrix)
    out = []
    for k, i in enumerate(range(0, matrix.shape[0], chunk_rows), start=1):
        npz = os.path.join(directory, f"{name}_{split}_counts_{k}.npz")
        pkl = os.path.join(directory, f"{name}_{split}_metadata_{k}.pkl")
        sp.save_npz(npz, matrix[i:i + chunk_rows], compressed=compressed)
        metadata.iloc[i:i + chunk_rows].reset_index(drop=True).to_pickle(pkl)
        out += [npz, pkl]
    return out
