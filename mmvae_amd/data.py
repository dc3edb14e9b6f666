"""Data feed of the training step: chunked npz-CSR counts + pickled metadata -> `(x, metadata, expert_id)`
batches, the tuples `CMMVAEModel.training_step` consumes.

Mirrors the reference's local datapipes without torchdata (not installed here; data/local/cellxgene_datapipe.py):
  * chunk discovery  -- `FileLister(root, masks, non_deterministic=False)` zipped pairwise (:295-318): files of one
    directory matched by fnmatch masks, sorted, counts chunk i paired with metadata chunk i;
  * chunk loading    -- `scipy.sparse.load_npz` + `pickle.load` of a DataFrame (:60-86);
  * shuffling        -- chunk order and, per chunk, one row permutation applied to matrix and DataFrame alike (:110-122);
  * batching         -- consecutive `batch_size` rows as `torch.sparse_csr_tensor(indptr, indices, data)` (int64 indices,
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


class SpeciesChunks:
    """Iterable over the batches of one modality's chunk files: `(x, metadata, name)`.

    `rank` / `world`: data-parallel sharding -- every rank walks the same (seeded) chunk order and row permutations
    and keeps batches rank, rank + world, ... of each chunk, so that all ranks see the same number of batches of the
    same modality schedule (the reference has no distributed sampler)."""

    def __init__(self, directory_path: str, npz_masks, metadata_masks, batch_size: int, name: str,
                 allow_partials: bool = False, shuffle: bool = True, return_dense: bool = False, seed: int = 0,
                 device: Optional[Union[str, torch.device]] = None, prefetch: bool = True, rank: int = 0, world: int = 1):
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
        self.epoch = 0
        self._stage: dict = {}
        self._stage_flip = 0

    # ---- one chunk -> permuted scipy matrix + DataFrame
    def _prepared_chunks(self, rng: np.random.Generator) -> Iterator:
        order = list(range(len(self.chunks)))
        if self.shuffle:
            order = [int(i) for i in rng.permutation(len(order))]
        for ci in order:
            matrix, metadata = load_chunk(*self.chunks[ci])
            if self.shuffle:
                perm = rng.permutation(matrix.shape[0])
                metadata = metadata.iloc[perm].reset_index(drop=True)
                matrix = matrix[perm]
            yield matrix, metadata

    def _background(self, gen: Iterator) -> Iterator:
        """Run `gen` in a thread, one item ahead (loading + permuting a chunk overlaps the training on the last one)."""
        q: "queue.Queue" = queue.Queue(maxsize=1)
        done = object()

        def work():
            try:
                for item in gen:
                    q.put(item)
                q.put(done)
            except BaseException as e:  # noqa: BLE001 -- re-raised in the consumer
                q.put(e)

        threading.Thread(target=work, daemon=True).start()
        while True:
            item = q.get()
            if item is done:
                return
            if isinstance(item, BaseException):
                raise item
            yield item

    def _pinned(self, key: str, like: np.ndarray, dtype) -> torch.Tensor:
        """A reusable page-locked staging tensor per CSR component (grown on demand): allocating pinned memory per batch
        costs more than the copy.  Two alternating sets, so that the asynchronous H2D copy of batch i is not overwritten
        while batch i + 1 is being staged."""
        slot = self._stage_flip
        buf = self._stage.get((key, slot))
        if buf is None or buf.numel() < like.size:
            buf = torch.empty(max(int(like.size * 1.25), 16), dtype=dtype).pin_memory()
            self._stage[(key, slot)] = buf
        out = buf[:like.size]
        out.numpy()[...] = like  # converts the dtype on the way (int32 -> int64 indices)
        return out

    def _tensor(self, matrix, i: int) -> torch.Tensor:
        """Rows i .. i + batch_size of a scipy CSR chunk -> torch.sparse_csr (int64 indices as torch stores them, fp32
        values), optionally on the device.  Consecutive rows of a CSR matrix are one contiguous run of its index and
        value arrays: the batch is three views (plus the dtype conversion), not a scipy row slice."""
        j = min(i + self.batch_size, matrix.shape[0])
        lo, hi = int(matrix.indptr[i]), int(matrix.indptr[j])
        indptr, indices, data = matrix.indptr[i:j + 1] - lo, matrix.indices[lo:hi], matrix.data[lo:hi]
        shape = (j - i, matrix.shape[1])
        if self.device is not None and self.device.type == "cuda":
            self._stage_flip ^= 1
            crow = self._pinned("crow", indptr, torch.int64).to(self.device, non_blocking=True)
            col = self._pinned("col", indices, torch.int64).to(self.device, non_blocking=True)
            val = self._pinned("val", data, torch.float32).to(self.device, non_blocking=True)
        else:
            crow = torch.from_numpy(indptr.astype(np.int64))
            col = torch.from_numpy(indices.astype(np.int64))
            val = torch.from_numpy(data.astype(np.float32))
        t = torch.sparse_csr_tensor(crow, col, val, size=shape)
        if self.return_dense:
            from . import backend

            t = backend.to_dense(t) if t.values().is_cuda else t.to_dense()
        return t

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, pd.DataFrame, str]]:
        rng = np.random.default_rng([self.seed, self.epoch])
        self.epoch += 1
        chunks = self._prepared_chunks(rng)
        if self.prefetch:
            chunks = self._background(chunks)
        for matrix, metadata in chunks:
            n = matrix.shape[0]
            for b, i in enumerate(range(0, n, self.batch_size)):
                if i + self.batch_size > n and not self.allow_partials:
                    continue
                if b % self.world != self.rank:
                    continue
                yield self._tensor(matrix, i), metadata.iloc[i:i + self.batch_size].reset_index(drop=True), self.name

    def __len__(self) -> int:
        raise TypeError("SpeciesChunks streams chunk files: its length is not known without reading them")


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
