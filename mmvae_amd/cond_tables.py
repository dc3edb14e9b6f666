"""Host-side index tables of one conditional layer application, shared by the module path
(`ConditionalLayer._forward_grouped`) and the captured engine (`engine_cond.CondProgram`).

From the per-cell block index of a batch (`local`, what `ConditionalLayer.forward` derives from the metadata column,
components.py:365-413) the kernels of `csrc/cond_layers.hip` need:

  rows        the cells sorted by block, cells of a block in batch order (stable) -- the forward reads a shared block
              once per 8 sorted cells, the weight gradient walks a block's cells in this order;
  chunks      every present block's cells cut into pieces of at most CHUNK cells: (dst, beg, end) with [beg, end) a
              range of `rows`; dst >= 0: the piece is the whole block `dst`, its gradient is written directly;
              dst <= -2: one of several pieces, its partial gradient goes to scratch slot -2 - dst;
  reductions  (block, first slot, pieces) of the blocks cut into several pieces: their partials are summed in piece
              order -- a fixed tree, bitwise reproducible, no atomics -- by the second kernel.

A block shared by many cells (the species block: all of them) is thereby spread over workgroups instead of being walked
by one.  `pad=True` pads the chunk / reduction lists to their fixed maxima for a batch of R cells (captured programs
launch fixed grids; dst = -1 / block = -1 mark the unused slots)."""
import numpy as np

CHUNK = 32  # MMVAE_COND_DW_CHUNK of include/mmvae_hip.h


def max_chunks(R: int) -> int:
    return R + R // CHUNK + 1


def max_reductions(R: int) -> int:
    return R // (CHUNK + 1) + 1


def partial_slots(R: int) -> int:
    """Scratch slots that the chunks of multi-piece blocks of an R-cell batch can need."""
    return R // CHUNK + max_reductions(R) + 1


def words(R: int) -> int:
    """int32 words of one padded table set: cond, rows, 3 chunk arrays, 3 reduction arrays."""
    return 2 * R + 3 * max_chunks(R) + 3 * max_reductions(R)


def layout(R: int) -> dict:
    """Word offsets of the arrays inside one padded table set."""
    nc, nr = max_chunks(R), max_reductions(R)
    off, out = 0, {}
    for name, n in (("cond", R), ("rows", R), ("chunk_dst", nc), ("chunk_beg", nc), ("chunk_end", nc),
                    ("red_cond", nr), ("red_slot", nr), ("red_n", nr)):
        out[name] = off
        off += n
    return out


def group_tables(local: np.ndarray, base: int = 0) -> dict:
    """Unpadded tables of one application; `base` shifts block indices into a table of several banks.  Also returns
    `present` (the local indices of the blocks that took part, ascending)."""
    local = np.asarray(local, dtype=np.int32)
    B = len(local)
    rows = np.argsort(local, kind="stable").astype(np.int32)
    present, start = np.unique(local[rows], return_index=True)
    counts = np.diff(np.append(start, B))
    nch = (counts + CHUNK - 1) // CHUNK
    owner = np.repeat(np.arange(len(present)), nch)
    first = np.cumsum(nch) - nch
    k = np.arange(int(nch.sum())) - first[owner]
    beg = start[owner] + k * CHUNK
    end = np.minimum(beg + CHUNK, (start + counts)[owner])
    multi = nch > 1
    taken = np.where(multi, nch, 0)
    slot_first = np.cumsum(taken) - taken
    dst = np.where(multi[owner], -2 - (slot_first[owner] + k), present[owner] + base)
    return dict(cond=(local + base).astype(np.int32), rows=rows, present=present,
                chunk_dst=dst.astype(np.int32), chunk_beg=beg.astype(np.int32), chunk_end=end.astype(np.int32),
                red_cond=(present[multi] + base).astype(np.int32), red_slot=slot_first[multi].astype(np.int32),
                red_n=nch[multi].astype(np.int32))


def fill_padded(seg: np.ndarray, t: dict, R: int) -> None:
    """Write the tables `t` of an R-cell batch into the int32 segment `seg` (length words(R)), padded."""
    lay = layout(R)
    nc, nr = max_chunks(R), max_reductions(R)
    n, m = len(t["chunk_dst"]), len(t["red_cond"])
    if len(t["cond"]) != R or n > nc or m > nr:
        raise ValueError("conditional tables do not fit the padded layout")
    seg[lay["cond"]:lay["cond"] + R] = t["cond"]
    seg[lay["rows"]:lay["rows"] + R] = t["rows"]
    for name, fill, k, cap in (("chunk_dst", -1, n, nc), ("chunk_beg", 0, n, nc), ("chunk_end", 0, n, nc),
                               ("red_cond", -1, m, nr), ("red_slot", 0, m, nr), ("red_n", 0, m, nr)):
        o = lay[name]
        seg[o:o + k] = t[name]
        seg[o + k:o + cap] = fill


def fill_all(seg: np.ndarray, stride: int, local: np.ndarray, base: np.ndarray, R: int) -> list:
    """The padded tables of SEVERAL applications in one call: local [n, R] int32 (block index per cell and application),
    base [n]; application j fills seg[j * stride : j * stride + words(R)].  Returns the `present` array of each.  Native
    (libmmvae_feed.so, mmvae_feed_cond_tables: ~3 us per application where group_tables + fill_padded take ~50 us of
    numpy calls -- the captured conditional programs are host-bound, tools/time_conditional_host.py)."""
    from .data import feed_lib

    n = local.shape[0]
    local = np.ascontiguousarray(local, dtype=np.int32)
    base = np.ascontiguousarray(base, dtype=np.int32)
    if local.shape != (n, R) or seg.dtype != np.int32 or not seg.flags["C_CONTIGUOUS"] or seg.size < n * stride:
        raise ValueError("conditional tables: bad shapes")
    present = np.empty((n, R), dtype=np.int32)
    n_present = np.empty(n, dtype=np.int32)
    rc = feed_lib().mmvae_feed_cond_tables(n, R, local.ctypes.data, base.ctypes.data, seg.ctypes.data, stride,
                                           present.ctypes.data, n_present.ctypes.data)
    if rc != 0:
        raise ValueError(f"mmvae_feed_cond_tables failed with code {rc} (negative block index / tables do not fit)")
    return [present[j, :n_present[j]] for j in range(n)]


_LOOKUP = None


def lookup_i32(table: dict, values: list, out: np.ndarray) -> int:
    """out[i] = table[values[i]] (int32) through csrc/pylookup.c (CPython API under the interpreter lock: ~10 us per 512
    values where the interpreter takes 40-60).  Returns len(values) when every value was a key, else the index of the
    first one that was not (the caller extends `table` and calls again)."""
    global _LOOKUP
    if _LOOKUP is None:
        import ctypes as C
        import os
        import subprocess

        here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
        path = os.path.join(here, "_mmvae_pylookup.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", here, "_mmvae_pylookup.so"], check=True, capture_output=True)
        fn = C.PyDLL(path).mmvae_py_lookup_i32
        fn.restype = C.c_ssize_t
        fn.argtypes = [C.py_object, C.py_object, C.c_void_p, C.c_ssize_t]
        _LOOKUP = fn
    if out.dtype != np.int32 or not out.flags["C_CONTIGUOUS"] or out.size < len(values):
        raise ValueError("lookup_i32: out must be a contiguous int32 array of at least len(values)")
    return int(_LOOKUP(table, values, out.ctypes.data, len(values)))
