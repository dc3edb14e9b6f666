"""`predictions.h5` writer / reader: the step after `predict_step`.

File layout of the reference (`callbacks/prediction_writer.py:14-70`, written there through h5py):

    /<key>/data                 float32 [N, D]   extendable along axis 0, chunked   (RK.PREDICT_SAMPLES)
    /<key>/metadata/<column>    one 1-D extendable, chunked dataset per DataFrame column (RK.METADATA):
                                strings as variable-length UTF-8, integers int64, floats float64, bools as the
                                int8 enum {FALSE=0, TRUE=1}
    /<key>/umap_embeddings      optional, written by the UMAP runner (RK.UMAP_EMBEDDINGS); read back if present

h5py is not part of this image; the HDF5 C library is, so this module binds `libhdf5` directly with ctypes
(`MMVAE_HDF5_LIB` overrides the search).  There is no fallback format: without the library the writer raises.
"""
import ctypes
import ctypes.util
import os
import warnings
from typing import Any, Optional, Sequence

import numpy as np
import pandas as pd

from .constants import REGISTRY_KEYS as RK

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64
herr_t = ctypes.c_int
htri_t = ctypes.c_int
_P = ctypes.POINTER

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_EXCL = 0, 1, 4
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SELECT_SET = 0
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
H5T_VARIABLE = ctypes.c_size_t(-1).value
H5T_CSET_UTF8 = 1
H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_ENUM = 0, 1, 3, 8
H5_INDEX_NAME, H5_ITER_INC = 0, 0
_CHUNK_BYTES = 256 * 1024  # target chunk size of the extendable datasets

_SIGNATURES = {
    "H5open": (herr_t, []),
    "H5Eset_auto2": (herr_t, [hid_t, ctypes.c_void_p, ctypes.c_void_p]),
    "H5Fcreate": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]),
    "H5Fopen": (hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t]),
    "H5Fclose": (herr_t, [hid_t]),
    "H5Gcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]),
    "H5Gopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
    "H5Gclose": (herr_t, [hid_t]),
    "H5Gget_info": (herr_t, [hid_t, ctypes.c_void_p]),
    "H5Lexists": (htri_t, [hid_t, ctypes.c_char_p, hid_t]),
    "H5Lget_name_by_idx": (ctypes.c_ssize_t, [hid_t, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, hsize_t,
                                              ctypes.c_char_p, ctypes.c_size_t, hid_t]),
    "H5Screate_simple": (hid_t, [ctypes.c_int, _P(hsize_t), _P(hsize_t)]),
    "H5Sclose": (herr_t, [hid_t]),
    "H5Sget_simple_extent_ndims": (ctypes.c_int, [hid_t]),
    "H5Sget_simple_extent_dims": (ctypes.c_int, [hid_t, _P(hsize_t), _P(hsize_t)]),
    "H5Sselect_hyperslab": (herr_t, [hid_t, ctypes.c_int, _P(hsize_t), _P(hsize_t), _P(hsize_t), _P(hsize_t)]),
    "H5Pcreate": (hid_t, [hid_t]),
    "H5Pset_chunk": (herr_t, [hid_t, ctypes.c_int, _P(hsize_t)]),
    "H5Pclose": (herr_t, [hid_t]),
    "H5Dcreate2": (hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
    "H5Dopen2": (hid_t, [hid_t, ctypes.c_char_p, hid_t]),
    "H5Dclose": (herr_t, [hid_t]),
    "H5Dget_space": (hid_t, [hid_t]),
    "H5Dget_type": (hid_t, [hid_t]),
    "H5Dset_extent": (herr_t, [hid_t, _P(hsize_t)]),
    "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
    "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]),
    "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, ctypes.c_void_p]),
    "H5Tcopy": (hid_t, [hid_t]),
    "H5Tclose": (herr_t, [hid_t]),
    "H5Tset_size": (herr_t, [hid_t, ctypes.c_size_t]),
    "H5Tset_cset": (herr_t, [hid_t, ctypes.c_int]),
    "H5Tget_class": (ctypes.c_int, [hid_t]),
    "H5Tget_size": (ctypes.c_size_t, [hid_t]),
    "H5Tis_variable_str": (htri_t, [hid_t]),
    "H5Tenum_create": (hid_t, [hid_t]),
    "H5Tenum_insert": (herr_t, [hid_t, ctypes.c_char_p, ctypes.c_void_p]),
}

_lib = None


class HDF5Error(RuntimeError):
    pass


def hdf5_lib():
    """The HDF5 C library, loaded once.  Raises when it cannot be found: there is no other prediction format."""
    global _lib
    if _lib is not None:
        return _lib
    candidates = [os.environ.get("MMVAE_HDF5_LIB"), ctypes.util.find_library("hdf5"),
                  "/opt/conda/lib/libhdf5.so", "libhdf5.so"]
    errors = []
    for name in candidates:
        if not name:
            continue
        try:
            lib = ctypes.CDLL(name)
        except OSError as e:
            errors.append(f"{name}: {e}")
            continue
        for fn, (res, args) in _SIGNATURES.items():
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        if lib.H5open() < 0:
            raise HDF5Error("H5open failed")
        lib.H5Eset_auto2(0, None, None)  # errors surface as Python exceptions, not as stderr stack dumps
        _lib = lib
        return lib
    raise HDF5Error("libhdf5 not found (set MMVAE_HDF5_LIB); tried: " + "; ".join(errors))


def _global(name: str) -> int:
    return hid_t.in_dll(hdf5_lib(), name).value


def _check(status: int, what: str) -> int:
    if status < 0:
        raise HDF5Error(f"HDF5 call failed: {what}")
    return status


def _dims(*values) -> Any:
    return (hsize_t * len(values))(*values)


class _Handle:
    """An HDF5 identifier closed on scope exit."""

    def __init__(self, hid: int, closer: str, what: str):
        self.id = _check(hid, what)
        self._closer = closer

    def __enter__(self):
        return self.id

    def __exit__(self, *exc):
        getattr(hdf5_lib(), self._closer)(self.id)
        return False


def _open_file(path: str, mode: str) -> _Handle:
    h5 = hdf5_lib()
    p = os.fsencode(path)
    if mode == "r":
        return _Handle(h5.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT), "H5Fclose", f"open {path}")
    if os.path.exists(path):  # mode "a" of h5py: read/write if it exists, create otherwise
        return _Handle(h5.H5Fopen(p, H5F_ACC_RDWR, H5P_DEFAULT), "H5Fclose", f"open {path} for appending")
    return _Handle(h5.H5Fcreate(p, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT), "H5Fclose", f"create {path}")


def _exists(loc: int, name: str) -> bool:
    return _check(hdf5_lib().H5Lexists(loc, name.encode(), H5P_DEFAULT), f"H5Lexists {name}") > 0


def _group(loc: int, name: str, create: bool) -> _Handle:
    h5 = hdf5_lib()
    if _exists(loc, name):
        return _Handle(h5.H5Gopen2(loc, name.encode(), H5P_DEFAULT), "H5Gclose", f"open group {name}")
    if not create:
        raise KeyError(name)
    return _Handle(h5.H5Gcreate2(loc, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gclose",
                   f"create group {name}")


def _members(group: int) -> list:
    h5 = hdf5_lib()
    info = (ctypes.c_uint64 * 4)()  # H5G_info_t: storage_type, nlinks, max_corder, mounted
    _check(h5.H5Gget_info(group, info), "H5Gget_info")
    names = []
    for i in range(int(info[1])):
        n = _check(h5.H5Lget_name_by_idx(group, b".", H5_INDEX_NAME, H5_ITER_INC, i, None, 0, H5P_DEFAULT), "name")
        buf = ctypes.create_string_buffer(n + 1)
        h5.H5Lget_name_by_idx(group, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, n + 1, H5P_DEFAULT)
        names.append(buf.value.decode())
    return names


def _column(values) -> tuple:
    """(kind, contiguous buffer) of one metadata column, with h5py's type mapping: str -> variable-length UTF-8,
    integers -> int64, floats -> float64, bool -> int8 enum."""
    if hasattr(values, "to_list"):  # pandas extension arrays, as in the reference (`prediction_writer.py:40-41`)
        values = values.tolist()
    arr = np.asarray(values)
    if arr.dtype.kind in "OUS":
        items = arr.tolist()
        if not all(isinstance(v, (str, bytes)) for v in items):
            raise TypeError("metadata columns of dtype object must hold strings")  # h5py raises for these too
        return "str", [v if isinstance(v, bytes) else v.encode("utf-8") for v in items]
    if arr.dtype.kind == "b":
        return "bool", np.ascontiguousarray(arr, dtype=np.int8)
    if arr.dtype.kind in "iu":
        return "int", np.ascontiguousarray(arr, dtype=np.int64)
    if arr.dtype.kind == "f":
        return "float", np.ascontiguousarray(arr, dtype=np.float64)
    raise TypeError(f"metadata column of dtype {arr.dtype} has no HDF5 mapping")


def _mem_type(kind: str) -> _Handle:
    h5 = hdf5_lib()
    if kind == "str":
        t = _check(h5.H5Tcopy(_global("H5T_C_S1_g")), "H5Tcopy")
        h5.H5Tset_size(t, H5T_VARIABLE)
        h5.H5Tset_cset(t, H5T_CSET_UTF8)
        return _Handle(t, "H5Tclose", "string type")
    if kind == "bool":
        t = _check(h5.H5Tenum_create(_global("H5T_NATIVE_INT8_g")), "H5Tenum_create")
        for name, v in ((b"FALSE", 0), (b"TRUE", 1)):
            h5.H5Tenum_insert(t, name, ctypes.byref(ctypes.c_int8(v)))
        return _Handle(t, "H5Tclose", "bool enum")
    native = {"int": "H5T_NATIVE_INT64_g", "float": "H5T_NATIVE_DOUBLE_g", "f32": "H5T_NATIVE_FLOAT_g"}[kind]
    return _Handle(h5.H5Tcopy(_global(native)), "H5Tclose", native)


def _buffer(kind: str, data) -> Any:
    if kind == "str":
        return (ctypes.c_char_p * len(data))(*data)
    return data.ctypes.data_as(ctypes.c_void_p)


def _chunk_shape(shape: tuple, itemsize: int) -> tuple:
    row = itemsize * int(np.prod(shape[1:], dtype=np.int64)) if len(shape) > 1 else itemsize
    rows = max(1, min(max(int(shape[0]), 1), _CHUNK_BYTES // max(row, 1)))
    return (rows,) + tuple(max(int(s), 1) for s in shape[1:])


def _create_extendable(loc: int, name: str, kind: str, shape: tuple, data) -> None:
    """`create_dataset(name, data=..., maxshape=(None,) + shape[1:], chunks=True)`."""
    h5 = hdf5_lib()
    itemsize = {"str": 16, "bool": 1, "int": 8, "float": 8, "f32": 4}[kind]
    maxdims = (H5S_UNLIMITED,) + tuple(shape[1:])
    with _mem_type(kind) as t, \
            _Handle(h5.H5Screate_simple(len(shape), _dims(*shape), _dims(*maxdims)), "H5Sclose", "dataspace") as sp, \
            _Handle(h5.H5Pcreate(_global("H5P_CLS_DATASET_CREATE_ID_g")), "H5Pclose", "dcpl") as dcpl:
        _check(h5.H5Pset_chunk(dcpl, len(shape), _dims(*_chunk_shape(shape, itemsize))), "H5Pset_chunk")
        with _Handle(h5.H5Dcreate2(loc, name.encode(), t, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT), "H5Dclose",
                     f"create dataset {name}") as ds:
            if shape[0]:
                _check(h5.H5Dwrite(ds, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, _buffer(kind, data)), f"write {name}")


def _shape_of(ds: int) -> tuple:
    h5 = hdf5_lib()
    with _Handle(h5.H5Dget_space(ds), "H5Sclose", "H5Dget_space") as sp:
        nd = _check(h5.H5Sget_simple_extent_ndims(sp), "ndims")
        dims = (hsize_t * max(nd, 1))()
        h5.H5Sget_simple_extent_dims(sp, dims, None)
        return tuple(int(d) for d in dims[:nd])


def _append(loc: int, name: str, kind: str, rows: int, tail: tuple, data, new_size: Optional[int] = None) -> int:
    """`ds.resize(new_size, axis=0); ds[-rows:] = data`."""
    h5 = hdf5_lib()
    with _Handle(h5.H5Dopen2(loc, name.encode(), H5P_DEFAULT), "H5Dclose", f"open dataset {name}") as ds:
        old = _shape_of(ds)
        if tuple(old[1:]) != tuple(tail):
            raise ValueError(f"{name}: appended rows of shape {tail}, dataset holds {old[1:]}")
        total = old[0] + rows if new_size is None else new_size
        _check(h5.H5Dset_extent(ds, _dims(total, *tail)), f"resize {name}")
        if rows == 0:
            return total
        count = (rows,) + tuple(tail)
        start = (total - rows,) + (0,) * len(tail)
        with _mem_type(kind) as t, \
                _Handle(h5.H5Dget_space(ds), "H5Sclose", "filespace") as fs, \
                _Handle(h5.H5Screate_simple(len(count), _dims(*count), None), "H5Sclose", "memspace") as ms:
            _check(h5.H5Sselect_hyperslab(fs, H5S_SELECT_SET, _dims(*start), None, _dims(*count), None), "hyperslab")
            _check(h5.H5Dwrite(ds, t, ms, fs, H5P_DEFAULT, _buffer(kind, data)), f"append to {name}")
        return total


def save_to_hdf5(data: np.ndarray, metadata: pd.DataFrame, hdf5_filepath: str, key: str, strict: bool = True):
    """Append `data` [n, D] and the columns of `metadata` (n rows) under group `key`, creating the extendable
    datasets on first use (reference `prediction_writer.py:14-70`; same error for an unknown column when strict)."""
    data = np.ascontiguousarray(data)
    kind = {"f": "f32" if data.dtype == np.float32 else "float", "i": "int", "u": "int", "b": "bool"}[data.dtype.kind]
    if kind == "float":
        data = data.astype(np.float64)
    elif kind == "int":
        data = data.astype(np.int64)
    elif kind == "bool":
        data = data.astype(np.int8)
    with _open_file(hdf5_filepath, "a") as f, _group(f, key, create=True) as g:
        if _exists(g, RK.PREDICT_SAMPLES) and _exists(g, RK.METADATA):
            new_size = _append(g, RK.PREDICT_SAMPLES, kind, data.shape[0], data.shape[1:], data)
            with _group(g, RK.METADATA, create=False) as mg:
                for col in metadata.columns:
                    col = str(col)
                    ckind, cdata = _column(metadata[col].values)
                    if not _exists(mg, col):
                        if strict:
                            raise RuntimeError(
                                f"metadata column {col} not in h5file for group_key {key}/{RK.METADATA}/{col}")
                        continue
                    _append(mg, col, ckind, len(cdata), (), cdata, new_size=new_size)
        else:
            _create_extendable(g, RK.PREDICT_SAMPLES, kind, data.shape, data)
            with _group(g, RK.METADATA, create=True) as mg:
                for col in metadata.columns:
                    ckind, cdata = _column(metadata[col].values)
                    _create_extendable(mg, str(col), ckind, (len(cdata),), cdata)


def _read_dataset(loc: int, name: str) -> np.ndarray:
    h5 = hdf5_lib()
    with _Handle(h5.H5Dopen2(loc, name.encode(), H5P_DEFAULT), "H5Dclose", f"open dataset {name}") as ds, \
            _Handle(h5.H5Dget_type(ds), "H5Tclose", "H5Dget_type") as ft:
        shape = _shape_of(ds)
        cls, size = h5.H5Tget_class(ft), h5.H5Tget_size(ft)
        n = int(np.prod(shape, dtype=np.int64))
        if cls == H5T_STRING:
            if h5.H5Tis_variable_str(ft) <= 0:
                out = np.empty(shape, dtype=f"S{size}")
                if n:
                    _check(h5.H5Dread(ds, ft, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)), name)
                return out
            out = np.empty(n, dtype=object)  # bytes objects, what h5py >= 3 returns for variable-length strings
            if n:
                buf = (ctypes.c_void_p * n)()
                with _mem_type("str") as t, _Handle(h5.H5Dget_space(ds), "H5Sclose", "space") as sp:
                    _check(h5.H5Dread(ds, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), f"read {name}")
                    for i in range(n):
                        out[i] = ctypes.string_at(buf[i]) if buf[i] else b""
                    h5.H5Dvlen_reclaim(t, sp, H5P_DEFAULT, buf)
            return out.reshape(shape)
        if cls == H5T_ENUM and size == 1:
            kind, dtype = "bool", np.int8
        elif cls == H5T_INTEGER:
            kind, dtype = "int", np.int64
        elif cls == H5T_FLOAT:
            kind, dtype = ("f32", np.float32) if size == 4 else ("float", np.float64)
        else:
            raise HDF5Error(f"{name}: unsupported HDF5 type class {cls}")
        out = np.empty(shape, dtype=dtype)
        if n:
            with _mem_type(kind) as t:
                _check(h5.H5Dread(ds, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(ctypes.c_void_p)), name)
        return out.astype(bool) if kind == "bool" else out


def load_from_hdf5(hdf5_filepath: str, key: str):
    """`(data, metadata, embedding)` of group `key`; each None when absent (reference `prediction_writer.py:73-113`).
    String columns come back as `bytes`, as they do through h5py 3."""
    data = metadata = embedding = None
    with _open_file(hdf5_filepath, "r") as f, _group(f, key, create=False) as g:
        if _exists(g, RK.PREDICT_SAMPLES):
            data = _read_dataset(g, RK.PREDICT_SAMPLES)
        if _exists(g, RK.METADATA):
            with _group(g, RK.METADATA, create=False) as mg:
                metadata = pd.DataFrame({col: _read_dataset(mg, col) for col in _members(mg)})
        if _exists(g, RK.UMAP_EMBEDDINGS):
            embedding = _read_dataset(g, RK.UMAP_EMBEDDINGS)
    return data, metadata, embedding


class PredictionWriter:
    """Per-batch writer of `predict_step` outputs, with the callback surface of the reference's
    `PredictionWriter(BasePredictionWriter)` (`prediction_writer.py:116-203`): `on_predict_start`,
    `write_on_batch_end`, `on_predict_epoch_end`.  `mmvae_amd.trainer.Trainer.predict(..., writer=...)` drives it."""

    def __init__(self, root_dir: str, experiment_name: str = "", run_name: str = "",
                 hdf5_filename: str = "predictions.h5"):
        self.interval = "batch"
        self.root_dir = root_dir
        self.experiment_name = experiment_name
        self.run_name = run_name
        self.hdf5_filename = hdf5_filename
        self._curr_size = 0  # rows written so far

    @property
    def save_dir(self) -> str:
        return os.path.join(self.root_dir, self.experiment_name, self.run_name)

    @property
    def hdf5_filepath(self) -> str:
        return os.path.join(self.save_dir, self.hdf5_filename)

    def write_on_batch_end(self, trainer, pl_module, prediction: Any, batch_indices: Optional[Sequence[int]] = None,
                           batch: Any = None, batch_idx: int = 0, dataloader_idx: int = 0) -> None:
        import torch

        if isinstance(prediction, tuple):
            prediction = prediction[0]
        if not isinstance(prediction, dict) or not all(
                isinstance(p, (tuple, list)) and len(p) == 2
                and isinstance(p[0], (torch.Tensor, np.ndarray)) and isinstance(p[1], pd.DataFrame)
                for p in prediction.values()):
            raise ValueError("Prediction must be a dictionary of type 'dict[str, tuple[torch.Tensor, pd.DataFrame]]' "
                             f"(got {type(prediction)})")
        for key, (data, metadata) in prediction.items():
            data = data.detach().cpu().numpy() if isinstance(data, torch.Tensor) else data
            data[np.isposinf(data)] = np.finfo(np.float32).max  # infinities saturate, in place as in the reference
            data[np.isneginf(data)] = np.finfo(np.float32).min
            save_to_hdf5(data.astype(np.float32), metadata, self.hdf5_filepath, key)
        self._curr_size += list(prediction.values())[0][0].shape[0]

    def on_predict_start(self, trainer=None, pl_module=None) -> None:
        n = 0
        while os.path.exists(self.hdf5_filepath):
            if n == 0:
                warnings.warn("PredictionWriter initialized with hdf5_filepath that already exists: "
                              f"{self.hdf5_filepath}")
            n += 1
            # the reference renames to f"{hdf5_filename[:1]}{n}" (first character + counter, `:190`); kept as is
            self.hdf5_filename = f"{self.hdf5_filename[:1]}{n}"
        os.makedirs(self.save_dir, exist_ok=True)

    def on_predict_epoch_end(self, trainer=None, pl_module=None) -> None:
        self._curr_size = 0
