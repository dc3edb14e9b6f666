"""SURVEY 8(f4), writer side: the `predictions.h5` layout of the reference's PredictionWriter
(`callbacks/prediction_writer.py:14-203`), written through libhdf5, read back through this package's reader and,
independently, through the HDF5 command-line tools of the image when they are present."""
import os
import shutil
import subprocess

import numpy as np
import pandas as pd
import pytest
import torch

from mmvae_amd import predictions as P
from mmvae_amd.constants import REGISTRY_KEYS as RK

try:
    P.hdf5_lib()
except P.HDF5Error as e:  # pragma: no cover - image without libhdf5
    pytest.skip(f"libhdf5 not available: {e}", allow_module_level=True)


def _batch(seed, n, d=16):
    rng = np.random.default_rng(seed)
    data = rng.standard_normal((n, d)).astype(np.float32)
    meta = pd.DataFrame({
        "cell_type": [f"type_{i % 3}_é" for i in range(n)],  # non-ASCII on purpose: stored as UTF-8
        "donor_id": pd.Categorical([f"d{seed}_{i % 2}" for i in range(n)]),
        "n_genes": rng.integers(0, 5000, n),
        "frac": rng.random(n),
        "is_primary": rng.random(n) > 0.5,
        "species": ["human"] * n,
    })
    return data, meta


def _tool(name):
    for cand in (shutil.which(name), f"/opt/conda/bin/{name}"):
        if cand and os.path.exists(cand):
            return cand
    return None


def test_append_layout_and_round_trip(tmp_path):
    path = str(tmp_path / "predictions.h5")
    batches = [_batch(s, n) for s, n in ((0, 33), (1, 512), (2, 7))]
    for data, meta in batches:
        P.save_to_hdf5(data, meta, path, "z")
    P.save_to_hdf5(batches[0][0], batches[0][1], path, "human_xhat")  # second key in the same file
    data, meta, emb = P.load_from_hdf5(path, "z")
    want = np.concatenate([b[0] for b in batches])
    want_meta = pd.concat([b[1] for b in batches], ignore_index=True)
    assert emb is None
    assert data.dtype == np.float32 and np.array_equal(data, want)
    assert sorted(meta.columns) == sorted(want_meta.columns)
    assert [v.decode("utf-8") for v in meta["cell_type"]] == want_meta["cell_type"].tolist()
    assert [v.decode() for v in meta["donor_id"]] == want_meta["donor_id"].astype(str).tolist()
    assert meta["n_genes"].dtype == np.int64 and np.array_equal(meta["n_genes"], want_meta["n_genes"])
    assert meta["frac"].dtype == np.float64 and np.array_equal(meta["frac"], want_meta["frac"])
    assert meta["is_primary"].dtype == bool and np.array_equal(meta["is_primary"], want_meta["is_primary"])
    d2, m2, _ = P.load_from_hdf5(path, "human_xhat")
    assert np.array_equal(d2, batches[0][0]) and len(m2) == 33
    with pytest.raises(KeyError):
        P.load_from_hdf5(path, "absent")


def test_file_is_plain_hdf5_for_other_readers(tmp_path):
    """h5ls / h5dump (the C tools, no code of this package) must see the reference's layout: unlimited first axis,
    chunked storage, variable-length UTF-8 strings, float32 samples."""
    h5ls, h5dump = _tool("h5ls"), _tool("h5dump")
    if not (h5ls and h5dump):
        pytest.skip("HDF5 command-line tools not in this image")
    path = str(tmp_path / "predictions.h5")
    for s, n in ((0, 5), (1, 4)):
        P.save_to_hdf5(*_batch(s, n, d=3), path, "z")
    listing = subprocess.run([h5ls, "-r", path], capture_output=True, text=True, check=True).stdout
    assert f"/z/{RK.PREDICT_SAMPLES}" in listing and "{9/Inf, 3}" in listing
    assert f"/z/{RK.METADATA}/cell_type" in listing and "{9/Inf}" in listing
    header = subprocess.run([h5dump, "-H", "-p", path], capture_output=True, text=True, check=True).stdout
    assert "H5T_IEEE_F32LE" in header and "CHUNKED" in header
    assert "STRSIZE H5T_VARIABLE" in header and "CSET H5T_CSET_UTF8" in header
    assert "H5T_STD_I64LE" in header and "H5T_IEEE_F64LE" in header and "H5T_ENUM" in header
    dump = subprocess.run([h5dump, "-d", f"/z/{RK.PREDICT_SAMPLES}", "-m", "%.9g", path], capture_output=True,
                          text=True, check=True).stdout
    body = dump.split("DATA {")[1].rsplit("}", 2)[0]
    values = [float(tok.strip().rstrip(",")) for line in body.splitlines() if ":" in line
              for tok in line.split(":", 1)[1].split(",") if tok.strip()]
    want = np.concatenate([_batch(0, 5, d=3)[0], _batch(1, 4, d=3)[0]]).ravel()
    assert np.array_equal(np.asarray(values, dtype=np.float32), want)
    species = subprocess.run([h5dump, "-d", f"/z/{RK.METADATA}/species", path], capture_output=True, text=True,
                             check=True).stdout
    assert species.count('"human"') == 9


def test_strict_append_rejects_unknown_column(tmp_path):
    path = str(tmp_path / "p.h5")
    data, meta = _batch(0, 8)
    P.save_to_hdf5(data, meta, path, "z")
    meta2 = meta.assign(extra=1)
    with pytest.raises(RuntimeError, match="metadata column extra not in h5file"):
        P.save_to_hdf5(data, meta2, path, "z")
    P.save_to_hdf5(data, meta2, path, "z", strict=False)  # the column is skipped, rows still appended
    d, m, _ = P.load_from_hdf5(path, "z")
    # the strict failure had already grown /z/data (reference order of operations): 8 + 8 + 8 rows
    assert d.shape[0] == 24 and "extra" not in m.columns
    with pytest.raises(ValueError):
        P.save_to_hdf5(data[:, :5], meta, path, "z")


def test_prediction_writer_callback(tmp_path):
    w = P.PredictionWriter(str(tmp_path), "exp", "run")
    w.on_predict_start()
    assert os.path.isdir(w.save_dir)
    z = torch.randn(6, 4)
    z[0, 0], z[1, 1] = float("inf"), float("-inf")
    meta = pd.DataFrame({"species": ["mouse"] * 6})
    w.write_on_batch_end(None, None, {RK.Z: (z, meta)})
    w.write_on_batch_end(None, None, ({RK.Z: (z.numpy().copy(), meta)},))
    assert w._curr_size == 12
    data, m, _ = P.load_from_hdf5(w.hdf5_filepath, RK.Z)
    assert data.shape == (12, 4) and data[0, 0] == np.finfo(np.float32).max and data[1, 1] == np.finfo(np.float32).min
    assert list(m["species"]) == [b"mouse"] * 12
    with pytest.raises(ValueError, match="Prediction must be a dictionary"):
        w.write_on_batch_end(None, None, {RK.Z: z})
    w.on_predict_epoch_end()
    assert w._curr_size == 0
    w2 = P.PredictionWriter(str(tmp_path), "exp", "run")
    with pytest.warns(UserWarning, match="already exists"):
        w2.on_predict_start()
    assert w2.hdf5_filename == "p1"  # the reference's renaming rule, first character + counter
