"""Device RNG state for the Philox kernels (k15): one {seed, offset} pair per device, advanced on the device by the
kernels themselves (so a captured hipGraph draws fresh numbers on every replay).  Seeded from torch's global seed at
first use, so `torch.manual_seed(s)` followed by `mmvae_amd.rng.reseed()` makes runs reproducible."""
from __future__ import annotations

import torch

_states: dict = {}

STREAM_DROPOUT = 0x44524F50  # "DROP"
STREAM_NORMAL = 0x4E4F524D  # "NORM"


def state(device) -> torch.Tensor:
    device = torch.device(device)
    key = (device.type, device.index)
    st = _states.get(key)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        _states[key] = st
    return st


def reseed(seed: int | None = None) -> None:
    for st in _states.values():
        st.copy_(torch.tensor([(torch.initial_seed() if seed is None else seed) & 0x7FFFFFFFFFFFFFFF, 0],
                              dtype=torch.int64))
