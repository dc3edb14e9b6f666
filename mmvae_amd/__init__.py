"""mmvae_amd -- MI355X-native (gfx950) implementation of the MMVAE training step of zdebruine/MMVAE (`cmmvae`).

Layout (only what the hot path needs):
  csrc/      hand-written HIP kernels + the C-ABI of include/mmvae_hip.h  -> libmmvae_hip.so
  _lib.py    ctypes binding (fails loudly when the .so is missing)
  ops.py     one thin Python wrapper per C-ABI entry point
  ...        host-side mirror of the reference's model / trainer plugin surface
"""
__version__ = "0.1.0"
