"""gamma_amd -- MI355X-native implementation of vearch/gamma's retrieval hot path
(IVFPQ list scan, flat L2/IP, top-k) behind the RetrievalModel plugin boundary.

Layout:
  csrc/      hand-written gfx950 HIP kernels + the C ABI (include/gamma_hip.h) -> libgamma_hip.so
  host/      C++ RetrievalModel plugin (HIPIVFPQ / HIPFLAT) that calls the C ABI
  api.py     ctypes face of the C ABI for tests / bench
  dist.py    list-sharded multi-GPU search (RCCL all-gather of per-shard top-k)
  synth.py   portable synthetic data (bench / tests)

Importing the package does not load the HIP library; gamma_amd.api.GammaHip() does and raises
if it is missing (there is no CPU fallback).
"""
__all__ = ["api", "synth"]
