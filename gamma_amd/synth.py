"""Portable synthetic data for the hot-path benchmarks.

No dataset can be downloaded, so the bench and the parity tests use a counter-based
(Philox) generator: the same (seed, n, d) gives the same vectors on every machine with this
numpy, independent of how many vectors are requested at once (fixed 65 536-row blocks).

"SIFT-shaped": d=128, mixture of K=2048 Gaussian clusters, centres ~ U[32,160]^d, per-cluster
sigma ~ U[12,28], weights ∝ (i+1)^-0.5, values clipped to [0,255] and rounded to integers
stored as float32.  Mixture seed 99, base seed 1234, query seed 4321.
"""
import numpy as np

_BLOCK = 65536


def _mixture(d, K, mix_seed):
    g = np.random.Generator(np.random.Philox(key=mix_seed))
    centres = g.uniform(32.0, 160.0, size=(K, d)).astype(np.float32)
    sigma = g.uniform(12.0, 28.0, size=K).astype(np.float32)
    w = 1.0 / np.sqrt(np.arange(1, K + 1, dtype=np.float64))
    return centres, sigma, np.cumsum(w / w.sum())


def sift_like(n, d=128, seed=1234, mix_seed=99, K=2048, start=0):
    """Rows [start, start+n) of the infinite SIFT-shaped stream for `seed`."""
    centres, sigma, cdf = _mixture(d, K, mix_seed)
    out = np.empty((n, d), dtype=np.float32)
    b0, b1 = start // _BLOCK, (start + n + _BLOCK - 1) // _BLOCK
    for b in range(b0, b1):
        g = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, b, 1]))
        u = g.random(_BLOCK)
        z = g.standard_normal((_BLOCK, d), dtype=np.float32)
        c = np.minimum(np.searchsorted(cdf, u), K - 1)
        blk = centres[c] + sigma[c][:, None] * z
        np.clip(blk, 0.0, 255.0, out=blk)
        np.rint(blk, out=blk)
        lo, hi = max(start, b * _BLOCK), min(start + n, (b + 1) * _BLOCK)
        out[lo - start:hi - start] = blk[lo - b * _BLOCK:hi - b * _BLOCK]
    return out


def embedding_like(n, d=768, seed=1234, mix_seed=99, K=4096, start=0):
    """Unit-normalised Gaussian-mixture vectors (config 5 shape, inner-product search)."""
    g0 = np.random.Generator(np.random.Philox(key=mix_seed + 7))
    centres = g0.standard_normal((K, d), dtype=np.float32)
    out = np.empty((n, d), dtype=np.float32)
    b0, b1 = start // _BLOCK, (start + n + _BLOCK - 1) // _BLOCK
    for b in range(b0, b1):
        g = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, b, 2]))
        c = g.integers(0, K, size=_BLOCK)
        blk = centres[c] + 0.6 * g.standard_normal((_BLOCK, d), dtype=np.float32)
        blk /= np.linalg.norm(blk, axis=1, keepdims=True)
        lo, hi = max(start, b * _BLOCK), min(start + n, (b + 1) * _BLOCK)
        out[lo - start:hi - start] = blk[lo - b * _BLOCK:hi - b * _BLOCK]
    return out


# ---- device-side streams for the 10^7 .. 10^8-vector configurations --------------------------------------------------
# The numpy streams above cost ~2-6 us per row on one host core (100 M rows: minutes, on every rank of a sharded job).  The
# same mixtures drawn on the GPU: rows [start, start+n) of a stream keyed by (seed, block of 65 536 rows), so every rank
# of a job -- and a test that re-reads a chunk -- gets the same rows whatever the chunking.  torch is plumbing here
# (a random-number source); the values are NOT those of the numpy streams (other generator), the distribution is.
_dev_cache = {}


def _dev_block(kind, b, d, seed, mix_seed, K, device, noise):
    import torch
    key = (kind, d, mix_seed, K, str(device))
    if key not in _dev_cache:
        if kind == "sift":
            centres, sigma, cdf = _mixture(d, K, mix_seed)
            _dev_cache[key] = (torch.from_numpy(centres).to(device), torch.from_numpy(sigma).to(device),
                               torch.from_numpy(cdf.astype(np.float32)).to(device))
        else:
            g0 = torch.Generator(device=device)
            g0.manual_seed(mix_seed + 7)
            _dev_cache[key] = (torch.randn((K, d), device=device, generator=g0),)
    g = torch.Generator(device=device)
    g.manual_seed((seed << 24) + b)
    if kind == "sift":
        centres, sigma, cdf = _dev_cache[key]
        u = torch.rand((_BLOCK,), device=device, generator=g)
        c = torch.clamp(torch.searchsorted(cdf, u), max=K - 1)
        blk = centres[c] + sigma[c][:, None] * torch.randn((_BLOCK, d), device=device, generator=g)
        return torch.round(torch.clamp(blk, 0.0, 255.0))
    (centres,) = _dev_cache[key]
    c = torch.randint(0, K, (_BLOCK,), device=device, generator=g)
    blk = centres[c] + noise * torch.randn((_BLOCK, d), device=device, generator=g)
    return torch.nn.functional.normalize(blk, dim=1)


def _dev_rows(kind, n, d, seed, mix_seed, K, start, device, noise=0.7):
    import torch
    device = torch.device(device)
    out = torch.empty((n, d), dtype=torch.float32, device=device)
    for b in range(start // _BLOCK, (start + n + _BLOCK - 1) // _BLOCK):
        blk = _dev_block(kind, b, d, seed, mix_seed, K, device, noise)
        lo, hi = max(start, b * _BLOCK), min(start + n, (b + 1) * _BLOCK)
        out[lo - start:hi - start] = blk[lo - b * _BLOCK:hi - b * _BLOCK]
    return out


def sift_like_device(n, d=128, seed=1234, mix_seed=99, K=2048, start=0, device="cuda:0"):
    """Rows [start, start+n) of the SIFT-shaped device stream for `seed` (a torch tensor on `device`)."""
    return _dev_rows("sift", n, d, seed, mix_seed, K, start, device)


def embedding_like_device(n, d=768, seed=1234, mix_seed=99, K=4096, start=0, device="cuda:0", noise=0.7):
    """Rows [start, start+n) of the unit-normalised embedding-shaped device stream (configuration 5; `noise` = the
    per-coordinate sigma before normalisation, tools/c5_scale.py's C5_NOISE)."""
    return _dev_rows("emb", n, d, seed, mix_seed, K, start, device, noise)
