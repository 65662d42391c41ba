"""TEST HELPER (not part of the package): a quick Lloyd k-means in torch for tests that only need SOME trained
centroids / codebooks in a hurry (the differential fuzz trains hundreds of small indexes).  The product trains with
faiss's own procedure on the device (gamma_hip_ivfpq_train / the plugins' Indexing()); bench.py, the C3 fixture and the
tools use that.  Search parity is defined GIVEN the trained state, so either trainer serves a parity test.
Deterministic for a given seed / device."""
import numpy as np
import torch


def kmeans(x, k, niter=10, seed=1234, device=None):
    """x: [n, d] float32 (numpy or torch).  Returns centroids [k, d] float32 numpy."""
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    xt = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    n = xt.shape[0]
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(n, generator=g)[:k].to(dev)
    c = xt[perm].clone()
    xn = (xt * xt).sum(1)
    for _ in range(niter):
        assign = torch.empty(n, dtype=torch.long, device=dev)
        bs = max(1, min(n, (1 << 27) // max(k, 1)))
        cn = (c * c).sum(1)
        for i0 in range(0, n, bs):
            xb = xt[i0:i0 + bs]
            dist = xn[i0:i0 + bs, None] + cn[None, :] - 2.0 * (xb @ c.T)
            assign[i0:i0 + bs] = dist.argmin(1)
        sums = torch.zeros_like(c)
        sums.index_add_(0, assign, xt)
        cnt = torch.bincount(assign, minlength=k).to(torch.float32)
        empty = cnt == 0
        c = torch.where(empty[:, None], c, sums / cnt.clamp(min=1)[:, None])
        if empty.any():  # re-seed empty clusters from random points
            ne = int(empty.sum())
            idx = torch.randint(0, n, (ne,), generator=g).to(dev)
            c[empty] = xt[idx]
    return c.cpu().numpy().astype(np.float32)


def assign_l2(x, c, device=None):
    dev = device or ("cuda" if torch.cuda.is_available() else "cpu")
    xt = torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    ct = torch.as_tensor(np.ascontiguousarray(c), dtype=torch.float32, device=dev)
    out = torch.empty(xt.shape[0], dtype=torch.long, device=dev)
    cn = (ct * ct).sum(1)
    bs = max(1, (1 << 27) // max(ct.shape[0], 1))
    for i0 in range(0, xt.shape[0], bs):
        xb = xt[i0:i0 + bs]
        out[i0:i0 + bs] = ((xb * xb).sum(1)[:, None] + cn[None, :] - 2.0 * (xb @ ct.T)).argmin(1)
    return out.cpu().numpy()


def train_ivfpq(xt, nlist, M, niter=10, pq_niter=25, seed=1234, device=None):
    """Coarse centroids [nlist, d] + PQ codebooks [M, 256, d/M] trained on residuals
    (by_residual = true, gamma_index_ivfpq.cc:179; cp.niter = 10 :175; pq 25 iterations)."""
    xt = np.ascontiguousarray(xt, dtype=np.float32)
    d = xt.shape[1]
    dsub = d // M
    cc = kmeans(xt, nlist, niter=niter, seed=seed, device=device)
    a = assign_l2(xt, cc, device=device)
    res = xt - cc[a]
    # at most 256 points per PQ centroid, like faiss's max_points_per_centroid
    if res.shape[0] > 256 * 256:
        rs = np.random.Generator(np.random.Philox(key=seed)).permutation(res.shape[0])[:256 * 256]
        res = res[rs]
    pq = np.empty((M, 256, dsub), dtype=np.float32)
    for m in range(M):
        pq[m] = kmeans(res[:, m * dsub:(m + 1) * dsub], 256, niter=pq_niter, seed=seed + 1 + m,
                       device=device)
    return cc, pq
