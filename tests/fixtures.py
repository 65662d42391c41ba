"""Shared builders for the parity tests: same trained state + same inverted lists loaded
into the CPU oracle and (on a GPU box) into libgamma_hip.so."""
import numpy as np

from gamma_amd import synth
from oracle import binding as B

_cache = {}


def trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2, seed=1234, normalize=False,
                 bucket_init_size=1000):
    """Synthetic base/queries, coarse centroids + PQ codebooks trained the way GammaIVFPQIndex::Indexing trains them
    (IndexIVFPQ::train: the oracle's go_ivfpq_train, bit-identical to the compiled library, tests/test_training_cpu.py),
    and an oracle index with everything added through the oracle's own Add path."""
    key = (d, nlist, M, N, nq, metric, seed, normalize, bucket_init_size)
    if key in _cache:
        return _cache[key]
    base = synth.sift_like(N, d=d, seed=seed)
    q = synth.sift_like(nq, d=d, seed=4321)
    if normalize:
        base = (base / np.maximum(np.linalg.norm(base, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        q = (q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-9)).astype(np.float32)
    ntrain = min(N, max(nlist * 40, 5000))
    cc, pq = B.ivfpq_train(base[:ntrain], nlist, M)
    o = B.OracleIVFPQ(d, nlist, M, 8, metric, bucket_init_size=bucket_init_size)
    o.set_trained(cc, pq, None)
    B.lib().go_set_assign_mode(0)
    assert o.add(base)
    o.set_raw(base)
    case = dict(d=d, nlist=nlist, M=M, N=N, nq=nq, metric=metric, base=base, q=q, cc=cc, pq=pq, oracle=o)
    _cache[key] = case
    return case


def load_hip(case, device=0, table_from_oracle=False, bucket_init_size=1000):
    """Build a libgamma_hip handle holding exactly the oracle's state."""
    from gamma_amd import api
    g = api.GammaHip(device)
    g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], bucket_init_size)
    g.ivfpq_set_trained(case["cc"], case["pq"], case["oracle"].table() if table_from_oracle else None)
    o = case["oracle"]
    lists, counts, vids, codes = [], [], [], []
    for l in range(case["nlist"]):
        ids, cds = o.get_list(l)
        if len(ids):
            lists.append(l)
            counts.append(len(ids))
            vids.append(ids)
            codes.append(cds)
    if lists:
        g.add_keys_batch(lists, counts, np.concatenate(vids), np.concatenate(codes))
    g.raw_init(case["d"])
    g.raw_append(case["base"])
    return g
