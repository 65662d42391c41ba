"""GPU: training on the device (gamma_hip_kmeans, csrc/gamma_hip_train.cpp; the plugins' Indexing()).  The device runs
faiss::Clustering::train -- the library's subsampling, seeds, centroid sums and empty-cluster splits, the assignment step
on the coarse quantizer's kernels -- and must equal the oracle's restatement of it BIT FOR BIT; the restatement is pinned
against the compiled faiss in tests/test_training_cpu.py (bit-identical with the exact assignment form, within 1 % of
the quantisation error with the GEMM form)."""
import numpy as np
import pytest

from gamma_amd import api, synth
from oracle import binding as B

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,k,niter", [
    (3000, 16, 32, 10),
    (70000, 8, 256, 6),     # more than 256 points per centroid: subsample_training_set
    (520, 4, 256, 25),      # clusters left empty: split_clusters and its RandomGenerator(1234)
    (64, 8, 64, 3),         # as many points as clusters: the copy corner case
    (20000, 128, 512, 10),  # the coarse quantizer's shape
    (15, 8, 4, 5),          # fewer than 20 points: the exact assignment form
])
def test_device_kmeans_is_the_oracles(n, d, k, niter):
    x = synth.sift_like(n, d=d, seed=7)
    g = api.GammaHip(0)
    try:
        cen_g, obj_g = g.kmeans(x, k, niter, seed=1234)
    finally:
        g.close()
    cen_o, obj_o = B.kmeans(x, k, niter, seed=1234)
    assert cen_g.tobytes() == cen_o.tobytes()
    assert obj_g == obj_o


def test_plugin_indexing_is_the_oracles_ivfpq_train():
    """HIPIVFPQ::Indexing() on the first indexing_size vectors of the store == IndexIVFPQ::train as restated by the
    oracle (coarse k-means niter 10, residuals of at most 65536 points, 25-iteration k-means per sub-quantizer)."""
    from gamma_amd import plugin
    d, nlist, M, N = 32, 64, 8, 12000
    base = synth.sift_like(N, d=d, seed=21)
    m = plugin.PluginModel("HIPIVFPQ", d, '{"ncentroids": %d, "nsubvector": %d, "nprobe": 8, "metric_type": "L2"}' % (nlist, M),
                           indexing_size=9000)
    try:
        m.store(base)
        assert m.indexing() == 0
        cc, pq = m.trained_state(nlist, M)
    finally:
        m.close()
    cc_o, pq_o = B.ivfpq_train(base[:9000], nlist, M)
    assert cc.tobytes() == cc_o.tobytes()
    assert pq.tobytes() == pq_o.tobytes()
