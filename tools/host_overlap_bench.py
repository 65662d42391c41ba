"""T client threads x one large host-buffer Search each through the HIPIVFPQ plugin (C3 index): what overlapping callers buy.
usage: python tools/host_overlap_bench.py   (env GAMMA_HIP_NO_HOST_OVERLAP / GAMMA_HIP_HOST_PIN_X for A/B)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gamma_amd import plugin, synth

N, d, nlist, M, nq = 1000000, 128, 4096, 16, 16384
base = synth.sift_like(N, d=d, seed=1234)
q = synth.sift_like(4 * nq, d=d, seed=4321)
m = plugin.PluginModel("HIPIVFPQ", d, '{"ncentroids": %d, "nsubvector": %d, "nprobe": 32, "metric_type": "L2", "bucket_init_size": 1000}' % (nlist, M),
                       indexing_size=nlist * 64)
m.store(base)
assert m.indexing() == 0
for i0 in range(0, N, 10000):
    assert m.add(base[i0:i0 + 10000])
rp = '{"metric_type": "L2", "recall_num": 200, "nprobe": 32}'
m.search(q[:nq], 10, rp)
for T in (1, 2, 3, 4):
    calls = max(8, 40 // T)
    dt, lat = m.concurrent_clients(q, rp, T, calls, nq_call=nq, k=10)
    print("%d client threads x %d queries per call: %.2f M queries/s, per-call latency median %.3f ms (mean %.3f)" % (
        T, nq, T * calls * nq / dt / 1e6, np.median(lat) / 1e3, lat.mean() / 1e3), flush=True)
