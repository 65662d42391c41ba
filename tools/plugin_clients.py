"""Closed-loop single-query clients against the HIPIVFPQ plugin (C++ threads calling RetrievalModel::Search,
the pattern of the reference's tools/perf.cc), C3 index.  Run once as is and once with GAMMA_HIP_NO_COMBINE=1
to see what request combining in libgamma_hip.so buys."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gamma_amd import plugin, synth
N, d, nlist, M = 1000000, 128, 4096, 16
base = synth.sift_like(N, d=d, seed=1234)
q = synth.sift_like(4096, d=d, seed=4321)
m = plugin.PluginModel("HIPIVFPQ", d, '{"ncentroids": %d, "nsubvector": %d, "nprobe": 32, "metric_type": "L2"}' % (nlist, M),
                       indexing_size=nlist * 64)
m.store(base)
assert m.indexing() == 0
for i0 in range(0, N, 10000):
    assert m.add(base[i0:i0 + 10000])
params = '{"metric_type": "L2", "recall_num": 200, "nprobe": 32}'
m.search(q[:4], 10, params)
TS = [int(v) for v in os.environ['PC_THREADS'].split(',')] if os.environ.get('PC_THREADS') else (1, 8, 32, 128, 512)
for T in TS:
    calls = max(100, int(os.environ.get('PC_CALLS', 20000)) // T)
    dt, lat = m.concurrent_clients(q, params, T, calls)
    lat = np.sort(lat)
    print("%3d client threads x 1 query per call: %8.0f queries/s, latency median %.0f us, p99 %.0f us (mean %.0f, p99.9 %.0f, max %.0f; %d calls in %.2f s)" % (
        T, T * calls / dt, np.median(lat), lat[int(0.99 * len(lat))], lat.mean(), lat[int(0.999 * len(lat))], lat[-1], len(lat), dt), flush=True)
N0 = N
for T in (() if os.environ.get('PC_THREADS') else (8, 32, 128)):
    bad, sec = m.concurrent_filtered_check(q, params, T, max(50, 4000 // T), N0 // (T + 4), N0 // 4)
    print("%3d client threads x 1 query, each with its own range filter: %8.0f queries/s (%d results differ from the call made alone)" % (
        T, T * max(50, 4000 // T) / sec, bad), flush=True)
