"""End-to-end through the RetrievalModel plugin at C3 size: Init -> store -> Indexing (device
k-means) -> Add in engine-sized batches -> Search; prints timings and recall@10."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gamma_amd import plugin, synth
N, d, nlist, M = 1000000, 128, 4096, 16
base = synth.sift_like(N, d=d, seed=1234)
q = synth.sift_like(1024, d=d, seed=4321)
m = plugin.PluginModel("HIPIVFPQ", d, '{"ncentroids": %d, "nsubvector": %d, "nprobe": 32, "metric_type": "L2"}' % (nlist, M),
                       indexing_size=nlist * 64)
t0 = time.time(); m.store(base); print("store %.1fs" % (time.time() - t0))
t0 = time.time(); assert m.indexing() == 0; print("Indexing() %.1fs" % (time.time() - t0))
t0 = time.time()
for i0 in range(0, N, 10000):
    assert m.add(base[i0:i0 + 10000])
dt = time.time() - t0
print("Add %d vectors in batches of 10000: %.1fs = %.0f vectors/s" % (N, dt, N / dt))
params = '{"metric_type": "L2", "recall_num": 200, "nprobe": 32}'
D, I = m.search(q, 10, params)
t0 = time.time()
for _ in range(10):
    D, I = m.search(q, 10, params)
dt = (time.time() - t0) / 10
print("Search 1024 queries: %.2f ms = %.0f queries/s (host buffers)" % (dt * 1e3, 1024 / dt))
Df, If = m.search(q[:200], 10, '{"metric_type": "L2"}', brute_force=True)
rec = np.mean([len(set(I[i].tolist()) & set(If[i].tolist())) / 10.0 for i in range(200)])
print("recall@10 = %.4f" % rec)
