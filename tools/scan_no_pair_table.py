"""What the list scan costs WITHOUT the per-pair table build: the C3 index searched with the inner-product metric, whose
scan uses the query's table as it is (written to LDS once per workgroup, no T2 row, no fma, no barriers in the probe
loop) -- the cost a filter pass of the two-pass formulation in DESIGN section 9 would have, less its 4 extra bytes per
code.  Prints the stage times of the L2 and of the inner-product search of the same 16384 queries."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
dev = torch.device("cuda", 0)
N, d, nlist, M, P, R, k, nq = 1000000, 128, 4096, 16, 32, 200, 10, 16384
base = synth.sift_like(N, d=d, seed=1234)
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
q = torch.from_numpy(synth.sift_like(nq, d=d, seed=4321)).to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
for name, metric in (("L2", api.METRIC_L2), ("inner product", api.METRIC_IP)):
    g = api.GammaHip(0)
    g.ivfpq_init(d, nlist, M, 8, metric, bucket_init_size=700)
    g.ivfpq_set_trained(cc, pq, None)
    g.raw_init(d)
    for i0 in range(0, N, 200000):
        g.raw_append(base[i0:i0 + 200000])
        g.add(base[i0:i0 + 200000], i0)
    args = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30)
    for i in range(3):
        g.ivfpq_search_device(q.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    g.profile_enable(True); g.profile_reset()
    t0 = time.perf_counter()
    for i in range(10):
        g.ivfpq_search_device(q.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    dt = (time.perf_counter() - t0) / 10
    prof = g.profile()
    print("%s: %.3f ms per %d queries; stage us per step: %s" % (
        name, dt * 1e3, nq, {n: round(prof[n][0] / 10 * 1e3, 1) for n in ("coarse", "tables", "scan", "select", "rerank")}), flush=True)
    g.close()
