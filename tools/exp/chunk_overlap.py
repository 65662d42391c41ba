"""C3: does cutting one 16384-query call into internal chunks (the tie replay of chunk i runs beside chunk i + 1) beat the
single chunk?  The workspace budget forces the chunk size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gamma_amd import api, synth
N, d, nlist, M = 1000000, 128, 4096, 16
base = synth.sift_like(N, d=d, seed=1234)
g = api.GammaHip(0)
cc, pq = g.ivfpq_train(base[:nlist * 64], nlist, M)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for i0 in range(0, N, 250000):
    g.raw_append(base[i0:i0 + 250000])
    g.add(base[i0:i0 + 250000], i0)
nq, k = 16384, 10
q = synth.sift_like(2 * nq, d=d, seed=4321)
dev = torch.device("cuda", 0)
dq = torch.from_numpy(q).to(dev)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True, min_score=0.0, max_score=1e30)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
stride = 32 * g.max_list_len() * 4
ref = None
for chunk in (16384, 8192, 5462, 4096, 2048):
    g.set_dist_budget(max(stride * chunk + 1024, 1 << 20))
    for i in range(5):
        g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    dt = (time.perf_counter() - t0) / 20
    out = (D.cpu().numpy().tobytes(), I.cpu().numpy().tobytes())
    same = ref is None or out == ref
    ref = ref or out
    print("chunks of %5d queries: %.3f ms per %d-query call = %.2f M q/s, results %s" % (chunk, dt * 1e3, nq, nq / dt / 1e6, "identical" if same else "DIFFER"))
