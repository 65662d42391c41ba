"""Latency of small Search calls through the host-buffer entry point (what the plugin calls): C3 index by default
(LAT_N / LAT_D / LAT_NLIST / LAT_M / LAT_NPROBE / LAT_METRIC=ip pick another shape, LAT_SMALL=0 the regular chain),
nq in {1, 4, 16, 64}, median / p99 over 300 calls each, and the device-buffer entry point for comparison."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
dev = torch.device("cuda", 0)
E = lambda n, v: int(os.environ.get(n, v))
N, d, nlist, M, P, R, k = E("LAT_N", 1000000), E("LAT_D", 128), E("LAT_NLIST", 4096), E("LAT_M", 16), E("LAT_NPROBE", 32), 200, 10
metric = api.METRIC_IP if os.environ.get("LAT_METRIC", "l2") == "ip" else api.METRIC_L2
base = synth.sift_like(N, d=d, seed=1234)
cc, pq = api.train_ivfpq(base[:nlist * 40], nlist, M)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, metric, bucket_init_size=700)
g.set_small_path(E("LAT_SMALL", 1) != 0)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for i0 in range(0, N, 200000):
    g.raw_append(base[i0:i0 + 200000])
    g.add(base[i0:i0 + 200000], i0)
q = synth.sift_like(4096, d=d, seed=4321)
args = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30,
                      coarse_mode=E('LAT_CM', -1))
dq = torch.from_numpy(q).to(dev)
for nq in ([E("LAT_NQ", 0)] if E("LAT_NQ", 0) else (1, 4, 16, 64)):
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    for mode in ("host", "device"):
        ts = []
        for i in range(320):
            xb = q[(i * nq) % 4000:(i * nq) % 4000 + nq]
            t0 = time.perf_counter()
            if mode == "host":
                g.ivfpq_search(xb, k, args)
            else:
                g.ivfpq_search_device(dq[(i * nq) % 4000:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
                g.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts[20:])) * 1e6
        print("nq=%-3d %-6s median %.1f us  p99 %.1f us" % (nq, mode, np.median(ts), ts[int(0.99 * len(ts))]))
