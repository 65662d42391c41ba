"""Stage times of the IVFPQ search over a sweep of shapes (metric, d, M, nprobe, recall_num, k, batch) on one
GPU: looks for shape-dependent cliffs away from the benchmarked C3 point.  1M vectors per shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
dev = torch.device("cuda", 0)
N = 1000000
SHAPES = [
    # metric, d, M, nlist, P, R, k, nq
    ("L2", 128, 16, 4096, 32, 200, 10, 8192),
    ("IP", 128, 16, 4096, 32, 200, 10, 8192),
    ("L2", 128, 64, 4096, 32, 200, 10, 8192),
    ("L2", 128, 8, 4096, 32, 200, 10, 8192),
    ("L2", 96, 24, 4096, 32, 200, 10, 8192),
    ("L2", 128, 16, 4096, 128, 200, 10, 4096),
    ("L2", 128, 16, 4096, 32, 1000, 100, 4096),
    ("L2", 128, 16, 1024, 16, 100, 10, 8192),
    ("L2", 256, 32, 2048, 32, 100, 10, 4096),
    ("IP", 768, 64, 2048, 32, 100, 10, 2048),
    ("L2", 128, 16, 4096, 32, 200, 10, 64),
    ("L2", 128, 16, 4096, 32, 200, 10, 1),
]
cache = {}
for metric, d, M, nlist, P, R, k, nq in SHAPES:
    key = (metric, d, M, nlist)
    if key not in cache:
        cache.clear()
        n = N if d <= 256 else 300000
        base = synth.sift_like(n, d=d, seed=1234)
        if metric == "IP":
            base = (base / np.maximum(np.linalg.norm(base, axis=1, keepdims=True), 1e-9)).astype(np.float32)
        cc, pq = api.train_ivfpq(base[:nlist * 40], nlist, M)
        g = api.GammaHip(0)
        mt = api.METRIC_L2 if metric == "L2" else api.METRIC_IP
        g.ivfpq_init(d, nlist, M, 8, mt, bucket_init_size=max(200, int(2.5 * n / nlist)))
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        for i0 in range(0, n, 200000):
            g.raw_append(base[i0:i0 + 200000])
            g.add(base[i0:i0 + 200000], i0)
        cache[key] = (g, mt, n)
        del base
    g, mt, n = cache[key]
    q = synth.sift_like(nq * 2, d=d, seed=4321)
    if metric == "IP":
        q = (q / np.maximum(np.linalg.norm(q, axis=1, keepdims=True), 1e-9)).astype(np.float32)
    dq = torch.from_numpy(q).to(dev)
    D = torch.empty((nq, k), dtype=torch.float32, device=dev)
    I = torch.empty((nq, k), dtype=torch.int64, device=dev)
    args = api.SearchArgs(metric=mt, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30)
    for i in range(3):
        g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    g.profile_enable(True); g.profile_reset()
    steps = 10
    t0 = time.perf_counter()
    for i in range(steps):
        g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = g.profile()
    g.profile_enable(False)
    st = {s: round(prof[s][0] / steps, 3) for s in ("coarse", "tables", "scan", "select", "rerank")}
    gb = prof["scan_bytes"] / steps / 1e9
    print("%s d=%d M=%d nlist=%d P=%d R=%d k=%d nq=%d N=%d: %.3f ms = %.0f q/s | %s | scan %.2f GB -> %.2f TB/s" % (
        metric, d, M, nlist, P, R, k, nq, n, dt * 1e3, nq / dt, st, gb, gb / max(1e-9, st["scan"])), flush=True)
