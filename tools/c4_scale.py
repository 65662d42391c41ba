"""C4-shaped scale run on one GPU: nlist 16384, M 32, nprobe 64 over N vectors (default 20M)
generated and added in 1M chunks (the host never holds the whole base).  Reports build time,
QPS at 8192-query steps, recall@10 against the exact flat search on the GPU, stage times."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
d, nlist, M, P, R, k, nq = 128, 16384, 32, 64, 100, 10, 8192
dev = torch.device("cuda", 0)
CH = 1000000
t0 = time.time()
first = synth.sift_like(CH, d=d, seed=1234)      # synth blocks are position-keyed: chunk c = rows [c*CH, (c+1)*CH)
cc, pq = api.train_ivfpq(first[:nlist * 40], nlist, M)
print("train %.1fs" % (time.time() - t0)); t0 = time.time()
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=max(200, int(1.3 * N / nlist)))
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for c in range(0, N, CH):
    xb = first if c == 0 else synth.sift_like(min(CH, N - c), d=d, seed=1234, start=c)
    g.raw_append(xb)
    g.add(xb, c)
print("generate + add %d vectors %.1fs, device bytes %.1f GB" % (N, time.time() - t0, g.total_mem_bytes() / 1e9))
q = synth.sift_like(nq * 2, d=d, seed=4321)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=0.0, max_score=1e30)
dq = torch.from_numpy(q).to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
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
print("search: %.2f ms per %d queries = %.0f queries/s" % (dt * 1e3, nq, nq / dt))
print("stage ms per step:", {n: round(prof[n][0] / steps, 3) for n in ("coarse", "tables", "scan", "select", "rerank")},
      "scan GB/step %.2f" % (prof["scan_bytes"] / steps / 1e9))
scan_ms = prof["scan"][0] / max(1, prof["scan"][1])
print("scan roofline: %.2f GB of codes per launch in %.3f ms -> %.2f TB/s = %.3f of 8 TB/s" % (
    prof["scan_bytes"] / max(1, prof["scan"][1]) / 1e9, scan_ms, prof["scan_bytes"] / max(1, prof["scan"][1]) / scan_ms / 1e9,
    prof["scan_bytes"] / max(1, prof["scan"][1]) / scan_ms / 1e9 / 8.0))
NR = 256
qr = q[(steps - 1) % 2 * nq:][:NR]
Df, If = g.flat_search(qr, k, api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30))
Ih = I[:NR].cpu().numpy()
rec = np.mean([len(set(Ih[i].tolist()) & set(If[i].tolist())) / float(k) for i in range(NR)])
print("recall@10 vs flat on %d queries: %.3f" % (NR, rec))
# the metric's bar is recall@10 >= 0.95: recall_num is what limits it at this shape (the PQ short-list, as on the CPU path)
g.profile_enable(False)
for R2 in (150, 200, 300, 400):
    a2 = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R2, has_rank=True, min_score=0.0, max_score=1e30)
    for i in range(2):
        g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, a2, D.data_ptr(), I.data_ptr())
    g.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, a2, D.data_ptr(), I.data_ptr())
    g.synchronize()
    dt2 = (time.perf_counter() - t0) / 4
    Ih = I[:NR].cpu().numpy()       # the last step searched batch 1 = the rows of qr when steps is even ... compare on its own queries
    Df2, If2 = g.flat_search(q[nq:][:NR], k, api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30))
    rec2 = np.mean([len(set(Ih[i].tolist()) & set(If2[i].tolist())) / float(k) for i in range(NR)])
    print("recall_num %d: recall@10 %.3f, %.2f ms per %d queries = %.0f queries/s" % (R2, rec2, dt2 * 1e3, nq, nq / dt2))
g.profile_enable(True)
if os.environ.get("C4_FILTER"):   # a request bitmap that keeps every tenth document (what the engine's range index hands over)
    keep = np.arange(0, N, 10, dtype=np.int64)
    fargs = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=0.0, max_score=1e30,
                           range_filters=[api.make_range_filter(keep)])
    for mode, name in (("0", "predicate per scored code"), (None, "lists compacted per call")):
        if mode is None:
            os.environ.pop("GAMMA_HIP_LIST_COMPACT", None)
        else:
            os.environ["GAMMA_HIP_LIST_COMPACT"] = mode
        for i in range(2):
            g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, fargs, D.data_ptr(), I.data_ptr())
        g.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, fargs, D.data_ptr(), I.data_ptr())
        g.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("10 %% range filter, %s: %.2f ms per %d queries = %.0f queries/s" % (name, dt * 1e3, nq, nq / dt))
# small calls (serving latency): device-buffer entry point, synchronised per call; the small-batch chain against the
# regular one
g.profile_enable(False)
for nqs in (1, 16, 64, 256):
    for small in (1, 0):
        g.set_small_path(small)
        ts = []
        for i in range(120):
            off = (i * nqs) % (nq - nqs)
            t0 = time.perf_counter()
            g.ivfpq_search_device(dq[off:].data_ptr(), nqs, k, args, D.data_ptr(), I.data_ptr())
            g.synchronize()
            ts.append(time.perf_counter() - t0)
        ts = np.sort(np.array(ts[20:])) * 1e6
        print("latency nq=%-4d %s median %.1f us  p99 %.1f us" % (nqs, "small-batch chain" if small else "regular chain    ", np.median(ts), ts[98]))
g.set_small_path(1)
