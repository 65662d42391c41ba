"""Per-step latency distribution of the C3 search (debug aid): syncs after every step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
N, d, nlist, M, nq = 1000000, 128, 4096, 16, 8192
base = synth.sift_like(N, d=d, seed=1234)
q = synth.sift_like(nq * 4, d=d, seed=4321)
dev = torch.device("cuda", 0)
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
g.ivfpq_set_trained(cc, pq, None)
g.add(base, 0)
g.raw_init(d)
g.raw_append(base)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True, min_score=0.0, max_score=1e30)
dq = torch.from_numpy(q).to(dev)
D = torch.empty((nq, 10), dtype=torch.float32, device=dev)
I = torch.empty((nq, 10), dtype=torch.int64, device=dev)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
    xb = dq[(i % 4) * nq:(i % 4 + 1) * nq]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.ivfpq_search_device(xb.data_ptr(), nq, 10, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts)
print("steps %d  ms: first5 %s  median %.3f  p99 %.3f  max %.3f  (argmax %d)" % (
    len(ts), np.round(ts[:5], 3), np.median(ts), np.percentile(ts, 99), ts.max(), ts.argmax()))
print("steps over 2x median:", np.nonzero(ts > 2 * np.median(ts))[0][:20], np.round(ts[ts > 2 * np.median(ts)][:20], 2))
