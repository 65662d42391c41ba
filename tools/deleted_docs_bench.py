"""C3 index with 5 % of the documents deleted: 16384-query searches with the per-code delete-bitmap test
(GAMMA_HIP_LIST_COMPACT=0) and over lists compacted once per call (default)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
dev = torch.device("cuda", 0)
N, d, nlist, M, P, R, k, nq = 1000000, 128, 4096, 16, 32, 200, 10, 16384
base = synth.sift_like(N, d=d, seed=1234)
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=700)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for i0 in range(0, N, 200000):
    g.raw_append(base[i0:i0 + 200000])
    g.add(base[i0:i0 + 200000], i0)
rng = np.random.default_rng(1)
dead = rng.choice(N, N // 20, replace=False)
bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
g.bitmap_upload(bm, N)
g.delete(dead)
q = torch.from_numpy(synth.sift_like(nq, d=d, seed=4321)).to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30)
for mode in ("0", None):
    if mode is None:
        os.environ.pop("GAMMA_HIP_LIST_COMPACT", None)
    else:
        os.environ["GAMMA_HIP_LIST_COMPACT"] = mode
    for i in range(3):
        g.ivfpq_search_device(q.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    t0 = time.perf_counter()
    for i in range(20):
        g.ivfpq_search_device(q.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
    g.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("5 %% of the documents deleted, %s: %.3f ms per %d queries = %.2f M queries/s" % (
        "per-code bitmap test" if mode == "0" else "lists compacted per call", dt * 1e3, nq, nq / dt / 1e6), flush=True)
