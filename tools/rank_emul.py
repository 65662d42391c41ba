"""One rank of a W-GPU list-sharded C3 job, emulated on one GPU (no communication): the lists rank 0 of
`gamma_amd.dist.balance_lists(.., W)` would own, W*nq queries per step.  Times the three compute legs of
dist.sharded_search -- coarse on the own slice, shard scan of the whole batch, merge + re-rank of the own
slice -- so the compute side of weak scaling can be read without an 8-GPU node.
usage: python tools/rank_emul.py [W=8] [nq=8192]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
from gamma_amd import dist as gdist
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
N, d, nlist, M, P, R, k = 1000000, 128, 4096, 16, 32, 200, 10
dev = torch.device("cuda", 0)
base = synth.sift_like(N, d=d, seed=1234)
gnq = nq * W
queries = synth.sift_like(gnq * 2, d=d, seed=4321)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=max(1000, int(2.5 * N / nlist)))
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
g.ivfpq_set_trained(cc, pq, None)
lno, codes = g.encode(base)
list_sizes = np.bincount(lno, minlength=nlist)
owner = gdist.balance_lists(list_sizes, W)
vids = np.nonzero(owner[lno] == 0)[0].astype(np.int64)
order = np.argsort(lno[vids], kind="stable")
lists, counts = np.unique(lno[vids], return_counts=True)
g.add_keys_batch(lists, counts, vids[order], codes[vids][order])
g.raw_init(d)
for i0 in range(0, N, 1 << 18):
    g.raw_append(base[i0:i0 + (1 << 18)])
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=0.0, max_score=1e30,
                      coarse_mode=1)
dq = torch.from_numpy(queries).to(dev)
f32, i32, i64 = torch.float32, torch.int32, torch.int64
cdis = torch.empty((2, gnq, P), dtype=f32, device=dev)
probe = torch.empty((2, gnq, P), dtype=i32, device=dev)
for b in range(2):
    for s in range(W):
        r0 = b * gnq + s * nq
        g.ivfpq_coarse_device(dq[r0:].data_ptr(), nq, args, cdis[b, s * nq:].data_ptr(), probe[b, s * nq:].data_ptr())
rdis = torch.empty((gnq, R), dtype=f32, device=dev)
rids = torch.empty((gnq, R), dtype=i64, device=dev)
D = torch.empty((nq, k), dtype=f32, device=dev)
I = torch.empty((nq, k), dtype=i64, device=dev)


def step(i):
    b = i % 2
    x = dq[b * gnq:(b + 1) * gnq]
    g.ivfpq_coarse_device(x.data_ptr(), nq, args, cdis[b].data_ptr(), probe[b].data_ptr())
    g.ivfpq_search_shard_preassigned(x.data_ptr(), gnq, cdis[b].data_ptr(), probe[b].data_ptr(), k, args,
                                     rdis.data_ptr(), rids.data_ptr())
    # the W candidate tables of the own slice: stand-ins of the right shape (rows of the local result)
    g.ivfpq_merge_rerank(W, nq, x.data_ptr(), k, args, rdis.data_ptr(), rids.data_ptr(), 0, nq, D.data_ptr(),
                         I.data_ptr())


for i in range(4):
    step(i)
g.synchronize()
for leg in ("all", "coarse", "shard", "merge"):
    g.profile_enable(True); g.profile_reset()
    steps = 20
    t0 = time.perf_counter()
    for i in range(steps):
        b = i % 2
        x = dq[b * gnq:(b + 1) * gnq]
        if leg in ("all", "coarse"):
            g.ivfpq_coarse_device(x.data_ptr(), nq, args, cdis[b].data_ptr(), probe[b].data_ptr())
        if leg in ("all", "shard"):
            g.ivfpq_search_shard_preassigned(x.data_ptr(), gnq, cdis[b].data_ptr(), probe[b].data_ptr(), k, args,
                                             rdis.data_ptr(), rids.data_ptr())
        if leg in ("all", "merge"):
            g.ivfpq_merge_rerank(W, nq, x.data_ptr(), k, args, rdis.data_ptr(), rids.data_ptr(), 0, nq,
                                 D.data_ptr(), I.data_ptr())
    g.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof = g.profile()
    print("W=%d leg %-6s %.3f ms per step of %d queries (%d per rank) -> compute-only %.0f queries/s per rank" % (
        W, leg, dt * 1e3, gnq, nq, nq / dt))
    print("   stage ms/step:", {n: round(prof[n][0] / steps, 3) for n in prof if isinstance(prof[n], tuple) and prof[n][1]})
