"""One rank of W list shards, emulated on ONE GPU at the C4 shape (nlist 16384, M 32, nprobe 64; N vectors, default 20 M),
single-phase (gamma_hip_ivfpq_search_shard_preassigned: the shard bounds its own recall_num-th best) against TWO-PHASE
(gamma_hip_ivfpq_search_shard_bounded: producers -> reduction of the bounds across the shards -> consumers against the
global bound).  The reduction is emulated: the bounds every shard's first phase exports are computed once, outside the
timed region, under each shard's list mask, and the timed step's reduce callback hands shard 0 their minimum -- what the
all-reduce leaves in its buffer.  A step is what dist.sharded_search has ONE rank compute for a batch of W x nq queries:
the coarse quantizer of its own slice, the shard scan + local selection of all W x nq queries over its lists, the merge +
re-rank of its slice.  Weak scaling: per-rank compute efficiency = (one GPU, nq queries, whole index) / (this step).
usage: python tools/shard_two_phase.py [N=2e7] [W list=2,4,8] [R=100]      env: G1=1,2,4 (two-phase producer group sizes)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gamma_amd import api, synth
from gamma_amd import dist as gdist

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000000
WS = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "2,4,8").split(",")]
R = int(sys.argv[3]) if len(sys.argv) > 3 else 100
C5 = os.environ.get("SHAPE", "c4") == "c5"     # SHAPE=c5: d 768 inner product, nlist 4096, M 64, nprobe 64, 4096 queries per rank
d, nlist, M, P, k, nq = (768, 4096, 64, 64, 10, 4096) if C5 else (128, 16384, 32, 64, 10, 8192)
METRIC = api.METRIC_IP if C5 else api.METRIC_L2
STEPS = int(os.environ.get("STEPS", "6"))
dev = torch.device("cuda", 0)
CH = 200000 if C5 else 1000000
f32, i32, i64 = torch.float32, torch.int32, torch.int64
t0 = time.time()


def rows(n, start, seed):
    if C5:
        return synth.embedding_like_device(n, d=d, seed=seed, start=start, device=dev).cpu().numpy()
    return synth.sift_like(n, d=d, seed=seed, start=start)


first = rows(CH, 0, 1234)
cc, pq = api.train_ivfpq(first[:nlist * 40], nlist, M)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, METRIC, bucket_init_size=max(200, int((1.5 if C5 else 1.3) * N / nlist)))
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for c in range(0, N, CH):
    xb = first[:min(CH, N)] if c == 0 else rows(min(CH, N - c), c, 1234)
    g.raw_append(xb)
    g.add(xb, c)
print("# %s shape, %d vectors (%d codes per list), recall_num %d: train + add %.1f s" % ("C5" if C5 else "C4", N, N // nlist, R, time.time() - t0), flush=True)
args = api.SearchArgs(metric=METRIC, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30 if C5 else 0.0, max_score=1e30)
sizes = np.array([g.list_size(l) for l in range(nlist)], dtype=np.int64)
stream = torch.cuda.ExternalStream(g.stream(), device=dev)


def timed(fn, tag):
    for _ in range(2):
        fn()
    g.synchronize()
    g.profile_enable(True)
    g.profile_reset()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        fn()
    g.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    pr = g.profile()
    g.profile_enable(False)
    st = {n: round(pr[n][0] / STEPS, 3) for n in ("coarse", "tables", "scan", "select", "rerank") if pr[n][1]}
    return dt * 1e3, st


q1 = torch.from_numpy(rows(nq, 0, 4321)).to(dev)
D = torch.empty((nq, k), dtype=f32, device=dev)
I = torch.empty((nq, k), dtype=i64, device=dev)
ONLY_TWO = os.environ.get("ONLY_TWO") is not None   # (kernel profiles of the two-phase step alone)
base_ms, st = (1.0, {}) if ONLY_TWO else timed(lambda: g.ivfpq_search_device(q1.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr()), "one GPU")
print("one GPU, whole index: %.2f ms per %d queries; stage ms/step %s" % (base_ms, nq, st), flush=True)
for W in WS:
    owner = gdist.balance_lists(sizes, W)
    gnq = nq * W
    dqq = torch.from_numpy(rows(gnq, 0, 4321)).to(dev)
    cdis = torch.empty((gnq, P), dtype=f32, device=dev)
    probe = torch.empty((gnq, P), dtype=i32, device=dev)
    g.set_list_mask(None)
    for s_ in range(W):      # the assignment of the whole batch (the other ranks' coarse results, all-gathered)
        g.ivfpq_coarse_device(dqq[s_ * nq:].data_ptr(), nq, args, cdis[s_ * nq:].data_ptr(), probe[s_ * nq:].data_ptr())
    rdis = torch.empty((gnq, R), dtype=f32, device=dev)
    rids = torch.empty((gnq, R), dtype=i64, device=dev)
    D2 = torch.empty((nq, k), dtype=f32, device=dev)
    I2 = torch.empty((nq, k), dtype=i64, device=dev)
    bound = torch.empty((gnq,), dtype=f32, device=dev)
    glob = [None]

    def estep(two):
        g.ivfpq_coarse_device(dqq.data_ptr(), nq, args, cdis.data_ptr(), probe.data_ptr())
        if two:
            g.ivfpq_search_shard_bounded(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                         rids.data_ptr(), bound.data_ptr(), reduce_cb)
        else:
            g.ivfpq_search_shard_preassigned(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                             rids.data_ptr())
        g.ivfpq_merge_rerank(W, nq, dqq.data_ptr(), k, args, rdis.data_ptr(), rids.data_ptr(), 0, nq, D2.data_ptr(), I2.data_ptr())

    def reduce_cb(n, take_max):
        with torch.cuda.stream(stream):
            bound.copy_(glob[0], non_blocking=True)

    g.set_list_mask((owner == 0).astype(np.uint8))
    ms, st = (1.0, {}) if ONLY_TWO else timed(lambda: estep(False), "single")
    print("W=%d single-phase: %.2f ms per step of %d queries (%d per rank) -> per-rank compute efficiency %.0f %%; stage ms/step %s; "
          "exchange %d candidates per query" % (W, ms, gnq, nq, 100.0 * base_ms / ms, st, W * R), flush=True)
    for g1 in os.environ.get("G1", "2").split(","):
        os.environ["GAMMA_HIP_SHARD_G1"] = g1
        own = []
        cnt = torch.zeros((), dtype=torch.float64, device=dev)
        for s_ in range(W):      # phase 1 of every shard: its own bounds
            g.set_list_mask((owner == s_).astype(np.uint8))
            b_ = torch.empty((gnq,), dtype=f32, device=dev)
            g.ivfpq_search_shard_bounded(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                         rids.data_ptr(), b_.data_ptr(), None)
            g.synchronize()
            own.append(b_)
        glob[0] = (torch.stack(own).max(dim=0).values if C5 else torch.stack(own).min(dim=0).values).contiguous()
        for s_ in range(W):      # what the exchange would have to carry: every shard's candidates within the global bound
            g.set_list_mask((owner == s_).astype(np.uint8))
            g.ivfpq_search_shard_bounded(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                         rids.data_ptr(), bound.data_ptr(), reduce_cb)
            g.synchronize()
            cnt += ((rids >= 0) & ((rdis >= glob[0][:, None]) if C5 else (rdis <= glob[0][:, None]))).sum().double()
        g.set_list_mask((owner == 0).astype(np.uint8))
        ms, st = timed(lambda: estep(True), "two")
        print("W=%d TWO-PHASE (G1 %s): %.2f ms per step -> per-rank compute efficiency %.0f %%; stage ms/step %s; shard 0's own bound is "
              "the global one for %.1f %% of the queries; candidates within the global bound: %.1f per query over all shards "
              "(single-phase exchange: %d)" % (W, g1, ms, 100.0 * base_ms / ms, st, 100.0 * (own[0] == glob[0]).float().mean().item(),
                                              cnt.item() / gnq, W * R), flush=True)
        os.environ.pop("GAMMA_HIP_SHARD_G1", None)
g.set_list_mask(None)
