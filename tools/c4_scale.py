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
def _emul_measure(g, step_fn, two, Genv, W, sizes, owner, gnq, nq, base_ms):
    if not two and Genv != "0":
        os.environ["GAMMA_HIP_SCAN_G"] = Genv     # probes per workgroup forced (the library reads it per call)
        print("GAMMA_HIP_SCAN_G=%s:" % Genv, end=" ")
    for _ in range(2):
        step_fn()
    g.synchronize()
    g.profile_enable(True); g.profile_reset()
    t0 = time.perf_counter()
    for _ in range(4):
        step_fn()
    g.synchronize()
    de = (time.perf_counter() - t0) / 4
    pr = g.profile()
    os.environ.pop("GAMMA_HIP_SCAN_G", None)
    print("emulated rank of W=%d%s (lists of shard 0: %.1f M of %.1f M vectors): %.2f ms per step of %d queries (%d per rank) "
          "against %.2f ms for %d queries on one GPU -> per-rank compute efficiency %.0f %%; stage ms/step %s" % (
              W, ", TWO-PHASE (global bound)" if two else "", sizes[owner == 0].sum() / 1e6, sizes.sum() / 1e6, de * 1e3, gnq, nq,
              base_ms, nq, 100.0 * base_ms / (de * 1e3),
              {n: round(pr[n][0] / 4, 3) for n in pr if isinstance(pr[n], tuple) and pr[n][1]}))
    g.profile_enable(False)


if os.environ.get("C4_EMUL"):
    # One rank of a W-GPU LIST-SHARDED job on this index, emulated on one GPU (no communication; tools/rank_emul.py is
    # the C3 version): the handle works under the list mask of shard 0 of gamma_amd.dist.balance_lists(sizes, W) -- it
    # scans only the lists that rank would own -- and a step is what dist.sharded_search has one rank compute for a batch
    # of W x nq queries: the coarse quantizer for its own slice of nq queries, the shard scan + local top-recall_num of
    # ALL W x nq queries over its lists, the merge + re-rank of its slice (stand-in candidate tables of the right shape).
    # Weak scaling: per-rank compute efficiency = (one GPU, nq queries, whole index) / (this step).
    from gamma_amd import dist as gdist
    f32, i32, i64 = torch.float32, torch.int32, torch.int64
    sizes = np.array([g.list_size(l) for l in range(nlist)], dtype=np.int64)
    base_ms = dt * 1e3
    for W in [int(v) for v in os.environ["C4_EMUL"].split(",")]:
        owner = gdist.balance_lists(sizes, W)
        g.set_list_mask((owner == 0).astype(np.uint8))
        gnq = nq * W
        qq = synth.sift_like(gnq, d=d, seed=4321)
        dqq = torch.from_numpy(qq).to(dev)
        cdis = torch.empty((gnq, P), dtype=f32, device=dev)
        probe = torch.empty((gnq, P), dtype=i32, device=dev)
        g.set_list_mask(None)
        for s_ in range(W):      # the assignment of the whole batch (the other ranks' coarse results, all-gathered)
            g.ivfpq_coarse_device(dqq[s_ * nq:].data_ptr(), nq, args, cdis[s_ * nq:].data_ptr(), probe[s_ * nq:].data_ptr())
        g.set_list_mask((owner == 0).astype(np.uint8))
        rdis = torch.empty((gnq, R), dtype=f32, device=dev)
        rids = torch.empty((gnq, R), dtype=i64, device=dev)
        D2 = torch.empty((nq, k), dtype=f32, device=dev)
        I2 = torch.empty((nq, k), dtype=i64, device=dev)

        def estep():
            g.ivfpq_coarse_device(dqq.data_ptr(), nq, args, cdis.data_ptr(), probe.data_ptr())
            g.ivfpq_search_shard_preassigned(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                             rids.data_ptr())
            g.ivfpq_merge_rerank(W, nq, dqq.data_ptr(), k, args, rdis.data_ptr(), rids.data_ptr(), 0, nq, D2.data_ptr(), I2.data_ptr())
        # two-phase shard search (gamma_hip_ivfpq_search_shard_bounded): the reduction across the W shards is emulated -- the
        # bounds every shard's first phase exports are computed once, outside the timed region, under each shard's list mask;
        # the timed step's reduce callback hands shard 0 their minimum (what the all-reduce would leave in its buffer)
        two_modes = [False]
        if os.environ.get("C4_EMUL_TWO"):
            two_modes = [False, True]
            bound = torch.empty((gnq,), dtype=f32, device=dev)
            stream = torch.cuda.ExternalStream(g.stream(), device=dev)
            glob_box = [None]

            def compute_glob():
                own = []
                for s_ in range(W):
                    g.set_list_mask((owner == s_).astype(np.uint8))
                    b_ = torch.empty((gnq,), dtype=f32, device=dev)
                    g.ivfpq_search_shard_bounded(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                                 rids.data_ptr(), b_.data_ptr(), None)
                    g.synchronize()
                    own.append(b_)
                glob_box[0] = torch.stack(own).min(dim=0).values.contiguous()
                g.set_list_mask((owner == 0).astype(np.uint8))
                print("two-phase: shard 0's own bound is the global one for %.1f %% of the queries; mean ratio global / own %.3f" % (
                    100.0 * (own[0] == glob_box[0]).float().mean().item(),
                    (glob_box[0] / own[0].clamp(min=1e-9)).clamp(max=1.0).mean().item()))

            def reduce_cb(n, take_max):
                with torch.cuda.stream(stream):
                    bound.copy_(glob_box[0], non_blocking=True)

            def estep2():
                g.ivfpq_coarse_device(dqq.data_ptr(), nq, args, cdis.data_ptr(), probe.data_ptr())
                g.ivfpq_search_shard_bounded(dqq.data_ptr(), gnq, cdis.data_ptr(), probe.data_ptr(), k, args, rdis.data_ptr(),
                                             rids.data_ptr(), bound.data_ptr(), reduce_cb)
                g.ivfpq_merge_rerank(W, nq, dqq.data_ptr(), k, args, rdis.data_ptr(), rids.data_ptr(), 0, nq, D2.data_ptr(), I2.data_ptr())
            # the coarse call above overwrote rows [0, nq) of the assignment with shard 0's own slice: the same values
        for two in two_modes:
          step_fn = estep2 if two else estep
          for Genv in os.environ.get("C4_EMUL_G1" if two else "C4_EMUL_G", "0").split(","):
            if two and Genv != "0":
                os.environ["GAMMA_HIP_SHARD_G1"] = Genv
                print("GAMMA_HIP_SHARD_G1=%s:" % Genv, end=" ")
            if two:
                compute_glob()
            _emul_measure(g, step_fn, two, Genv, W, sizes, owner, gnq, nq, base_ms)
            os.environ.pop("GAMMA_HIP_SHARD_G1", None)
    g.set_list_mask(None)
g.profile_enable(True)
if os.environ.get("C4_DELETE"):   # 5 % of the documents deleted (the engine's delete bitmap): every search after it tests the bit
    rng = np.random.default_rng(5)
    dead = np.sort(rng.choice(N, N // 20, replace=False)).astype(np.int64)
    bm = np.zeros((N + 7) // 8, dtype=np.uint8)
    np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
    g.bitmap_upload(bm, N)
    for mode, name in (("0", "bit tested per scored code"), (None, "lists compacted (kept until the next write)")):
        if mode is None:
            os.environ.pop("GAMMA_HIP_LIST_COMPACT", None)
        else:
            os.environ["GAMMA_HIP_LIST_COMPACT"] = mode
        for i in range(3):
            g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
        g.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            g.ivfpq_search_device(dq[(i % 2) * nq:].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
        g.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("5 %% deleted, %s: %.2f ms per %d queries = %.0f queries/s" % (name, dt * 1e3, nq, nq / dt))
    g.bitmap_upload(np.zeros((N + 7) // 8, dtype=np.uint8), N)
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
