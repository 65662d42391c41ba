"""C5-shaped run on one GPU: d=768 unit-normalised embedding-shaped vectors, inner product, nlist 4096, M 64
(dsub 12), nprobe 64, recall_num 100, N vectors (default 2M) added in 100k chunks, with and without a 10 %
range filter on an int column (device-side field filter).  Reports build time, QPS at 4096-query steps,
recall@10 against the exact flat search, stage times.
C5_NOISE=<sigma> (default 0.7): the per-coordinate noise of the mixture before normalisation -- at 0.7 a vector is two
thirds noise by energy and 64 x 8-bit codes of a 768-d unit vector keep little of a neighbour ranking; C5_SWEEP=1: the
recall_num / nprobe sweep towards the metric's recall@10 >= 0.95 with the QPS at each point."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
d, nlist, M, P, R, k, nq = 768, (int(sys.argv[2]) if len(sys.argv) > 2 else 4096), 64, 64, 100, 10, 4096
dev = torch.device("cuda", 0)
CH = 100000
NOISE = float(os.environ.get("C5_NOISE", "0.7"))
gen = torch.Generator(device=dev); gen.manual_seed(99)
centres = torch.randn((4096, d), device=dev, generator=gen)


def chunk(n, seed):
    g2 = torch.Generator(device=dev); g2.manual_seed(seed)
    lab = torch.randint(0, 4096, (n,), device=dev, generator=g2)
    v = centres[lab] + NOISE * torch.randn((n, d), device=dev, generator=g2)
    return torch.nn.functional.normalize(v, dim=1).cpu().numpy()


t0 = time.time()
first = chunk(max(CH, nlist * 40), 1000)
cc, pq = api.train_ivfpq(first[:nlist * 40], nlist, M)
print("train %.1fs" % (time.time() - t0)); t0 = time.time()
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_IP, bucket_init_size=max(200, int(1.5 * N / nlist)))
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
rng = np.random.default_rng(3)
done = 0
c = 0
col = []
while done < N:
    n = min(CH, N - done)
    xb = chunk(n, 2000 + c)
    g.raw_append(xb)
    g.add(xb, done)
    col.append(rng.integers(0, 1000000, size=n).astype(np.int64))
    g.field_append(0, col[-1])
    done += n
    c += 1
col = np.concatenate(col)
print("generate + add %d vectors %.1fs, device bytes %.1f GB" % (N, time.time() - t0, g.total_mem_bytes() / 1e9))
q = chunk(nq * 2, 777)
dq = torch.from_numpy(q).to(dev)
D = torch.empty((nq, k), dtype=torch.float32, device=dev)
I = torch.empty((nq, k), dtype=torch.int64, device=dev)
legs = [("no filter", None, None), ("10% range filter", [(0, 0, 99999, True, True)], None)]
if os.environ.get("C5_BITMAP_LEG"):   # the same documents as a request bitmap (what the engine's range index hands over)
    legs.append(("10% range filter as a bitmap", None, [api.make_range_filter(np.nonzero(col <= 99999)[0])]))
for name, ff, rf in legs:
    args = api.SearchArgs(metric=api.METRIC_IP, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30,
                          field_filters=ff, range_filters=rf)
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
    print("%s: %.2f ms per %d queries = %.0f queries/s" % (name, dt * 1e3, nq, nq / dt))
    print("   stage ms per step:", {n: round(prof[n][0] / steps, 3) for n in ("coarse", "tables", "scan", "select", "rerank")},
          "scan GB/step %.2f -> %.2f TB/s" % (prof["scan_bytes"] / steps / 1e9, prof["scan_bytes"] / steps / 1e9 / max(1e-9, prof["scan"][0] / steps)))
    Ih = I[:64].cpu().numpy()
    Df, If = g.flat_search(q[(steps - 1) % 2 * nq:][:64], k, api.SearchArgs(metric=api.METRIC_IP, min_score=-1e30, max_score=1e30, field_filters=ff))
    rec = np.mean([len(set(Ih[i].tolist()) & set(If[i].tolist()) - {-1}) / float(max(1, (If[i] >= 0).sum())) for i in range(64)])
    print("   recall@10 vs flat on 64 queries: %.3f" % rec)
if os.environ.get("C5_SWEEP"):
    g.profile_enable(False)
    NR = 256
    fl = api.SearchArgs(metric=api.METRIC_IP, min_score=-1e30, max_score=1e30)
    Df, If = g.flat_search(q[:NR], k, fl)
    print("sweep (noise %.2f, %d vectors): recall@10 vs flat on %d queries" % (NOISE, N, NR))
    for P2, R2 in ((64, 100), (64, 200), (64, 400), (64, 1000), (64, 2000), (64, 4000), (128, 1000), (128, 4000), (256, 4000)):
        if P2 > nlist:
            continue
        a2 = api.SearchArgs(metric=api.METRIC_IP, nprobe=P2, recall_num=R2, has_rank=True, min_score=-1e30, max_score=1e30)
        for i in range(2):
            g.ivfpq_search_device(dq.data_ptr(), nq, k, a2, D.data_ptr(), I.data_ptr())
        g.synchronize()
        t0 = time.perf_counter()
        for i in range(4):
            g.ivfpq_search_device(dq.data_ptr(), nq, k, a2, D.data_ptr(), I.data_ptr())
        g.synchronize()
        dt2 = (time.perf_counter() - t0) / 4
        Ih = I[:NR].cpu().numpy()
        rec2 = np.mean([len(set(Ih[i].tolist()) & set(If[i].tolist())) / float(k) for i in range(NR)])
        print("   nprobe %3d recall_num %4d: recall@10 %.3f, %.2f ms per %d queries = %.0f queries/s" % (P2, R2, rec2, dt2 * 1e3, nq, nq / dt2))
    g.profile_enable(True)
# small calls (serving latency): device-buffer entry point, synchronised per call; small-batch chain / regular chain
g.profile_enable(False)
args = api.SearchArgs(metric=api.METRIC_IP, nprobe=P, recall_num=R, has_rank=True, min_score=-1e30, max_score=1e30)
for nqs in (1, 16, 64):
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
