"""bench_scale.py -- the two 8-GPU configurations of BASELINE.json as a runnable, streamed, multi-rank job
(`python bench.py --workload c4|c5 --gpus N`; bench.py starts the ranks and calls run() here).

  c4: IVFPQ nlist=16384 m=32, 100 M x 128 SIFT-shaped synthetic, nprobe=64, lists sharded over the ranks, RCCL exchange of
      per-shard top-recall_num (gamma_amd/dist.py sharded_search).
  c5: IVFPQ 10 M x 768 inner product (embedding-shaped), nlist 4096, M 64, nprobe 64, scalar range filters of 1 / 10 / 50 %
      on an int column, searches under a realtime insert stream of 10 000 vectors/s.

How the index gets there (reference precedent: index/impl/gpu/gamma_gpu_cloner.cpp:200-269 fills the shards from one host
copy, faiss:IndexShards.cpp:283-345 searches them): NO rank ever holds the base on the host.  Vectors come from the
counter-based device streams of gamma_amd/synth.py in chunks; every rank draws the SAME chunk on its own GPU and hands it
to the product's Add under its list mask (HipShardBackend.add: the handle encodes the chunk and keeps the entries of the
lists this rank owns -- an insert reaches the owner of its list with no exchange; gamma_index_ivfpq.cc:424-512).  The
owner table is the greedy sum(len) balance (dist.balance_lists) over list sizes ESTIMATED from a sample assigned before
the first Add (lists are owned before they are filled).

Raw vectors (exact re-rank, gamma_index_ivfpq.cc:642-697) are REPLICATED on every rank (`config.raw_placement`; bytes in
the line): the merge + re-rank of a query slice runs at the slice's owner, which needs rows of every shard's candidates.
A step = one sharded Search of nq x N queries (weak scaling: per-GPU scan work constant).  Rank 0 prints ONE JSON line.
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SPEC = {
    "c4": dict(d=128, nlist=16384, M=32, P=64, R=150, k=10, nq=8192, n=100000000, metric="L2", chunk=2000000,
               name="C4: IVFPQ nlist=16384 m=32, %dx128 synthetic (SIFT-shaped device stream), nprobe=64"),
    "c5": dict(d=768, nlist=4096, M=64, P=64, R=1200, k=10, nq=4096, n=10000000, metric="IP", chunk=250000,
               name="C5: IVFPQ %dx768 inner product (embedding-shaped device stream), nlist=4096 m=64 nprobe=64, range filter "
                    "on an int column, realtime inserts during search"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def run(a):
    import torch
    import torch.distributed as dist

    from gamma_amd import api, synth
    from gamma_amd import dist as gdist

    sp = dict(SPEC[a.workload])
    N = int(a.scale_n) if a.scale_n else sp["n"]
    d, nlist, M, P, k = sp["d"], sp["nlist"], sp["M"], sp["P"], sp["k"]
    R = a.scale_recall_num or sp["R"]
    nq = a.scale_nq or sp["nq"]
    if a.scale_nlist:
        nlist = a.scale_nlist
    l2 = sp["metric"] == "L2"
    metric = api.METRIC_L2 if l2 else api.METRIC_IP
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29713")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(a.backend)
    comm = "rccl over %d devices" % world if (a.backend == "nccl" and world > 1 and not a.one_gpu) else (
        "%s, %d rank(s)%s -- no RCCL communicator of >= 2 ranks was formed in this run" % (
            a.backend, world, " sharing cuda:0" if a.one_gpu and world > 1 else ""))

    def rows(n, start, seed):
        if l2:
            return synth.sift_like_device(n, d=d, seed=seed, start=start, device=dev)
        return synth.embedding_like_device(n, d=d, seed=seed, start=start, device=dev)

    def bcast(t):
        t = t.to(dev) if a.backend == "nccl" else t.cpu()
        if world > 1:
            dist.broadcast(t, 0)
        return t

    # ---- training (rank 0, the product's IndexIVFPQ::train on the device) + the owner table --------------------------
    t0 = time.time()
    g = api.GammaHip(local_rank)
    ntrain = min(N, nlist * 40)
    g.ivfpq_init(d, nlist, M, 8, metric, bucket_init_size=max(200, int((1.3 if l2 else 1.5) * N / nlist)))
    if rank == 0:
        cc, pq = g.ivfpq_train(rows(ntrain, 0, 1234).cpu().numpy(), nlist, M)
        st = [torch.from_numpy(cc), torch.from_numpy(pq)]
    else:
        st = [torch.empty((nlist, d), dtype=torch.float32), torch.empty((M, 256, d // M), dtype=torch.float32)]
    cc, pq = [bcast(t).cpu().numpy() for t in st]
    g.ivfpq_set_trained(cc, pq, None)
    train_s = time.time() - t0
    # list sizes estimated from a sample (every rank assigns the same rows: no exchange), scaled to N
    ns = int(min(N, max(4 * nlist, min(2000000, N // 8))))
    lno, _ = g.encode(rows(ns, 0, 1234).cpu().numpy())
    est = np.bincount(lno[(lno >= 0) & (lno < nlist)], minlength=nlist).astype(np.float64) * (float(N) / ns)
    owner = gdist.balance_lists(np.round(est).astype(np.int64), world)
    owned = (owner == rank).astype(np.uint8)
    if world > 1:
        g.set_list_mask(owned)
    raw_sharded = world > 1 and getattr(a, "raw_placement", "replicated") == "sharded"
    backend = gdist.HipShardBackend(g, local_rank, raw_sharded=raw_sharded, owned=owned)
    g.raw_init(d)

    # ---- streamed Add: every rank draws the same chunk on its GPU and keeps the entries of its own lists ----------------
    t0 = time.time()
    col_rng = np.random.default_rng(3)
    col_all = []
    CH = int(min(sp["chunk"], N))
    for c in range(0, N, CH):
        xb = rows(min(CH, N - c), c, 1234).cpu().numpy()
        backend.add(xb, c)
        if not l2:   # the int column the range filters select on (replicated like the delete bitmap)
            col = col_rng.integers(0, 1000000, size=len(xb)).astype(np.int64)
            g.field_append(0, col)
            col_all.append(col)
    del xb
    torch.cuda.empty_cache()
    add_s = time.time() - t0
    mine = np.array([g.list_size(l) for l in range(nlist)], dtype=np.int64)
    tot = torch.tensor([int(mine.sum())], dtype=torch.int64, device=dev if a.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tot)
    assert int(tot.item()) == N, "the shards hold %d of %d vectors" % (int(tot.item()), N)
    log("[rank %d] train %.1fs, streamed Add of %d vectors %.1fs (%.0f vec/s through every rank's encode); this shard %d "
        "vectors in %d lists; device bytes %.1f GB" % (rank, train_s, N, add_s, N / add_s, int(mine.sum()), int(owned.sum()),
                                                       g.total_mem_bytes() / 1e9))

    # ---- the steps -----------------------------------------------------------------------------------------------
    gnq = nq * world
    nbatches = 2
    d_q = rows(gnq * nbatches, 0, 4321)
    win = dict(min_score=0.0, max_score=1e30) if l2 else dict(min_score=-1e30, max_score=1e30)
    args = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=True, **win)
    g.set_exact_ties(not a.no_exact_ties)

    d_D = torch.empty((gnq, k), dtype=torch.float32, device=dev)
    d_I = torch.empty((gnq, k), dtype=torch.int64, device=dev)
    plain = world == 1 and not a.force_dist     # one GPU: the ordinary Search (the sharded orchestration with --force-dist)

    def step(i, sargs=args):
        xb = d_q[(i % nbatches) * gnq:(i % nbatches + 1) * gnq]
        if plain:
            g.ivfpq_search_device(xb.data_ptr(), gnq, k, sargs, d_D.data_ptr(), d_I.data_ptr())
            return d_D, d_I
        return gdist.sharded_search(backend, xb, k, sargs)

    def timed(sargs, steps, warmup, profile=True):
        for i in range(warmup):
            step(i, sargs)
        g.profile_enable(1 if profile else 0)
        g.profile_reset()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(steps):
            step(warmup + i, sargs)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t1
        prof = g.profile()
        g.profile_enable(False)
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, prof

    # recall@10 against the exact flat search (every rank holds every raw row; rank 0 reports)
    recall = None
    nrq = min(a.recall_queries, 256, gnq)
    Dg, Ig = step(0)
    torch.cuda.synchronize()
    if rank == 0 and a.scale_dump:     # step 0's result table, for a comparison with an unsharded index (tests/test_gpu_dist.py)
        np.savez(a.scale_dump, D=Dg.cpu().numpy(), I=Ig.cpu().numpy(), cc=cc, pq=pq, q=d_q[:gnq].cpu().numpy())
    if rank == 0 and nrq > 0 and not raw_sharded:   # (the exact flat search needs every row on this rank)
        Ih = Ig[:nrq].cpu().numpy()
        Df, If = g.flat_search(d_q[:nrq].cpu().numpy(), k, api.SearchArgs(metric=metric, **win))
        recall = float(np.mean([len(set(Ih[i].tolist()) & set(If[i].tolist())) / float(k) for i in range(nrq)]))
        log("[rank 0] recall@%d = %.4f over %d queries at recall_num %d" % (k, recall, nrq, R))

    # what this rank's candidate exchange carries in one step (two-phase scan + packed exchange: gamma_amd/dist.py)
    backend.exchange_stats = {}
    if not plain:
        step(1)
        torch.cuda.synchronize()
    xst = backend.exchange_stats
    backend.exchange_stats = None
    dt, prof = timed(args, a.steps, a.warmup)
    stages = {n: round(prof[n][0] / max(1, prof[n][1]) * 1e3, 1) for n in ("coarse", "tables", "scan", "select", "rerank") if prof[n][1]}
    scan_ms, scan_n = prof["scan"]
    bytes_per_launch = prof["scan_bytes"] / max(1, scan_n)
    achieved = bytes_per_launch / max(1e-9, scan_ms / 1e3 / max(1, scan_n)) / 1e9
    # every rank's stage times and scan rate, for the line
    mine_line = json.dumps({"rank": rank, "shard_vectors": int(mine.sum()), "stage_us_per_launch": stages,
                            "scan_launches_per_step": scan_n / float(a.steps), "scan_gb_per_step": prof["scan_bytes"] / a.steps / 1e9,
                            "scan_tb_s": round(achieved / 1e3, 3), "device_gb": round(g.total_mem_bytes() / 1e9, 2)})
    per_rank = [None] * world
    if world > 1:
        dist.all_gather_object(per_rank, mine_line)
    else:
        per_rank = [mine_line]

    extra = {}
    if a.workload == "c5" and not a.no_extra:
        col = np.concatenate(col_all)
        # (a) the range filters, no inserts
        filt = {}
        for frac in (0.01, 0.10, 0.50):
            hi = int(frac * 1000000) - 1
            fa = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=True, field_filters=[(0, 0, hi, True, True)], **win)
            Df_, If_ = step(0, fa)
            torch.cuda.synchronize()
            Ih = If_[:512].cpu().numpy()
            ok = bool((col[Ih[Ih >= 0]] <= hi).all())
            dtf, _ = timed(fa, max(2, a.steps // 2), 2, profile=False)
            filt["%g%%" % (frac * 100)] = {"qps": round(gnq * max(2, a.steps // 2) / dtf, 1), "results_inside_the_filter": ok}
        extra["range_filter"] = filt
        # (b) searches under a realtime insert stream: engine-sized Add batches (<= 1000 vectors,
        #     vector/vector_manager.cc:305-349) from a held-out pool, paced to 10 000 vectors/s on a writer thread of every
        #     rank (the same batches everywhere: each rank's handle keeps the entries of its own lists), while the search
        #     steps run with the 10 % filter and without one
        pool_n = int(a.insert_seconds * a.insert_rate) + 2000
        pool = rows(pool_n, N, 1234).cpu().numpy()            # rows N.. of the base stream: never added before
        pcol = col_rng.integers(0, 1000000, size=pool_n).astype(np.int64)
        stop = threading.Event()
        done = {"n": 0, "t": 0.0, "err": None}

        def writer():
            t_start = time.perf_counter()
            nb = 0
            try:
                while not stop.is_set() and (nb + 1) * 1000 <= pool_n:
                    due = t_start + nb * (1000.0 / a.insert_rate)
                    now = time.perf_counter()
                    if now < due:
                        time.sleep(min(0.01, due - now))
                        continue
                    sl = slice(nb * 1000, (nb + 1) * 1000)
                    backend.add(pool[sl], N + nb * 1000)
                    g.field_append(0, pcol[sl])
                    nb += 1
                    done["n"] = nb * 1000
                    done["t"] = time.perf_counter() - t_start
            except Exception as e:      # noqa: BLE001
                done["err"] = str(e)

        ins = {}
        th = threading.Thread(target=writer, daemon=True)
        if world > 1:
            dist.barrier()
        th.start()
        for name, sargs in (("no_filter", args),
                            ("10%_filter", api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=True,
                                                          field_filters=[(0, 0, 99999, True, True)], **win))):
            nst = 0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            while True:
                step(nst, sargs)
                nst += 1
                torch.cuda.synchronize()
                # (every rank runs the same number of steps: the loop ends for all when any rank's clock says so)
                flag = torch.tensor([1 if time.perf_counter() - t1 < a.insert_seconds / 2.0 - 0.5 else 0], dtype=torch.int32,
                                    device=dev if a.backend == "nccl" else "cpu")
                if world > 1:
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    break
            el = time.perf_counter() - t1
            ins[name] = {"qps": round(gnq * nst / el, 1), "steps": nst}
        stop.set()
        th.join()
        ins["insert_rate_asked_vec_s"] = a.insert_rate
        ins["insert_rate_achieved_vec_s"] = round(done["n"] / max(1e-9, done["t"]), 1)
        ins["inserted"] = done["n"]
        if done["err"]:
            ins["writer_error"] = done["err"]
        now_total = torch.tensor([int(sum(g.list_size(l) for l in range(nlist)))], dtype=torch.int64,
                                 device=dev if a.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(now_total)
        ins["vectors_in_the_shards_after"] = int(now_total.item())
        extra["search_during_inserts"] = ins

    if rank != 0:
        dist.destroy_process_group()
        return
    qps = gnq * a.steps / dt
    peak = 8000.0
    line = {
        "metric": "queries/sec @ recall@10>=0.95, IVFPQ nlist=%d nprobe=%d" % (nlist, P),
        "value": round(qps, 1), "unit": "queries/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": dict({
            "workload": (sp["name"] % N) + ", recall_num=%d has_rank=true k=%d, %d queries per step (%d per GPU)" % (R, k, gnq, nq),
            "placement": "one GPU: the whole index, plain Search" if plain else "shard: lists by greedy sum(len) over sizes estimated from %d sample rows; per-shard top-recall_num "
                         "exchanged all-to-all; merge + re-rank at the query slice's owner" % ns,
            "raw_placement": ("sharded: every rank keeps the raw rows of the vectors in ITS lists only (%.1f GB of rows + a 4-byte vid -> row "
                              "table on rank 0; replicated it would be %.1f GB per rank); the exact re-rank distances are computed by the "
                              "shard that holds the row and travel with the packed candidates (16 instead of 12 bytes per entry); recall is "
                              "not computed in this mode (the exact flat search needs every row on one rank)"
                              % (g.raw_stats()["rows"] * d * 4 / 1e9, N * d * 4 / 1e9)) if raw_sharded else
                             ("replicated: %.1f GB of raw vectors on EVERY rank (re-rank at the slice's owner reads rows of every "
                              "shard's candidates); codes + ids + sums sharded; --raw-placement sharded keeps a rank's own rows only"
                              % (N * d * 4 / 1e9)),
            "communicator": comm,
            "exchange_bytes_per_step": {"assignment_all_gather": gnq * P * 8,
                                        "candidates_all_to_all_per_rank": xst.get("exchange_bytes", (gnq // world) * R * 12 * (world - 1)),
                                        "candidates_per_query_sent_by_rank_0": None if not xst.get("queries") else round(xst["exchange_entries"] / xst["queries"], 1),
                                        "whole_tables_would_be_per_rank": (gnq // world) * R * 12 * (world - 1),
                                        "bounds_all_reduce": gnq * 4, "tightening_all_reduce": gnq * 32 * 4 if l2 else 0,
                                        "results_all_gather": gnq * k * 12},
            "shard_scan": "two phases around a min-all-reduce of one float per query (gamma_hip_ivfpq_search_shard_bounded); "
                          "GAMMA_DIST_TWO_PHASE=0 / GAMMA_DIST_PACKED=0 for the round-5 path",
            "build": {"train_s": round(train_s, 2), "streamed_add_s": round(add_s, 1), "add_vec_per_s_per_rank": round(N / add_s, 0)},
            "recall_at_10": recall, "recall_queries": nrq,
            "per_rank": [json.loads(s) for s in per_rank],
            "exact_ties": not a.no_exact_ties,
        }, **extra),
        "cpu_baseline": None,     # the CPU leg belongs to the default (C3) line; this line is the sharded job's
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s", "frac": round(achieved / peak, 4),
                     "traffic": None, "note": "rank 0's scan launches: device-counted sum(len) x %d B / HIP-event duration -- ALGORITHMIC bytes by the "
                     "contract (every (query, list) pair counts the list's codes); the list-major byte-table pass reads a list's codes once "
                     "per tile of 8 queries, so with enough queries per list the figure exceeds the HBM peak" % M},
    }
    print(json.dumps(line), flush=True)
    dist.destroy_process_group()
