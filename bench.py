#!/usr/bin/env python3
"""bench.py -- QPS of the MI355X-native Gamma IVFPQ search path.

Workload (BASELINE.json metric, configs[2] "C3"): IVFPQ nlist=4096 m=16 nbits=8 over 1M x 128
SIFT1M-shaped synthetic vectors, nprobe=32, recall_num=200 + exact re-rank (the operating point
that reaches recall@10 >= 0.95 on this data), k=10, L2.  One "step" = one
Search call over a batch of `--nq` synthetic queries already resident in HBM.

  python bench.py --gpus N --steps K --warmup W          (N=1 default)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N>1: the index is sharded by IVF list across the ranks (gamma_amd/dist.py): every rank scans the
lists it owns for the whole batch, an RCCL all-to-all hands each rank the candidates of its query
slice, merge + re-rank there.  A step then holds N x nq queries (per-GPU scan work constant:
"weak" scaling).  Rank 0 prints ONE JSON line (plus diagnostics on stderr).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("MKL_THREADING_LAYER", "GNU")   # oracle/_ref links MKL; its Intel OpenMP layer next to libgomp is ~10x slower


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nq", type=int, default=16384, help="queries per Search call (one step)")
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--recall-num", type=int, default=200)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--no-rank", action="store_true")
    ap.add_argument("--coarse-mode", type=int, default=-1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--recall-queries", type=int, default=1000)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the sharded orchestration (gamma_amd.dist) even on one GPU")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend; 'gloo' with --one-gpu runs all ranks on GPU 0 (functional "
                         "check of the multi-rank path on a single-GPU box, not a measurement)")
    ap.add_argument("--one-gpu", action="store_true", help="every rank uses cuda:0")
    ap.add_argument("--timed-events", default="all", choices=["scan", "all", "none"],
                    help="HIP events inside the timed region: around every stage (default; roofline.achieved is the scan launch "
                         "measured live), around the scan launch only, or none (everything from an untimed pass over the same steps)")
    ap.add_argument("--deferred-replay", action="store_true",
                    help="exact ties: replay the flagged queries of a step beside the NEXT step's coarse quantizer "
                         "(gamma_hip_set_deferred_replay: a streamed schedule of device-pointer calls whose flagged rows "
                         "complete one call later).  Default: every call is complete when it returns -- what the plugin "
                         "boundary hands back -- and `value` is that figure; the streamed figure is config.exact_ties.qps")
    ap.add_argument("--placement", default="auto", choices=["auto", "shard", "replicate"],
                    help="--gpus N > 1: 'shard' = the lists split over the ranks (greedy sum(len)), candidates exchanged "
                         "(gamma_amd.dist.sharded_search); 'replicate' = every rank holds the whole index and answers its "
                         "slice of the batch (replicated_search: one all-gather of the results); 'auto' = replicate while "
                         "the lists are under 2 GiB per GPU -- list sharding is for indexes that need it (C4), at C3 size "
                         "it multiplies the per-query fixed work by N (DESIGN.md, multi-GPU)")
    ap.add_argument("--in-process", action="store_true",
                    help="--gpus N through ONE process: the in-process group of handles (gamma_hip_group_*, what the "
                         "plugins run with \"devices\"), tools/group_bench.py")
    ap.add_argument("--no-exact-ties", action="store_true",
                    help="run the timed region WITHOUT the reference's heap order inside exact ties (the library default "
                         "is on: labels identical to the reference at every rank)")
    ap.add_argument("--dup-queries", type=int, default=0,
                    help="experiment: every batch repeats its first N queries (the lists they probe stay cache "
                         "resident: what the scan costs without its table traffic)")
    ap.add_argument("--batches", default="1,32,1024", help="batch sizes of the qps_by_batch leg (BASELINE.md protocol)")
    ap.add_argument("--no-plugin", action="store_true", help="skip the leg that drives the HIPIVFPQ plugin (Indexing, Add, Search with host buffers)")
    ap.add_argument("--no-shapes", action="store_true", help="skip the reduced C4 / C5 shape legs (child processes, ~1 min)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra legs (exact ties, batch sizes 1/32/1024, coarse_mode 0, C2 flat)")
    ap.add_argument("--pmc-traffic", type=float, default=None,
                    help="HBM bytes per scan launch from a separate rocprofv3 --pmc pass")
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "c5"],
                    help="c3 (default): the metric's configuration, this file.  c4 / c5: BASELINE.json configs[3] / configs[4] as a "
                         "streamed list-sharded job over --gpus ranks (bench_scale.py): vectors drawn on the device in chunks, every "
                         "rank encodes each chunk under its list mask, no host holds the base")
    ap.add_argument("--scale-n", type=float, default=0, help="--workload c4 / c5: vectors (default 1e8 / 1e7)")
    ap.add_argument("--scale-nq", type=int, default=0, help="--workload c4 / c5: queries per GPU and step (default 8192 / 4096)")
    ap.add_argument("--scale-nlist", type=int, default=0, help="--workload c4 / c5: lists (default 16384 / 4096)")
    ap.add_argument("--scale-recall-num", type=int, default=0, help="--workload c4 / c5: short-list (default 150 / 1000: recall@10 >= 0.95)")
    ap.add_argument("--raw-placement", default="replicated", choices=["replicated", "sharded"],
                    help="--workload c4 / c5 with several ranks: 'sharded' = every rank keeps the raw rows of the vectors in ITS lists "
                         "only (gamma_hip_raw_put) and the exact re-rank distances travel with the candidates (C4 on 8 GPUs: 6.4 GB of "
                         "rows per GPU instead of 51.2); 'replicated' = every rank holds every row")
    ap.add_argument("--scale-dump", default="", help="--workload c4 / c5: rank 0 writes step 0's (D, I), the trained state and the queries here (.npz)")
    ap.add_argument("--insert-rate", type=float, default=10000.0, help="--workload c5: vectors per second of the realtime insert leg")
    ap.add_argument("--insert-seconds", type=float, default=12.0, help="--workload c5: length of the insert leg")
    a = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: this process only starts the N ranks (one fresh process per
    # GPU, RCCL rendezvous on 127.0.0.1) -- BEFORE anything here touches the GPU -- and waits for them; rank 0
    # prints the line.  Under torchrun (WORLD_SIZE set) the world size must BE --gpus.
    if a.in_process:   # one process, N handles: a child of its own (this process has not touched the GPU yet)
        import subprocess
        cmd = [sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "group_bench.py"),
               "--gpus", str(a.gpus), "--steps", str(a.steps), "--warmup", str(a.warmup), "--nq", str(a.nq), "--placement", a.placement]
        if a.one_gpu:
            cmd.append("--one-gpu")
        sys.exit(subprocess.call(cmd))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus:
        log("error: --gpus %d but WORLD_SIZE=%s" % (a.gpus, os.environ.get("WORLD_SIZE")))
        sys.exit(2)

    if a.workload != "c3":
        import bench_scale
        if a.steps == 50 and a.warmup == 10:     # the defaults are C3's: a step here is 26 ms .. seconds
            a.steps, a.warmup = 10, 3
        return bench_scale.run(a)

    import torch
    import torch.distributed as dist

    from gamma_amd import api, synth
    from gamma_amd import dist as gdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if a.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29711")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(a.backend)
        if dist.get_world_size() != a.gpus:
            log("error: process group of %d ranks for --gpus %d" % (dist.get_world_size(), a.gpus))
            sys.exit(2)

    t0 = time.time()
    N, d, nlist, M = a.n, a.d, a.nlist, a.m
    base = synth.sift_like(N, d=d, seed=1234)
    nbatches = 4
    # weak scaling over GPUs: every rank contributes a.nq queries to each step's batch, the
    # lists (scan work) are split, so per-GPU scan work per step is constant
    gnq = a.nq * world
    queries = synth.sift_like(gnq * nbatches, d=d, seed=4321)
    if a.dup_queries > 0:
        queries = np.ascontiguousarray(np.tile(queries[:a.dup_queries], (queries.shape[0] // a.dup_queries + 1, 1))[:queries.shape[0]])
    log("[rank %d] data %.1fs" % (rank, time.time() - t0))

    g = api.GammaHip(local_rank)
    g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=max(1000, int(2.5 * N / nlist)))

    # ---- training + encoding (setup, untimed): rank 0 trains, everyone gets the same state ----
    t0 = time.time()
    ntrain = min(N, nlist * 64)
    train_s = None
    if rank == 0:
        # trained the way a Gamma table's Indexing() trains it: faiss's IndexIVFPQ::train on the first nlist * 64 vectors, on
        # the device (gamma_hip_ivfpq_train; bit-identical to the compiled library's training, tests/test_training_cpu.py)
        t1 = time.time()
        cc, pq = g.ivfpq_train(base[:ntrain], nlist, M)
        train_s = time.time() - t1
        log("[rank 0] IndexIVFPQ::train on the device: %.2fs for %d x %d vectors" % (train_s, ntrain, d))
        state = [torch.from_numpy(cc).to(dev), torch.from_numpy(pq).to(dev)]
    else:
        state = [torch.empty((nlist, d), dtype=torch.float32, device=dev),
                 torch.empty((M, 256, d // M), dtype=torch.float32, device=dev)]
    if world > 1:
        for t in state:
            dist.broadcast(t, 0)
    cc, pq = state[0].cpu().numpy(), state[1].cpu().numpy()
    g.ivfpq_set_trained(cc, pq, None)  # precomputed table built on device
    if rank == 0:
        lno, codes = g.encode(base)     # device assign + residual + PQ encode
        enc = [torch.from_numpy(lno).to(dev), torch.from_numpy(codes).to(dev)]
    else:
        enc = [torch.empty(N, dtype=torch.int64, device=dev),
               torch.empty((N, M), dtype=torch.uint8, device=dev)]
    if world > 1:
        for t in enc:
            dist.broadcast(t, 0)
    lno, codes = enc[0].cpu().numpy(), enc[1].cpu().numpy()
    del enc, state
    list_sizes = np.bincount(lno, minlength=nlist)
    # Several GPUs: an index whose lists fit one GPU (<= 2 GiB of codes + ids: C3) is loaded WHOLE on every rank and BOTH
    # placements are timed in this run -- "shard" (the north star's design: lists split by greedy sum(len), every rank scans
    # its lists for the whole batch, all-to-all of per-shard top-recall_num, merge + re-rank at the query's owner; the
    # handle then works under the list mask of its shard) and "replicate" (every rank answers its slice of the batch on the
    # whole index, one all-gather of the top-k).  `value` is the one --placement names (auto: the faster of the two, named
    # in config.placement); both figures are in the line.  An index beyond that size is list-sharded, period.
    fits = N * (M + 12) <= (2 << 30)
    both = use_dist and fits and a.placement == "auto"
    replicate = use_dist and (a.placement == "replicate" or both)
    shard_owner = gdist.balance_lists(list_sizes, world) if use_dist else None
    owner = gdist.balance_lists(list_sizes, 1 if replicate else world)
    mine = owner[lno] == (0 if replicate else rank)
    vids = np.nonzero(mine)[0].astype(np.int64)
    order = np.argsort(lno[vids], kind="stable")
    lists, counts = np.unique(lno[vids], return_counts=True)
    g.add_keys_batch(lists, counts, vids[order], codes[vids][order])
    g.raw_init(d)
    for i0 in range(0, N, 1 << 18):
        g.raw_append(base[i0:i0 + (1 << 18)])
    log("[rank %d] train+encode+load %.1fs; lists mean %.1f max %d; device bytes %.0f MB" % (
        rank, time.time() - t0, list_sizes.mean(), list_sizes.max(), g.total_mem_bytes() / 1e6))

    k, R = a.k, max(a.recall_num, a.k)
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=a.nprobe, recall_num=a.recall_num,
                          has_rank=not a.no_rank, min_score=0.0, max_score=1e30,
                          coarse_mode=a.coarse_mode)
    d_q = torch.from_numpy(queries).to(dev)
    d_D = torch.empty((gnq, k), dtype=torch.float32, device=dev)
    d_I = torch.empty((gnq, k), dtype=torch.int64, device=dev)
    backend = gdist.HipShardBackend(g, local_rank) if use_dist else None
    g.set_exact_ties(not a.no_exact_ties)
    # the steps of the timed loop are back-to-back device-pointer calls: the handful of queries a step flags for the heap
    # replay are redone beside the NEXT step's coarse quantizer / query tables; everything is complete at the
    # synchronize that ends the timed region (include/gamma_hip.h, gamma_hip_set_deferred_replay)
    # (replicated ranks do the same: gamma_amd.dist.ReplicatedStream gathers a step's results one step behind, after the
    #  next step's search has been enqueued, and flush() -- inside the timed region -- gathers the last step's)
    deferred = (not use_dist or replicate) and (not a.no_exact_ties) and a.deferred_replay
    rstream = gdist.ReplicatedStream(backend, k, args) if (use_dist and replicate and deferred) else None
    if rstream is None:
        g.set_deferred_replay(deferred)

    mode = {"replicate": replicate}     # the placement the steps run (both: switched between the two timed regions)

    def set_placement(rep):
        mode["replicate"] = rep
        if both:   # a shard = the whole index under the list mask of this rank
            g.set_list_mask(None if rep else (shard_owner == rank).astype(np.uint8))

    def step(i, stream=True):
        xb = d_q[(i % nbatches) * gnq:(i % nbatches + 1) * gnq]
        if not use_dist:
            g.ivfpq_search_device(xb.data_ptr(), gnq, k, args, d_D.data_ptr(), d_I.data_ptr())
            return d_D, d_I
        if mode["replicate"]:
            if rstream is not None and stream:
                return rstream.submit(xb)      # the PREVIOUS step's results (None for the first)
            if rstream is None:
                return gdist.replicated_search(backend, xb, k, args)
            rstream.flush()
            g.set_deferred_replay(False)       # a call on its own: complete when it returns
            out = gdist.replicated_search(backend, xb, k, args)
            g.set_deferred_replay(True)
            return out
        return gdist.sharded_search(backend, xb, k, args)

    # ---- recall@10 against exact flat search on the GPU (rank 0 data is complete: raw replicated)
    recall = None
    nrq = min(a.recall_queries, gnq)
    if nrq > 0:
        Dg, Ig = step(0, stream=False)
        torch.cuda.synchronize()
        Ig = Ig[:nrq].cpu().numpy()
        fargs = api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30)
        Df, If = g.flat_search(queries[:nrq], k, fargs)
        hits = sum(len(set(Ig[i].tolist()) & set(If[i].tolist())) for i in range(nrq))
        recall = hits / float(nrq * k)
        log("[rank %d] recall@%d = %.4f over %d queries" % (rank, k, recall, nrq))

    # both placements answer step 0: the list-sharded result table (two-phase scan, packed exchange, merge at the owners)
    # must be the replicated one -- labels and distance bits -- so that a scaling run carries its own correctness bit
    sharded_equals_replicated = None
    exchange_per_step = None
    if both:
        set_placement(False)
        backend.exchange_stats = {}
        Ds0, Is0 = step(0, stream=False)
        Ds0, Is0 = Ds0.clone(), Is0.clone()
        es = backend.exchange_stats
        backend.exchange_stats = None
        set_placement(True)
        Dr0, Ir0 = step(0, stream=False)
        torch.cuda.synchronize()
        same = torch.tensor([1 if (torch.equal(Is0, Ir0) and torch.equal(Ds0, Dr0)) else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
        sharded_equals_replicated = bool(int(same.item()) == 1)
        exchange_per_step = {"candidates_sent_by_rank_0_bytes": es.get("exchange_bytes"),
                             "candidates_sent_by_rank_0_entries_per_query": None if not es.get("queries") else round(es["exchange_entries"] / es["queries"], 1),
                             "whole_tables_would_be_bytes": (gnq // world) * R * 12 * (world - 1)}
        log("[rank %d] sharded == replicated on step 0: %s" % (rank, sharded_equals_replicated))
    elif use_dist and not replicate:
        backend.exchange_stats = {}
        step(0, stream=False)
        es = backend.exchange_stats
        backend.exchange_stats = None
        exchange_per_step = {"candidates_sent_by_rank_0_bytes": es.get("exchange_bytes"),
                             "candidates_sent_by_rank_0_entries_per_query": None if not es.get("queries") else round(es["exchange_entries"] / es["queries"], 1),
                             "whole_tables_would_be_bytes": (gnq // world) * R * 12 * (world - 1)}

    # the streamed schedule of the replicated ranks against a plain call on the same batch, once, before anything is
    # timed: a mismatch on ANY rank sends every rank back to the plain schedule (replay at the end of each call)
    if rstream is not None:
        Dp, Ip = step(0, stream=False)
        Dp, Ip = Dp.clone(), Ip.clone()
        rstream.submit(d_q[:gnq])
        Ds, Is = rstream.flush()
        torch.cuda.synchronize()
        same = torch.tensor([1 if (torch.equal(Ds, Dp) and torch.equal(Is, Ip)) else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
        if int(same.item()) != 1:
            log("[rank %d] ReplicatedStream disagrees with replicated_search: falling back to the plain schedule" % rank)
            rstream.close()
            rstream = None
            deferred = False
            g.set_deferred_replay(False)

    # ---- timed region ----
    def timed_region():
        """W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides; max over the ranks"""
        for i in range(a.warmup):
            step(i)
        # stage events inside the timed region (roofline.achieved is the scan kernel's launch duration measured live there).
        # --timed-events scan / none: fewer / no events in the timed region, the rest from an untimed pass over the same
        # batches -- the step time is the same within run-to-run noise (all 1.629, scan 1.642, none 1.635 ms, 60 steps each)
        g.profile_enable({"scan": 2, "all": 1, "none": 0}[a.timed_events])
        g.profile_reset()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + i)
        if rstream is not None and mode["replicate"]:
            rstream.flush()                  # the last step's results: gathered inside the timed region
        t_enq = time.perf_counter() - t0     # host time to enqueue the steps (log only)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_ = time.perf_counter() - t0
        log("[rank %d] %s: host enqueue %.3f ms/step of %.3f ms/step" % (
            rank, "single GPU" if not use_dist else ("replicate" if mode["replicate"] else "shard"), t_enq / a.steps * 1e3,
            dt_ / a.steps * 1e3))
        if world > 1:
            tt = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, g.profile()

    placement_qps = {}
    if both:
        set_placement(False)
        dt_s, prof_s = timed_region()
        set_placement(True)
        dt_r, prof_r = timed_region()
        placement_qps = {"shard": gnq * a.steps / dt_s, "replicate": gnq * a.steps / dt_r}
        if dt_s <= dt_r:          # `value` = the faster placement, named in config.placement
            set_placement(False)
            dt, prof_first = dt_s, prof_s
        else:
            dt, prof_first = dt_r, prof_r
        replicate = mode["replicate"]
    else:
        dt, prof_first = timed_region()
        if use_dist:
            placement_qps = {("replicate" if replicate else "shard"): gnq * a.steps / dt}
    prof = prof_timed = prof_first
    if a.timed_events != "all":
        # every stage + the algorithmic bytes of the same launches: the same steps again, untimed
        g.profile_enable(True)
        g.profile_reset()
        for i in range(a.steps):
            step(a.warmup + i)
        if rstream is not None:
            rstream.flush()
        torch.cuda.synchronize()
        prof = g.profile()
        if prof_timed["scan"][1]:
            prof["scan"] = prof_timed["scan"]     # the launch duration the roofline block uses: from the timed region
    g.profile_enable(False)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        _flush_c_stdio()      # (this rank's RCCL banner: out now, long before rank 0 prints the line)
        return

    qps = gnq * a.steps / dt
    scan_ms, scan_n = prof["scan"]
    bytes_per_launch = prof["scan_bytes"] / max(1, scan_n)
    avg_s = (scan_ms / 1e3) / max(1, scan_n)
    achieved = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else 0.0
    peak = 8000.0
    stages = {n: round(prof[n][0] / max(1, prof[n][1]) * 1e3, 2) for n in
              ("coarse", "tables", "scan", "select", "rerank") if prof[n][1]}
    log("stage avg us per launch:", stages, "scan bytes/launch %.0f" % bytes_per_launch)

    # host-buffer entry point (what the RetrievalModel boundary hands over): H2D queries + D2H
    # results included.  Reported for DESIGN.md; never `value`.
    host_qps = None
    if world == 1:
        qh = queries[:a.nq]
        g.ivfpq_search(qh, k, args)
        t1 = time.perf_counter()
        for _ in range(5):
            g.ivfpq_search(qh, k, args)
        host_qps = 5 * a.nq / (time.perf_counter() - t1)

    # batch 0 through the device's default path, every call complete -- kept for the label-by-label comparison with the CPU
    # baseline (the compiled library at its default BLAS threshold) -- and the plugin leg: both BEFORE the legs below
    # insert into and delete from the index
    gpu_res = None
    if world == 1:
        g.set_exact_ties(True)
        g.set_deferred_replay(False)
        g.ivfpq_search_device(d_q.data_ptr(), a.nq, k, args, d_D.data_ptr(), d_I.data_ptr())
        torch.cuda.synchronize()
        gpu_res = (d_D[:a.nq].cpu().numpy().copy(), d_I[:a.nq].cpu().numpy().copy())
        g.set_exact_ties(not a.no_exact_ties)
        g.set_deferred_replay(deferred)
    # ---- what the RetrievalModel boundary delivers: the HIPIVFPQ plugin driven the way VectorManager drives a model
    #      (gamma_amd/host/harness_c_api.cc): vectors stored, Indexing(), Add() in the engine's batches of 10 000, then
    #      Search() with HOST buffers in and out -- 16384-query and 1024-query calls.  Never `value` (PCIe inclusive).
    plugin_leg = None
    if world == 1 and not a.no_extra and not a.no_plugin:
        try:
            plugin_leg = plugin_bench(a, base, queries, cc, pq, g, d_q, d_D, d_I, k, args)
        except Exception as e:      # noqa: BLE001 -- a leg never fails the bench
            plugin_leg = {"error": str(e)}
        log("plugin leg: %s" % json.dumps(plugin_leg))

    # ---- the other operating points BASELINE.md / SURVEY 8d name, each a short bounded leg on the same index
    #      (single GPU only; none of them is `value`) ----
    extra = {}
    if world == 1 and not a.no_extra:
        def timed(fn, n, warm=3):
            for _ in range(warm):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / n

        # (a) exact ties (the default: the queries whose result a tie can change are replayed through the reference's
        #     heaps, csrc/ties.hip): how many are flagged per batch, and what the step costs with the mode off
        #     (the three timings walk the same query batches as the timed region, no stage events)
        turn = [0]

        def one_step():
            xb = d_q[(turn[0] % nbatches) * gnq:(turn[0] % nbatches + 1) * gnq]
            turn[0] += 1
            g.ivfpq_search_device(xb.data_ptr(), gnq, k, args, d_D.data_ptr(), d_I.data_ptr())

        g.set_exact_ties(True)
        g.set_deferred_replay(True)      # the streamed schedule: a step's flagged rows complete beside the next step
        sec = timed(one_step, 2, 3)
        g.tie_stats(reset=True)
        nst = 4 * nbatches
        turn[0] = 0
        sec = timed(one_step, nst, 0)
        ts = g.tie_stats()
        g.set_deferred_replay(False)     # every call complete on return (the timed region's default)
        turn[0] = 0
        sec_inline = timed(one_step, nst, 3)
        g.set_deferred_replay(deferred)
        g.set_exact_ties(False)
        turn[0] = 0
        sec_off = timed(one_step, nst, 3)
        g.set_exact_ties(True)
        extra["exact_ties"] = {"qps_streamed_deferred_replay": round(gnq / sec, 1), "ms_per_step_streamed": round(sec * 1e3, 4),
                               "qps_call_complete": round(gnq / sec_inline, 1),
                               "qps_with_ties_off": round(gnq / sec_off, 1),
                               "ms_per_step_with_ties_off": round(sec_off * 1e3, 4),
                               "flagged_per_batch": {"coarse_rows_redone": round(ts["coarse_rows"] / nst, 1),
                                                     "recall_num_cut_ties": round(ts["cut_ties"] / nst, 1),
                                                     "queries_replayed": round(ts["replayed"] / nst, 1)},
                               "batch": gnq}
        # (a'') TWO (and three) caller threads on the one handle, every call COMPLETE when it returns to its caller
        #       (gamma_hip_ivfpq_search_device_wait): one caller's tie replay runs beside the other's coarse quantizer, query
        #       tables and scan -- the reference's own calling pattern (re-entrant Search, tests/test.h:1033-1062)
        import threading
        callers = {}
        for T in (1, 2, 3):
            per_t = max(8, 48 // T)
            bufs = [(torch.empty((gnq, k), dtype=torch.float32, device=dev), torch.empty((gnq, k), dtype=torch.int64, device=dev)) for _ in range(T)]

            def caller(t, n):
                for i in range(n):
                    xb = d_q[((t + i) % nbatches) * gnq:((t + i) % nbatches + 1) * gnq]
                    g.ivfpq_search_device_wait(xb.data_ptr(), gnq, k, args, bufs[t][0].data_ptr(), bufs[t][1].data_ptr())
            for t in range(T):
                caller(t, 2)
            th = [threading.Thread(target=caller, args=(t, per_t)) for t in range(T)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for x_ in th:
                x_.start()
            for x_ in th:
                x_.join()
            el = time.perf_counter() - t0
            callers[str(T)] = {"qps": round(T * per_t * gnq / el, 1), "ms_per_call_per_caller": round(el / per_t * 1e3, 4)}
        # the last call of caller 0 against the plain call on the same batch
        xb = d_q[((0 + per_t - 1) % nbatches) * gnq:((0 + per_t - 1) % nbatches + 1) * gnq]
        g.ivfpq_search_device(xb.data_ptr(), gnq, k, args, d_D.data_ptr(), d_I.data_ptr())
        torch.cuda.synchronize()
        callers["identical_to_the_plain_call"] = bool(torch.equal(bufs[0][1], d_I) and torch.equal(bufs[0][0], d_D))
        extra["exact_ties"]["caller_threads_each_call_complete_on_return"] = callers
        # (b) batch sizes of the BASELINE.md protocol: one Search call of nq queries, device buffers
        byb = {}
        for nqb in [int(v) for v in a.batches.split(",")]:
            sec = timed(lambda: g.ivfpq_search_device(d_q.data_ptr(), nqb, k, args, d_D.data_ptr(), d_I.data_ptr()),
                        200 if nqb < 1024 else 100, 10)
            byb[str(nqb)] = {"qps": round(nqb / sec, 1), "us_per_call": round(sec * 1e6, 1)}
        extra["qps_by_batch"] = byb
        # (b') the same single query through the host-buffer entry point (what the plugin calls): pageable numpy
        #      buffers in and out, the call returns when the results are there
        hq = np.ascontiguousarray(queries[:1])
        ts = []
        for i in range(220):
            t0 = time.perf_counter()
            g.ivfpq_search(hq, k, args)
            ts.append(time.perf_counter() - t0)
        extra["single_query_host_call_us"] = round(float(np.median(ts[20:])) * 1e6, 1)
        # (c) the coarse path that is bit-identical to the compiled reference at every batch size (exact
        #     fvec_L2sqr per pair instead of the GEMM form; faiss itself switches to sgemm at 20 queries)
        args0 = api.SearchArgs(metric=api.METRIC_L2, nprobe=a.nprobe, recall_num=a.recall_num,
                               has_rank=not a.no_rank, min_score=0.0, max_score=1e30, coarse_mode=0)
        sec = timed(lambda: g.ivfpq_search_device(d_q.data_ptr(), gnq, k, args0, d_D.data_ptr(), d_I.data_ptr()), 10, 3)
        extra["exact_ties_coarse_mode_0"] = {"qps": round(gnq / sec, 1), "ms_per_step": round(sec * 1e3, 4),
                                             "note": "exact ties on + the coarse path that is bit-identical to compiled faiss"}
        g.set_exact_ties(not a.no_exact_ties)
        # (d) C2: flat L2 over the same 1M x 128 raw vectors, 1024 queries per call, k = 100, exact
        #     fvec_L2sqr operation order (1 sub + 1 fma per element pair = 3 flops): bound by the fp32 vector rate
        if N * d * 4 <= (2 << 30):
            fk, fnq = 100, 1024
            fD = torch.empty((fnq, fk), dtype=torch.float32, device=dev)
            fI = torch.empty((fnq, fk), dtype=torch.int64, device=dev)
            fargs = api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30)
            sec = timed(lambda: g.flat_search_device(d_q.data_ptr(), fnq, fk, fargs, fD.data_ptr(), fI.data_ptr()), 5, 2)
            g.set_exact_ties(False)
            sec_off = timed(lambda: g.flat_search_device(d_q.data_ptr(), fnq, fk, fargs, fD.data_ptr(), fI.data_ptr()), 5, 2)
            g.set_exact_ties(not a.no_exact_ties)
            # two caller threads, every call complete on return (gamma_hip_flat_search_device_wait: one caller's heap replay
            # beside the other's filter passes)
            c2_callers = {}
            for T in (1, 2):
                fb = [(torch.empty((fnq, fk), dtype=torch.float32, device=dev), torch.empty((fnq, fk), dtype=torch.int64, device=dev)) for _ in range(T)]

                def fcaller(t, n):
                    for i in range(n):
                        xq_ = d_q[((t + i) % 8) * fnq:((t + i) % 8 + 1) * fnq]
                        g.flat_search_device_wait(xq_.data_ptr(), fnq, fk, fargs, fb[t][0].data_ptr(), fb[t][1].data_ptr())
                for t in range(T):
                    fcaller(t, 2)
                th_ = [threading.Thread(target=fcaller, args=(t, 16 // T)) for t in range(T)]
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for x_ in th_:
                    x_.start()
                for x_ in th_:
                    x_.join()
                el_ = time.perf_counter() - t0_
                c2_callers[str(T)] = {"ms_per_call": round(el_ / 16 * 1e3, 3), "qps": round(16 * fnq / el_, 1)}
            xq_ = d_q[((0 + 16 // 2 - 1) % 8) * fnq:((0 + 16 // 2 - 1) % 8 + 1) * fnq]
            g.flat_search_device(xq_.data_ptr(), fnq, fk, fargs, fD.data_ptr(), fI.data_ptr())
            torch.cuda.synchronize()
            c2_callers["identical_to_the_plain_call"] = bool(torch.equal(fb[0][1], fI) and torch.equal(fb[0][0], fD))
            extra["c2_flat_callers"] = c2_callers
            extra["c2_flat"] = {"workload": "C2: flat L2, %dx%d, %d queries/call, k=%d" % (N, d, fnq, fk),
                                "ms_per_call": round(sec * 1e3, 3), "qps": round(fnq / sec, 1),
                                "ms_per_call_with_ties_off": round(sec_off * 1e3, 3),
                                "how": "first 16384 rows exactly for every query; the rest through a bf16 hi/lo filter on the matrix "
                                       "pipe (3 products, proven margin, csrc/flat_mfma.hip) + the reference's exact arithmetic for "
                                       "the ~k survivors per query and pass; queries with equal distances among their k + 1 best "
                                       "replayed through the reference's heap",
                                # SURVEY 8d counts the GEMM form's 2 nq N d flops against the fp32 matrix / vector peak
                                "roofline_gemm_form_flops": {"bound": "mfma", "achieved": round(2.0 * fnq * N * d / sec / 1e12, 2),
                                                             "peak": 157.3, "unit": "TFLOP/s",
                                                             "frac": round(2.0 * fnq * N * d / sec / 1e12 / 157.3, 4)},
                                # the filter kernel's own work: 3 bf16 products per element pair of the rows behind the
                                # first chunk, against the dense bf16 MFMA peak; its launch times are in profiles/ (rocprofv3)
                                "filter_bf16_flops_per_call": 6.0 * fnq * max(0, N - 16384) * d}

        # (d') C5's flat fallback / HIPFLAT on embedding-shaped rows: inner product over 1 M x 768 unit vectors, 1024 queries,
        #      k = 100 (GammaFLATIndex::Search is dimension-agnostic, gamma_index_flat.cc:183-291): the long-row variant of the
        #      matrix-pipe filter (flat_mfma.hip, k_flat_filter_big: a block of 32 queries' bf16 hi / lo image in LDS, rows streamed)
        try:
            fN, fd, fk, fnq = 1000000, 768, 100, 1024
            gf5 = api.GammaHip(local_rank)
            gf5.raw_init(fd)
            for c0 in range(0, fN, 250000):
                gf5.raw_append(synth.embedding_like_device(250000, d=fd, seed=1234, start=c0, device=dev).cpu().numpy())
            fq5 = synth.embedding_like_device(fnq, d=fd, seed=4321, device=dev)
            fD5 = torch.empty((fnq, fk), dtype=torch.float32, device=dev)
            fI5 = torch.empty((fnq, fk), dtype=torch.int64, device=dev)
            fa5 = api.SearchArgs(metric=api.METRIC_IP, min_score=-1e30, max_score=1e30)
            sec5 = timed(lambda: gf5.flat_search_device(fq5.data_ptr(), fnq, fk, fa5, fD5.data_ptr(), fI5.data_ptr()), 3, 1)
            extra["c5_flat"] = {"workload": "C5 flat: inner product, %dx%d unit-norm embedding-shaped rows, %d queries/call, k=%d" % (fN, fd, fnq, fk),
                                "ms_per_call": round(sec5 * 1e3, 3), "qps": round(fnq / sec5, 1),
                                "roofline_gemm_form_flops": {"bound": "mfma", "achieved": round(2.0 * fnq * fN * fd / sec5 / 1e12, 2),
                                                             "peak": 157.3, "unit": "TFLOP/s",
                                                             "frac": round(2.0 * fnq * fN * fd / sec5 / 1e12 / 157.3, 4)},
                                "filter_bf16_flops_per_call": 6.0 * fnq * max(0, fN - 1024) * fd,
                                "exact_vector_kernel_ms_per_call_round4": "~1200 (k_pairwise_generic at ~1.3 TFLOP/s, profiles/r04_c5_shape_2m_kernel_stats.txt)"}
            gf5.close()
        except Exception as e:      # noqa: BLE001 -- a leg never fails the bench
            extra["c5_flat"] = {"error": str(e)}

        # (d2) the IVFFLAT model (f4) on the same vectors, centroids and nprobe: exact distances of every entry of the
        #      probed lists, rows gathered from the raw store
        if N * d * 4 <= (2 << 30):
            gf = api.GammaHip(local_rank)
            gf.ivfflat_init(d, nlist, api.METRIC_L2, bucket_init_size=max(1000, int(2.5 * N / nlist)))
            gf.ivfflat_set_trained(cc)
            gf.raw_init(d)
            for i0 in range(0, N, 1 << 16):
                gf.raw_append(base[i0:i0 + (1 << 16)])
                gf.add(base[i0:i0 + (1 << 16)], i0)
            vnq = 4096
            vargs = api.SearchArgs(metric=api.METRIC_L2, nprobe=a.nprobe, min_score=0.0, max_score=1e30)
            sec = timed(lambda: gf.ivfflat_search_device(d_q.data_ptr(), vnq, k, vargs, d_D.data_ptr(), d_I.data_ptr()), 5, 2)
            extra["ivfflat"] = {"workload": "IVFFLAT nlist=%d nprobe=%d, %dx%d, %d queries/call, k=%d" % (nlist, a.nprobe, N, d, vnq, k),
                                "ms_per_call": round(sec * 1e3, 3), "qps": round(vnq / sec, 1)}
            gf.close()

        # (e) C5's other half: realtime inserts at 10 k vectors/s WHILE searching (writers run on their own stream
        #     and publish versioned list tables; tests/test_gpu_concurrent.py checks that every search sees a prefix)
        import threading
        ins = synth.sift_like(20000, d=d, seed=777)
        sstream = torch.cuda.ExternalStream(g.stream(), device=dev)
        done = {"t": None}

        def writer():
            t1 = time.perf_counter()
            for b in range(20):
                lo = b * 1000
                g.raw_append(ins[lo:lo + 1000])
                g.add(ins[lo:lo + 1000], N + lo)
                pause = t1 + (b + 1) * 0.1 - time.perf_counter()    # 1000 vectors every 0.1 s
                if pause > 0:
                    time.sleep(pause)
            done["t"] = time.perf_counter() - t1

        wt = threading.Thread(target=writer)
        nsteps, t1 = 0, time.perf_counter()
        wt.start()
        while wt.is_alive():
            g.ivfpq_search_device(d_q.data_ptr(), gnq, k, args, d_D.data_ptr(), d_I.data_ptr())
            sstream.synchronize()
            nsteps += 1
        el = time.perf_counter() - t1
        wt.join()
        extra["search_during_inserts"] = {"qps": round(gnq * nsteps / el, 1), "insert_rate_vectors_per_s": round(20000 / done["t"], 1),
                                          "inserted": 20000, "batch": gnq, "steps": nsteps}

    # (g) the other BASELINE shapes at a size that builds in seconds, so that the driver's line carries a timing of
    #     them too (full size -- 100 M x 128 and 10 M x 768 -- takes minutes to generate: tools/c4_scale.py 1e8,
    #     tools/c5_scale.py 1e7 16384).  Child processes: each tool builds its own index.
    if world == 1 and not a.no_extra and not a.no_shapes:
        import re
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))

        def run_tool(argv, seconds):
            try:
                r = subprocess.run([sys.executable] + argv, cwd=here, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                   timeout=seconds, universal_newlines=True, env=dict(os.environ, C4_FILTER="1"))
                return r.stdout
            except Exception as e:   # a shape leg never fails the bench
                return "failed: %s" % e

        def first(pattern, text, cast=float):
            m = re.search(pattern, text)
            return cast(m.group(1)) if m else None

        t0 = time.time()
        out = run_tool([os.path.join("tools", "c4_scale.py"), "8e6"], 150)
        extra["c4_shape_8m"] = {
            "workload": "C4 shape at 8 M vectors: 8000000x128, nlist 16384, M 32, nprobe 64, recall_num 100, 8192 queries/call",
            "qps": first(r"= (\d+) queries/s", out), "ms_per_call": first(r"search: ([0-9.]+) ms", out),
            "scan_gb_per_call": first(r"scan GB/step ([0-9.]+)", out), "recall_at_10_vs_flat": first(r"recall@10 vs flat on 64 queries: ([0-9.]+)", out),
            "qps_with_10pct_filter": first(r"lists compacted per call: [0-9.]+ ms per \d+ queries = (\d+) queries/s", out),
            "roofline": {"bound": "hbm", "kernel": "k_ivfpq_scan_pair (MT 32)", "unit": "GB/s", "peak": 8000.0,
                         "achieved": (first(r"-> ([0-9.]+) TB/s = [0-9.]+ of 8 TB/s", out) or 0) * 1000.0,
                         "frac": first(r"TB/s = ([0-9.]+) of 8 TB/s", out)},
            "recall_num_sweep": re.findall(r"recall_num (\d+): recall@10 ([0-9.]+), ([0-9.]+) ms per \d+ queries = (\d+) queries/s", out),
            "single_query_us": first(r"latency nq=1\s+small-batch chain median ([0-9.]+)", out), "seconds": round(time.time() - t0, 1)}
        t0 = time.time()
        out = run_tool([os.path.join("tools", "c5_scale.py"), "2e6"], 150)
        extra["c5_shape_2m"] = {
            "workload": "C5 shape at 2 M vectors: 2000000x768 inner product, nlist 4096, M 64, nprobe 64, recall_num 100, 4096 queries/call",
            "qps": first(r"no filter: [0-9.]+ ms per \d+ queries = (\d+) queries/s", out),
            "qps_with_10pct_range_filter": first(r"10% range filter: [0-9.]+ ms per \d+ queries = (\d+) queries/s", out),
            "scan_tb_per_s": first(r"-> ([0-9.]+) TB/s", out),
            "roofline": {"bound": "hbm", "kernel": "k_ivfpq_scan_pair (MT 64, inner product)", "unit": "GB/s", "peak": 8000.0,
                         "achieved": (first(r"-> ([0-9.]+) TB/s", out) or 0) * 1000.0,
                         "frac": round((first(r"-> ([0-9.]+) TB/s", out) or 0) / 8.0, 4)},
            "single_query_us": first(r"latency nq=1\s+small-batch chain median ([0-9.]+)", out), "seconds": round(time.time() - t0, 1)}
        log("shape legs: %s" % json.dumps({k2: extra[k2] for k2 in ("c4_shape_8m", "c5_shape_2m")}))

    cpu = None
    labels_vs_cpu = None
    if world == 1 and a.cpu_seconds > 0:
        # (gpu_res: batch 0 through the device's default path, taken before the legs that change the index)
        cpu = cpu_baseline(a, base, queries, cc, pq, g, lno, codes, list_sizes, gpu_res)
        labels_vs_cpu = cpu.pop("labels_equal_to_cpu_baseline", None) if cpu else None

    # (f') the same index with 5 % of its documents deleted (the LAST leg: it changes the index): large batches run over lists
    #      cut down to the live entries once per call (csrc/kernels.hip k_compact_lists) instead of testing the delete
    #      bitmap for every scored code
    if world == 1 and not a.no_extra:
        rng_d = np.random.default_rng(5)
        dead = rng_d.choice(N, N // 20, replace=False)
        bmd = np.zeros((N >> 3) + 1, dtype=np.uint8)
        np.bitwise_or.at(bmd, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
        g.bitmap_upload(bmd, N)
        g.delete(dead)
        sec = timed(lambda: g.ivfpq_search_device(d_q.data_ptr(), gnq, k, args, d_D.data_ptr(), d_I.data_ptr()), 10, 3)
        extra["deleted_5pct"] = {"qps": round(gnq / sec, 1), "ms_per_step": round(sec * 1e3, 4), "batch": gnq}

    # HBM-side traffic of the scan kernel cannot be read from inside the process: it comes from
    # the separate rocprofv3 --pmc passes (tools/pmc_traffic.sh) committed under profiles/, and is
    # only reported when that measurement was taken on this exact workload
    traffic = a.pmc_traffic
    traffic_source = "--pmc-traffic argument" if traffic is not None else None
    if traffic is None and world == 1:
        try:
            pj = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles",
                                             "pmc_traffic_scan.json")))
            wl = pj["workload"]
            if (wl["n"], wl["d"], wl["nlist"], wl["m"], wl["nprobe"], wl["nq"], wl["recall_num"]) == (
                    N, d, nlist, M, a.nprobe, a.nq, a.recall_num):
                traffic = pj["traffic_bytes_per_launch"]
                traffic_source = pj.get("source", "profiles/pmc_traffic_scan.json") + " (separate rocprofv3 --pmc passes of this workload; not measured in this run)"
        except (OSError, KeyError, ValueError):
            traffic = None

    out = {
        "metric": "queries/sec @ recall@10>=0.95, IVFPQ nlist=4096 nprobe=32",
        "value": round(qps, 1),
        "unit": "queries/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "C3: IVFPQ nlist=%d m=%d nbits=8, %dx%d SIFT1M-shaped synthetic, nprobe=%d, "
                        "recall_num=%d, has_rank=%s, k=%d, L2, batch=%d queries/step (%d per GPU)" % (
                            nlist, M, N, d, a.nprobe, a.recall_num, str(not a.no_rank).lower(), k, gnq, a.nq),
            "recall_at_10": None if recall is None else round(recall, 4),
            "placement": None if not use_dist else ("replicate" if replicate else "shard"),
            "sharded_equals_replicated": sharded_equals_replicated,
            "communicator": None if not use_dist else {"backend": a.backend + (" (RCCL)" if a.backend == "nccl" else ""), "ranks": dist.get_world_size(),
                                                       "devices": 1 if a.one_gpu else world},
            "exchange_per_step": exchange_per_step,
            "sharded_qps": None if "shard" not in placement_qps else round(placement_qps["shard"], 1),
            "replicated_qps": None if "replicate" not in placement_qps else round(placement_qps["replicate"], 1),
            "parallelism": "single GPU" if world == 1 else (
                ("query-parallel x%d over REPLICATED lists (%.0f MB of lists per GPU: --placement %s), every rank answers "
                 "its slice of the batch; RCCL all-gather of the top-k; the list-sharded placement timed in the same run: "
                 "config.sharded_qps" % (world, N * (M + 12) / 1e6, a.placement))
                if replicate else
                ("IVF lists sharded x%d (greedy by size), queries sliced x%d; RCCL all-gather of "
                 "the coarse assignment, all-to-all of per-shard top-recall_num, all-gather of "
                 "top-k" % (world, world))),
            "stage_us": stages,
            "stage_timing": "HIP events in the timed region: " + a.timed_events,
            "value_is": "device-pointer Search calls, each COMPLETE when it returns (exact ties on, replay inside the call)" if not deferred else
                        "a STREAM of device-pointer calls with the deferred tie replay (--deferred-replay): flagged rows complete one call later",
            "indexing_seconds": None if train_s is None else round(train_s, 2),
            "tie_replay": ("exact ties on; the replay of a step's flagged queries runs on a side stream beside the next "
                           "step's coarse quantizer and query tables (gamma_hip_set_deferred_replay); every step's results "
                           "are complete inside the timed region (it ends with a device-wide synchronize)"
                           + ("; replicated ranks: a step's results are all-gathered one step behind, the last step's before "
                              "the timed region ends (gamma_amd.dist.ReplicatedStream)" if rstream is not None else "")) if deferred else
                          ("exact ties on; replay at the end of every call" if not a.no_exact_ties else "exact ties off"),
            "pcie_inclusive_qps": None if host_qps is None else round(host_qps, 1),
            "labels_equal_to_cpu_baseline": labels_vs_cpu,
            "plugin": plugin_leg,
            # T closed-loop client threads x ONE query per Search call -- the HIPIVFPQ plugin (host buffers,
            # through RetrievalModel::Search) beside the compiled faiss 1.7.1 + re-rank under the same T on this box's host cores
            "closed_loop": {"pattern": "T client threads x 1 query per call (tools/perf.cc:364-395)",
                            "plugin_HIPIVFPQ": (plugin_leg or {}).get("closed_loop_threads_x_1_query") if isinstance(plugin_leg, dict) else None,
                            "cpu_reference_faiss": (cpu or {}).get("closed_loop_threads_x_1_query") if isinstance(cpu, dict) else None},
            **extra,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "k_ivfpq_scan_pair_c8",
            "achieved": round(achieved, 1),
            "peak": peak,
            "unit": "GB/s",
            "frac": round(achieved / peak, 4),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": round(bytes_per_launch),
            "avg_launch_us": round(avg_s * 1e6, 2),
        },
        "cpu_baseline": cpu,
    }
    # the JSON line is the LAST thing on stdout: RCCL writes a version banner to the C stdout of every process that
    # created a communicator, buffered until something flushes it
    if use_dist:
        dist.destroy_process_group()
    _flush_c_stdio()
    print(json.dumps(out), flush=True)


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:     # noqa: BLE001
        pass
    sys.stdout.flush()


def spawn_ranks(n):
    """Start n copies of this script as ranks 0..n-1 of one job (what `torch.distributed.run --nnodes=1
    --nproc-per-node n` would do) and return the worst exit code.  Children inherit stdout / stderr."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while rc == 0 and any(p.poll() is None for p in procs):
            time.sleep(0.2)
            rc = max(abs(p.returncode) for p in procs if p.returncode is not None) if any(
                p.returncode is not None for p in procs) else 0
        rc = max([rc] + [abs(p.returncode) for p in procs if p.returncode is not None])
    finally:
        for p in procs:                 # a rank died: the others would wait in a collective forever
            if p.poll() is None:
                p.kill()
    return rc


def _closed_loop_threads():
    """T of the closed-loop legs: 1, 8 and as many client threads as this process has cores (CPU quota respected)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if qv != "max":
            n = min(n, max(1, int(round(float(qv) / float(pv)))))
    except Exception:
        pass
    return sorted({1, 8, max(1, n)})


def plugin_bench(a, base, queries, cc, pq, g, d_q, d_D, d_I, k, args):
    """HIPIVFPQ behind the reference's plugin interface (index/retrieval_model.h:218-310; vector/vector_manager.cc:161-349,
    433-617): Init -> (store) -> Indexing -> Add in batches of 10 000 -> Search with the caller's host buffers."""
    import torch
    from gamma_amd import plugin
    N, d = base.shape
    params = ('{"ncentroids": %d, "nsubvector": %d, "nprobe": %d, "metric_type": "L2", "bucket_init_size": %d}'
              % (a.nlist, a.m, a.nprobe, max(1000, int(2.5 * N / a.nlist))))
    m = plugin.PluginModel("HIPIVFPQ", d, params, indexing_size=min(N, a.nlist * 64))
    out = {}
    try:
        t0 = time.time()
        m.store(base)
        out["store_seconds"] = round(time.time() - t0, 2)
        t0 = time.time()
        if m.indexing() != 0:
            raise RuntimeError("Indexing() failed")
        out["indexing_seconds"] = round(time.time() - t0, 2)      # the reference: 11.3 s on 8 cores
        st = m.trained_state(a.nlist, a.m)
        out["same_trained_state_as_the_handle"] = bool(st is not None and st[0].tobytes() == cc.tobytes() and st[1].tobytes() == pq.tobytes())
        t0 = time.time()
        for i0 in range(0, N, 10000):
            if not m.add(base[i0:i0 + 10000]):
                raise RuntimeError("Add() failed")
        out["add_vectors_per_s"] = round(N / (time.time() - t0), 1)
        rp = '{"nprobe": %d, "recall_num": %d, "metric_type": "L2"}' % (a.nprobe, a.recall_num)
        for nqb, reps in ((a.nq, 20), (1024, 60)):
            qh = np.ascontiguousarray(queries[:nqb])
            Dp, Ip = m.search(qh, k, rp, has_rank=not a.no_rank, min_score=0.0, max_score=1e30)
            g.ivfpq_search_device(d_q.data_ptr(), nqb, k, args, d_D.data_ptr(), d_I.data_ptr())
            torch.cuda.synchronize()
            same = bool(np.array_equal(Ip, d_I[:nqb].cpu().numpy()) and Dp.tobytes() == d_D[:nqb].cpu().numpy().tobytes())
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                m.search(qh, k, rp, has_rank=not a.no_rank, min_score=0.0, max_score=1e30)
                ts.append(time.perf_counter() - t0)
            # median: the host side of a call is CPU work (8 MB in, 2 MB out), and this process runs under a CPU quota --
            # a throttled period inside a short loop would otherwise be the figure
            sec = float(np.median(ts))
            out["search_%d" % nqb] = {"qps": round(nqb / sec, 1), "ms_per_call": round(sec * 1e3, 4),
                                      "ms_per_call_mean": round(float(np.mean(ts)) * 1e3, 4),
                                      "identical_to_the_device_pointer_call": same}
        # the reference's own load pattern (tools/perf.cc:364-395): T closed-loop client threads, ONE query
        # per Search call, through the plugin boundary (C++ threads in the harness: no GIL).  Concurrent small calls share
        # device batches (gamma_hip_combine.cpp).
        cl = {}
        pool = np.ascontiguousarray(queries[:4096])
        for T in _closed_loop_threads():
            calls = max(150, 6000 // T)
            dtc, lat = m.concurrent_clients(pool, rp, T, calls, k=k, has_rank=not a.no_rank)
            lat = np.sort(lat)
            cl[str(T)] = {"qps": round(T * calls / dtc, 1), "p50_us": round(float(np.median(lat)), 1),
                          "p99_us": round(float(lat[int(0.99 * (len(lat) - 1))]), 1), "calls": int(len(lat))}
        out["closed_loop_threads_x_1_query"] = cl
        # T client threads x one LARGE call each (the batch of `value`, host buffers in and out, every call complete on
        # return): the queries of one caller go up and its results come down beside the other caller's kernels, and one
        # caller's tie replay runs beside the other's coarse quantizer / tables (ivfpq_search_host_overlap)
        big = {}
        poolb = np.ascontiguousarray(queries)
        for T in (1, 2, 3):
            calls = max(10, 48 // T)
            m.concurrent_clients(poolb, rp, T, 3, nq_call=a.nq, k=k, has_rank=not a.no_rank)   # warm-up (staging slots)
            dtc, lat = m.concurrent_clients(poolb, rp, T, calls, nq_call=a.nq, k=k, has_rank=not a.no_rank)
            big[str(T)] = {"qps": round(T * calls * a.nq / dtc, 1), "ms_per_call_p50": round(float(np.median(lat)) / 1e3, 4)}
        out["client_threads_x_%d_queries" % a.nq] = big
    finally:
        m.close()
    return out


def cpu_baseline(a, base, queries, cc, pq, g, lno, codes, list_sizes, gpu_res=None):
    """The reference's CPU path on the host cores of this box, on a bounded sample of the same workload.

    kind "reference" (when oracle/_ref is there -- it is built from /root/reference and travels with the
    snapshot): the REAL faiss 1.7.1 the reference links, configured the way GammaIVFPQIndex::Init configures it,
    trained and filled by faiss itself on the same vectors, searched through ref_driver.cpp's
    ref_ivfpq_search_rerank (IndexIVFPQ::search with k = recall_num -- MKL sgemm coarse quantizer at this batch
    size, OpenMP over queries -- plus compute_dis with the library's fvec_L2sqr and heaps).
    kind "port" otherwise: the CPU restatement (oracle/gamma_oracle.c).  The port's own figures, for both coarse
    forms, are reported next to it either way."""
    from oracle import binding as B
    budget = max(2.0, a.cpu_seconds)
    nb = queries.shape[0] // a.nq
    cores = B.lib().go_num_threads()
    # a container CPU quota (cgroup v2 cpu.max "quota period"): with more OpenMP threads than the quota carries, every
    # thread runs for a fraction of each period and is then throttled.  The baseline is then ALSO timed with as many
    # threads as the quota is worth, and the better of the two is reported, with the quota next to it.
    quota_cores = None
    try:
        qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if qv != "max":
            quota_cores = max(1, int(round(float(qv) / float(pv))))
    except Exception:
        pass

    def set_threads(n):
        try:
            import ctypes
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
            return True
        except Exception:
            return False

    def timed(fn, seconds):
        fn(0)   # warm-up
        done, i, t0 = 0, 0, time.perf_counter()
        while True:
            fn(i % nb)
            done += a.nq
            i += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                return done / el, i, el

    t0 = time.time()
    o = B.OracleIVFPQ(a.d, a.nlist, a.m, 8, B.METRIC_L2,
                      bucket_init_size=max(1000, int(list_sizes.max()) + 1))
    o.set_trained(cc, pq, g.ivfpq_table())
    order = np.argsort(lno, kind="stable")
    starts = np.concatenate([[0], np.cumsum(list_sizes)])
    for l in range(a.nlist):
        if list_sizes[l]:
            sl = order[starts[l]:starts[l + 1]]
            o.add_keys(l, sl.astype(np.int64), codes[sl])
    o.set_raw(base)
    ctx = B.make_ctx(min_score=0.0, max_score=1e30)
    log("cpu baseline: oracle index built in %.1fs, %d threads" % (time.time() - t0, cores))
    all_threads = cores
    if quota_cores and quota_cores < cores and set_threads(quota_cores):
        cores = quota_cores   # the port's legs and the first reference leg run with what the quota carries
    port = {}
    for mode, name in ((1, "gemm_form_coarse"), (0, "exact_coarse")):
        qps, n, el = timed(lambda b: o.search(queries[b * a.nq:(b + 1) * a.nq], a.k, a.nprobe, recall_num=a.recall_num,
                                               has_rank=not a.no_rank, metric=B.METRIC_L2, ctx=ctx, coarse_mode=mode),
                           budget / 3)
        port[name] = {"value": round(qps, 1), "per_thread": round(qps / cores, 1),
                      "sample": "%d Search calls of %d queries in %.1fs" % (n, a.nq, el)}
    out = None
    if B.have_ref() and hasattr(B.ref(), "ref_ivfpq_search_rerank") and not a.no_rank:
        try:
            t0 = time.time()
            r = B.RefIVFPQ(a.d, a.nlist, a.m, 8, B.METRIC_L2)
            r.train(base[:min(len(base), a.nlist * 64)])
            r.add(base)
            log("cpu baseline: faiss trained + filled in %.1fs" % (time.time() - t0))
            fn = lambda b: r.search_rerank(queries[b * a.nq:(b + 1) * a.nq], a.k, a.recall_num, a.nprobe, base)
            agree = None
            if gpu_res is not None:
                # The library trained itself on the same vectors -- and the device's training is the library's bit for
                # bit -- so the two indexes are the same index, and the library's answer to batch 0 (MKL coarse path at
                # this batch size) can be compared with the device's label by label.  (ref_ivfpq_search_rerank feeds the
                # k-heap from the SORTED recall table, Gamma from the heap's array: inside exact ties the order can
                # differ, hence the set figure beside the strict one; the oracle run on the library's own coarse
                # assignment is Gamma's order exactly.)
                same_index = (r.coarse_centroids().tobytes() == np.ascontiguousarray(cc, np.float32).tobytes() and
                              r.pq_centroids().tobytes() == np.ascontiguousarray(pq, np.float32).tobytes())
                Dc, Ic = fn(0)
                Dg, Ig = gpu_res
                strict = float((Ic == Ig).all(axis=1).mean())
                sets = float(np.mean([set(x_.tolist()) == set(y_.tolist()) for x_, y_ in zip(Ic, Ig)]))
                cdl, cil = r.coarse(queries[:a.nq], a.nprobe)
                Do, Io = o.search(queries[:a.nq], a.k, a.nprobe, recall_num=a.recall_num, has_rank=True, metric=B.METRIC_L2,
                                  ctx=ctx, preassigned=(cdl, cil))
                agree = {"queries": int(a.nq), "same_trained_state_as_the_library": bool(same_index),
                         "identical_to_faiss_search_rerank": round(strict, 6), "identical_as_sets": round(sets, 6),
                         "identical_to_gamma_order_on_the_librarys_coarse_assignment": round(float((Io == Ig).all(axis=1).mean()), 6),
                         "distance_bits_identical": round(float((Do.view(np.uint32) == Dg.view(np.uint32)).all(axis=1).mean()), 6)}
                log("cpu baseline: label agreement with the device on batch 0: %s" % json.dumps(agree))
            # the same library under the reference's closed-loop pattern (T client threads x 1 query per call: ctypes drops the
            # GIL for the call; OpenMP to one thread per call -- a one-query call has one iteration to hand out) and at the
            # 1024-query batch of BASELINE.md 3
            closed = {}
            try:
                import threading
                set_threads(1)
                for T in _closed_loop_threads():
                    lats = [[] for _ in range(T)]
                    stop_at = time.perf_counter() + 1.5

                    def client(t):
                        set_threads(1)   # (the OpenMP thread-count setting is per calling thread: a new thread starts from the default)
                        i = t * 977
                        while time.perf_counter() < stop_at:
                            t1 = time.perf_counter()
                            r.search_rerank(queries[i % a.nq:i % a.nq + 1], a.k, a.recall_num, a.nprobe, base)
                            lats[t].append(time.perf_counter() - t1)
                            i += 1
                    th = [threading.Thread(target=client, args=(t,)) for t in range(T)]
                    t1 = time.perf_counter()
                    for x_ in th:
                        x_.start()
                    for x_ in th:
                        x_.join()
                    el_c = time.perf_counter() - t1
                    lat = np.sort(np.concatenate([np.asarray(v) for v in lats])) * 1e6
                    closed[str(T)] = {"qps": round(len(lat) / el_c, 1), "p50_us": round(float(np.median(lat)), 1),
                                      "p99_us": round(float(lat[int(0.99 * (len(lat) - 1))]), 1), "calls": int(len(lat))}
            finally:
                set_threads(cores)
            q1024 = timed(lambda b: r.search_rerank(queries[(b * 1024) % (a.nq - 1024):(b * 1024) % (a.nq - 1024) + 1024], a.k,
                                                    a.recall_num, a.nprobe, base), 1.5)
            nq1024 = {"qps": round(q1024[0] * 1024.0 / a.nq, 1), "calls": q1024[1]}
            qps, n, el = timed(fn, budget / 3)
            used = cores
            alt = None
            if cores != all_threads and set_threads(all_threads):   # and with every hardware thread, throttled by the quota
                q2, n2, el2 = timed(fn, budget / 3)
                alt = {"threads": all_threads, "value": round(q2, 1)}
                set_threads(cores)
                if q2 > qps:
                    alt = {"threads": cores, "value": round(qps, 1)}
                    qps, n, el, used = q2, n2, el2, all_threads
            out = {"value": round(qps, 1), "unit": "queries/s", "cores": used, "kind": "reference",
                   "per_thread": round(qps / used, 1),
                   "sample": "faiss 1.7.1 (oracle/_ref) IndexIVFPQ::search k=%d + compute_dis re-rank, its own training "
                             "on the same vectors: %d Search calls of %d queries in %.1fs" % (a.recall_num, n, a.nq, el),
                   "port": port}
            if quota_cores:
                out["cpu_quota_cores"] = quota_cores
                out["sample"] += "; the container's CPU quota is %d cores (cgroup cpu.max)" % quota_cores
            if alt:
                out["other_thread_count"] = alt
            if agree:
                out["labels_equal_to_cpu_baseline"] = agree
            out["closed_loop_threads_x_1_query"] = closed
            out["search_1024"] = nq1024
        except Exception as e:   # a prebuilt _ref that does not load here: fall back to the port
            log("cpu baseline: reference library unusable (%s), using the port" % e)
    if out is None:
        best = max(port.values(), key=lambda v: v["value"])
        out = {"value": best["value"], "unit": "queries/s", "cores": cores, "kind": "port",
               "per_thread": best["per_thread"], "sample": best["sample"] + " (the faster of the two coarse forms)",
               "port": port}
    return out


if __name__ == "__main__":
    main()
