"""GPU: search concurrent with realtime inserts (BASELINE configs[4]: 768-d inner product, inserts at 10 k
vectors/s DURING search).  The reference's lists are lock-free for readers: a writer appends, then publishes the
new length (realtime/realtime_mem_data.cc:279-300).  Here writers run on their own stream and publish a new
VERSION of the lists' (offset, length) tables; a search reads the version current when it was enqueued.  Checked:
every Search call made while the writer runs equals the oracle at SOME prefix of the insert log -- never a torn
state -- prefixes never go backwards within a client thread, and the writer sustains the rate while clients search."""
import threading
import time

import numpy as np
import pytest

from gamma_amd import api, synth
from tests import lloyd as train
from oracle import binding as B
from tests.parity import compare_exact

pytestmark = pytest.mark.gpu
WIDE = dict(min_score=-3e38, max_score=3e38)


def _norm(x):
    return (x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-9)).astype(np.float32)


def test_search_during_inserts_sees_a_prefix_of_the_log():
    d, nlist, M, N0, nb, bs = 768, 64, 64, 6000, 16, 1000
    rng = np.random.default_rng(11)
    base = _norm(rng.standard_normal((N0 + nb * bs, d)).astype(np.float32) + 0.3 * rng.standard_normal((1, d)).astype(np.float32))
    q = _norm(rng.standard_normal((8, d)).astype(np.float32))
    cc, pq = train.train_ivfpq(base[:4000], nlist, M, niter=4, pq_niter=4, seed=3, device="cpu")
    k, nprobe, R = 10, 8, 60
    # the oracle at every prefix of the insert log (Add in the engine's batches: n >= 20 -> GEMM-form assignment)
    B.lib().go_set_assign_mode(1)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_IP, bucket_init_size=100)
    o.set_trained(cc, pq, None)
    assert o.add(base[:N0])
    ctx = B.make_ctx(**WIDE)
    expect = []
    for b in range(nb + 1):
        if b:
            assert o.add(base[N0 + (b - 1) * bs:N0 + b * bs])
        o.set_raw(base[:N0 + b * bs])
        expect.append(o.search(q, k, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_IP, ctx=ctx, coarse_mode=0))
    B.lib().go_set_assign_mode(0)
    assert any(not np.array_equal(expect[b][1], expect[b + 1][1]) for b in range(nb))   # the log changes the answers

    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_IP, 100)     # small buckets: lists grow (extents move) while searched
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base[:N0])
        g.add(base[:N0], 0)
        args = api.SearchArgs(metric=api.METRIC_IP, nprobe=nprobe, recall_num=R, has_rank=True, coarse_mode=0, **WIDE)
        stop = threading.Event()
        results = [[] for _ in range(3)]
        errors = []

        def client(slot):
            try:
                while not stop.is_set():
                    results[slot].append(g.ivfpq_search(q, k, args))
            except Exception as e:     # noqa: BLE001
                errors.append(e)

        th = [threading.Thread(target=client, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        time.sleep(0.05)
        t0 = time.perf_counter()
        for b in range(nb):            # the indexing thread: store first (re-rank reads it), then the index
            lo = N0 + b * bs
            g.raw_append(base[lo:lo + bs])
            g.add(base[lo:lo + bs], lo)
        dt = time.perf_counter() - t0
        time.sleep(0.05)
        stop.set()
        for t in th:
            t.join()
        assert not errors, errors
        rate = nb * bs / dt
        assert rate >= 10000, "insert rate %.0f vectors/s" % rate
        nsearch = sum(len(r) for r in results)
        assert nsearch >= 30
        seen = set()
        for slot in range(3):
            last = 0
            for D, I in results[slot]:
                hit = None
                for b in range(last, nb + 1):
                    try:
                        compare_exact(expect[b][0], expect[b][1], D, I)
                        hit = b
                        break
                    except AssertionError:
                        continue
                assert hit is not None, "a search saw a state that is no prefix of the insert log (after prefix %d)" % last
                last = hit
                seen.add(hit)
        assert len(seen) >= 2, seen            # searches really ran while the index was growing
        # and the final state is the whole log
        D, I = g.ivfpq_search(q, k, args)
        compare_exact(expect[nb][0], expect[nb][1], D, I)
    finally:
        g.close()


def test_raw_store_grows_in_place_under_search():
    """The raw store (re-rank rows, flat search) grows by mapping physical memory behind its rows (virtual memory
    management): no reallocation, no copy, no wait for the searches in flight -- a searcher thread keeps getting exact
    answers over a prefix of the rows while 600 MB arrive in 40 appends."""
    import threading
    d, rows_per, nb = 128, 30000, 40
    rng = np.random.default_rng(5)
    blocks = [rng.integers(0, 255, size=(rows_per, d)).astype(np.float32) for _ in range(4)]
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(blocks[0])
        st0 = g.raw_stats()
        q = blocks[0][:8] + 1.0
        args = api.SearchArgs(metric=api.METRIC_L2, min_score=-3e38, max_score=3e38)
        stop, bad, calls = threading.Event(), [], [0]

        def searcher():
            while not stop.is_set():
                D, I = g.flat_search(q, 1, args)
                calls[0] += 1
                # row i of block 0 is the nearest of q[i] (distance d) whatever has been appended since
                if not (np.array_equal(I[:, 0], np.arange(8)) and (D[:, 0] == float(d)).all()):
                    bad.append((I[:, 0].copy(), D[:, 0].copy()))

        t = threading.Thread(target=searcher)
        t.start()
        try:
            for b in range(1, nb):
                g.raw_append(blocks[b % 4] + np.float32(1000.0 * b))   # far from the queries
        finally:
            stop.set()
            t.join()
        st = g.raw_stats()
        assert st["rows"] == rows_per * nb and not bad and calls[0] > 0
        if st["in_place"]:
            assert st["moves"] == 0 and st0["moves"] == 0
    finally:
        g.close()


@pytest.mark.parametrize("mapped", [True, False])
def test_list_arena_grows_in_place_under_search(mapped, monkeypatch):
    """The inverted-list arena (codes, ids, code sums) grows by mapping physical memory behind its three arrays: no
    reallocation, no copy, no exclusive lock (the reference grows bucket by bucket,
    realtime/realtime_mem_data.cc:152-188,426-474).  A searcher thread keeps getting the answer of a prefix state while
    400 000 entries arrive in lists that start with 16 slots; the final state is the oracle's.  mapped = False: the
    reallocating fallback (GAMMA_HIP_NO_ARENA_VMM) gives the same results and reports its moves."""
    import threading
    if not mapped:
        monkeypatch.setenv("GAMMA_HIP_NO_ARENA_VMM", "1")
    d, nlist, M, nb, per = 32, 64, 8, 40, 10000
    rng = np.random.default_rng(11)
    base = rng.integers(0, 255, size=(nb * per, d)).astype(np.float32)
    cc, pq = api.train_ivfpq(base[:8000], nlist, M)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 16)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base[:per])
        g.add(base[:per], 0)
        g0 = g.arena_growth()
        q = base[:16] + 0.25     # row i of the first block is the nearest of q[i], whatever arrives later
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=nlist, recall_num=50, has_rank=True, min_score=-3e38,
                              max_score=3e38)
        stop, bad, calls = threading.Event(), [], [0]

        def searcher():
            while not stop.is_set():
                D, I = g.ivfpq_search(q, 1, args)
                calls[0] += 1
                if not np.array_equal(I[:, 0], np.arange(16)):
                    bad.append(I[:, 0].copy())

        t = threading.Thread(target=searcher)
        t.start()
        try:
            for b in range(1, nb):
                xb = base[b * per:(b + 1) * per] + np.float32(4000.0)    # far from the queries
                g.raw_append(xb)
                g.add(xb, b * per)
        finally:
            stop.set()
            t.join()
        assert not bad and calls[0] > 0
        st, gr = g.arena_stats(), g.arena_growth()
        assert st["used"] >= nb * per
        assert gr["mapped"] == (mapped and g0["mapped"])
        if gr["mapped"]:
            assert gr["moves"] == 0
        else:
            assert gr["moves"] > 0
        # final state against the oracle built the same way
        allx = np.concatenate([base[:per]] + [base[b * per:(b + 1) * per] + np.float32(4000.0) for b in range(1, nb)])
        B.lib().go_set_assign_mode(1)     # Add in the engine's batches: n >= 20 -> GEMM-form assignment
        o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2, bucket_init_size=16)
        o.set_trained(cc, pq, None)
        for b in range(nb):
            assert o.add(allx[b * per:(b + 1) * per])
        B.lib().go_set_assign_mode(0)
        o.set_raw(allx)
        qq = np.concatenate([q, allx[5 * per:5 * per + 16] + np.float32(0.25)])
        a2 = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=50, has_rank=True, min_score=-3e38, max_score=3e38,
                            coarse_mode=0)
        D, I = g.ivfpq_search(qq, 10, a2)
        Do, Io = o.search(qq, 10, 8, recall_num=50, has_rank=True, metric=B.METRIC_L2, ctx=B.make_ctx(**WIDE), coarse_mode=0)
        compare_exact(Do, Io, D, I)
    finally:
        g.close()


def test_searches_stay_exact_while_everything_else_moves():
    """The retrieval contract under stress (Search from any number of threads beside ONE indexing thread): three
    searcher threads (single queries, small and large batches, device chains of every kind) keep getting the oracle's answer
    over a fixed set of vectors while the writer thread adds far-away vectors (lists grow: extents move, the mapped arena
    grows), moves them between lists (Update), deletes them, compacts, and -- repack threshold 1 -- has the whole arena moved
    to the other set of address ranges again and again under the exclusive lock.  The searches carry a range filter that
    admits exactly the fixed set (tie-free data), so nothing the writer does may change an answer."""
    import threading
    d, nlist, M, nbase, nnoise = 32, 32, 8, 6000, 30000
    rng = np.random.default_rng(2024)
    base = rng.standard_normal((nbase, d)).astype(np.float32)
    noise = (rng.standard_normal((nnoise, d)) * 0.5 + 60.0).astype(np.float32)          # far from every query
    cc, pq = train.train_ivfpq(base[:4000], nlist, M, niter=4, pq_niter=4, seed=3, device="cpu")
    B.lib().go_set_assign_mode(1)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2, bucket_init_size=16)
    o.set_trained(cc, pq, None)
    assert o.add(base)
    B.lib().go_set_assign_mode(0)
    o.set_raw(base)
    allx = np.concatenate([base, noise])
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 16)                 # tiny buckets: constant growth
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(allx)
        g.add(base, 0)
        nbits = nbase + nnoise + 64
        g.bitmap_upload(np.zeros(nbits // 8 + 1, np.uint8), nbits)
        g.set_repack_threshold(1)
        ctx = B.make_ctx(**WIDE)
        jobs = []
        for nq, P, R, k in ((1, nlist, 50, 10), (7, nlist, 64, 5), (300, nlist, 100, 10), (4200, nlist, 40, 3)):
            q = rng.standard_normal((nq, d)).astype(np.float32)
            mode = 1 if nq >= 20 else 0
            D, I = o.search(q, k, P, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=mode)
            # (a range filter that admits exactly the fixed set: the far-away entries' PQ codes decode to ordinary points --
            #  a product quantizer cannot represent an outlier -- and would crowd the short-list otherwise)
            jobs.append((q, k, api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True,
                                              range_filters=[api.make_range_filter(np.arange(nbase, dtype=np.int64))], **WIDE), D, I))
        stop, bad, calls = threading.Event(), [], [0, 0, 0]

        def searcher(slot):
            i = slot
            while not stop.is_set():
                q, k, a, D, I = jobs[i % len(jobs)]
                Dg, Ig = g.ivfpq_search(q, k, a)
                if Dg.tobytes() != D.tobytes() or not np.array_equal(Ig, I):
                    rows = np.nonzero((Ig != I).any(axis=1) | (Dg != D).any(axis=1))[0]
                    r0 = int(rows[0]) if len(rows) else -1
                    bad.append((slot, i % len(jobs), len(rows), r0, I[r0].tolist(), Ig[r0].tolist(), D[r0].tolist(), Dg[r0].tolist()))
                    stop.set()
                calls[slot] += 1
                i += 1

        th = [threading.Thread(target=searcher, args=(s,)) for s in range(3)]
        for t in th:
            t.start()
        try:
            lno, codes = g.encode(noise)
            lno = np.where((lno < 0) | (lno >= nlist), np.arange(nnoise) % nlist, lno)
            added = 0
            t_end = time.time() + 4.0
            wr = np.random.default_rng(7)
            while time.time() < t_end and not stop.is_set():
                op = wr.random()
                if op < 0.45 and added < nnoise:
                    n = int(min(nnoise - added, wr.integers(50, 1500)))
                    g.add(noise[added:added + n], nbase + added)
                    added += n
                elif op < 0.65 and added:
                    j = int(wr.integers(0, added))     # (its own code: far in every list -- a random code could decode near a query)
                    g.update(int(wr.integers(0, nlist)), nbase + j, codes[j])
                elif op < 0.85 and added:
                    dead = (nbase + wr.integers(0, added, size=int(wr.integers(1, 200)))).astype(np.int64)
                    g.bitmap_set(dead, 1)
                    g.delete(dead)
                else:
                    g.compact_if_need()
        finally:
            stop.set()
            for t in th:
                t.join()
        assert not bad, bad
        assert min(calls) > 3 and g.arena_stats()["repacks"] > 0, (calls, g.arena_stats())
    finally:
        g.close()


@pytest.mark.parametrize("nthreads", [2, 3])
def test_concurrent_callers_each_get_a_complete_result(nthreads):
    """gamma_hip_ivfpq_search_device_wait (round 6): every call is complete when it returns to ITS caller, while the tie
    replay of one caller's call runs beside the next caller's coarse quantizer / tables / scan.  Tie-heavy data (duplicated
    vectors: equal ADC values at the recall_num cut, equal exact distances at the k cut) so that every call has flagged
    queries to replay; each thread's batches differ, every result is compared -- labels and distance bits at every rank --
    with the same batch through the plain call made alone."""
    import torch
    d, nlist, M, N = 32, 64, 8, 40000
    base = synth.sift_like(N, d=d, seed=31)
    base[N // 2:] = base[:N // 2]            # every vector twice
    cc, pq = B.ivfpq_train(base[:6000], nlist, M)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base)
        g.add(base, 0)
        dev = torch.device("cuda", 0)
        nq, k, nb = 4200, 10, 3
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=100, has_rank=True, coarse_mode=1, **WIDE)
        qs = [[torch.from_numpy(synth.sift_like(nq, d=d, seed=1000 + 17 * t + b)).to(dev) for b in range(nb)] for t in range(nthreads)]
        want = [[None] * nb for _ in range(nthreads)]
        D0 = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I0 = torch.empty((nq, k), dtype=torch.int64, device=dev)
        before = g.tie_stats()
        for t in range(nthreads):
            for b in range(nb):
                g.ivfpq_search_device(qs[t][b].data_ptr(), nq, k, args, D0.data_ptr(), I0.data_ptr())
                g.synchronize()
                want[t][b] = (D0.cpu().numpy().copy(), I0.cpu().numpy().copy())
        after = g.tie_stats()
        assert after["replayed"] > before["replayed"], "the data was meant to flag queries for the replay"
        errors = []

        def client(t):
            try:
                Dt = torch.empty((nq, k), dtype=torch.float32, device=dev)
                It = torch.empty((nq, k), dtype=torch.int64, device=dev)
                for rep in range(12):
                    b = rep % nb
                    g.ivfpq_search_device_wait(qs[t][b].data_ptr(), nq, k, args, Dt.data_ptr(), It.data_ptr())
                    # complete on return: read back WITHOUT any further synchronisation of the handle
                    Dh = torch.empty((nq, k), dtype=torch.float32).pin_memory()
                    Ih = torch.empty((nq, k), dtype=torch.int64).pin_memory()
                    st = torch.cuda.Stream(device=dev)
                    with torch.cuda.stream(st):
                        Dh.copy_(Dt, non_blocking=True)
                        Ih.copy_(It, non_blocking=True)
                    st.synchronize()
                    compare_exact(want[t][b][0], want[t][b][1], Dh.numpy(), Ih.numpy())
            except BaseException as e:   # noqa: B902 (reported by the main thread)
                errors.append((t, repr(e)))

        th = [threading.Thread(target=client, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors[:3]
        # and a plain call afterwards still joins whatever replay is pending
        g.ivfpq_search_device(qs[0][0].data_ptr(), nq, k, args, D0.data_ptr(), I0.data_ptr())
        g.synchronize()
        compare_exact(want[0][0][0], want[0][0][1], D0.cpu().numpy(), I0.cpu().numpy())
        # small batches (the small-batch chain replays inside its tail kernel) and an empty call through the same entry point
        for n in (1, 7, 300):
            g.ivfpq_search_device_wait(qs[1][1].data_ptr(), n, k, args, D0.data_ptr(), I0.data_ptr())
            a_small = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=100, has_rank=True, coarse_mode=1, **WIDE)
            Dp = torch.empty((n, k), dtype=torch.float32, device=dev)
            Ip = torch.empty((n, k), dtype=torch.int64, device=dev)
            g.ivfpq_search_device(qs[1][1].data_ptr(), n, k, a_small, Dp.data_ptr(), Ip.data_ptr())
            g.synchronize()
            compare_exact(Dp.cpu().numpy(), Ip.cpu().numpy(), D0[:n].cpu().numpy(), I0[:n].cpu().numpy())
        g.ivfpq_search_device_wait(qs[1][1].data_ptr(), 0, k, args, D0.data_ptr(), I0.data_ptr())
    finally:
        g.close()


def test_large_host_buffer_calls_from_several_threads_overlap_and_stay_exact():
    """gamma_hip_ivfpq_search with more than 1 MB of queries (the plugin's large Search): staged through two slots so that
    concurrent callers overlap (ivfpq_search_host_overlap) -- every caller gets the result of ITS call, bit for bit what the
    call gives when made alone; tie-heavy data, so that every call has a replay pending when the next one starts"""
    d, nlist, M, N = 32, 64, 8, 40000
    base = synth.sift_like(N, d=d, seed=32)
    base[N // 2:] = base[:N // 2]
    cc, pq = B.ivfpq_train(base[:6000], nlist, M)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base)
        g.add(base, 0)
        nq, k, nthreads, nb = 8400, 10, 3, 2          # 8400 x 32 x 4 B > 1 MB: the staged path
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=100, has_rank=True, coarse_mode=1, **WIDE)
        qs = [[synth.sift_like(nq, d=d, seed=2000 + 13 * t + b) for b in range(nb)] for t in range(nthreads)]
        want = [[g.ivfpq_search(qs[t][b], k, args) for b in range(nb)] for t in range(nthreads)]
        assert g.tie_stats()["replayed"] > 0
        errors = []

        def client(t):
            try:
                for rep in range(8):
                    b = rep % nb
                    D, I = g.ivfpq_search(qs[t][b], k, args)
                    compare_exact(want[t][b][0], want[t][b][1], D, I)
            except BaseException as e:   # noqa: B902
                errors.append((t, repr(e)))
        th = [threading.Thread(target=client, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors[:3]
        # a writer between the calls: the pending replay of the last call is joined before the lists change
        g.add(base[:100] + 1.0, N)
        D, I = g.ivfpq_search(qs[0][0], k, args)
        assert D.shape == (nq, k)
    finally:
        g.close()


@pytest.mark.parametrize("nthreads", [2, 3])
def test_concurrent_flat_callers_each_get_a_complete_result(nthreads):
    """gamma_hip_flat_search_device_wait: the flat call's heap replay on the side stream, the next caller in the other bank of
    what it reads; tie-heavy rows (every row twice) so that every call replays queries; every result compared -- labels and
    distance bits -- with the plain call's on the same batch; an IVFPQ call in between joins whatever is pending"""
    import torch
    d, N = 64, 80000
    base = synth.sift_like(N, d=d, seed=41)
    base[N // 2:] = base[:N // 2]
    g = api.GammaHip(0)
    try:
        g.raw_init(d)
        g.raw_append(base)
        dev = torch.device("cuda", 0)
        nq, k, nb = 700, 20, 3
        args = api.SearchArgs(metric=api.METRIC_L2, **WIDE)
        qs = [[torch.from_numpy(synth.sift_like(nq, d=d, seed=3000 + 11 * t + b)).to(dev) for b in range(nb)] for t in range(nthreads)]
        D0 = torch.empty((nq, k), dtype=torch.float32, device=dev)
        I0 = torch.empty((nq, k), dtype=torch.int64, device=dev)
        want = [[None] * nb for _ in range(nthreads)]
        g.tie_stats(reset=True)
        for t in range(nthreads):
            for b in range(nb):
                g.flat_search_device(qs[t][b].data_ptr(), nq, k, args, D0.data_ptr(), I0.data_ptr())
                g.synchronize()
                want[t][b] = (D0.cpu().numpy().copy(), I0.cpu().numpy().copy())
        assert g.tie_stats()["replayed"] > 0
        errors = []

        def client(t):
            try:
                Dt = torch.empty((nq, k), dtype=torch.float32, device=dev)
                It = torch.empty((nq, k), dtype=torch.int64, device=dev)
                st = torch.cuda.Stream(device=dev)
                for rep in range(10):
                    b = rep % nb
                    g.flat_search_device_wait(qs[t][b].data_ptr(), nq, k, args, Dt.data_ptr(), It.data_ptr())
                    with torch.cuda.stream(st):
                        Dh, Ih = Dt.to("cpu", non_blocking=False), It.to("cpu", non_blocking=False)
                    compare_exact(want[t][b][0], want[t][b][1], Dh.numpy(), Ih.numpy())
            except BaseException as e:   # noqa: B902
                errors.append((t, repr(e)))
        th = [threading.Thread(target=client, args=(t,)) for t in range(nthreads)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errors, errors[:3]
        g.flat_search_device(qs[0][1].data_ptr(), nq, k, args, D0.data_ptr(), I0.data_ptr())
        g.synchronize()
        compare_exact(want[0][1][0], want[0][1][1], D0.cpu().numpy(), I0.cpu().numpy())
    finally:
        g.close()
