"""GPU: exact-ties mode (gamma_hip_set_exact_ties, csrc/ties.hip).  With it on, labels must be the reference's
at EVERY rank -- which members of a group of equal distances survive the nprobe / recall_num / k cuts and the
order equal distances come out in -- checked with tests/parity.compare_exact (no tie tolerance, no excluded
queries) against the tie-heavy golden built with the real faiss heaps and against the pinned oracle."""
import os

import numpy as np
import pytest

from gamma_amd import api, synth
from oracle import binding as B
from tests import fixtures
from tests.parity import compare_exact
from tests.test_oracle_golden import load_ties

pytestmark = pytest.mark.gpu
WIDE = dict(min_score=-3e38, max_score=3e38)


def _device_for(z, tag, base, metric):
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    g = api.GammaHip(0)
    g.ivfpq_init(d, nlist, M, 8, metric)
    g.ivfpq_set_trained(z["cc_" + tag], z["pq_" + tag], None)
    sizes = z["list_sizes_" + tag]
    nz = np.nonzero(sizes)[0]
    g.add_keys_batch(nz, sizes[nz], z["list_ids_" + tag], z["list_codes_" + tag])
    g.raw_init(d)
    g.raw_append(base)
    g.set_exact_ties(True)
    return g


@pytest.mark.parametrize("name", ["ivfpq_ties_d32", "ivfpq_ties_c4shape"])
@pytest.mark.parametrize("tag", ["l2", "ip"])
def test_tie_heavy_golden_on_device(tag, name):
    """tests/golden/ivfpq_ties_d32.npz: most queries have equal ADC distances across the recall_num cut and
    equal exact distances across the k cut.  Small call: the small-batch chain flags and replays inside its
    kernels.  ivfpq_ties_c4shape.npz: the same with the cuts of the C4 configuration -- 4160 lists (rows of the
    coarse matrix longer than the wave selection's chunk: the generic tie flags), 64 probes, recall_num 100."""
    z, o, base, metric = load_ties(tag, name)
    nprobe, R, k = int(z["nprobe"]), int(z["R"]), int(z["k"])
    g = _device_for(z, tag, base, metric)
    try:
        for has_rank, nm in ((True, "rank"), (False, "norank")):
            g.tie_stats(reset=True)
            args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **WIDE)
            Dg, Ig = g.ivfpq_search(z["q"], k, args)
            sg = g.last_stages(len(z["q"]), nprobe, R)
            assert sg["coarse_dis"].tobytes() == z["coarse_dis_" + tag].tobytes()
            assert np.array_equal(sg["coarse_idx"], z["coarse_idx_" + tag])
            compare_exact(z["D_%s_%s" % (nm, tag)], z["I_%s_%s" % (nm, tag)], Dg, Ig)
            st = g.tie_stats()
            assert st["replayed"] >= int(z["ncut_" + tag][0])
            g.set_small_path(0)   # the same call through the regular chain (matrix coarse path, k_tie_replay)
            Dg2, Ig2 = g.ivfpq_search(z["q"], k, args)
            g.set_small_path(1)
            compare_exact(z["D_%s_%s" % (nm, tag)], z["I_%s_%s" % (nm, tag)], Dg2, Ig2)
            if not has_rank:     # the replay leaves the recall-stage table in heap_reorder order
                rows = np.array([i for i in range(len(z["q"]))])
                compare_exact(z["rdis_" + tag][rows], z["rids_" + tag][rows], sg["recall_dis"][rows],
                              sg["recall_ids"][rows])
        # without the mode the same call is only tie-tolerantly equal -- the test data really needs the replay
        g.set_exact_ties(False)
        Dg, Ig = g.ivfpq_search(z["q"], k, api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R,
                                                         has_rank=True, coarse_mode=0, **WIDE))
        assert not np.array_equal(Ig, z["I_rank_" + tag])
    finally:
        g.close()


@pytest.mark.parametrize("tag,nprobe,R,k,reps", [("l2", 12, 60, 10, 100), ("ip", 12, 60, 10, 100),
                                                  ("l2", 16, 200, 10, 90), ("l2", 6, 40, 40, 100)])
def test_tie_heavy_batches_take_the_bounded_scan(tag, nprobe, R, k, reps):
    """The same tie-heavy index at batch size: thousands of queries go through the scan with the threshold
    pre-filter (survivor slices, k_select_final flags the cut ties); the replay then rebuilds the candidate stream
    from the first probe group's distances + the slices.  Expected: the pinned oracle, strictly."""
    z, o, base, metric = load_ties(tag)
    q = np.tile(z["q"], (reps, 1))
    q[len(z["q"]):] += np.float32(0)    # exact copies: identical answers, different rows
    g = _device_for(z, tag, base, metric)
    try:
        ctx = B.make_ctx(**WIDE)
        for has_rank in (True, False):
            D, I, st = o.search(z["q"], k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx,
                                coarse_mode=0, want_stages=True)
            args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **WIDE)
            g.tie_stats(reset=True)
            Dg, Ig = g.ivfpq_search(q, k, args)
            compare_exact(np.tile(D, (reps, 1)), np.tile(I, (reps, 1)), Dg, Ig)
            assert g.tie_stats()["replayed"] > 0
    finally:
        g.close()


def test_ties_with_filters_and_score_window():
    """Deleted docs and a range filter remove candidates BEFORE the heaps (the reference `continue`s), the score
    window applies at the k-heap: the replay must see the same stream."""
    z, o, base, metric = load_ties("l2")
    nprobe, R, k = 8, 50, 10
    N = len(base)
    rng = np.random.default_rng(3)
    dead = rng.choice(N, size=N // 10, replace=False)
    bm = np.zeros(N // 8 + 1, np.uint8)
    for v in dead:
        bm[v >> 3] |= 1 << (v & 7)
    allowed = np.nonzero(rng.random(N) < 0.6)[0]
    g = _device_for(z, "l2", base, metric)
    try:
        g.bitmap_upload(bm, N)
        o.set_docids_bitmap(bm)
        rf = B.make_range_filter(allowed)
        q = np.tile(z["q"], (12, 1))
        for has_rank in (True, False):
            for win in (dict(min_score=-3e38, max_score=3e38), dict(min_score=2000.0, max_score=60000.0)):
                ctx = B.make_ctx(docids_bitmap=bm, range_filters=[rf], **win)
                D, I = o.search(z["q"], k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx,
                                coarse_mode=0)
                args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0,
                                      range_filters=[api.make_range_filter(allowed)], **win)
                for qq, rep in ((z["q"], 1), (q, 12)):
                    Dg, Ig = g.ivfpq_search(qq, k, args)
                    compare_exact(np.tile(D, (rep, 1)), np.tile(I, (rep, 1)), Dg, Ig)
    finally:
        g.close()


def test_plugin_exact_ties_keys():
    """Through the RetrievalModel boundary: exact ties are the model's default, `"exact_ties": 0` in the model's
    retrieval_param or in a request's retrieval parameters (Parse) turns them off for the index / that request."""
    from gamma_amd import plugin
    z, _, base, metric = load_ties("l2")
    d, nlist, M = int(z["d"]), int(z["nlist"]), int(z["M"])
    nprobe, R, k = int(z["nprobe"]), int(z["R"]), int(z["k"])
    q = np.tile(z["q"], (3, 1))                       # 3 x the golden queries: >= 20 -> GEMM-form coarse on both sides
    o = B.OracleIVFPQ(d, nlist, M, 8, metric)
    o.set_trained(z["cc_l2"], z["pq_l2"], None)
    B.lib().go_set_assign_mode(1)                     # GammaIVFPQIndex::Add of >= 20 vectors: faiss's BLAS assign rule
    try:
        assert o.add(base)
    finally:
        B.lib().go_set_assign_mode(0)
    o.set_raw(base)
    D, I = o.search(q, k, nprobe, recall_num=R, has_rank=True, metric=metric, ctx=B.make_ctx(), coarse_mode=-1)
    req = '{"metric_type": "L2", "recall_num": %d, "nprobe": %d%s}'
    model = '{"ncentroids": %d, "nsubvector": %d, "nprobe": %d, "metric_type": "L2"%s}'
    for model_extra, req_extra, exact in (("", "", True), ("", ', "exact_ties": 0', False),
                                          (', "exact_ties": 0', "", False), (', "exact_ties": 0', ', "exact_ties": 1', True)):
        m = plugin.PluginModel("HIPIVFPQ", d, model % (nlist, M, nprobe, model_extra), indexing_size=len(base))
        try:
            m.store(base)
            assert m.set_trained(z["cc_l2"], z["pq_l2"]) == 0
            assert m.add(base)
            Dg, Ig = m.search(q, k, req % (R, nprobe, req_extra), has_rank=True)
            if exact:
                compare_exact(D, I, Dg, Ig)
            else:
                # (distances agree up to the members of cut ties; the labels of this tie-heavy data do not)
                assert not np.array_equal(Ig, I)
        finally:
            m.close()


@pytest.mark.parametrize("tag", ["l2", "ip"])
def test_c4_shape_ties_at_batch_size(tag):
    """The C4-shaped tie-heavy index at a batch size that takes the matrix-free coarse quantizer (>= 4096 queries,
    GEMM form, 4160 lists, 64 probes: k_coarse_fused -> k_coarse_final flags rows with equal keys near the cut -> their
    rows recomputed by the MFMA chain and walked through faiss's result heap) and the bounded scan + k_tie_replay.
    Expected: the pinned oracle with the GEMM-form coarse distances, labels strictly."""
    z, o, base, metric = load_ties(tag, "ivfpq_ties_c4shape")
    nprobe, R, k = int(z["nprobe"]), int(z["R"]), int(z["k"])
    reps = 4096 // len(z["q"]) + 1
    q = np.tile(z["q"], (reps, 1))
    g = _device_for(z, tag, base, metric)
    try:
        ctx = B.make_ctx(**WIDE)
        for has_rank in (True, False):
            D, I, st = o.search(z["q"], k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx,
                                coarse_mode=1, want_stages=True)
            args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=-1, **WIDE)
            g.tie_stats(reset=True)
            Dg, Ig = g.ivfpq_search(q, k, args)
            sg = g.last_stages(len(q), nprobe, R)
            assert sg["coarse_dis"].tobytes() == np.tile(st["coarse_dis"], (reps, 1)).tobytes()
            assert np.array_equal(sg["coarse_idx"], np.tile(st["coarse_idx"], (reps, 1)))
            compare_exact(np.tile(D, (reps, 1)), np.tile(I, (reps, 1)), Dg, Ig)
            ts = g.tie_stats()
            assert ts["coarse_rows"] > 0 and ts["replayed"] > 0
    finally:
        g.close()


def _ivfflat_for(z, tag, base, metric):
    g = api.GammaHip(0)
    g.ivfflat_init(int(z["d"]), int(z["nlist"]), metric, 1000)
    g.ivfflat_set_trained(z["cc_" + tag])
    sizes = z["list_sizes_" + tag]
    nz = np.nonzero(sizes)[0]
    g.add_keys_batch(nz, sizes[nz], z["list_ids_" + tag], np.zeros((len(z["list_ids_" + tag]), 1), np.uint8))
    g.raw_init(int(z["d"]))
    g.raw_append(base)
    return g


@pytest.mark.parametrize("tag", ["l2", "ip"])
def test_ivfflat_exact_ties(tag):
    """IVFFLAT on the tie-heavy data (every base vector four times): the scanner's k-heap takes an entry with
    heap_pop + heap_push (gamma_index_ivfflat.h:52-75) and heap_reorder orders the result -- labels strictly the
    oracle's at every rank, through the small-batch chain (replay inside k_small_tail), the pair kernel and the
    list-major kernel (k_flag_cut_ties + k_tie_replay), with deletes, a range filter and a score window."""
    z, o, base, metric = load_ties(tag)
    g = _ivfflat_for(z, tag, base, metric)
    N = len(base)
    rng = np.random.default_rng(5)
    try:
        for step in range(2):
            ctx_kw, kw_f = {}, {}
            if step == 1:
                dead = rng.choice(N, N // 9, replace=False)
                bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
                np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
                g.bitmap_upload(bm, N)
                docs = rng.choice(N, 2 * N // 3, replace=False)
                ctx_kw = dict(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)])
                kw_f = dict(range_filters=[api.make_range_filter(docs)])
            for q, reps in ((z["q"][:4], 1), (z["q"][:5], 1), (z["q"], 1), (z["q"], 30)):
                for P, k in ((6, 10), (3, 40), (16, 100), (1, 3)):
                    wins = [WIDE]
                    Dw, _ = B.ivfflat_search(o, q, k, P, metric, B.make_ctx(**WIDE, **ctx_kw), coarse_mode=0)
                    fin = Dw[np.abs(Dw) < 1e37]
                    if len(fin) > 4 and reps == 1:
                        wins.append(dict(min_score=float(np.quantile(fin, 0.3)), max_score=float(np.quantile(fin, 0.9))))
                    for win in wins:
                        D, I = B.ivfflat_search(o, q, k, P, metric, B.make_ctx(**win, **ctx_kw), coarse_mode=0)
                        args = api.SearchArgs(metric=metric, nprobe=P, coarse_mode=0, **win, **kw_f)
                        g.tie_stats(reset=True)
                        Dg, Ig = g.ivfflat_search(np.tile(q, (reps, 1)), k, args)
                        compare_exact(np.tile(D, (reps, 1)), np.tile(I, (reps, 1)), Dg, Ig)
                        if k > 3:
                            assert g.tie_stats()["replayed"] > 0
        # the data needs it: with the mode off for the request only the distances agree
        args = api.SearchArgs(metric=metric, nprobe=6, coarse_mode=0, exact_ties=-1, **WIDE, **kw_f)
        D, I = B.ivfflat_search(o, z["q"], 10, 6, metric, B.make_ctx(**WIDE, **ctx_kw), coarse_mode=0)
        Dg, Ig = g.ivfflat_search(z["q"], 10, args)
        assert Dg.tobytes() == D.tobytes() and not np.array_equal(Ig, I)
    finally:
        g.close()


@pytest.mark.parametrize("tag", ["l2", "ip"])
def test_flat_exact_ties(tag):
    """GammaFLATIndex::Search on tie-heavy data (gamma_index_flat.cc:118-300: rows in vid order through heap_pop +
    heap_push, heap_reorder): labels strictly the oracle's -- the small-batch chain (replay inside k_small_tail), the
    slab path (one row chunk) and the running-bound path over several row chunks (both run for k + 1 results, queries
    with equal distances among them replayed over a recomputed distance row)."""
    z, _, base, metric = load_ties(tag)
    d = int(z["d"])
    rng = np.random.default_rng(11)
    for mult in (1, 12):     # 6 000 rows: one chunk; 72 000 rows: the running bound, two row chunks
        b = np.ascontiguousarray(np.tile(base, (mult, 1)))
        if mult > 1:
            b = np.ascontiguousarray(b[rng.permutation(len(b))])
        N = len(b)
        g = api.GammaHip(0)
        g.raw_init(d)
        g.raw_append(b)
        try:
            for step in range(2):
                ctx_kw, kw_f = {}, {}
                if step == 1:
                    dead = rng.choice(N, N // 9, replace=False)
                    bm = np.zeros((N >> 3) + 1, dtype=np.uint8)
                    np.bitwise_or.at(bm, dead >> 3, (1 << (dead & 7)).astype(np.uint8))
                    g.bitmap_upload(bm, N)
                    docs = rng.choice(N, 2 * N // 3, replace=False)
                    ctx_kw = dict(docids_bitmap=bm, range_filters=[B.make_range_filter(docs)])
                    kw_f = dict(range_filters=[api.make_range_filter(docs)])
                for q, reps in ((z["q"][:3], 1), (z["q"], 1), (z["q"], 3)):     # 144 queries: past the small-batch chain
                    for k in (10, 1, 100):
                        wins = [WIDE]
                        Dw, _ = B.flat_search(b, q, k, metric, B.make_ctx(**WIDE, **ctx_kw))
                        fin = Dw[np.abs(Dw) < 1e37]
                        if len(fin) > 4 and k == 10:
                            wins.append(dict(min_score=float(np.quantile(fin, 0.3)), max_score=float(np.quantile(fin, 0.9))))
                        for win in wins:
                            D, I = B.flat_search(b, q, k, metric, B.make_ctx(**win, **ctx_kw))
                            args = api.SearchArgs(metric=metric, **win, **kw_f)
                            g.tie_stats(reset=True)
                            Dg, Ig = g.flat_search(np.tile(q, (reps, 1)), k, args)
                            compare_exact(np.tile(D, (reps, 1)), np.tile(I, (reps, 1)), Dg, Ig)
                            if k > 1:
                                assert g.tie_stats()["replayed"] > 0
            args = api.SearchArgs(metric=metric, exact_ties=-1, **WIDE, **kw_f)
            D, I = B.flat_search(b, z["q"], 10, metric, B.make_ctx(**WIDE, **ctx_kw))
            Dg, Ig = g.flat_search(np.tile(z["q"], (3, 1)), 10, args)
            assert Dg[:len(D)].tobytes() == D.tobytes() and not np.array_equal(Ig[:len(I)], I)
        finally:
            g.close()


def test_deferred_replay_streams_of_device_calls():
    """gamma_hip_set_deferred_replay: back-to-back device-pointer calls whose flagged queries are replayed on the side
    stream beside the next call's first stages.  Different query sets alternate through the same result buffers; after
    join / the next call every call's rows are the oracle's, strictly; a host-buffer call in between, an Add and a small
    call do not disturb it."""
    import torch
    z, o, base, metric = load_ties("l2")
    nprobe, R, k = 12, 60, 10
    g = _device_for(z, "l2", base, metric)
    try:
        ctx = B.make_ctx(**WIDE)
        qa = z["q"]
        qb = np.ascontiguousarray(z["q"][::-1])
        exp = {}
        for nm, qq in (("a", qa), ("b", qb)):
            exp[nm] = o.search(qq, k, nprobe, recall_num=R, has_rank=True, metric=metric, ctx=ctx, coarse_mode=0)
        reps = 100
        dev = torch.device("cuda", 0)
        d_q = {nm: torch.from_numpy(np.tile(qq, (reps, 1))).to(dev) for nm, qq in (("a", qa), ("b", qb))}
        nq = len(qa) * reps
        outs = [(torch.empty((nq, k), dtype=torch.float32, device=dev), torch.empty((nq, k), dtype=torch.int64, device=dev))
                for _ in range(2)]
        args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=True, coarse_mode=0, **WIDE)
        g.set_deferred_replay(True)
        g.tie_stats(reset=True)
        seq = ["a", "b", "b", "a", "a", "b"]
        for i, nm in enumerate(seq):
            D, I = outs[i & 1]
            g.ivfpq_search_device(d_q[nm].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
            if i >= 1:
                # the call before this one is complete once this one is enqueued on the handle's stream ... after a join of
                # this call's own replay only the synchronize below says so; check call i - 1 at the next turn
                pass
            if i == 2:
                Dh, Ih = g.ivfpq_search(qa, k, args)          # host-buffer call: joins, replays inline
                compare_exact(exp["a"][0], exp["a"][1], Dh, Ih)
                Ds, Is = g.ivfpq_search(qa[:3], k, args)      # small-batch chain
                compare_exact(exp["a"][0][:3], exp["a"][1][:3], Ds, Is)
            if i == 3:
                g.join()
                torch.cuda.synchronize()                       # (the handle's stream is not torch's)
                D3, I3 = outs[i & 1]
                compare_exact(np.tile(exp[nm][0], (reps, 1)), np.tile(exp[nm][1], (reps, 1)), D3.cpu().numpy(), I3.cpu().numpy())
        g.synchronize()
        for i in (len(seq) - 2, len(seq) - 1):
            D, I = outs[i & 1]
            e = exp[seq[i]]
            compare_exact(np.tile(e[0], (reps, 1)), np.tile(e[1], (reps, 1)), D.cpu().numpy(), I.cpu().numpy())
        assert g.tie_stats()["replayed"] > 0
        # a call cut into many chunks (small slab budget): the replay of chunk i runs beside chunk i + 1
        g.set_dist_budget(4 << 20)
        for nm in ("a", "b"):
            D, I = outs[0]
            g.ivfpq_search_device(d_q[nm].data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr())
            g.synchronize()
            compare_exact(np.tile(exp[nm][0], (reps, 1)), np.tile(exp[nm][1], (reps, 1)), D.cpu().numpy(), I.cpu().numpy())
        g.set_deferred_replay(False)
    finally:
        g.close()


def _shard_of(z, tag, base, metric, owner, s, raw_sharded=False):
    g = api.GammaHip(0)
    g.ivfpq_init(int(z["d"]), int(z["nlist"]), int(z["M"]), 8, metric)
    g.ivfpq_set_trained(z["cc_" + tag], z["pq_" + tag], None)
    sizes = z["list_sizes_" + tag]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    lists, counts, vids, codes = [], [], [], []
    for l in range(int(z["nlist"])):
        if owner[l] == s and sizes[l]:
            lists.append(l)
            counts.append(int(sizes[l]))
            vids.append(z["list_ids_" + tag][offs[l]:offs[l + 1]])
            codes.append(z["list_codes_" + tag][offs[l]:offs[l + 1]])
    g.add_keys_batch(lists, counts, np.concatenate(vids), np.concatenate(codes))
    mask = (np.asarray(owner) == s).astype(np.uint8)
    g.set_list_mask(mask)
    g.raw_init(int(z["d"]))
    if raw_sharded:   # raw vectors sharded with their lists: the rows of this shard's vectors only (duplicates in the golden's
        mine = np.unique(np.concatenate(vids) & 0x7fffffffffffffff)   # lists -- moved entries -- are one row)
        g.raw_put(mine, base[mine])
    else:
        g.raw_append(base)
    return g


@pytest.mark.parametrize("tag,W,reps,has_rank,small_budget,raw_sharded", [
    ("l2", 2, 1, True, False, False), ("l2", 3, 13, True, False, False), ("ip", 2, 13, True, False, False), ("l2", 2, 13, False, False, False),
    ("l2", 4, 90, True, False, False), ("l2", 2, 13, True, True, False), ("ip", 4, 90, True, True, False),
    # raw vectors sharded with their lists (round 6): the exact distances travel with the candidates and with the exported streams
    ("l2", 2, 1, True, False, True), ("l2", 3, 13, True, False, True), ("ip", 2, 13, True, False, True), ("l2", 4, 90, True, False, True),
    ("ip", 4, 90, True, True, True)])
def test_exact_ties_across_list_shards(tag, W, reps, has_rank, small_budget, raw_sharded):
    """W shards emulated on one GPU through the C ABI (what gamma_hip_group / dist.py drive): coarse per slice, shard scans,
    merge + re-rank at the slice's owner -- then the tie phase: the owner lists the queries a tie can change
    (gamma_hip_ivfpq_merge_flagged), every shard exports their candidate streams over the lists it owns
    (gamma_hip_ivfpq_shard_export), the owner assembles and replays them (gamma_hip_ivfpq_merge_replay).  Expected: the
    pinned oracle on the unsharded index, labels strictly.  90 repetitions: >= 4096 queries per shard call (one workgroup per
    query whatever the lists' length).  small_budget: the workspace budget holds the slab of ~40 queries at the general
    stride (nprobe x the longest list), so the shard call measures the longest candidate row of the batch on the device,
    sizes its chunks by that, and gathers the chunks' cut-tie flags."""
    import torch
    from gamma_amd import dist as gdist
    z, o, base, metric = load_ties(tag)
    nprobe, R, k = 12, 60, 10
    sizes = z["list_sizes_" + tag]
    owner = gdist.balance_lists(sizes, W)
    shards = [_shard_of(z, tag, base, metric, owner, s, raw_sharded) for s in range(W)]
    try:
        q1 = z["q"]
        D1, I1 = o.search(q1, k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=B.make_ctx(**WIDE), coarse_mode=0)
        qh = np.tile(q1, (reps, 1))[:len(q1) * reps - (1 if reps > 1 else 0)]     # not a multiple of W
        nq = len(qh)
        Dexp = np.tile(D1, (reps, 1))[:nq]
        Iexp = np.tile(I1, (reps, 1))[:nq]
        dev = torch.device("cuda", 0)
        x = torch.from_numpy(qh).to(dev)
        args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **WIDE)
        from tests.shard_emul import sharded_search_emulated
        # (W == 3 runs without the shards' flags: every table that ends at the cut value counts as a tie)
        D, I, flagged = sharded_search_emulated(shards, x, k, args, use_shard_flags=(W != 3), raw_sharded=raw_sharded)
        assert flagged > 0
        compare_exact(Dexp, Iexp, D.cpu().numpy(), I.cpu().numpy())
        if small_budget:
            for g in shards:
                g.set_dist_budget(40 * nprobe * max(1, g.max_list_len()) * 4)
            D2, I2, flagged2 = sharded_search_emulated(shards, x, k, args, use_shard_flags=True, raw_sharded=raw_sharded)
            assert flagged2 == flagged   # the chunks' own flags reached the merge (not "every query may have cut a tie")
            compare_exact(Dexp, Iexp, D2.cpu().numpy(), I2.cpu().numpy())
    finally:
        for g in shards:
            g.close()


def test_plugin_flat_and_ivfflat_exact_ties_keys():
    """HIPFLAT / HIPIVFFLAT behind the RetrievalModel boundary: exact ties are the models' default, `"exact_ties": 0` in
    the model's or a request's parameters turns them off (tie-heavy data: labels strictly the oracle's, or not)."""
    from gamma_amd import plugin
    z, o, base, metric = load_ties("l2")
    d, nlist = int(z["d"]), int(z["nlist"])
    q = z["q"]
    Df, If = B.flat_search(base, q, 10, B.METRIC_L2, B.make_ctx())
    for model_extra, req_extra, exact in (("", "", True), ("", ', "exact_ties": 0', False), (', "exact_ties": 0', "", False),
                                          (', "exact_ties": 0', ', "exact_ties": 1', True)):
        m = plugin.PluginModel("HIPFLAT", d, '{"metric_type": "L2"%s}' % model_extra)
        try:
            assert m.add(base)
            Dg, Ig = m.search(q, 10, '{"metric_type": "L2"%s}' % req_extra)
            assert Dg.tobytes() == Df.tobytes()
            assert np.array_equal(Ig, If) == exact
        finally:
            m.close()
    # IVFFLAT: the oracle's lists through Indexing-free set-up is not exposed by the plugin harness; its own k-means then
    # (same seeds on both sides are not needed: the expected result comes from the plugin with the mode on, the oracle pins
    # that path in test_ivfflat_exact_ties) -- here only that the keys reach the device
    ms = []
    try:
        for extra in ("", ', "exact_ties": 0'):
            m = plugin.PluginModel("HIPIVFFLAT", d, '{"ncentroids": %d, "nprobe": 6, "metric_type": "L2"%s}' % (nlist, extra),
                                   indexing_size=len(base))
            ms.append(m)
            m.store(base)
            assert m.indexing() == 0
            assert m.add(base)
        D1, I1 = ms[0].search(q, 10, '{"metric_type": "L2", "nprobe": 6}')
        D0, I0 = ms[1].search(q, 10, '{"metric_type": "L2", "nprobe": 6}')
        Dr, Ir = ms[0].search(q, 10, '{"metric_type": "L2", "nprobe": 6, "exact_ties": 0}')
        assert D1.tobytes() == D0.tobytes() == Dr.tobytes()
        assert not np.array_equal(I1, I0) and np.array_equal(I0, Ir)
    finally:
        for m in ms:
            m.close()


@pytest.mark.parametrize("tag", ["l2", "ip"])
def test_heaps_beyond_1024_entries(tag):
    """recall_num and k above 1024 (up to the 4096 the ABI accepts; the reference has no limit, faiss:utils/Heap.h:103-131,
    gamma_index_ivfpq.cc:762-770): the regular chain + k_tie_replay's <.., 4096, 4096> variant (sort buffer of 4096 items,
    heaps in > 64 KB of LDS).  Tie-heavy data, labels strictly the oracle's: IVFPQ with and without rank, IVFFLAT, flat."""
    z, o, base, metric = load_ties(tag)
    q = z["q"][:12]
    g = _device_for(z, tag, base, metric)
    try:
        ctx = B.make_ctx(**WIDE)
        for nprobe, R, k, has_rank in ((16, 1500, 1100, True), (16, 2000, 10, True), (12, 1030, 1030, False),
                                       (16, 4096, 4096, True), (12, 3000, 200, False)):
            D, I = o.search(q, k, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=ctx, coarse_mode=0)
            args = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **WIDE)
            g.tie_stats(reset=True)
            Dg, Ig = g.ivfpq_search(q, k, args)
            compare_exact(D, I, Dg, Ig)
            assert g.tie_stats()["replayed"] > 0
        assert g.ties_not_honoured() == 0
    finally:
        g.close()
    gf = _ivfflat_for(z, tag, base, metric)
    try:
        for P, k in ((3, 2000), (16, 1100), (16, 4096)):
            D, I = B.ivfflat_search(o, q, k, P, metric, B.make_ctx(**WIDE), coarse_mode=0)
            Dg, Ig = gf.ivfflat_search(q, k, api.SearchArgs(metric=metric, nprobe=P, coarse_mode=0, **WIDE))
            compare_exact(D, I, Dg, Ig)
        assert gf.ties_not_honoured() == 0
    finally:
        gf.close()
    g2 = api.GammaHip(0)
    try:
        g2.raw_init(int(z["d"]))
        g2.raw_append(base)
        for k in (1100, 2500, 4095):
            D, I = B.flat_search(base, q[:5], k, metric, B.make_ctx(**WIDE))
            Dg, Ig = g2.flat_search(q[:5], k, api.SearchArgs(metric=metric, **WIDE))
            compare_exact(D, I, Dg, Ig)
        assert g2.ties_not_honoured() == 0
    finally:
        g2.close()


def test_a_shape_beyond_the_replay_is_never_silent():
    """nprobe > 1024 (and a flat search for k = 4096) are outside the exact-ties mode.  A request that asks for the mode
    explicitly fails with GAMMA_HIP_EUNSUPPORTED; one that inherits the handle's default runs with the (distance, position)
    order inside ties -- still the reference's distances at every rank -- and is counted (gamma_hip_ties_not_honoured)."""
    from tests.parity import compare_topk
    case = fixtures.trained_case(d=32, nlist=1056, M=8, N=45000, nq=64, metric=B.METRIC_L2)
    g = fixtures.load_hip(case)
    try:
        q = case["q"][:9]
        P, R, k = 1030, 100, 10
        ctx = B.make_ctx(**WIDE)
        D, I = case["oracle"].search(q, k, P, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=0)
        assert g.ties_not_honoured() == 0
        Dg, Ig = g.ivfpq_search(q, k, api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, coarse_mode=0, **WIDE))
        compare_topk(D, I, Dg, Ig)                 # ties off on purpose: the tie-tolerant comparison
        assert g.ties_not_honoured() == 1
        with pytest.raises(RuntimeError, match="ties"):
            g.ivfpq_search(q, k, api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, coarse_mode=0, exact_ties=1, **WIDE))
        g.ivfpq_search(q, k, api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, coarse_mode=0, exact_ties=-1, **WIDE))
        assert g.ties_not_honoured() == 1          # asked to be off: nothing to report
        g.ivfpq_search(q, k, api.SearchArgs(metric=api.METRIC_L2, nprobe=1024, recall_num=R, coarse_mode=0, exact_ties=1, **WIDE))
        assert g.ties_not_honoured(reset=True) == 1 and g.ties_not_honoured() == 0
        Df, If = B.flat_search(case["base"], q[:2], 4096, B.METRIC_L2, ctx)
        Dg, Ig = g.flat_search(q[:2], 4096, api.SearchArgs(metric=api.METRIC_L2, **WIDE))
        compare_topk(Df, If, Dg, Ig)
        assert g.ties_not_honoured() == 1
        with pytest.raises(RuntimeError, match="ties"):
            g.flat_search(q[:2], 4096, api.SearchArgs(metric=api.METRIC_L2, exact_ties=1, **WIDE))
    finally:
        g.close()


@pytest.mark.parametrize("nprobe,nq,coarse_mode", [(100, 24, 0), (128, 24, 1), (128, 700, 0), (200, 24, 0), (200, 700, 1),
                                                     (256, 300, 0), (100, 4200, -1), (256, 4200, -1), (99, 700, 0),
                                                     (300, 300, 0), (512, 24, 0), (600, 700, 1), (640, 4200, -1)])
def test_coarse_ties_from_100_probes_on_are_the_reservoirs(nprobe, nq, coarse_mode):
    """From 100 probes on faiss's knn_L2sqr collects the coarse assignment through ReservoirTopN instead of the result
    heap (faiss:utils/distances.cpp:341-358): which of the centroids at the same distance are probed, and in which order
    their lists are scanned, is the reservoir's doing (oracle: go_reservoir_stream, pinned against the compiled library
    in tests/test_oracle_vs_ref.py and tests/golden/reservoir_ties.npz).  Integer centroids and queries on a small
    grid: nearly every row has equal keys around the nprobe cut.  Small calls (the small-batch chain walks in-kernel up
    to 128 probes), matrix-path calls and the matrix-free coarse quantizer (>= 4096 queries); 99 probes: the heap; beyond
    256 probes (one row per workgroup, up to 1024)."""
    d, nlist, M, N, R, k = 16, 640, 4, 30000, 120, 10
    rng = np.random.default_rng(nprobe * 7 + nq)
    cc = rng.integers(0, 4, size=(nlist, d)).astype(np.float32)
    base = rng.integers(0, 4, size=(N, d)).astype(np.float32) + rng.standard_normal((N, d)).astype(np.float32) * 0.05
    pq = rng.standard_normal((M, 256, d // M)).astype(np.float32) * 0.3
    q = rng.integers(0, 4, size=(nq, d)).astype(np.float32)
    B.lib().go_set_assign_mode(1)
    o = B.OracleIVFPQ(d, nlist, M, 8, B.METRIC_L2)
    o.set_trained(cc, pq, None)
    assert o.add(base)
    B.lib().go_set_assign_mode(0)
    o.set_raw(base)
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2)
        g.ivfpq_set_trained(cc, pq, None)
        g.raw_init(d)
        g.raw_append(base)
        g.add(base, 0)
        g.set_exact_ties(True)
        omode = coarse_mode if coarse_mode >= 0 else (1 if nq >= 20 else 0)
        D, I, st = o.search(q, k, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=B.make_ctx(**WIDE),
                            coarse_mode=omode, want_stages=True)
        # the data does what the test is for: rows with equal keys across the nprobe cut or among the probed
        full, _ = B.knn_L2sqr(q[:64], cc, nprobe + 1, mode=omode)
        assert (full[:, 1:] == full[:, :-1]).any(axis=1).mean() > 0.9
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=nprobe, recall_num=R, has_rank=True, coarse_mode=coarse_mode, **WIDE)
        g.tie_stats(reset=True)
        Dg, Ig = g.ivfpq_search(q, k, args)
        sg = g.last_stages(nq, nprobe, R)
        from tests.parity import compare_search_exact
        compare_search_exact(D, I, st, Dg, Ig, sg)
        assert g.tie_stats()["coarse_rows"] > 0
        assert g.ties_not_honoured() == 0
    finally:
        g.close()
