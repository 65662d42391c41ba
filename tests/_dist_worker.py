"""world_size-2 gloo worker: runs gamma_amd.dist.sharded_search with an oracle-backed shard
backend on CPU tensors and checks it against the unsharded oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from gamma_amd import api  # noqa: E402
from gamma_amd import dist as gdist  # noqa: E402
from oracle import binding as B  # noqa: E402
from tests import fixtures  # noqa: E402
from tests.parity import compare_topk  # noqa: E402


class OracleShardBackend:
    def __init__(self, case, owner, rank):
        self.case = case
        o = B.OracleIVFPQ(case["d"], case["nlist"], case["M"], 8, case["metric"])
        o.set_trained(case["cc"], case["pq"], None)
        for l in range(case["nlist"]):
            if owner[l] == rank:
                ids, codes = case["oracle"].get_list(l)
                if len(ids):
                    o.add_keys(l, ids, codes)
        self.raw = case["base"].copy()
        o.set_raw(self.raw)
        self.o = o

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

    # ---- sharded_update ----
    class _Store:
        def __init__(self, o):
            self.o = o
            self.has_vid, self.remove, self.add_keys = o.has_vid, o.remove, o.add_keys

        def update(self, lno, vid, code):
            self.o.update_code(lno, vid, code)

    def has_vid(self, vids):
        return torch.from_numpy(self.o.has_vid(vids))

    def update_one(self, vid, vec, owned, held_somewhere):
        self.raw[vid] = vec
        lno, code = self.o.encode(vec[None])
        gdist.route_update(self._Store(self.o), int(lno[0]), vid, code[0], owned, held_somewhere)

    def compact_if_need(self):
        self.o.compact_if_need()

    def coarse(self, x, args, cdis, probe):
        if x.shape[0] == 0:
            return
        p = args.p
        Dc, Ic = B.knn_L2sqr(x.numpy(), self.case["cc"], p.nprobe, mode=0)
        cdis[:x.shape[0]] = torch.from_numpy(Dc)
        probe[:x.shape[0]] = torch.from_numpy(Ic.astype(np.int32))

    def search_shard(self, x, cdis, probe, k, args, rdis, rids):
        p = args.p
        ctx = B.make_ctx(min_score=p.min_score, max_score=p.max_score)
        _, _, st = self.o.search(x.numpy(), k, p.nprobe, recall_num=p.recall_num, has_rank=False,
                                 metric=p.metric, ctx=ctx, coarse_mode=0, want_stages=True)
        # the oracle recomputes the (deterministic) coarse assignment: it must be what the
        # owning ranks computed and the all-gather delivered
        assert np.array_equal(st["coarse_idx"], probe.numpy().astype(np.int64))
        assert st["coarse_dis"].tobytes() == cdis.numpy().tobytes()
        rdis[:] = torch.from_numpy(st["recall_dis"])
        rids[:] = torch.from_numpy(st["recall_ids"])

    G1 = 2
    n_tightened = 0

    def search_shard_bounded(self, x, cdis, probe, k, args, rdis, rids, bound, reduce):
        """the two phases of gamma_hip_ivfpq_search_shard_bounded restated on the oracle: the bound of this shard's
        recall_num-th best from the query's nearest G1 probes it owns, the caller's reduction, the local candidates
        within the reduced bound (best first, padded)"""
        p = args.p
        R = max(p.recall_num, k)
        l2 = p.metric == B.METRIC_L2
        ctx = B.make_ctx(min_score=p.min_score, max_score=p.max_score)
        xs, pr, cd = x.numpy(), probe.numpy().astype(np.int64), cdis.numpy()
        nq, P = pr.shape
        own = np.array([self.o.list_size(l) > 0 for l in range(self.case["nlist"])])
        p1, c1 = np.full((nq, P), -1, np.int64), np.zeros((nq, P), np.float32)
        for qi in range(nq):
            mine = [j for j in range(P) if pr[qi, j] >= 0 and own[pr[qi, j]]][:self.G1]
            p1[qi, :len(mine)] = pr[qi, mine]
            c1[qi, :len(mine)] = cd[qi, mine]
        _, _, s1 = self.o.search(xs, k, P, recall_num=R, has_rank=False, metric=p.metric, ctx=ctx, want_stages=True,
                                 preassigned=(c1, p1))
        full = s1["recall_ids"][:, R - 1] >= 0
        none = np.float32(np.inf if l2 else -np.inf)
        bound[:] = torch.from_numpy(np.where(full, s1["recall_dis"][:, R - 1], none).astype(np.float32))
        own_bound = bound.numpy().copy()
        reduce(not l2)
        _, _, st = self.o.search(xs, k, P, recall_num=R, has_rank=False, metric=p.metric, ctx=ctx, want_stages=True,
                                 preassigned=(cd, pr))
        self.n_tightened += int((bound.numpy() != own_bound).sum())
        b = bound.numpy()[:, None]
        rd, ri = st["recall_dis"].copy(), st["recall_ids"].copy()
        drop = (ri < 0) | ((rd > b) if l2 else (rd < b))
        rd[drop] = np.float32(3.4028235e38 if l2 else -3.4028235e38)
        ri[drop] = -1
        rdis[:] = torch.from_numpy(rd)
        rids[:] = torch.from_numpy(ri)

    def merge_rerank(self, all_dis, all_ids, x, k, args, nql, D, I):
        p = args.p
        R = max(p.recall_num, k)
        ks = 1 if p.metric == B.METRIC_L2 else 0
        L = B.lib()
        ad, ai = all_dis.numpy(), all_ids.numpy()
        d = self.case["d"]
        for qi in range(nql):
            dis = ad[:, qi, :].reshape(-1)
            ids = ai[:, qi, :].reshape(-1)
            keep = ids >= 0
            dis, ids = dis[keep], ids[keep]
            order = np.argsort(dis if ks else -dis, kind="stable")[:R]
            ids = ids[order]
            xq = np.ascontiguousarray(x[qi].numpy())
            fn = L.go_fvec_L2sqr if ks else L.go_fvec_inner_product
            ex = np.array([fn(B._fp(xq), B._fp(np.ascontiguousarray(self.raw[i])), d) for i in ids],
                          dtype=np.float32)
            ov, oi = np.empty(k, np.float32), np.empty(k, np.int64)
            ids = np.ascontiguousarray(ids)
            L.go_heap_pop_push_stream(ks, k, len(ids), B._fp(ex), B._ip(ids), B._fp(ov), B._ip(oi))
            D[qi] = torch.from_numpy(ov)
            I[qi] = torch.from_numpy(oi)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=8000, nq=33, metric=B.METRIC_L2)
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, world)
    be = OracleShardBackend(case, owner, rank)
    k, nprobe, R = 10, 8, 60
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=nprobe, recall_num=R, has_rank=True,
                          min_score=-3e38, max_score=3e38, coarse_mode=0)
    x = torch.from_numpy(case["q"])
    ctx = B.make_ctx(min_score=-3e38, max_score=3e38)
    Dr, Ir = case["oracle"].search(case["q"], k, nprobe, recall_num=R, has_rank=True,
                                   metric=B.METRIC_L2, ctx=ctx, coarse_mode=0)
    Dr0 = Dr.copy()
    # two and three interleaved sub-batches (asynchronous collectives, padded last slices), then the
    # plain single pass
    for pipeline in (2, 3, None):
        D, I = gdist.sharded_search(be, x, k, args, pipeline=pipeline)
        compare_topk(Dr, Ir, D.numpy(), I.numpy())
    # the packed exchange carries what lies within the (tightened) global bound: about recall_num entries per query over
    # ALL shards, not world x recall_num; the plain all-to-all of whole tables gives the same result
    be.exchange_stats = {}
    D, I = gdist.sharded_search(be, x, k, args, pipeline=1)
    compare_topk(Dr, Ir, D.numpy(), I.numpy())
    st = be.exchange_stats
    per_q = st["exchange_entries"] / float(st["queries"])
    assert per_q < 1.3 * R * (world - 1) / world, (per_q, R, world)     # this rank's share of ~ R (+ one edge's worth) per query
    os.environ["GAMMA_DIST_PACKED"] = "0"
    be.exchange_stats = {}
    D, I = gdist.sharded_search(be, x, k, args, pipeline=1)
    compare_topk(Dr, Ir, D.numpy(), I.numpy())
    assert be.exchange_stats["exchange_entries"] >= (world - 1) * R * (len(x) // world)
    os.environ.pop("GAMMA_DIST_PACKED")
    be.exchange_stats = None
    # the single-phase shard scan (every shard against its own bound) gives the same table
    os.environ["GAMMA_DIST_TWO_PHASE"] = "0"
    D, I = gdist.sharded_search(be, x, k, args)
    compare_topk(Dr, Ir, D.numpy(), I.numpy())
    os.environ.pop("GAMMA_DIST_TWO_PHASE")
    # the reduction did something: some query's global bound is tighter than this shard's own
    assert be.n_tightened > 0 or world == 1
    # query-parallel over replicated lists: every rank answers its slice on the whole index, one all-gather
    class Full:
        def empty(self, shape, dtype):
            return torch.empty(shape, dtype=dtype)

        def search_all(self, xs, kk, a, Dd, Ii):
            Ds, Is = case["oracle"].search(xs.numpy(), kk, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx,
                                           coarse_mode=0)
            Dd.copy_(torch.from_numpy(Ds))
            Ii.copy_(torch.from_numpy(Is))
    Dp, Ip = gdist.replicated_search(Full(), x, k, args)
    assert Dp.numpy().tobytes() == Dr.tobytes() and np.array_equal(Ip.numpy(), Ir)
    # the same for a stream of batches, gathered one batch behind (ReplicatedStream: what runs with the deferred tie
    # replay on the device); batch sizes that do and do not divide by the world size
    rs = gdist.ReplicatedStream(Full(), k, args)
    batches = [x, x[:5], x[3:], x]
    outs = [rs.submit(xb) for xb in batches] + [rs.flush()]
    assert outs[0] is None and rs.flush() is None
    for xb, (Ds, Is) in zip(batches, outs[1:]):
        De, Ie = case["oracle"].search(xb.numpy(), k, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx,
                                       coarse_mode=0)
        assert Ds.numpy().tobytes() == De.tobytes() and np.array_equal(Is.numpy(), Ie)
    rs.close()
    # the default coarse_mode (-1) is resolved on the size of the WHOLE batch before the slices are searched (faiss's
    # 20-query rule, faiss:utils/distances.cpp:303,346) and restored afterwards: 20 <= nq < 20 * world
    seen = []

    class Spy(Full):
        def search_all(self, xs, kk, a, Dd, Ii):
            seen.append((xs.shape[0], a.p.coarse_mode))
            Full.search_all(self, xs, kk, a, Dd, Ii)
    a_def = api.SearchArgs(metric=api.METRIC_L2, nprobe=nprobe, recall_num=R, has_rank=True, min_score=-3e38, max_score=3e38,
                           coarse_mode=-1)
    nd = 20 + 4 * (world - 1)
    assert nd <= x.shape[0]
    gdist.replicated_search(Spy(), x[:nd], k, a_def)
    rs2 = gdist.ReplicatedStream(Spy(), k, a_def)
    rs2.submit(x[:nd])
    rs2.submit(x[:7])
    rs2.flush()
    rs2.close()
    assert a_def.p.coarse_mode == -1
    assert [m for n_, m in seen] == [1, 1, 0] and all(n_ < 20 for n_, m in seen), seen
    # every rank holds the full, identical result
    gathered = [torch.empty_like(I) for _ in range(world)]
    dist.all_gather(gathered, I)
    for g in gathered:
        assert torch.equal(g, I)
    # Update across shards: every rank is handed the same (vids, vectors); most of them change list, many change
    # shard; one vid is updated twice in the batch, one was never added (ignored, realtime_mem_data.cc:307-311)
    rng = np.random.default_rng(99)
    N = len(case["base"])
    vids = rng.choice(N, 300, replace=False)
    vids = np.concatenate([vids, vids[:1], [N + 5]])
    newv = case["base"][rng.integers(0, N, len(vids))] + rng.integers(-3, 4, (len(vids), case["d"])).astype(np.float32)
    owned = (owner == rank).astype(np.uint8)
    gdist.sharded_update(be, vids, newv, owned)
    ref = B.OracleIVFPQ(case["d"], case["nlist"], case["M"], 8, case["metric"])
    ref.set_trained(case["cc"], case["pq"], None)
    for l in range(case["nlist"]):
        ids, codes = case["oracle"].get_list(l)
        if len(ids):
            ref.add_keys(l, ids, codes)
    raw2 = case["base"].copy()
    for v, nv in zip(vids, newv):
        if v < N:
            raw2[v] = nv
        ref.update(int(v), nv)
    ref.set_raw(raw2)
    live = sum(int(be.o.has_vid([v])[0]) for v in range(N))
    tot = torch.tensor([live])
    dist.all_reduce(tot)
    assert int(tot) == N, int(tot)            # every vid lives on exactly one shard afterwards
    Dr, Ir = ref.search(case["q"], k, nprobe, recall_num=R, has_rank=True, metric=B.METRIC_L2, ctx=ctx, coarse_mode=0)
    assert not np.array_equal(Dr, Dr0)        # the updates are visible
    D, I = gdist.sharded_search(be, x, k, args)
    compare_topk(Dr, Ir, D.numpy(), I.numpy())
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
