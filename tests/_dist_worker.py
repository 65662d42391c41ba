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
        o.set_raw(case["base"])
        self.o = o

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype)

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

    def merge_rerank(self, all_dis, all_ids, x, k, args, nql, D, I):
        p = args.p
        R = max(p.recall_num, k)
        ks = 1 if p.metric == B.METRIC_L2 else 0
        L = B.lib()
        ad, ai = all_dis.numpy(), all_ids.numpy()
        base, d = self.case["base"], self.case["d"]
        for qi in range(nql):
            dis = ad[:, qi, :].reshape(-1)
            ids = ai[:, qi, :].reshape(-1)
            keep = ids >= 0
            dis, ids = dis[keep], ids[keep]
            order = np.argsort(dis if ks else -dis, kind="stable")[:R]
            ids = ids[order]
            xq = np.ascontiguousarray(x[qi].numpy())
            fn = L.go_fvec_L2sqr if ks else L.go_fvec_inner_product
            ex = np.array([fn(B._fp(xq), B._fp(np.ascontiguousarray(base[i])), d) for i in ids],
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
    # two and three interleaved sub-batches (asynchronous collectives, padded last slices), then the
    # plain single pass
    for pipeline in (2, 3, None):
        D, I = gdist.sharded_search(be, x, k, args, pipeline=pipeline)
        compare_topk(Dr, Ir, D.numpy(), I.numpy())
    # every rank holds the full, identical result
    gathered = [torch.empty_like(I) for _ in range(world)]
    dist.all_gather(gathered, I)
    for g in gathered:
        assert torch.equal(g, I)
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
