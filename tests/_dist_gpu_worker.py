"""RCCL worker (one process per GPU): gamma_amd.dist.sharded_search with the HIP backend against
the unsharded device search of a handle that holds every list."""
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
from tests.parity import compare_exact  # noqa: E402


def main():
    backend = os.environ.get("GAMMA_TEST_BACKEND", "nccl")
    # gloo: several ranks share GPU 0 (RCCL refuses that) -- the multi-rank path with real exchanges on a
    # single-GPU box; the collectives then run through gloo on the same CUDA tensors
    local = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, world)
    full = fixtures.load_hip(case, device=local)
    g = api.GammaHip(local)
    g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
    g.ivfpq_set_trained(case["cc"], case["pq"], None)
    lists, counts, vids, codes = [], [], [], []
    for l in range(case["nlist"]):
        if owner[l] == rank:
            ids, cds = case["oracle"].get_list(l)
            if len(ids):
                lists.append(l)
                counts.append(len(ids))
                vids.append(ids)
                codes.append(cds)
    g.add_keys_batch(lists, counts, np.concatenate(vids), np.concatenate(codes))
    g.raw_init(case["d"])
    g.raw_append(case["base"])
    be = gdist.HipShardBackend(g, local)
    k, P, R, nq = 10, 8, 100, 61
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=-3e38,
                          max_score=3e38, coarse_mode=1)
    x = torch.from_numpy(case["q"][:nq]).to(dev)
    Dref = torch.empty((nq, k), dtype=torch.float32, device=dev)
    Iref = torch.empty((nq, k), dtype=torch.int64, device=dev)
    full.ivfpq_search_device(x.data_ptr(), nq, k, args, Dref.data_ptr(), Iref.data_ptr())
    full.synchronize()
    # single pass (twice: the second call reuses workspaces), then 2 and 3 interleaved sub-batches
    for pipeline in (None, None, 2, 3, 2):
        D, I = gdist.sharded_search(be, x, k, args, pipeline=pipeline)
        torch.cuda.synchronize()
        compare_exact(Dref.cpu().numpy(), Iref.cpu().numpy(), D.cpu().numpy(), I.cpu().numpy())
    # query-parallel over replicated lists: every rank answers its slice on the whole index (the single-handle path,
    # exact ties included), one all-gather of the results
    D, I = gdist.replicated_search(gdist.HipShardBackend(full, local), x, k, args)
    torch.cuda.synchronize()
    assert D.cpu().numpy().tobytes() == Dref.cpu().numpy().tobytes() and np.array_equal(I.cpu().numpy(), Iref.cpu().numpy())
    # the DEFAULT coarse_mode with 20 <= nq < 20 * world: faiss decides on the size of the whole call (GEMM form here), so
    # the slices -- each below 20 queries -- must not fall back to the exact form by themselves
    args_def = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=-3e38, max_score=3e38,
                              coarse_mode=-1)
    nd = min(nq, 20 + 9 * (world - 1))
    xd = x[:nd].contiguous()
    Dd = torch.empty((nd, k), dtype=torch.float32, device=dev)
    Id = torch.empty((nd, k), dtype=torch.int64, device=dev)
    full.ivfpq_search_device(xd.data_ptr(), nd, k, args_def, Dd.data_ptr(), Id.data_ptr())
    full.synchronize()
    D, I = gdist.replicated_search(gdist.HipShardBackend(full, local), xd, k, args_def)
    torch.cuda.synchronize()
    assert args_def.p.coarse_mode == -1
    assert D.cpu().numpy().tobytes() == Dd.cpu().numpy().tobytes() and np.array_equal(I.cpu().numpy(), Id.cpu().numpy())
    rs_d = gdist.ReplicatedStream(gdist.HipShardBackend(full, local), k, args_def)
    rs_d.submit(xd)
    Ds, Is = rs_d.flush()
    torch.cuda.synchronize()
    assert Ds.cpu().numpy().tobytes() == Dd.cpu().numpy().tobytes() and np.array_equal(Is.cpu().numpy(), Id.cpu().numpy())
    rs_d.close()
    # a stream of batches with the deferred tie replay: gathered one batch behind, every batch bit for bit the unsharded
    # handle's answer (tie-heavy queries included: x holds base vectors, whose nearest neighbours are at distance 0 ...)
    rs = gdist.ReplicatedStream(gdist.HipShardBackend(full, local), k, args)
    batches = [x, x[:7].contiguous(), x[3:].contiguous(), x]
    refs = []
    for xb in batches:
        Dx = torch.empty((xb.shape[0], k), dtype=torch.float32, device=x.device)
        Ix = torch.empty((xb.shape[0], k), dtype=torch.int64, device=x.device)
        full.ivfpq_search_device(xb.data_ptr(), xb.shape[0], k, args, Dx.data_ptr(), Ix.data_ptr())
        full.synchronize()
        refs.append((Dx.cpu().numpy(), Ix.cpu().numpy()))
    outs = []
    def keep(o):   # results are ready in the order of the backend's stream (as replicated_search's): wait, then copy
        torch.cuda.synchronize()
        return None if o is None else (o[0].clone(), o[1].clone())
    for xb in batches:
        outs.append(keep(rs.submit(xb)))
    outs.append(keep(rs.flush()))
    torch.cuda.synchronize()
    assert outs[0] is None
    for (Dx, Ix), (Ds, Is) in zip(refs, outs[1:]):
        assert Ds.cpu().numpy().tobytes() == Dx.tobytes() and np.array_equal(Is.cpu().numpy(), Ix)
    rs.close()
    # Add with ONE encode per batch (sharded_add: the encoding rank broadcasts list numbers + codes): the shards' lists are
    # those of the unsharded handle after the same Add, entry by entry
    from gamma_amd import synth
    extra = synth.sift_like(3000, d=case["d"], seed=77)
    v0 = len(case["base"])
    for t, i0 in enumerate(range(0, len(extra), 1000)):
        gdist.sharded_add(be, extra[i0:i0 + 1000], v0 + i0, (np.asarray(owner) == rank).astype(np.uint8), turn=t)
        full.raw_append(extra[i0:i0 + 1000])
        full.add(extra[i0:i0 + 1000], v0 + i0)
    for l in range(case["nlist"]):
        if owner[l] == rank:
            ia, ca = g.get_list(l)
            ib, cb = full.get_list(l)
            assert np.array_equal(ia, ib) and np.array_equal(ca, cb), l
    # exact ties across the ranks' shards: tie-heavy data (every base vector four times), labels strictly the pinned
    # oracle's on the unsharded index (gamma_amd.dist.tie_phase: flagged queries broadcast, candidate streams exported by
    # every rank, gathered, replayed by the slice's owner)
    from tests.test_oracle_golden import load_ties
    z, o, base, metric = load_ties("l2")
    sizes = z["list_sizes_l2"]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    own = gdist.balance_lists(sizes, world)
    gt = api.GammaHip(local)
    gt.ivfpq_init(int(z["d"]), int(z["nlist"]), int(z["M"]), 8, metric)
    gt.ivfpq_set_trained(z["cc_l2"], z["pq_l2"], None)
    ls = [l for l in range(int(z["nlist"])) if own[l] == rank and sizes[l]]
    gt.add_keys_batch(ls, [int(sizes[l]) for l in ls], np.concatenate([z["list_ids_l2"][offs[l]:offs[l + 1]] for l in ls]),
                      np.concatenate([z["list_codes_l2"][offs[l]:offs[l + 1]] for l in ls]))
    gt.set_list_mask((np.asarray(own) == rank).astype(np.uint8))
    gt.raw_init(int(z["d"]))
    gt.raw_append(base)
    bt = gdist.HipShardBackend(gt, local)
    wide = dict(min_score=-3e38, max_score=3e38)
    for has_rank, reps in ((True, 1), (True, 9), (False, 9)):
        nprobe, R, kk = 12, 60, 10
        D1, I1 = o.search(z["q"], kk, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=B.make_ctx(**wide), coarse_mode=0)
        qh = np.tile(z["q"], (reps, 1))[:len(z["q"]) * reps - (reps > 1)]
        at = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **wide)
        xt = torch.from_numpy(qh).to(dev)
        for pipeline in (None, 2):
            Dt, It = gdist.sharded_search(bt, xt, kk, at, pipeline=pipeline)
            torch.cuda.synchronize()
            compare_exact(np.tile(D1, (reps, 1))[:len(qh)], np.tile(I1, (reps, 1))[:len(qh)], Dt.cpu().numpy(), It.cpu().numpy())
    # ---- raw vectors SHARDED with their lists (round 6): every rank keeps the rows of its lists only, the exact distances
    #      travel with the candidates (packed exchange) and with the tie phase's exported streams ----
    # (a) tie-heavy data, rows by gamma_hip_raw_put; labels strictly the pinned oracle's
    gs = api.GammaHip(local)
    gs.ivfpq_init(int(z["d"]), int(z["nlist"]), int(z["M"]), 8, metric)
    gs.ivfpq_set_trained(z["cc_l2"], z["pq_l2"], None)
    gs.add_keys_batch(ls, [int(sizes[l]) for l in ls], np.concatenate([z["list_ids_l2"][offs[l]:offs[l + 1]] for l in ls]),
                      np.concatenate([z["list_codes_l2"][offs[l]:offs[l + 1]] for l in ls]))
    own_mask = (np.asarray(own) == rank).astype(np.uint8)
    gs.set_list_mask(own_mask)
    gs.raw_init(int(z["d"]))
    mine = np.unique(np.concatenate([z["list_ids_l2"][offs[l]:offs[l + 1]] for l in ls]) & 0x7fffffffffffffff)
    gs.raw_put(mine, base[mine])
    assert gs.raw_stats()["rows"] == len(mine) and (len(mine) < len(base) or world == 1)
    bs = gdist.HipShardBackend(gs, local, raw_sharded=True, owned=own_mask)
    for has_rank, reps in ((True, 1), (True, 9), (False, 9)):
        nprobe, R, kk = 12, 60, 10
        D1, I1 = o.search(z["q"], kk, nprobe, recall_num=R, has_rank=has_rank, metric=metric, ctx=B.make_ctx(**wide), coarse_mode=0)
        qh = np.tile(z["q"], (reps, 1))[:len(z["q"]) * reps - (reps > 1)]
        at = api.SearchArgs(metric=metric, nprobe=nprobe, recall_num=R, has_rank=has_rank, coarse_mode=0, **wide)
        xt = torch.from_numpy(qh).to(dev)
        for pipeline, env in ((None, {}), (2, {}), (None, {"GAMMA_DIST_PACKED": "0"}), (None, {"GAMMA_DIST_TWO_PHASE": "0"})):
            os.environ.update(env)
            Dt, It = gdist.sharded_search(bs, xt, kk, at, pipeline=pipeline)
            torch.cuda.synchronize()
            for kx in env:
                os.environ.pop(kx)
            compare_exact(np.tile(D1, (reps, 1))[:len(qh)], np.tile(I1, (reps, 1))[:len(qh)], Dt.cpu().numpy(), It.cpu().numpy())
    # (b) the product's Add under the rank's list mask keeps entries AND rows of its own lists (HipShardBackend.add): the
    #     sharded result is the unsharded handle's, which holds every row
    ga = api.GammaHip(local)
    ga.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
    ga.ivfpq_set_trained(case["cc"], case["pq"], None)
    owned_c = (np.asarray(owner) == rank).astype(np.uint8)
    ga.set_list_mask(owned_c)
    ga.raw_init(case["d"])
    ba = gdist.HipShardBackend(ga, local, raw_sharded=True, owned=owned_c)
    allv = np.concatenate([case["base"], extra])
    fa = api.GammaHip(local)
    fa.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
    fa.ivfpq_set_trained(case["cc"], case["pq"], None)
    fa.raw_init(case["d"])
    for i0 in range(0, len(allv), 4000):
        ba.add(allv[i0:i0 + 4000], i0)
        fa.raw_append(allv[i0:i0 + 4000])
        fa.add(allv[i0:i0 + 4000], i0)
    tot = torch.tensor([ga.raw_stats()["rows"]], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(tot)
    assert int(tot.item()) == len(allv) and (ga.raw_stats()["rows"] < len(allv) or world == 1)
    xq = torch.from_numpy(synth.sift_like(700, d=case["d"], seed=78)).to(dev)
    a2 = api.SearchArgs(metric=api.METRIC_L2, nprobe=16, recall_num=100, has_rank=True, min_score=-3e38, max_score=3e38, coarse_mode=1)
    Dr2 = torch.empty((700, k), dtype=torch.float32, device=dev)
    Ir2 = torch.empty((700, k), dtype=torch.int64, device=dev)
    fa.ivfpq_search_device(xq.data_ptr(), 700, k, a2, Dr2.data_ptr(), Ir2.data_ptr())
    fa.synchronize()
    D, I = gdist.sharded_search(ba, xq, k, a2)
    torch.cuda.synchronize()
    compare_exact(Dr2.cpu().numpy(), Ir2.cpu().numpy(), D.cpu().numpy(), I.cpu().numpy())
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
