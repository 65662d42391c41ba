"""GPU tests of the list-sharded path (gamma_amd/dist.py + the shard entry points of the C ABI).
Two shards are emulated on ONE GPU with two handles, the exchange done by tensor indexing; a
second test runs the real orchestration over RCCL with world_size 1."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import binding as B
from tests import fixtures
from tests.parity import compare_exact

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard_handle(case, owner, s, raw_sharded=False):
    from gamma_amd import api
    g = api.GammaHip(0)
    g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
    g.ivfpq_set_trained(case["cc"], case["pq"], None)
    lists, counts, vids, codes = [], [], [], []
    for l in range(case["nlist"]):
        if owner[l] != s:
            continue
        ids, cds = case["oracle"].get_list(l)
        if len(ids):
            lists.append(l)
            counts.append(len(ids))
            vids.append(ids)
            codes.append(cds)
    g.add_keys_batch(lists, counts, np.concatenate(vids), np.concatenate(codes))
    g.raw_init(case["d"])
    if raw_sharded:   # the rows of this shard's vectors only, in list order (any order will do)
        mine = np.concatenate(vids)
        for i0 in range(0, len(mine), 3000):   # several puts: the vid -> row table grows and is patched
            g.raw_put(mine[i0:i0 + 3000], case["base"][mine[i0:i0 + 3000]])
    else:
        g.raw_append(case["base"])
    return g


def _sharded_vs_full(case, shards, full, metric, has_rank, W, nq, P, raw_sharded=False):
    import torch
    from gamma_amd import api
    from gamma_amd import dist as gdist
    k, R = 10, 100                            # nq is not a multiple of W: padded slices
    args = api.SearchArgs(metric=metric, nprobe=P, recall_num=R, has_rank=has_rank, min_score=-3e38,
                          max_score=3e38, coarse_mode=1)
    dev = torch.device("cuda", 0)
    from gamma_amd import synth
    qh = case["q"][:nq] if nq <= len(case["q"]) else synth.sift_like(nq, d=case["d"], seed=777)
    x = torch.from_numpy(qh).to(dev)
    Dref = torch.empty((nq, k), dtype=torch.float32, device=dev)
    Iref = torch.empty((nq, k), dtype=torch.int64, device=dev)
    full.ivfpq_search_device(x.data_ptr(), nq, k, args, Dref.data_ptr(), Iref.data_ptr())
    full.synchronize()
    from tests.shard_emul import sharded_search_emulated
    D, I, _ = sharded_search_emulated(shards, x, k, args, raw_sharded=raw_sharded)      # incl. the tie phase across the shards
    compare_exact(Dref.cpu().numpy(), Iref.cpu().numpy(), D[:nq].cpu().numpy(), I[:nq].cpu().numpy())


@pytest.mark.parametrize("metric,has_rank,W,nq,P,d,M", [
    (B.METRIC_L2, True, 2, 61, 8, 32, 8), (B.METRIC_L2, False, 2, 61, 8, 32, 8), (B.METRIC_IP, True, 2, 61, 8, 32, 8),
    # enough queries x probes for 4 probes per scan workgroup: compacted probe lists + threshold
    # pre-filter inside every shard
    (B.METRIC_L2, True, 4, 603, 32, 32, 8), (B.METRIC_L2, False, 3, 603, 32, 32, 8),
    (B.METRIC_IP, True, 4, 603, 32, 32, 8),
    # enough queries that ONE workgroup takes all of a query's probes on the shard (bound from its own
    # candidates, no consumers); with M = 16 / 32 the query table is computed inside the scan
    (B.METRIC_L2, True, 4, 4200, 32, 64, 16), (B.METRIC_IP, True, 3, 4200, 32, 64, 16),
    (B.METRIC_L2, False, 4, 4200, 32, 64, 32), (B.METRIC_L2, True, 2, 4200, 16, 32, 8),
    # the reference's default nprobe (80) and beyond 128 probes (no pre-filter)
    (B.METRIC_L2, True, 4, 700, 80, 32, 8), (B.METRIC_IP, True, 2, 4200, 80, 32, 8), (B.METRIC_L2, True, 3, 300, 150, 32, 8),
])
def test_shards_on_one_gpu(metric, has_rank, W, nq, P, d, M):
    import torch
    from gamma_amd import api
    from gamma_amd import dist as gdist
    case = fixtures.trained_case(d=d, nlist=64 if P <= 64 else 160, M=M, N=20000, nq=64, metric=B.METRIC_L2)
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, W)
    full = fixtures.load_hip(case)
    shards = [_shard_handle(case, owner, s) for s in range(W)]
    _sharded_vs_full(case, shards, full, metric, has_rank, W, nq, P)
    for g in shards + [full]:
        g.close()


@pytest.mark.parametrize("metric,W,nq,P,d,M,two_phase", [
    (B.METRIC_L2, 2, 61, 8, 32, 8, True), (B.METRIC_L2, 3, 603, 32, 32, 8, True), (B.METRIC_IP, 2, 603, 32, 32, 8, True),
    (B.METRIC_L2, 4, 4200, 32, 64, 16, True), (B.METRIC_L2, 3, 4200, 16, 32, 8, False), (B.METRIC_IP, 3, 4200, 32, 64, 16, True),
])
def test_shards_with_their_own_raw_rows(metric, W, nq, P, d, M, two_phase, monkeypatch):
    """Raw vectors sharded with their lists (round 6): every shard holds the rows of ITS lists only, the exact
    distances of compute_dis are computed where the row is and travel with the candidates; results are those of ONE handle
    that holds everything -- labels and distance bits at every rank, the tie phase included.  And what needs every row
    refuses on such a handle."""
    import torch
    from gamma_amd import api
    from gamma_amd import dist as gdist
    monkeypatch.setenv("GAMMA_TEST_TWO_PHASE", "1" if two_phase else "0")
    case = fixtures.trained_case(d=d, nlist=64, M=M, N=20000, nq=64, metric=B.METRIC_L2)
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, W)
    full = fixtures.load_hip(case)
    shards = [_shard_handle(case, owner, s, raw_sharded=True) for s in range(W)]
    try:
        rows = sum(g.raw_stats()["rows"] for g in shards)
        assert rows == 20000 and all(g.raw_stats()["rows"] < 20000 * 0.75 for g in shards)
        _sharded_vs_full(case, shards, full, metric, True, W, nq, P, raw_sharded=True)
        _sharded_vs_full(case, shards, full, metric, False, W, min(nq, 603), P, raw_sharded=True)   # no re-rank: nothing travels
        # never silent: a search that would re-rank from the handle's own store, flat search, row reads
        a = api.SearchArgs(metric=metric, nprobe=P, recall_num=100, has_rank=True, min_score=-3e38, max_score=3e38)
        with pytest.raises(api.GammaHipError, match="raw rows only|rows only"):
            shards[0].ivfpq_search(case["q"][:300], 10, a)
        with pytest.raises(api.GammaHipError, match="rows only"):
            shards[0].flat_search(case["q"][:4], 10, a)
        with pytest.raises(api.GammaHipError, match="rows only"):
            shards[0].raw_append(case["base"][:2])
    finally:
        for g in shards + [full]:
            g.close()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_real_ranks_sharing_one_gpu(world):
    """The orchestration with REAL ranks (separate processes, each owning a shard, every exchange a real
    collective) on a single-GPU box: the ranks share GPU 0 and the collectives go through gloo on the CUDA
    tensors (RCCL refuses two ranks on one device)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", GAMMA_TEST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29680 + world),
                        os.path.join(ROOT, "tests", "_dist_gpu_worker.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for rk in range(world):
        assert "rank %d ok" % rk in r.stdout


def test_sharded_search_over_rccl_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", "29671",
                        os.path.join(ROOT, "tests", "_dist_gpu_worker.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 0 ok" in r.stdout


@pytest.mark.parametrize("metric,W,R,nql", [
    (B.METRIC_L2, 8, 200, 37), (B.METRIC_IP, 7, 100, 64), (B.METRIC_L2, 2, 256, 5), (B.METRIC_L2, 3, 33, 129),
    (B.METRIC_L2, 16, 200, 9),       # W * R > 2048: the general gather + select path
])
def test_merge_of_shard_tables(metric, W, R, nql):
    """gamma_hip_ivfpq_merge_rerank without re-rank and k = R returns the merged table itself: the R best
    of W*R in (distance, shard, rank) order; rows unsorted, with ties, padding and rows of padding only."""
    import torch
    from gamma_amd import api
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    g = fixtures.load_hip(case)
    rng = np.random.default_rng(W * 1000 + R)
    nq, q0 = nql + 11, 7                       # a slice inside a longer batch
    l2 = metric == B.METRIC_L2
    dis = rng.standard_normal((W, nq, R)).astype(np.float32)
    dis = (np.round(dis * 8) / 8 + 0.0).astype(np.float32) if R > 100 else dis   # many exact ties (+ 0.0: no -0.0, which
                                                                                 # the device key orders before +0.0)
    ids = rng.integers(0, 1 << 40, size=(W, nq, R)).astype(np.int64)
    pad = rng.random((W, nq, R)) < 0.15
    pad[:, q0 + 1] = True                      # a query nobody has candidates for
    pad[0, q0 + 2] = True
    ids[pad] = -1
    dev = torch.device("cuda", 0)
    d_dis, d_ids = torch.from_numpy(dis).to(dev), torch.from_numpy(ids).to(dev)
    x = torch.zeros((nq, case["d"]), dtype=torch.float32, device=dev)
    D = torch.empty((nql, R), dtype=torch.float32, device=dev)
    I = torch.empty((nql, R), dtype=torch.int64, device=dev)
    args = api.SearchArgs(metric=metric, nprobe=8, recall_num=R, has_rank=False, min_score=-3e38, max_score=3e38)
    g.ivfpq_merge_rerank(W, nq, x.data_ptr(), R, args, d_dis.data_ptr(), d_ids.data_ptr(), q0, nql, D.data_ptr(),
                         I.data_ptr())
    g.synchronize()
    D, I = D.cpu().numpy(), I.cpu().numpy()
    for i in range(nql):
        q = q0 + i
        dv = dis[:, q].reshape(-1)
        iv = ids[:, q].reshape(-1)
        e = np.arange(W * R)
        ok = iv >= 0
        order = np.lexsort((e[ok], dv[ok] if l2 else -dv[ok]))[:R]
        want_i = iv[ok][order]
        want_d = dv[ok][order]
        m = len(order)
        assert np.array_equal(I[i, :m], want_i), (i, m)
        assert np.array_equal(D[i, :m], want_d)
        assert np.all(I[i, m:] == -1)
    g.close()


def test_sharded_realtime_inserts_route_to_the_list_owner():
    """realtime inserts go to the GPU that owns the assigned list.  Every shard is handed the same
    Add batches (engine-sized, 1000 vectors) and keeps what it owns (list mask): its lists must equal those of
    an unsharded handle fed the same batches, the other lists stay empty, and the sharded search agrees."""
    from gamma_amd import api
    from gamma_amd import dist as gdist
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    W = 3
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, W)
    base = case["base"]

    def build(mask):
        g = api.GammaHip(0)
        g.ivfpq_init(case["d"], case["nlist"], case["M"], 8, case["metric"], 1000)
        g.ivfpq_set_trained(case["cc"], case["pq"], None)
        if mask is not None:
            g.set_list_mask(mask)
        g.raw_init(case["d"])
        for i0 in range(0, len(base), 1000):
            g.raw_append(base[i0:i0 + 1000])
            g.add(base[i0:i0 + 1000], i0)
        return g

    full = build(None)
    shards = [build((owner == s).astype(np.uint8)) for s in range(W)]
    try:
        for l in range(case["nlist"]):
            ids, codes = full.get_list(l)
            for s in range(W):
                if owner[l] == s:
                    gi, gc = shards[s].get_list(l)
                    assert np.array_equal(ids, gi) and np.array_equal(codes, gc)
                else:
                    assert shards[s].list_size(l) == 0
        _sharded_vs_full(case, shards, full, B.METRIC_L2, True, W, 603, 32)
    finally:
        for g in shards + [full]:
            g.close()


def test_update_across_shards():
    """GammaIVFPQIndex::Update on a list-sharded index (gamma_amd/dist.py sharded_update / route_update): every shard
    is handed the same (vid, vector); the holder of the old entry flags it, the owner of the new list appends.
    Owned lists must equal an unsharded handle's after the same updates, entry for entry, and searches agree."""
    from gamma_amd import api
    from gamma_amd import dist as gdist
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    W = 3
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, W)
    full = fixtures.load_hip(case)
    shards = [_shard_handle(case, owner, s) for s in range(W)]
    backs = [gdist.HipShardBackend(g, 0) for g in shards]
    masks = [(owner == s).astype(np.uint8) for s in range(W)]
    rng = np.random.default_rng(5)
    N = len(case["base"])
    vids = rng.choice(N, 400, replace=False)
    vids = np.concatenate([vids, vids[:2], [N + 9]])          # two vids twice, one never added
    newv = (case["base"][rng.integers(0, N, len(vids))] + rng.integers(-3, 4, (len(vids), case["d"]))).astype(np.float32)
    try:
        moved_shard = 0
        for v, x in zip(vids, newv):
            lno, code = full.encode(x[None])
            before = [int(g.has_vid([v])[0]) for g in shards]
            held = any(before)
            if held:
                full.raw_write(int(v), x[None])
                full.update(int(lno[0]), int(v), code[0])
                for s in range(W):
                    backs[s].update_one(int(v), x, masks[s], True)
                after = [int(g.has_vid([v])[0]) for g in shards]
                assert sum(after) == 1 and after[owner[lno[0]]] == 1
                moved_shard += before != after
            else:
                assert v >= N
        assert moved_shard > 50
        for g in shards + [full]:
            g.compact_if_need()
        for l in range(case["nlist"]):
            ids, codes = full.get_list(l)
            for s in range(W):
                if owner[l] == s:
                    gi, gc = shards[s].get_list(l)
                    assert np.array_equal(ids, gi) and np.array_equal(codes, gc), l
                else:
                    assert shards[s].list_size(l) == 0
        _sharded_vs_full(case, shards, full, B.METRIC_L2, True, W, 603, 32)
    finally:
        for g in shards + [full]:
            g.close()


@pytest.mark.parametrize("workload,extra", [
    ("c4", ["--scale-n", "400000", "--scale-nlist", "1024", "--scale-nq", "1500", "--scale-recall-num", "150"]),
    ("c5", ["--scale-n", "120000", "--scale-nlist", "256", "--scale-nq", "700", "--scale-recall-num", "300", "--insert-seconds", "5",
            "--insert-rate", "4000"]),
    # raw vectors sharded with their lists (round 6): every rank keeps its own rows, exact distances travel with the candidates
    ("c4", ["--scale-n", "400000", "--scale-nlist", "1024", "--scale-nq", "1500", "--scale-recall-num", "150", "--raw-placement", "sharded"]),
    ("c5", ["--scale-n", "120000", "--scale-nlist", "256", "--scale-nq", "700", "--scale-recall-num", "300", "--insert-seconds", "5",
            "--insert-rate", "4000", "--raw-placement", "sharded"]),
])
def test_bench_workload_c4_c5_streamed_two_ranks_on_one_gpu(workload, extra, tmp_path):
    """`bench.py --workload c4|c5 --gpus 2` (bench_scale.py), the entry point of the two 8-GPU configurations, at reduced N
    with both ranks on cuda:0 over gloo: the streamed Add under the ranks' list masks must leave every vector in exactly one
    shard (the job asserts it), the line must carry the placement / exchange / per-rank blocks, and step 0's result table
    must equal, at every rank, what ONE unsharded handle returns for the same device streams (exact ties on: identical
    labels) -- gamma_gpu_cloner.cpp:200-269, faiss:IndexShards.cpp:283-345."""
    import json
    from gamma_amd import api, synth
    dump = str(tmp_path / "step0.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--gpus", "2", "--one-gpu", "--backend",
                        "gloo", "--steps", "2", "--warmup", "1", "--scale-dump", dump] + extra,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    cfg = line["config"]
    assert len(cfg["per_rank"]) == 2 and sum(p["shard_vectors"] for p in cfg["per_rank"]) == int(float(extra[1]))
    sharded_raw = "sharded" in extra
    assert "no RCCL communicator" in cfg["communicator"] and cfg["raw_placement"].startswith("sharded" if sharded_raw else "replicated")
    if sharded_raw:   # each rank holds about half of the rows (+ the 4-byte table): well under the replicated store
        assert all(p["device_gb"] < 0.8 * line["config"]["per_rank"][0]["device_gb"] + 1.0 for p in cfg["per_rank"])
    if workload == "c5":
        ins = cfg["search_during_inserts"]
        assert ins["inserted"] > 0 and "writer_error" not in ins, ins
        assert ins["vectors_in_the_shards_after"] == int(float(extra[1])) + ins["inserted"]
        assert all(v["results_inside_the_filter"] for v in cfg["range_filter"].values())
    # the same index in ONE handle
    z = np.load(dump)
    N = int(float(extra[1]))
    l2 = workload == "c4"
    d, M = (128, 32) if l2 else (768, 64)
    nlist, R = int(extra[3]), int(extra[7])
    g = api.GammaHip(0)
    try:
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2 if l2 else api.METRIC_IP, bucket_init_size=1000)
        g.ivfpq_set_trained(z["cc"], z["pq"], None)
        g.raw_init(d)
        gen = synth.sift_like_device if l2 else synth.embedding_like_device
        for c in range(0, N, 100000):
            xb = gen(min(100000, N - c), d=d, seed=1234, start=c, device="cuda:0").cpu().numpy()
            g.raw_append(xb)
            g.add(xb, c)
        win = dict(min_score=0.0, max_score=1e30) if l2 else dict(min_score=-1e30, max_score=1e30)
        D, I = g.ivfpq_search(z["q"], 10, api.SearchArgs(metric=api.METRIC_L2 if l2 else api.METRIC_IP, nprobe=64, recall_num=R,
                                                         has_rank=True, **win))
        compare_exact(D, I, z["D"], z["I"])
    finally:
        g.close()
