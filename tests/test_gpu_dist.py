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
from tests.parity import compare_topk

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _shard_handle(case, owner, s):
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
    g.raw_append(case["base"])
    return g


@pytest.mark.parametrize("metric,has_rank,W,nq,P", [
    (B.METRIC_L2, True, 2, 61, 8), (B.METRIC_L2, False, 2, 61, 8), (B.METRIC_IP, True, 2, 61, 8),
    # enough queries x probes for 4 probes per scan workgroup: compacted probe lists + threshold
    # pre-filter inside every shard
    (B.METRIC_L2, True, 4, 603, 32), (B.METRIC_L2, False, 3, 603, 32), (B.METRIC_IP, True, 4, 603, 32),
])
def test_shards_on_one_gpu(metric, has_rank, W, nq, P):
    import torch
    from gamma_amd import api
    from gamma_amd import dist as gdist
    case = fixtures.trained_case(d=32, nlist=64, M=8, N=20000, nq=64, metric=B.METRIC_L2)
    sizes = np.array([case["oracle"].list_size(l) for l in range(case["nlist"])])
    owner = gdist.balance_lists(sizes, W)
    full = fixtures.load_hip(case)
    shards = [_shard_handle(case, owner, s) for s in range(W)]
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
    per = (nq + W - 1) // W
    backs = [gdist.HipShardBackend(g, 0) for g in shards]
    cdis = torch.zeros((W * per, P), dtype=torch.float32, device=dev)
    probe = torch.full((W * per, P), -1, dtype=torch.int32, device=dev)
    for s in range(W):
        q0, q1, _ = gdist.query_slice(nq, s, W)
        backs[s].coarse(x[q0:q1], args, cdis[s * per:(s + 1) * per], probe[s * per:(s + 1) * per])
        shards[s].synchronize()
    rd, ri = [], []
    for s in range(W):
        rdis = torch.zeros((W * per, R), dtype=torch.float32, device=dev)
        rids = torch.full((W * per, R), -1, dtype=torch.int64, device=dev)
        backs[s].search_shard(x, cdis[:nq], probe[:nq], k, args, rdis[:nq], rids[:nq])
        shards[s].synchronize()
        rd.append(rdis.view(W, per, R))
        ri.append(rids.view(W, per, R))
    D = torch.zeros((W * per, k), dtype=torch.float32, device=dev)
    I = torch.full((W * per, k), -1, dtype=torch.int64, device=dev)
    for r in range(W):       # what all_to_all delivers to rank r: block r of every shard
        q0, q1, _ = gdist.query_slice(nq, r, W)
        all_dis = torch.stack([rd[s][r] for s in range(W)]).contiguous()
        all_ids = torch.stack([ri[s][r] for s in range(W)]).contiguous()
        backs[r].merge_rerank(all_dis, all_ids, x[q0:q1], k, args, q1 - q0, D[r * per:(r + 1) * per],
                              I[r * per:(r + 1) * per])
        shards[r].synchronize()
    compare_topk(Dref.cpu().numpy(), Iref.cpu().numpy(), D[:nq].cpu().numpy(), I[:nq].cpu().numpy())
    for g in shards + [full]:
        g.close()


def test_sharded_search_over_rccl_world1():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                        "--master-addr", "127.0.0.1", "--master-port", "29671",
                        os.path.join(ROOT, "tests", "_dist_gpu_worker.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rank 0 ok" in r.stdout
