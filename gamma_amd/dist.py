"""List-sharded multi-GPU IVFPQ search.

One process per GPU.  Every rank holds the replicated small state (coarse centroids, PQ
codebooks, delete bitmap, raw vectors for the re-rank) and the inverted lists it OWNS.  A
search batch runs in three steps:

  1. every rank: coarse quantizer for the whole batch (cheap, replicated), scan of the owned
     probed lists only, local top-recall_num per query            (gamma_hip_ivfpq_search_shard)
  2. RCCL all-gather over xGMI of the per-shard (distance, id) tables, nq*R*(4+8) bytes per GPU
     -- the one real exchange step of the path
  3. every rank merges the gathered tables into the global top-recall_num and runs compute_dis
     (re-rank / truncate) for ITS slice of the queries             (gamma_hip_ivfpq_merge_rerank)
     followed by a small all-gather of the [nq_slice, k] results.

The global top-recall_num of the union equals the single-GPU top-recall_num (same ADC
distances, disjoint lists), so results are identical to one GPU up to the order inside exact
ties.  The orchestration is backend-agnostic so the world_size-2 gloo test on CPU exercises
the same code with an oracle-based backend (tests/test_dist_cpu.py).
"""
import numpy as np
import torch
import torch.distributed as dist


def balance_lists(list_sizes, nshards):
    """owner[l] by greedy longest-processing-time on list size: probe popularity follows list
    size, so balancing sum(len) balances scan bytes (l % nshards would not)."""
    list_sizes = np.asarray(list_sizes, dtype=np.int64)
    owner = np.zeros(len(list_sizes), dtype=np.int32)
    load = np.zeros(nshards, dtype=np.int64)
    for l in np.argsort(-list_sizes, kind="stable"):
        s = int(np.argmin(load))
        owner[l] = s
        load[s] += list_sizes[l] + 1
    return owner


def query_slice(nq, rank, world):
    per = (nq + world - 1) // world
    q0 = min(nq, rank * per)
    return q0, min(nq, q0 + per), per


class HipShardBackend:
    """Device backend: tensors live on the GPU, compute goes through the C ABI on the handle's
    stream, which is made torch's current stream so RCCL orders against it."""

    def __init__(self, g, device):
        self.g = g
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.ExternalStream(g.stream(), device=self.device)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def search_shard(self, x, k, args):
        nq = x.shape[0]
        R = max(args.p.recall_num, k)
        rdis = self.empty((nq, R), torch.float32)
        rids = self.empty((nq, R), torch.int64)
        self.g.ivfpq_search_shard(x.data_ptr(), nq, k, args, rdis.data_ptr(), rids.data_ptr())
        return rdis, rids

    def merge_rerank(self, all_dis, all_ids, x, k, args, q0, nql, out_rows):
        W, nq = all_dis.shape[0], all_dis.shape[1]
        D = self.empty((out_rows, k), torch.float32)
        I = self.empty((out_rows, k), torch.int64)
        if nql > 0:
            self.g.ivfpq_merge_rerank(W, nq, x.data_ptr(), k, args, all_dis.data_ptr(),
                                      all_ids.data_ptr(), q0, nql, D.data_ptr(), I.data_ptr())
        return D, I


def sharded_search(backend, x, k, args, group=None, gather_results=True):
    """x: [nq, d] tensor on the backend's device (same on every rank).  Returns (D, I) for all nq
    queries on every rank when gather_results, else this rank's slice."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = x.shape[0]
    stream_ctx = torch.cuda.stream(backend.stream) if hasattr(backend, "stream") else _Null()
    with stream_ctx:
        rdis, rids = backend.search_shard(x, k, args)
        # concatenation along dim 0 == [shard][nq][R] row-major (the layout merge_rerank reads)
        all_dis = backend.empty((world * nq, rdis.shape[1]), rdis.dtype)
        all_ids = backend.empty((world * nq, rids.shape[1]), rids.dtype)
        dist.all_gather_into_tensor(all_dis, rdis, group=group)
        dist.all_gather_into_tensor(all_ids, rids, group=group)
        all_dis = all_dis.view(world, nq, -1)
        all_ids = all_ids.view(world, nq, -1)
        q0, q1, per = query_slice(nq, rank, world)
        D, I = backend.merge_rerank(all_dis, all_ids, x, k, args, q0, q1 - q0, per)
        if not gather_results:
            return D[:q1 - q0], I[:q1 - q0]
        Dall = backend.empty((world * per, k), D.dtype)
        Iall = backend.empty((world * per, k), I.dtype)
        dist.all_gather_into_tensor(Dall, D, group=group)
        dist.all_gather_into_tensor(Iall, I, group=group)
    return Dall[:nq], Iall[:nq]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
