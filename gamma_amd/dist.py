"""List-sharded multi-GPU IVFPQ search.

One process per GPU.  Every rank holds the replicated small state (coarse centroids, PQ
codebooks, delete bitmap, raw vectors for the re-rank) and the inverted lists it OWNS.  The
batch of nq queries is also split into one contiguous slice per rank ("its" queries).  A search
runs in four steps:

  0. every rank: coarse quantizer for ITS query slice                (gamma_hip_ivfpq_coarse_device)
     + all-gather of the (distance, list) assignment, nq*nprobe*8 bytes in total
  1. every rank: query tables for the whole batch, scan of the owned probed lists only, local
     top-recall_num per query                            (gamma_hip_ivfpq_search_shard_preassigned)
  2. the one real exchange of the path: every rank sends each peer the candidates of the PEER'S
     query slice -- an RCCL all-to-all over xGMI, (nq/W)*R*(4+8) bytes per pair of GPUs, each
     pair on its own direct link (an all-gather of everything would move W times as much)
  3. every rank merges the W candidate tables of its slice into the global top-recall_num and
     runs compute_dis (re-rank / truncate)                           (gamma_hip_ivfpq_merge_rerank)
     followed by a small all-gather of the [nq/W, k] results.

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

    def coarse(self, x, args, cdis, probe):
        """coarse assignment of the rows of x into the preallocated cdis/probe [n, nprobe]"""
        if x.shape[0]:
            self.g.ivfpq_coarse_device(x.data_ptr(), x.shape[0], args, cdis.data_ptr(), probe.data_ptr())

    def search_shard(self, x, cdis, probe, k, args, rdis, rids):
        """local top-R over the owned lists for all rows of x into rdis/rids [n, R]"""
        self.g.ivfpq_search_shard_preassigned(x.data_ptr(), x.shape[0], cdis.data_ptr(), probe.data_ptr(),
                                              k, args, rdis.data_ptr(), rids.data_ptr())

    def merge_rerank(self, all_dis, all_ids, x, k, args, nql, D, I):
        """all_dis/all_ids [W, per, R]: candidates of this rank's slice from every shard"""
        W, per = all_dis.shape[0], all_dis.shape[1]
        if nql > 0:
            self.g.ivfpq_merge_rerank(W, per, x.data_ptr(), k, args, all_dis.data_ptr(),
                                      all_ids.data_ptr(), 0, nql, D.data_ptr(), I.data_ptr())


def _buffers(backend, world, per, P, R, k):
    """Exchange buffers, allocated once per shape and kept on the backend (a search step enqueues
    ~25 kernels and 6 collectives: per-step allocations would leave the GPU waiting for the host)."""
    cache = backend.__dict__.setdefault("_xbuf", {})
    key = (world, per, P, R, k)
    b = cache.get(key)
    if b is None:
        f32, i32, i64 = torch.float32, torch.int32, torch.int64
        b = dict(cdis_l=backend.empty((per, P), f32), probe_l=backend.empty((per, P), i32),
                 cdis=backend.empty((world * per, P), f32), probe=backend.empty((world * per, P), i32),
                 rdis=backend.empty((world * per, R), f32), rids=backend.empty((world * per, R), i64),
                 all_dis=backend.empty((world * per, R), f32), all_ids=backend.empty((world * per, R), i64),
                 D=backend.empty((per, k), f32), I=backend.empty((per, k), i64),
                 Dall=backend.empty((world * per, k), f32), Iall=backend.empty((world * per, k), i64))
        cache.clear()          # one shape at a time
        cache[key] = b
    return b


def sharded_search(backend, x, k, args, group=None, gather_results=True):
    """x: [nq, d] tensor on the backend's device (same on every rank).  Returns (D, I) for all nq
    queries on every rank when gather_results, else this rank's slice.  The returned tensors are
    views of buffers that the next call with the same shape overwrites."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = x.shape[0]
    P = args.p.nprobe
    R = max(args.p.recall_num, k)
    q0, q1, per = query_slice(nq, rank, world)
    nql = q1 - q0
    if args.p.coarse_mode < 0:
        # faiss chooses the coarse path from the size of the whole batch: the slices must agree
        args.p.coarse_mode = 0 if nq < 20 else 1
    stream_ctx = torch.cuda.stream(backend.stream) if hasattr(backend, "stream") else _Null()
    with stream_ctx:
        b = _buffers(backend, world, per, P, R, k)
        # 0. coarse quantizer on the own slice, assignment all-gathered (rows padded to W*per)
        cdis_l, probe_l = b["cdis_l"], b["probe_l"]
        if nql < per:            # padding rows: no valid list
            cdis_l.zero_()
            probe_l.fill_(-1)
        backend.coarse(x[q0:q1], args, cdis_l, probe_l)
        cdis, probe = b["cdis"], b["probe"]
        dist.all_gather_into_tensor(cdis, cdis_l, group=group)
        dist.all_gather_into_tensor(probe, probe_l, group=group)
        # 1. local top-R of every query over the owned lists, laid out [dest rank][per][R]
        rdis, rids = b["rdis"], b["rids"]
        if nq < world * per:     # padding rows carry no candidates
            rdis[nq:].zero_()
            rids[nq:].fill_(-1)
        backend.search_shard(x, cdis[:nq], probe[:nq], k, args, rdis[:nq], rids[:nq])
        # 2. all-to-all: block r of rdis goes to rank r; block s of all_dis came from shard s
        all_dis, all_ids = b["all_dis"], b["all_ids"]
        dist.all_to_all_single(all_dis, rdis, group=group)
        dist.all_to_all_single(all_ids, rids, group=group)
        # 3. merge + compute_dis for the own slice
        D, I = b["D"], b["I"]
        if nql < per:
            D.zero_()
            I.fill_(-1)
        backend.merge_rerank(all_dis.view(world, per, R), all_ids.view(world, per, R), x[q0:q1], k, args,
                             nql, D, I)
        if not gather_results:
            return D[:nql], I[:nql]
        Dall, Iall = b["Dall"], b["Iall"]
        dist.all_gather_into_tensor(Dall, D, group=group)
        dist.all_gather_into_tensor(Iall, I, group=group)
    return Dall[:nq], Iall[:nq]


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
