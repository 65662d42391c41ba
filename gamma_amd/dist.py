"""List-sharded multi-GPU IVFPQ search.

One process per GPU.  Every rank holds the replicated small state (coarse centroids, PQ
codebooks, delete bitmap, raw vectors for the re-rank) and the inverted lists it OWNS.  The
batch of nq queries is also split into one contiguous slice per rank ("its" queries).  A search
runs in four steps:

  0. every rank: coarse quantizer for ITS query slice                (gamma_hip_ivfpq_coarse_device)
     + all-gather of the (distance, list) assignment, nq*nprobe*8 bytes in total
  1. every rank: scan of the owned probed lists only, in TWO PHASES around a min-all-reduce of one float per query
     (gamma_hip_ivfpq_search_shard_bounded): the query's nearest owned probes bound the shard's recall_num-th best, the
     bounds are reduced across the ranks -- every shard's bound is an upper bound of the GLOBAL recall_num-th best, so
     the smallest one is too -- and the shard's other probes only keep what is within the global bound; local
     top-recall_num per query from that            (GAMMA_DIST_TWO_PHASE=0: gamma_hip_ivfpq_search_shard_preassigned,
     every shard against its own, W times looser bound)
  2. the one real exchange of the path: every rank sends each peer the candidates of the PEER'S
     query slice -- an RCCL all-to-all over xGMI, (nq/W)*R*(4+8) bytes per pair of GPUs, each
     pair on its own direct link (an all-gather of everything would move W times as much)
  3. every rank merges the W candidate tables of its slice into the global top-recall_num and
     runs compute_dis (re-rank / truncate)                           (gamma_hip_ivfpq_merge_rerank)
     followed by a small all-gather of the [nq/W, k] results.

Raw vectors: replicated on every rank by default; with HipShardBackend(raw_sharded=True) a rank keeps the rows of the vectors in
ITS lists only (gamma_hip_raw_put) and the exact distances of compute_dis are computed by the shard that holds the row
(gamma_hip_ivfpq_shard_exact) and travel with the candidates -- and, in the tie phase, with the exported streams
(gamma_hip_ivfpq_shard_export_exact / _merge_replay_exact).

Large batches run as two interleaved sub-batches with every collective asynchronous, so the
all-to-all of one overlaps the scan / merge of the other (sharded_search).

The global top-recall_num of the union equals the single-GPU top-recall_num (same ADC
distances, disjoint lists); the queries in which the order inside a group of exactly equal
distances can change the answer are replayed by their slice's owner over the candidate
streams every rank exports for them (tie_phase), so with exact ties on the results are those
of one GPU at every rank.  replicated_search is the other placement: every rank holds every
list and answers a slice of the queries.  The orchestration is backend-agnostic so the world_size-2 gloo test on CPU exercises
the same code with an oracle-based backend (tests/test_dist_cpu.py).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

METRIC_IP, METRIC_L2 = 0, 1   # gamma_hip_search_params.metric (include/gamma_hip.h)


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

    def __init__(self, g, device, raw_sharded=False, owned=None):
        """raw_sharded (round 6): this rank keeps the raw rows of the vectors in ITS lists only (gamma_hip_raw_put; `owned` = its
        list mask); the exact distances of compute_dis are then computed by the shard that holds the row and travel with the
        candidates (shard_exact / merge_rerank_exact; tie phase: shard_export_exact / merge_replay_exact)."""
        self.g = g
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.ExternalStream(g.stream(), device=self.device)
        self.raw_sharded = bool(raw_sharded)
        self.owned = None if owned is None else np.ascontiguousarray(owned, dtype=np.uint8)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def add(self, vecs, first_vid):
        """Realtime insert on a list-sharded index: EVERY rank calls this with the same batch (host
        array [n, d], vids first_vid ..).  The raw store is replicated; the handle, whose list mask
        says which lists this rank owns (GammaHip.set_list_mask), encodes the batch and keeps only
        the vectors assigned to its own lists -- the insert reaches the owner of the list without
        any exchange."""
        if self.raw_sharded:
            # the batch is encoded here (every rank does: no exchange); entries AND rows of the vectors assigned to this rank's
            # lists are kept -- lists in ascending order, entries in batch order, as the reference's std::map does
            vecs = np.ascontiguousarray(vecs, dtype=np.float32)
            lno, codes = self.g.encode(vecs)
            lno = np.asarray(lno, dtype=np.int64)
            nlist = len(self.owned)
            bad = (lno < 0) | (lno >= nlist)
            if bad.any():
                lno = lno.copy()
                lno[bad] = (first_vid + np.nonzero(bad)[0]) % nlist     # gamma_index_ivfpq.cc:475-481
            mine = np.nonzero(self.owned[lno] != 0)[0]
            if len(mine):
                order = mine[np.argsort(lno[mine], kind="stable")]
                lists, counts = np.unique(lno[order], return_counts=True)
                self.g.add_keys_batch(lists, counts, (first_vid + order).astype(np.int64), np.asarray(codes)[order])
                self.g.raw_put((first_vid + mine).astype(np.int64), vecs[mine])
            return
        self.g.raw_append(vecs)
        self.g.add(vecs, first_vid)

    # sharded_add
    def code_size(self):
        return self.g.code_size()

    def encode(self, vecs):
        return self.g.encode(vecs)

    def append_raw(self, vecs):
        self.g.raw_append(vecs)

    def add_keys_batch(self, lists, counts, vids, codes):
        self.g.add_keys_batch(lists, counts, np.ascontiguousarray(vids, dtype=np.int64), codes)

    def has_vid(self, vids):
        return torch.from_numpy(self.g.has_vid(vids)).to(self.device)

    def update_one(self, vid, vec, owned, held_somewhere):
        """One step of sharded_update on this shard; owned = the list mask handed to GammaHip.set_list_mask."""
        self.g.raw_write(vid, vec[None])                     # the raw store is replicated
        lno, code = self.g.encode(vec[None])                 # n = 1: the exact assign, as GammaIVFPQIndex::Update
        route_update(self.g, int(lno[0]), vid, code[0], owned, held_somewhere)

    def compact_if_need(self):
        self.g.compact_if_need()

    def search_all(self, x, k, args, D, I):
        """the whole search for the rows of x on this rank's (complete) index: replicated_search"""
        self.g.ivfpq_search_device(x.data_ptr(), x.shape[0], k, args, D.data_ptr(), I.data_ptr())

    def set_deferred_replay(self, on):
        self.g.set_deferred_replay(on)

    def join(self):
        self.g.join()

    def coarse(self, x, args, cdis, probe):
        """coarse assignment of the rows of x into the preallocated cdis/probe [n, nprobe]"""
        if x.shape[0]:
            self.g.ivfpq_coarse_device(x.data_ptr(), x.shape[0], args, cdis.data_ptr(), probe.data_ptr())

    def search_shard(self, x, cdis, probe, k, args, rdis, rids):
        """local top-R over the owned lists for all rows of x into rdis/rids [n, R]"""
        self.g.ivfpq_search_shard_preassigned(x.data_ptr(), x.shape[0], cdis.data_ptr(), probe.data_ptr(),
                                              k, args, rdis.data_ptr(), rids.data_ptr())

    def search_shard_bounded(self, x, cdis, probe, k, args, rdis, rids, bound, reduce):
        """search_shard in two phases (gamma_hip_ivfpq_search_shard_bounded): the shard's own bound per query lands in
        `bound` [n] float32, reduce(take_max) -- the caller's collective over `bound` -- runs once in between, the consumers
        then filter against the reduced, global bound"""
        self.g.ivfpq_search_shard_bounded(x.data_ptr(), x.shape[0], cdis.data_ptr(), probe.data_ptr(), k, args,
                                          rdis.data_ptr(), rids.data_ptr(), bound.data_ptr(),
                                          (lambda n, take_max: reduce(take_max)) if reduce is not None else None)

    def merge_rerank(self, all_dis, all_ids, x, k, args, nql, D, I, all_exact=None):
        """all_dis/all_ids [W, per, R]: candidates of this rank's slice from every shard (all_exact: their exact distances, when
        the raw vectors are sharded with the lists)"""
        W, per = all_dis.shape[0], all_dis.shape[1]
        if nql > 0 and all_exact is not None:
            self.g.ivfpq_merge_rerank_exact(W, per, x.data_ptr(), k, args, all_dis.data_ptr(), all_ids.data_ptr(),
                                            all_exact.data_ptr(), 0, nql, D.data_ptr(), I.data_ptr())
        elif nql > 0:
            self.g.ivfpq_merge_rerank(W, per, x.data_ptr(), k, args, all_dis.data_ptr(),
                                      all_ids.data_ptr(), 0, nql, D.data_ptr(), I.data_ptr())

    def shard_exact(self, x, ids, args, exact):
        """exact[q][r] = compute_dis's distance of candidate ids[q][r] where this shard holds the row (sentinel elsewhere)"""
        if x.shape[0]:
            self.g.ivfpq_shard_exact(x.data_ptr(), x.shape[0], ids.data_ptr(), ids.shape[1], args, exact.data_ptr())

    def shard_export_exact(self, xf, vals, ids, off, stride, bound_f, args, ex):
        self.g.ivfpq_shard_export_exact(xf.shape[0], xf.data_ptr(), vals.data_ptr(), ids.data_ptr(), off.data_ptr(), stride,
                                        bound_f.data_ptr(), args, ex.data_ptr())

    def merge_replay_exact(self, vals_all, ids_all, off_all, ex_all, x_slice, stride, k, args, d_list, D, I):
        W, nf = vals_all.shape[0], vals_all.shape[1]
        self.g.ivfpq_merge_replay_exact(W, nf, x_slice.data_ptr(), stride, vals_all.data_ptr(), ids_all.data_ptr(), off_all.data_ptr(),
                                        ex_all.data_ptr(), k, args, d_list, D.data_ptr(), I.data_ptr())

    # ---- exact ties across shards (include/gamma_hip.h; tie_phase below) ----
    def shard_cut_flags(self, n, flags):
        """flags[q] != 0: this shard's own top-R cut of query q went through a group of equal distances"""
        if n > 0:
            self.g.ivfpq_shard_cut_flags(n, flags.data_ptr())

    def merge_set_shard_flags(self, flags):
        self.g.ivfpq_merge_set_shard_flags(flags.data_ptr())

    def merge_flagged(self):
        """(number of the slice's queries a tie can change, device address of their slice-local indices); waits for the
        handle's stream"""
        return self.g.ivfpq_merge_flagged()

    def max_list_len(self):
        return self.g.max_list_len()

    def gather_rows(self, src, d_list, n, dst):
        self.g.gather_rows(src.data_ptr(), src.shape[1], d_list, n, dst.data_ptr())

    def shard_export_rows(self, pf, args):
        """the longest export row of these queries on this shard (entries of the probed lists it owns)"""
        return self.g.ivfpq_shard_export_rows(pf.shape[0], pf.data_ptr(), args)

    def shard_export(self, xf, cf, pf, stride, args, vals, ids, off):
        self.g.ivfpq_shard_export(xf.shape[0], xf.data_ptr(), cf.data_ptr(), pf.data_ptr(), stride, args, vals.data_ptr(),
                                  ids.data_ptr(), off.data_ptr())

    def merge_replay(self, vals_all, ids_all, off_all, x_slice, stride, k, args, d_list, D, I):
        W, nf = vals_all.shape[0], vals_all.shape[1]
        self.g.ivfpq_merge_replay(W, nf, x_slice.data_ptr(), stride, vals_all.data_ptr(), ids_all.data_ptr(), off_all.data_ptr(), k,
                                  args, d_list, D.data_ptr(), I.data_ptr())


def route_update(store, lno, vid, code, owned, held_somewhere):
    """RealTimeMemData::Update (realtime_mem_data.cc:305-327) when the list the vector leaves and the list it joins
    may belong to different shards.  store: has_vid / update / remove / add_keys of THIS shard."""
    held = bool(store.has_vid([vid])[0])
    if held and owned[lno]:
        store.update(lno, vid, code)          # both halves here: rewrite in place or move between two owned lists
    elif held:
        store.remove(vid)                     # leaves this shard: flag the old entry, the new owner appends
    elif owned[lno] and held_somewhere:
        store.add_keys(lno, np.array([vid], dtype=np.int64), np.asarray(code, dtype=np.uint8)[None])
    # a vid no shard holds is ignored (:307-311)


def sharded_add(backend, vecs, first_vid, owned, group=None, turn=0):
    """GammaIVFPQIndex::Add (gamma_index_ivfpq.cc:424-512) on a list-sharded index with ONE encode per batch: every rank
    calls this with the same host batch [n, d] and its list mask `owned`; rank `turn % W` runs the encode (coarse
    assignment + PQ codes) and broadcasts n x (8 + code_size) bytes, every rank appends the vectors to its replica of the
    raw store and the entries of the lists it owns -- lists in ascending order, entries in batch order, as the reference's
    std::map does.  (HipShardBackend.add is the exchange-free form: every rank encodes the batch itself.)"""
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    n = vecs.shape[0]
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    cs = backend.code_size()
    src_local = turn % world
    if rank == src_local:
        lno, codes = backend.encode(vecs)
        pack = torch.from_numpy(np.concatenate([lno.astype(np.int64).view(np.uint8).reshape(n, 8),
                                                np.ascontiguousarray(codes, dtype=np.uint8).reshape(n, cs)], axis=1))
    else:
        pack = torch.empty((n, 8 + cs), dtype=torch.uint8)
    pack = pack.to(backend.device) if hasattr(backend, "device") else pack
    if world > 1:
        dist.broadcast(pack, src=dist.get_global_rank(group, src_local) if group is not None else src_local, group=group)
    pack = pack.cpu().numpy()
    lno = np.ascontiguousarray(pack[:, :8]).view(np.int64).reshape(n)
    codes = np.ascontiguousarray(pack[:, 8:])
    backend.append_raw(vecs)
    # a key the quantizer could not assign goes to list vid % nlist (gamma_index_ivfpq.cc:475-481) -- the same rule on
    # every rank, before the owner mask is consulted
    nlist = len(owned)
    bad = (lno < 0) | (lno >= nlist)
    if bad.any():
        lno = lno.copy()
        lno[bad] = (first_vid + np.nonzero(bad)[0]) % nlist
    mine = np.nonzero(np.asarray(owned)[lno] != 0)[0]
    if len(mine):
        order = mine[np.argsort(lno[mine], kind="stable")]
        lists, counts = np.unique(lno[order], return_counts=True)
        backend.add_keys_batch(lists, counts, first_vid + order, codes[order])


def sharded_update(backend, vids, vecs, owned, group=None):
    """GammaIVFPQIndex::Update (gamma_index_ivfpq.cc:375-422) on a list-sharded index.  EVERY rank calls this with the
    same (vids [n], vecs [n, d] host arrays) and its own list mask.  Each rank encodes the vector itself (the
    coarse centroids and codebooks are replicated), so all ranks agree on the new list without exchanging it; the
    rank holding the old entry flags it, the owner of the new list appends.  The only exchange is n bytes of
    "somebody holds this vid" (Update ignores vids that were never added), all-reduced once per call."""
    vids = np.ascontiguousarray(vids, dtype=np.int64)
    vecs = np.ascontiguousarray(vecs, dtype=np.float32)
    held = backend.has_vid(vids)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(held, op=dist.ReduceOp.MAX, group=group)
    held = held.cpu().numpy()
    for i in range(len(vids)):
        if not held[i]:
            continue                          # never added: Update ignores it (:307-311)
        backend.update_one(int(vids[i]), vecs[i], owned, bool(held[i]))
    backend.compact_if_need()                 # gamma_index_ivfpq.cc:420


MIN_SUB = 2048   # queries per rank and sub-batch below which a batch is not split for overlap


def _buffers(backend, world, pers, P, R, k):
    """Exchange buffers, allocated once per shape and kept on the backend (a search step enqueues
    ~25 kernels and 3 collectives per sub-batch: per-step allocations would leave the GPU waiting
    for the host).  One set per sub-batch; the gathered results of all sub-batches share one
    [rows, k] table.  What travels together is packed: (coarse distance | list) of the assignment
    and (labels | distances) of the result are halves of ONE gathered buffer each."""
    cache = backend.__dict__.setdefault("_xbuf", {})
    key = (world, tuple(pers), P, R, k)
    b = cache.get(key)
    if b is None:
        f32, i32, i64, u8 = torch.float32, torch.int32, torch.int64, torch.uint8
        rows = world * sum(pers)
        b = dict(Dall=backend.empty((rows, k), f32), Iall=backend.empty((rows, k), i64), sub=[])
        for per in pers:
            cp_l = backend.empty((2, per, P), i32)
            nres = per * k
            res_bytes = (nres * 12 + 7) // 8 * 8          # labels (8 B) first: both halves stay aligned
            res_l = backend.empty((res_bytes,), u8)
            b["sub"].append(dict(
                cp_l=cp_l, cdis_l=cp_l[0].view(f32), probe_l=cp_l[1],
                cp=backend.empty((world, 2, per, P), i32),
                cdis=backend.empty((world * per, P), f32), probe=backend.empty((world * per, P), i32),
                rdis=backend.empty((world * per, R), f32), rids=backend.empty((world * per, R), i64),
                all_dis=backend.empty((world * per, R), f32), all_ids=backend.empty((world * per, R), i64),
                cutf=backend.empty((world * per,), u8), cutall=backend.empty((world * per,), u8),
                bound=backend.empty((world * per,), f32),
                rex=backend.empty((world * per, R), f32), all_exact=backend.empty((world * per, R), f32),
                res_l=res_l, I=res_l[:nres * 8].view(i64).view(per, k),
                D=res_l[nres * 8:nres * 12].view(f32).view(per, k),
                res=backend.empty((world, res_bytes), u8)))
        cache.clear()          # one shape at a time
        cache[key] = b
    return b


def _exchange(rdis, rids, all_dis, all_ids, group):
    """The all-to-all of the path: block r of (rdis, rids) goes to rank r, block s of (all_dis, all_ids)
    comes from shard s.  Two plain all-to-alls, issued back to back on the communication stream."""
    return [dist.all_to_all_single(all_dis, rdis, group=group, async_op=True),
            dist.all_to_all_single(all_ids, rids, group=group, async_op=True)]


NB_TIGHTEN = 32   # edges per query of the second, tightening reduction (L2)


def _packed_exchange(backend, rdis, rids, bound, n, world, per, R, l2, all_dis, all_ids, group, stats, exact_fn=None, all_exact=None):
    """The exchange of a two-phase step, PACKED: what a shard holds beyond the global bound cannot be in the global
    top-recall_num, so only the entries within it travel -- about recall_num per query over ALL shards instead of W x
    recall_num.  (1) L2: the bound is tightened first -- every shard counts its entries under 32 edges between 0 and
    the bound, the counts are sum-all-reduced (nq x 32 ints) and the lowest edge under which at least recall_num
    entries lie is the new bound: it still bounds the global recall_num-th best, now from all W shards' candidates
    instead of the best shard's producers'.  (2) every shard packs its entries within the bound, query by query; the
    W x W table of block sizes is all-gathered and read by the host (the one synchronisation of the path) and the blocks
    go out in one all-to-all with exact split sizes, next to the per-query counts (fixed size).  (3) the receiver
    unpacks into the [W, per, R] layout the merge takes, padding with (sentinel, -1): the merge, the cut flags and the
    tie phase are unchanged.  rdis / rids: [W * per, R] (rows >= n are padding), bound: [W * per] floats (n valid)."""
    dev_long = torch.int64
    rows = world * per
    tau = bound
    valid = rids >= 0
    if rows > n:
        valid[n:] = False
    if l2:
        # tightening: edges tau * j / NB, j = 1 .. NB (the last one is tau itself); rows are sorted ascending
        pos = torch.isfinite(tau) & (tau > 0)
        frac = torch.arange(1, NB_TIGHTEN + 1, device=tau.device, dtype=torch.float32) / NB_TIGHTEN
        edges = torch.where(pos, tau, torch.zeros_like(tau))[:, None] * frac[None, :]
        cnt_e = torch.searchsorted(rdis, edges, right=True).to(torch.int32)
        if rows > n:
            cnt_e[n:] = 0
        if world > 1:
            dist.all_reduce(cnt_e, op=dist.ReduceOp.SUM, group=group)
        ok = cnt_e >= R
        j = ok.to(torch.int32).argmax(dim=1, keepdim=True)
        tight = torch.where(ok.any(dim=1) & pos, edges.gather(1, j.to(dev_long)).squeeze(1), tau)
        within = valid & (rdis <= tight[:, None])
    else:
        within = valid & (rdis >= tau[:, None])
    cnt = within.sum(dim=1, dtype=torch.int32)                      # [rows] entries of every query that travel
    blk = cnt.view(world, per).sum(dim=1, dtype=dev_long)            # [W] block sizes: what goes to each owner
    table = torch.empty((world, world), dtype=dev_long, device=blk.device)
    if world > 1:
        dist.all_gather_into_tensor(table.view(-1), blk, group=group)
    else:
        table[0] = blk
    rank = dist.get_rank(group) if world > 1 else 0
    th = table.cpu()                                                 # the host needs the split sizes
    send = [int(v) for v in th[rank]]
    recv = [int(v) for v in th[:, rank]]
    pk_dis = rdis[within]                                            # row-major: by owner, query, rank
    pk_ids = rids[within]
    pk_ex = None
    if exact_fn is not None:      # raw vectors sharded with the lists: the exact distance of what travels, computed where the row is
        rex = exact_fn(torch.where(within, rids, torch.full_like(rids, -1)))
        pk_ex = rex[within]
    rc_dis = torch.empty((sum(recv),), dtype=rdis.dtype, device=rdis.device)
    rc_ids = torch.empty((sum(recv),), dtype=rids.dtype, device=rids.device)
    rc_cnt = torch.empty((world, per), dtype=torch.int32, device=cnt.device)
    rc_ex = torch.empty((sum(recv),), dtype=rdis.dtype, device=rdis.device) if pk_ex is not None else None
    if world > 1:
        w = [dist.all_to_all_single(rc_cnt.view(-1), cnt, group=group, async_op=True),
             dist.all_to_all_single(rc_dis, pk_dis, output_split_sizes=recv, input_split_sizes=send, group=group, async_op=True),
             dist.all_to_all_single(rc_ids, pk_ids, output_split_sizes=recv, input_split_sizes=send, group=group, async_op=True)]
        if pk_ex is not None:
            w.append(dist.all_to_all_single(rc_ex, pk_ex, output_split_sizes=recv, input_split_sizes=send, group=group, async_op=True))
        for x in w:
            x.wait()
    else:
        rc_cnt.view(-1).copy_(cnt)
        rc_dis, rc_ids, rc_ex = pk_dis, pk_ids, pk_ex
    # unpack: entry e of the received stream belongs to row rowid[e] (shard-major, then query) at column e - start[row]
    flat_cnt = rc_cnt.view(-1).to(dev_long)
    start = torch.cumsum(flat_cnt, 0) - flat_cnt
    total = sum(recv)
    all_dis.fill_(3.4028234663852886e38 if l2 else -3.4028234663852886e38)
    all_ids.fill_(-1)
    if total > 0:
        rowid = torch.repeat_interleave(torch.arange(world * per, device=flat_cnt.device), flat_cnt, output_size=total)
        col = torch.arange(total, device=flat_cnt.device) - start[rowid]
        all_dis.view(world * per, R)[rowid, col] = rc_dis
        all_ids.view(world * per, R)[rowid, col] = rc_ids
        if rc_ex is not None:
            all_exact.view(world * per, R)[rowid, col] = rc_ex
    if stats is not None:
        stats["exchange_entries"] = stats.get("exchange_entries", 0) + sum(send) - send[rank]
        stats["exchange_bytes"] = stats.get("exchange_bytes", 0) + (sum(send) - send[rank]) * (16 if pk_ex is not None else 12) + (world - 1) * per * 4 + \
            (rows * NB_TIGHTEN * 4 if l2 else 0)
        stats["queries"] = stats.get("queries", 0) + n


def plan_sub_batches(nq, world, nsub):
    """[(first query, end query)] of the contiguous sub-batches.  All but the last hold a multiple of
    `world` queries, so their gathered result rows carry no padding and the whole result stays one
    contiguous [nq, k] table."""
    nsub = max(1, min(nsub, nq // max(1, world)))
    bounds = [0]
    for j in range(nsub - 1):
        left = nq - bounds[-1]
        size = -(-left // (nsub - j))
        size = -(-size // world) * world
        bounds.append(min(nq, bounds[-1] + size))
    bounds.append(nq)
    return [(bounds[j], bounds[j + 1]) for j in range(len(bounds) - 1) if bounds[j + 1] > bounds[j] or j == 0]


def tie_phase(backend, x_slice, cdis_slice, probe_slice, nql, k, args, D, I, group=None, bound_slice=None, raw_sharded=False):
    """Exact ties across shards, after merge_rerank of a (sub-)batch: the slice's owner lists the queries whose result
    a tie can change; their vectors and assignment rows go to every rank (broadcast), every rank exports the candidate
    streams over the lists it owns (gamma_hip_ivfpq_shard_export), the exports are gathered and the owner replays the
    assembled streams through the reference's heaps (gamma_hip_ivfpq_merge_replay): rows of D / I rewritten.  Collective:
    every rank calls it.  A batch without a flagged query costs one all-gather of a counter (and a wait for the merge)."""
    if not hasattr(backend, "merge_flagged"):
        return
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    P = args.p.nprobe
    nf, d_list = backend.merge_flagged() if nql > 0 else (0, 0)
    meta = backend.empty((1,), torch.int64)
    meta[0] = nf
    allm = backend.empty((world,), torch.int64)
    dist.all_gather_into_tensor(allm, meta, group=group)
    counts = [int(v) for v in allm.cpu()]
    if sum(counts) == 0:
        return
    d = x_slice.shape[1]
    fcap = 256
    for o in range(world):
        src = dist.get_global_rank(group, o) if group is not None else o
        for f0 in range(0, counts[o], fcap):
            n = min(fcap, counts[o] - f0)
            xf = backend.empty((n, d), torch.float32)
            cf = backend.empty((n, P), torch.float32)
            pf = backend.empty((n, P), torch.int32)
            bf = backend.empty((n, 1), torch.float32)   # raw vectors sharded: the bound under which an entry can be a heap member
            if rank == o:
                lst = d_list + 4 * f0
                backend.gather_rows(x_slice, lst, n, xf)
                backend.gather_rows(cdis_slice.view(torch.float32), lst, n, cf)
                backend.gather_rows(probe_slice, lst, n, pf)
                if raw_sharded and bound_slice is not None:
                    backend.gather_rows(bound_slice.view(-1, 1), lst, n, bf)
                else:
                    bf.fill_(float("nan"))               # no bound: every entry
            for t in (xf, cf, pf) + ((bf,) if raw_sharded else ()):
                dist.broadcast(t, src=src, group=group)
            # one row stride for the exports of all ranks: the longest row any of them has for these queries
            meta[0] = backend.shard_export_rows(pf, args)
            dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
            stride = max(4, (int(meta.item()) + 3) // 4 * 4)
            vals = backend.empty((n, stride), torch.float32)
            ids = backend.empty((n, stride), torch.int64)
            off = backend.empty((n, P + 1), torch.int32)
            backend.shard_export(xf, cf, pf, stride, args, vals, ids, off)
            av = backend.empty((world, n, stride), torch.float32)
            ai = backend.empty((world, n, stride), torch.int64)
            ao = backend.empty((world, n, P + 1), torch.int32)
            dist.all_gather_into_tensor(av.view(-1), vals.view(-1), group=group)
            dist.all_gather_into_tensor(ai.view(-1), ids.view(-1), group=group)
            dist.all_gather_into_tensor(ao.view(-1), off.view(-1), group=group)
            if raw_sharded:
                ex = backend.empty((n, stride), torch.float32)
                backend.shard_export_exact(xf, vals, ids, off, stride, bf.view(-1), args, ex)
                ae = backend.empty((world, n, stride), torch.float32)
                dist.all_gather_into_tensor(ae.view(-1), ex.view(-1), group=group)
                if rank == o:
                    backend.merge_replay_exact(av, ai, ao, ae, x_slice, stride, k, args, d_list + 4 * f0, D, I)
            elif rank == o:
                backend.merge_replay(av, ai, ao, x_slice, stride, k, args, d_list + 4 * f0, D, I)


def sharded_search(backend, x, k, args, group=None, pipeline=None):
    """x: [nq, d] tensor on the backend's device (same on every rank).  Returns (D, I) for all nq
    queries on every rank.  The returned tensors are views of buffers that the next call with the
    same shape overwrites.

    A large batch is cut into two sub-batches whose steps are interleaved, every collective issued
    asynchronously: the all-to-all of one sub-batch (the only exchange whose volume matters,
    nq*R*12 bytes) runs over xGMI while the other sub-batch is being scanned or merged.
    `pipeline` overrides the number of sub-batches."""
    # faiss chooses the coarse path from the size of the whole batch: slices and sub-batches must agree.  The
    # resolved mode lives in the caller's parameter block only for the duration of this call -- a SearchArgs
    # reused for a batch on the other side of the 20-query rule must resolve again.
    with _whole_call_coarse_mode(args, x.shape[0]):
        return _sharded_search(backend, x, k, args, group, pipeline)


def _sharded_search(backend, x, k, args, group, pipeline):
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nq = x.shape[0]
    P = args.p.nprobe
    R = max(args.p.recall_num, k)
    nsub = pipeline or int(os.environ.get("GAMMA_DIST_PIPELINE", "0")) or (2 if world > 1 and nq >= 2 * world * MIN_SUB else 1)
    plan = plan_sub_batches(nq, world, nsub)
    pers = [max(1, -(-(e - s) // world)) for s, e in plan]
    two_phase = hasattr(backend, "search_shard_bounded") and os.environ.get("GAMMA_DIST_TWO_PHASE", "1") != "0"
    packed = two_phase and os.environ.get("GAMMA_DIST_PACKED", "1") != "0"
    stats = getattr(backend, "exchange_stats", None)   # a dict the caller hangs on the backend: entries / bytes this rank sent
    raw_sharded = bool(getattr(backend, "raw_sharded", False)) and bool(args.p.has_rank)
    stream_ctx = torch.cuda.stream(backend.stream) if hasattr(backend, "stream") else _Null()
    with stream_ctx:
        bufs = _buffers(backend, world, pers, P, R, k)
        Dall, Iall = bufs["Dall"], bufs["Iall"]
        subs = []
        row = 0
        for (s0, s1), per, b in zip(plan, pers, bufs["sub"]):
            n = s1 - s0
            q0, q1, _ = query_slice(n, rank, world)
            subs.append(dict(x=x[s0:s1], n=n, per=per, q0=q0, q1=q1, nql=q1 - q0, b=b, row=row))
            row += world * per
        # 0. coarse quantizer on the own slice, assignment all-gathered (rows padded to W*per)
        for sb in subs:
            b, per, nql = sb["b"], sb["per"], sb["nql"]
            cdis_l, probe_l = b["cdis_l"], b["probe_l"]
            if nql < per:            # padding rows: no valid list
                cdis_l.zero_()
                probe_l.fill_(-1)
            backend.coarse(sb["x"][sb["q0"]:sb["q1"]], args, cdis_l, probe_l)
            sb["w"] = [dist.all_gather_into_tensor(b["cp"].view(-1), b["cp_l"].view(-1), group=group, async_op=True)]
        # 1. local top-R of every query over the owned lists, laid out [dest rank][per][R];
        # 2. all-to-all: block r of rdis goes to rank r; block s of all_dis came from shard s
        for sb in subs:
            b, per, n = sb["b"], sb["per"], sb["n"]
            for w in sb["w"]:
                w.wait()
            # unpack [W][2][per][P] into the two contiguous [W*per, P] tables the shard search reads
            b["cdis"].view(world, per, P).copy_(b["cp"][:, 0].view(torch.float32))
            b["probe"].view(world, per, P).copy_(b["cp"][:, 1])
            rdis, rids = b["rdis"], b["rids"]
            if n < world * per:     # padding rows carry no candidates
                rdis[n:].zero_()
                rids[n:].fill_(-1)
            if two_phase:
                bound = b["bound"][:n]

                def reduce(take_max, bound=bound):
                    # (called once from inside the shard call, between its two phases, on the backend's stream)
                    dist.all_reduce(bound, op=dist.ReduceOp.MAX if take_max else dist.ReduceOp.MIN, group=group)
                backend.search_shard_bounded(sb["x"], b["cdis"][:n], b["probe"][:n], k, args, rdis[:n], rids[:n], bound, reduce)
            else:
                backend.search_shard(sb["x"], b["cdis"][:n], b["probe"][:n], k, args, rdis[:n], rids[:n])
            exact_fn = None
            if raw_sharded:
                def exact_fn(ids_m, sb=sb, b=b, n=n):
                    b["rex"].fill_(float("inf") if args.p.metric == METRIC_L2 else float("-inf"))
                    backend.shard_exact(sb["x"], ids_m[:n].contiguous(), args, b["rex"][:n])
                    return b["rex"]
            if two_phase and packed:
                _packed_exchange(backend, rdis, rids, b["bound"], n, world, per, R, args.p.metric == METRIC_L2,
                                 b["all_dis"], b["all_ids"], group, stats, exact_fn, b["all_exact"])
                sb["w"] = []
            else:
                sb["w"] = _exchange(rdis, rids, b["all_dis"], b["all_ids"], group)
                if raw_sharded:
                    sb["w"].append(dist.all_to_all_single(b["all_exact"], exact_fn(rids), group=group, async_op=True))
                if stats is not None:
                    stats["exchange_entries"] = stats.get("exchange_entries", 0) + (world - 1) * per * R
                    stats["exchange_bytes"] = stats.get("exchange_bytes", 0) + (world - 1) * per * R * 12
                    stats["queries"] = stats.get("queries", 0) + n
            if hasattr(backend, "shard_cut_flags") and args.p.exact_ties >= 0:
                # did this shard's own top-R cut of a query go through a tie?  One byte per query to the query's owner
                if n < world * per:
                    b["cutf"][n:].zero_()
                backend.shard_cut_flags(n, b["cutf"])
                sb["w"].append(dist.all_to_all_single(b["cutall"], b["cutf"], group=group, async_op=True))
        # 3. merge + compute_dis for the own slice, results all-gathered into the common table
        pending = []
        for sb in subs:
            b, per, nql = sb["b"], sb["per"], sb["nql"]
            for w in sb["w"]:
                w.wait()
            D, I = b["D"], b["I"]
            if nql < per:
                D.zero_()
                I.fill_(-1)
            if hasattr(backend, "shard_cut_flags") and args.p.exact_ties >= 0 and nql > 0:
                backend.merge_set_shard_flags(b["cutall"])
            if raw_sharded:
                backend.merge_rerank(b["all_dis"].view(world, per, R), b["all_ids"].view(world, per, R),
                                     sb["x"][sb["q0"]:sb["q1"]], k, args, nql, D, I, all_exact=b["all_exact"].view(world, per, R))
            else:
                backend.merge_rerank(b["all_dis"].view(world, per, R), b["all_ids"].view(world, per, R),
                                     sb["x"][sb["q0"]:sb["q1"]], k, args, nql, D, I)
            if args.p.exact_ties >= 0:   # (the handle's default is on; -1 = off for this request)
                o0 = rank * per
                tie_phase(backend, sb["x"][sb["q0"]:sb["q1"]], b["cdis"][o0:o0 + nql], b["probe"][o0:o0 + nql], nql, k, args, D, I,
                          group, bound_slice=(b["bound"][o0:o0 + nql] if two_phase else None), raw_sharded=raw_sharded)
            pending.append((sb, dist.all_gather_into_tensor(b["res"].view(-1), b["res_l"], group=group, async_op=True)))
        for sb, w in pending:
            w.wait()
            b, per = sb["b"], sb["per"]
            rows = slice(sb["row"], sb["row"] + world * per)
            nres = per * k
            Iall[rows].view(world, per, k).copy_(b["res"][:, :nres * 8].view(torch.int64).view(world, per, k))
            Dall[rows].view(world, per, k).copy_(b["res"][:, nres * 8:nres * 12].view(torch.float32).view(world, per, k))
    return Dall[:nq], Iall[:nq]


class _whole_call_coarse_mode:
    """faiss picks the coarse path (exact below 20 queries, GEMM form from 20 on, faiss:utils/distances.cpp:303,346) from
    the size of the WHOLE call.  A rank that searches a slice must not let the slice's size decide: the mode is resolved
    here on the batch size and lives in the caller's parameter block only for the duration of the call (a SearchArgs
    reused for a batch on the other side of the rule resolves again)."""

    def __init__(self, args, nq):
        self.args, self.nq = args, nq

    def __enter__(self):
        self.saved = self.args.p.coarse_mode
        if self.saved < 0:
            self.args.p.coarse_mode = 0 if self.nq < 20 else 1

    def __exit__(self, *a):
        self.args.p.coarse_mode = self.saved
        return False


def replicated_search(backend, x, k, args, group=None):
    """Query-parallel search over REPLICATED lists: every rank holds the whole index (no list mask) and answers its slice
    of the batch with the ordinary single-handle search -- exact ties and all -- and the [nq/W, k] results are
    all-gathered (the one collective of the path: nq*k*12 bytes in total).  This is what a deployment does with an index
    that is small next to a GPU's memory: at C3 size (24 MB of codes) list sharding leaves every shard ~1000 codes per
    query and the per-query fixed work (table, bound, selection) times W (multi-GPU); replicas scale with no
    exchange on the search path at all.  Inserts / updates / deletes go to every rank.  x: [nq, d] on the device, the
    same on every rank.  Returns (D, I) for all nq queries on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nq = x.shape[0]
    q0, q1, per = query_slice(nq, rank, world)
    stream_ctx = torch.cuda.stream(backend.stream) if hasattr(backend, "stream") else _Null()
    with stream_ctx:
        cache = backend.__dict__.setdefault("_rbuf", {})
        key = (world, per, k)
        b = cache.get(key)
        if b is None:
            nres = per * k
            res_bytes = (nres * 12 + 7) // 8 * 8
            res_l = backend.empty((res_bytes,), torch.uint8)
            b = dict(res_l=res_l, I=res_l[:nres * 8].view(torch.int64).view(per, k),
                     D=res_l[nres * 8:nres * 12].view(torch.float32).view(per, k),
                     res=backend.empty((world, res_bytes), torch.uint8),
                     Dall=backend.empty((world * per, k), torch.float32), Iall=backend.empty((world * per, k), torch.int64))
            cache.clear()
            cache[key] = b
        if q1 - q0 < per:
            b["D"].zero_()
            b["I"].fill_(-1)
        if q1 > q0:
            with _whole_call_coarse_mode(args, nq):
                backend.search_all(x[q0:q1], k, args, b["D"][:q1 - q0], b["I"][:q1 - q0])
        if world > 1:
            dist.all_gather_into_tensor(b["res"].view(-1), b["res_l"], group=group)
        else:
            b["res"][0].copy_(b["res_l"])
        nres = per * k
        b["Iall"].view(world, per, k).copy_(b["res"][:, :nres * 8].view(torch.int64).view(world, per, k))
        b["Dall"].view(world, per, k).copy_(b["res"][:, nres * 8:nres * 12].view(torch.float32).view(world, per, k))
    return b["Dall"][:nq], b["Iall"][:nq]


class ReplicatedStream:
    """replicated_search for a STREAM of batches of one shape, with the deferred tie replay (include/gamma_hip.h,
    gamma_hip_set_deferred_replay): the few queries of a batch that go through the reference's heaps again are replayed
    on the handle's side stream beside the NEXT batch's coarse quantizer and query tables instead of at the end of
    their own call, where the replay is the latency of one query's chain with the chip idle.  The results
    of a batch are therefore complete only once the next search has been enqueued, so the all-gather runs one batch
    behind:

        submit(x)  enqueues the search of this rank's slice of x and returns the gathered (D, I) of the PREVIOUS batch
                   (None for the first one);
        flush()    returns those of the last batch (joins its replay first).

    Every batch is searched once and gathered once; the buffers alternate between two sets: the (D, I) a call returns are
    views that stay valid until the second next submit, and are ready in the order of the backend's stream (as
    replicated_search's).  x must stay untouched until its results have been returned.  A backend without deferred replay (the CPU backends of the tests) runs the same
    schedule, one batch behind."""

    def __init__(self, backend, k, args, group=None):
        self.backend, self.k, self.args, self.group = backend, k, args, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.bufs = {}
        self.pending = None      # (slot, nq) of the batch searched but not gathered yet
        self.n = 0
        self.deferred = hasattr(backend, "set_deferred_replay")
        if self.deferred:
            backend.set_deferred_replay(True)

    def close(self):
        if self.deferred:
            self.backend.set_deferred_replay(False)

    def _ctx(self):
        return torch.cuda.stream(self.backend.stream) if hasattr(self.backend, "stream") else _Null()

    def _buf(self, slot, per):
        b = self.bufs.get(slot)
        if b is None or b["per"] != per:
            be, k, world = self.backend, self.k, self.world
            nres = per * k
            res_bytes = (nres * 12 + 7) // 8 * 8
            res_l = be.empty((res_bytes,), torch.uint8)
            b = dict(per=per, res_l=res_l, I=res_l[:nres * 8].view(torch.int64).view(per, k),
                     D=res_l[nres * 8:nres * 12].view(torch.float32).view(per, k),
                     res=be.empty((world, res_bytes), torch.uint8),
                     Dall=be.empty((world * per, k), torch.float32), Iall=be.empty((world * per, k), torch.int64))
            self.bufs[slot] = b
        return b

    def _gather(self, slot, nq):
        b, world, k = self.bufs[slot], self.world, self.k
        per, nres = b["per"], b["per"] * self.k
        if world > 1:
            dist.all_gather_into_tensor(b["res"].view(-1), b["res_l"], group=self.group)
        else:
            b["res"][0].copy_(b["res_l"])
        b["Iall"].view(world, per, k).copy_(b["res"][:, :nres * 8].view(torch.int64).view(world, per, k))
        b["Dall"].view(world, per, k).copy_(b["res"][:, nres * 8:nres * 12].view(torch.float32).view(world, per, k))
        return b["Dall"][:nq], b["Iall"][:nq]

    def submit(self, x):
        nq = x.shape[0]
        q0, q1, per = query_slice(nq, self.rank, self.world)
        slot = self.n & 1
        self.n += 1
        out = None
        with self._ctx():
            b = self._buf(slot, per)
            if q1 - q0 < per:
                b["D"].zero_()
                b["I"].fill_(-1)
            # (an empty slice still makes the call: a search call is what completes the previous one)
            with _whole_call_coarse_mode(self.args, nq):
                self.backend.search_all(x[q0:q1], self.k, self.args, b["D"][:q1 - q0], b["I"][:q1 - q0])
            if self.pending is not None:
                out = self._gather(*self.pending)
            self.pending = (slot, nq)
        return out

    def flush(self):
        if self.pending is None:
            return None
        with self._ctx():
            if self.deferred:
                self.backend.join()
            out = self._gather(*self.pending)
            self.pending = None
        return out


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
