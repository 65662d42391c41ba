"""W list shards emulated on ONE GPU through the C ABI -- the steps gamma_hip_group / gamma_amd.dist drive, with the
exchanges done by tensor indexing: coarse per query slice, shard scans, merge + re-rank at the slice's owner, then the
tie phase (include/gamma_hip.h, "exact ties across list shards"): the owner lists the queries a tie can change
(gamma_hip_ivfpq_merge_flagged), every shard exports their candidate streams over the lists it owns
(gamma_hip_ivfpq_shard_export), the owner assembles and replays them (gamma_hip_ivfpq_merge_replay).
Shared by tests/test_gpu_ties.py, tests/test_gpu_dist.py and the shard fuzz.
Several handles = several streams, and torch's own: every tensor torch fills or copies is complete (torch.cuda.synchronize)
before a handle's kernels are enqueued on it, every handle call is synchronised before torch or another handle reads its
output -- without the first half a zero fill could land AFTER the shard scan it was meant to precede (seen under load:
six test processes on one GPU)."""
import os

import torch

from gamma_amd import api
from gamma_amd import dist as gdist


def sharded_search_emulated(shards, x, k, args, use_shard_flags=True, two_phase=None, raw_sharded=False):
    """shards: W api.GammaHip handles, each holding the lists it owns.  x: [nq, d] float32 tensor on cuda:0.
    Returns (D, I, flagged): [nq, k] result tensors and the number of queries that went through the tie phase.
    two_phase (default: on, GAMMA_TEST_TWO_PHASE=0 turns it off): the shard scans run through
    gamma_hip_ivfpq_search_shard_bounded with the reduction of the bounds across the shards emulated in two passes -- every
    shard once with its own bounds only (they are what its first phase exports), then again with a reduction that hands
    it the minimum (L2) / maximum (inner product) over all shards."""
    if two_phase is None:
        two_phase = os.environ.get("GAMMA_TEST_TWO_PHASE", "1") != "0"
    # raw_sharded: every shard holds the raw rows of ITS lists only (gamma_hip_raw_put): the exact distances of compute_dis are
    # computed by the shard that holds the row and travel with its candidates (shard_exact -> merge_rerank_exact), in the tie
    # phase with the exported streams (shard_export_exact -> merge_replay_exact)
    raw_sharded = raw_sharded and bool(args.p.has_rank)
    rx = []
    W = len(shards)
    nq, d = x.shape
    P = args.p.nprobe
    R = max(args.p.recall_num, k)
    dev = x.device
    per = (nq + W - 1) // W
    backs = [gdist.HipShardBackend(g, 0) for g in shards]
    sl = [gdist.query_slice(nq, s, W) for s in range(W)]
    cd_parts, pr_parts = [], []
    for s in range(W):
        q0, q1, _ = sl[s]
        cdis = torch.zeros((max(1, q1 - q0), P), dtype=torch.float32, device=dev)
        probe = torch.full((max(1, q1 - q0), P), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        if q1 > q0:
            backs[s].coarse(x[q0:q1].contiguous(), args, cdis, probe)
            shards[s].synchronize()
        cd_parts.append(cdis[:q1 - q0])
        pr_parts.append(probe[:q1 - q0])
    cd_all = torch.cat(cd_parts).contiguous()      # the assignment in query order: what the all-gather delivers
    pr_all = torch.cat(pr_parts).contiguous()
    torch.cuda.synchronize()
    rd, ri, cf_ = [], [], []
    glob = None
    if two_phase:
        own = []
        for s in range(W):
            rdis = torch.zeros((nq, R), dtype=torch.float32, device=dev)
            rids = torch.full((nq, R), -1, dtype=torch.int64, device=dev)
            b = torch.zeros((nq,), dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            backs[s].search_shard_bounded(x, cd_all, pr_all, k, args, rdis, rids, b, None)
            shards[s].synchronize()
            own.append(b)
        st = torch.stack(own)
        glob = (st.max(dim=0).values if args.p.metric == api.METRIC_IP else st.min(dim=0).values).contiguous()
        torch.cuda.synchronize()
    for s in range(W):
        rdis = torch.zeros((nq, R), dtype=torch.float32, device=dev)
        rids = torch.full((nq, R), -1, dtype=torch.int64, device=dev)
        cutf = torch.zeros((nq,), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        if two_phase:
            b = torch.zeros((nq,), dtype=torch.float32, device=dev)

            def reduce(take_max, b=b, s=s):
                # what the all-reduce leaves in the buffer; the shard's own value must be one of the inputs
                # (no call back into the handle from here: the search holds its lock)
                torch.cuda.synchronize()
                assert torch.equal(b, own[s]), "phase 1 is not deterministic"
                assert take_max == (args.p.metric == api.METRIC_IP)
                b.copy_(glob)
                torch.cuda.synchronize()
            torch.cuda.synchronize()
            backs[s].search_shard_bounded(x, cd_all, pr_all, k, args, rdis, rids, b, reduce)
        else:
            backs[s].search_shard(x, cd_all, pr_all, k, args, rdis, rids)
        if raw_sharded:
            rex = torch.empty((nq, R), dtype=torch.float32, device=dev)
            shards[s].synchronize()
            shards[s].ivfpq_shard_exact(x.data_ptr(), nq, rids.data_ptr(), R, args, rex.data_ptr())
            rx.append(rex)
        backs[s].shard_cut_flags(nq, cutf)
        shards[s].synchronize()
        rd.append(rdis)
        ri.append(rids)
        cf_.append(cutf)
    D = torch.zeros((nq, k), dtype=torch.float32, device=dev)
    I = torch.full((nq, k), -1, dtype=torch.int64, device=dev)
    flagged = 0
    for r in range(W):       # what the all-to-all delivers to rank r: its slice of every shard's table
        q0, q1, _ = sl[r]
        nql = q1 - q0
        if nql == 0:
            continue
        all_dis = torch.stack([rd[s][q0:q1] for s in range(W)]).contiguous()
        all_ids = torch.stack([ri[s][q0:q1] for s in range(W)]).contiguous()
        xs = x[q0:q1].contiguous()
        Dr = torch.zeros((nql, k), dtype=torch.float32, device=dev)
        Ir = torch.full((nql, k), -1, dtype=torch.int64, device=dev)
        cut_all = torch.stack([cf_[s][q0:q1] for s in range(W)]).contiguous() if use_shard_flags else None
        torch.cuda.synchronize()
        if use_shard_flags:   # (without: every table that ends at the cut value counts as a tie)
            shards[r].ivfpq_merge_set_shard_flags(cut_all.data_ptr())
        if raw_sharded:
            all_ex = torch.stack([rx[s][q0:q1] for s in range(W)]).contiguous()
            torch.cuda.synchronize()
            shards[r].ivfpq_merge_rerank_exact(W, nql, xs.data_ptr(), k, args, all_dis.data_ptr(), all_ids.data_ptr(),
                                               all_ex.data_ptr(), 0, nql, Dr.data_ptr(), Ir.data_ptr())
        else:
            shards[r].ivfpq_merge_rerank(W, nql, xs.data_ptr(), k, args, all_dis.data_ptr(), all_ids.data_ptr(), 0, nql,
                                         Dr.data_ptr(), Ir.data_ptr())
        nf, d_list = shards[r].ivfpq_merge_flagged() if args.p.exact_ties >= 0 else (0, 0)
        flagged += nf
        if nf:
            xf = torch.empty((nf, d), dtype=torch.float32, device=dev)
            cf = torch.empty((nf, P), dtype=torch.float32, device=dev)
            pf = torch.empty((nf, P), dtype=torch.int32, device=dev)
            cds = cd_all[q0:q1].contiguous()
            prs = pr_all[q0:q1].contiguous()
            torch.cuda.synchronize()
            shards[r].gather_rows(xs.data_ptr(), d, d_list, nf, xf.data_ptr())
            shards[r].gather_rows(cds.data_ptr(), P, d_list, nf, cf.data_ptr())
            shards[r].gather_rows(prs.data_ptr(), P, d_list, nf, pf.data_ptr())
            shards[r].synchronize()
            stride = max(4, (max(g.ivfpq_shard_export_rows(nf, pf.data_ptr(), args) for g in shards) + 3) // 4 * 4)
            vals = torch.empty((W, nf, stride), dtype=torch.float32, device=dev)
            ids = torch.empty((W, nf, stride), dtype=torch.int64, device=dev)
            off = torch.empty((W, nf, P + 1), dtype=torch.int32, device=dev)
            exs = torch.empty((W, nf, stride), dtype=torch.float32, device=dev) if raw_sharded else None
            bf = None
            if raw_sharded:   # the bound under which an entry can still be a member of the recall_num-heap (NaN: every entry)
                bf = torch.full((nf,), float("nan"), dtype=torch.float32, device=dev)
                if glob is not None:
                    gs = glob[q0:q1].contiguous().view(-1, 1)
                    torch.cuda.synchronize()
                    shards[r].gather_rows(gs.data_ptr(), 1, d_list, nf, bf.data_ptr())
                    shards[r].synchronize()
            for s in range(W):
                shards[s].ivfpq_shard_export(nf, xf.data_ptr(), cf.data_ptr(), pf.data_ptr(), stride, args, vals[s].data_ptr(),
                                             ids[s].data_ptr(), off[s].data_ptr())
                if raw_sharded:
                    shards[s].ivfpq_shard_export_exact(nf, xf.data_ptr(), vals[s].data_ptr(), ids[s].data_ptr(), off[s].data_ptr(), stride,
                                                       bf.data_ptr(), args, exs[s].data_ptr())
                shards[s].synchronize()
            if raw_sharded:
                shards[r].ivfpq_merge_replay_exact(W, nf, xs.data_ptr(), stride, vals.data_ptr(), ids.data_ptr(), off.data_ptr(),
                                                   exs.data_ptr(), k, args, d_list, Dr.data_ptr(), Ir.data_ptr())
            else:
                shards[r].ivfpq_merge_replay(W, nf, xs.data_ptr(), stride, vals.data_ptr(), ids.data_ptr(), off.data_ptr(), k, args,
                                             d_list, Dr.data_ptr(), Ir.data_ptr())
        shards[r].synchronize()
        D[q0:q1] = Dr
        I[q0:q1] = Ir
    return D, I, flagged
