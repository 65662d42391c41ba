"""Tie-aware comparison of (distances, ids) result tables.

Bar (BASELINE.json north_star): ids/ranks bit-exact, distances within 1e-4.  Our kernels
reproduce the reference's fp32 operation order, so the tests demand MORE: distances must be
bit-identical position by position.  Ids must be identical wherever the distance order
determines them; inside a group of exactly equal distances the reference's order is an
artefact of its binary-heap mechanics (faiss:utils/Heap.h), so there the id SETS must match,
and a tie group cut by the k boundary may keep different members."""
import numpy as np


def compare_topk(D_ref, I_ref, D_got, I_got):
    """Returns dict(n_tie_queries=..., n_boundary=...).  Raises AssertionError on mismatch."""
    D_ref = np.ascontiguousarray(D_ref, dtype=np.float32)
    D_got = np.ascontiguousarray(D_got, dtype=np.float32)
    assert D_ref.shape == D_got.shape and I_ref.shape == I_got.shape
    a, b = D_ref.view(np.uint32), D_got.view(np.uint32)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        q, r = bad[0]
        raise AssertionError("distance bits differ at %d entries, first (q=%d, rank=%d): ref=%r got=%r"
                             % (len(bad), q, r, D_ref[q, r], D_got[q, r]))
    n_tie = n_boundary = 0
    nq, k = D_ref.shape
    for q in np.argwhere((I_ref != I_got).any(axis=1)).ravel():
        n_tie += 1
        d = D_ref[q]
        start = 0
        while start < k:
            end = start + 1
            while end < k and d[end] == d[start]:
                end += 1
            ref_set, got_set = set(I_ref[q, start:end].tolist()), set(I_got[q, start:end].tolist())
            if ref_set != got_set:
                if end == k and I_ref[q, k - 1] != -1:
                    n_boundary += 1  # tie group truncated by k: membership may differ
                else:
                    raise AssertionError("ids differ outside ties: q=%d ranks[%d:%d] ref=%s got=%s"
                                         % (q, start, end, sorted(ref_set), sorted(got_set)))
            start = end
    return dict(n_tie_queries=n_tie, n_boundary=n_boundary)


def compare_search(D, I, st, Dg, Ig, sg):
    """Final (D, I) vs (Dg, Ig) given the recall-stage tables st / sg (dicts with recall_dis,
    recall_ids).  The recall stage must agree (bit-identical distances, ids up to ties).  A query
    whose recall-stage id SET differs -- possible only when equal ADC distances straddle the
    recall_num boundary, where the reference keeps whichever tied entries its heap happens to hold
    -- is excluded from the final comparison (its re-rank input legitimately differs).
    Returns the number of such queries."""
    rd_o, rd_g = st["recall_dis"].copy(), sg["recall_dis"].copy()
    rd_o[st["recall_ids"] == -1] = 0      # oracle pads with FLT_MAX, device with +-inf
    rd_g[sg["recall_ids"] == -1] = 0
    # The same one stage earlier: two centroids at exactly the same fp32 distance straddling the nprobe
    # boundary (about 1e-5 of the queries on fp32 data).  The reference probes whichever of them its
    # heap happens to hold, the device the one with the lower list number; the sorted coarse distances
    # are identical, the probed SET differs, and everything downstream legitimately does too.
    keep = np.ones(len(rd_o), dtype=bool)
    if "coarse_idx" in st and "coarse_idx" in sg and st["coarse_idx"].shape == sg["coarse_idx"].shape:
        for qi, (a, b) in enumerate(zip(st["coarse_idx"], sg["coarse_idx"])):
            if set(a.tolist()) != set(b.tolist()):
                assert st["coarse_dis"][qi].tobytes() == sg["coarse_dis"][qi].tobytes(), \
                    "probed lists differ without a tie in the coarse distances (q=%d)" % qi
                keep[qi] = False
    compare_topk(rd_o[keep], st["recall_ids"][keep], rd_g[keep], sg["recall_ids"][keep])
    same = keep & np.array([set(a.tolist()) == set(b.tolist())
                            for a, b in zip(st["recall_ids"], sg["recall_ids"])])
    if same.any():
        compare_topk(D[same], I[same], Dg[same], Ig[same])
    return int((~same).sum())


def compare_exact(D_ref, I_ref, D_got, I_got):
    """Strict form (exact-ties mode, gamma_hip_set_exact_ties): distances bit-identical AND labels identical
    at every rank -- the order inside groups of equal distances, and which members of a group cut by k or
    recall_num survive, must be the reference heap's."""
    D_ref = np.ascontiguousarray(D_ref, dtype=np.float32)
    D_got = np.ascontiguousarray(D_got, dtype=np.float32)
    assert D_ref.shape == D_got.shape and I_ref.shape == I_got.shape
    a, b = D_ref.view(np.uint32), D_got.view(np.uint32)
    pad = (I_ref == -1) & (I_got == -1)      # the reference pads with +-FLT_MAX; either padding value is fine
    if not np.array_equal(a[~pad], b[~pad]):
        bad = np.argwhere((a != b) & ~pad)
        q, r = bad[0]
        raise AssertionError("distance bits differ at %d entries, first (q=%d, rank=%d): ref=%r got=%r"
                             % (len(bad), q, r, D_ref[q, r], D_got[q, r]))
    if not np.array_equal(I_ref, I_got):
        bad = np.argwhere(I_ref != I_got)
        q, r = bad[0]
        raise AssertionError("labels differ at %d entries (%d queries), first (q=%d, rank=%d): ref=%s got=%s"
                             % (len(bad), len(set(bad[:, 0].tolist())), q, r, I_ref[q].tolist(), I_got[q].tolist()))


def _stage_rows(dis, ids):
    """Recall-stage rows as canonically ordered (distance bits, id) pairs; padding (-1) normalised."""
    dis = np.ascontiguousarray(dis, dtype=np.float32).copy()
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    dis[ids == -1] = 0
    order = np.lexsort((ids, dis.view(np.uint32)), axis=1)
    return np.take_along_axis(dis.view(np.uint32), order, axis=1), np.take_along_axis(ids, order, axis=1)


def compare_search_exact(D, I, st, Dg, Ig, sg):
    """Strict form of compare_search (exact ties on, the default): NO query is excluded.  The probed lists must
    be the reference's in the reference's order (coarse distances bit-identical), the recall stage must hold
    the same (distance, id) pairs -- which members of a tie group cut by recall_num survive is the reference
    heap's choice and the device has to make the same one -- and the final table must pass compare_exact."""
    if "coarse_idx" in st and "coarse_idx" in sg and st["coarse_idx"].shape == sg["coarse_idx"].shape:
        if not np.array_equal(st["coarse_idx"], sg["coarse_idx"]):
            bad = np.argwhere((st["coarse_idx"] != sg["coarse_idx"]).any(axis=1)).ravel()
            raise AssertionError("probed lists differ for %d queries, first q=%d: ref=%s got=%s"
                                 % (len(bad), bad[0], st["coarse_idx"][bad[0]].tolist(), sg["coarse_idx"][bad[0]].tolist()))
        assert st["coarse_dis"].tobytes() == sg["coarse_dis"].tobytes(), "coarse distances differ"
    a_d, a_i = _stage_rows(st["recall_dis"], st["recall_ids"])
    b_d, b_i = _stage_rows(sg["recall_dis"], sg["recall_ids"])
    if not (np.array_equal(a_i, b_i) and np.array_equal(a_d, b_d)):
        bad = np.argwhere((a_i != b_i).any(axis=1) | (a_d != b_d).any(axis=1)).ravel()
        raise AssertionError("recall-stage (distance, id) sets differ for %d queries, first q=%d" % (len(bad), bad[0]))
    compare_exact(D, I, Dg, Ig)
    return 0
