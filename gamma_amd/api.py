"""Thin Python face of the libgamma_hip.so C ABI (include/gamma_hip.h), used by the tests,
bench.py and the multi-GPU driver.  numpy in / numpy out for host calls; raw device
pointers (ints) for the *_device entry points.  All compute happens in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import GammaHipError, RangeFilter, SearchParams

METRIC_IP = 0
METRIC_L2 = 1

FLT_TINY = float(np.finfo(np.float32).tiny)
FLT_MAX = float(np.finfo(np.float32).max)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def make_range_filter(docids, b_not_in=False):
    """RangeQueryResult-shaped filter (table/range_query_result.h:96-125) from matching docids.
    Returns (RangeFilter, keepalive array)."""
    docids = np.asarray(sorted(set(int(x) for x in docids)), dtype=np.int64)
    mn, mx = (int(docids[0]), int(docids[-1])) if len(docids) else (0, 0)
    min_aligned = (mn // 8) * 8
    max_aligned = (mx // 8 + 1) * 8 - 1
    bm = np.zeros(((max_aligned - min_aligned + 1) >> 3) + 1, dtype=np.uint8)
    rel = docids - min_aligned
    np.bitwise_or.at(bm, rel >> 3, (1 << (rel & 7)).astype(np.uint8))
    rf = RangeFilter(_p(bm, _lib.u8p), bm.size, mn, mx, min_aligned, 1 if b_not_in else 0)
    return rf, bm


class SearchArgs:
    """Builds a gamma_hip_search_params; keeps the filter buffers alive."""

    def __init__(self, metric=METRIC_L2, nprobe=1, recall_num=100, has_rank=True, min_score=None,
                 max_score=None, coarse_mode=-1, range_filters=None, field_filters=None, term_filters=None,
                 exact_ties=0):
        """field_filters: list of (field_id, lower, upper, include_lower, include_upper) evaluated on
        the device against columns loaded with GammaHip.field_append.
        term_filters: list of (field_id, op, item ids) -- op 0 And / 1 Or / 2 Not -- against columns loaded with
        GammaHip.term_append."""
        p = SearchParams()
        p.metric = metric
        p.nprobe = nprobe
        p.recall_num = recall_num
        p.has_rank = 1 if has_rank else 0
        # GammaSearchCondition defaults (common/gamma_common_data.h:50-51)
        p.min_score = FLT_TINY if min_score is None else min_score
        p.max_score = FLT_MAX if max_score is None else max_score
        p.coarse_mode = coarse_mode
        p.exact_ties = exact_ties   # 0: the handle's setting (default on), 1: on, -1: off
        self._keep = []
        if range_filters is not None:
            p.has_range = 1
            p.n_range = len(range_filters)
            arr = (RangeFilter * max(1, len(range_filters)))()
            for i, (rf, ka) in enumerate(range_filters):
                arr[i] = rf
                self._keep.append(ka)
            p.range = C.cast(arr, C.POINTER(RangeFilter))
            self._keep.append(arr)
        if field_filters:
            fa = (_lib.FieldFilter * len(field_filters))()
            for i, (fid, lo, hi, il, iu) in enumerate(field_filters):
                fa[i].field_id = fid
                fa[i].include_lower, fa[i].include_upper = int(bool(il)), int(bool(iu))
                fa[i].lower_i, fa[i].upper_i = int(lo), int(hi)
                fa[i].lower_f, fa[i].upper_f = float(lo), float(hi)
            p.n_field = len(field_filters)
            p.field = C.cast(fa, C.POINTER(_lib.FieldFilter))
            self._keep.append(fa)
        if term_filters:
            ta = (_lib.TermFilter * len(term_filters))()
            for i, (fid, op, items) in enumerate(term_filters):
                ta[i].field_id, ta[i].op, ta[i].n_items = fid, op, len(items)
                for j, it in enumerate(items):
                    ta[i].items[j] = int(it)
            p.n_term = len(term_filters)
            p.term = C.cast(ta, C.POINTER(_lib.TermFilter))
            self._keep.append(ta)
        self.p = p

    def ref(self):
        return C.byref(self.p)


class GammaHip:
    """One handle = one GPU (or one shard)."""

    def __init__(self, device=0):
        self.L = _lib.load()
        h = C.c_void_p()
        rc = self.L.gamma_hip_create(device, C.byref(h))
        if rc != 0:
            raise GammaHipError("gamma_hip_create(device=%d): %s" % (
                device, self.L.gamma_hip_strerror(rc).decode()))
        self.h = h
        self.d = None
        self.M = None

    def close(self):
        if getattr(self, "h", None):
            self.L.gamma_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise GammaHipError("%s: %s (%s)" % (
                what, self.L.gamma_hip_strerror(rc).decode(),
                self.L.gamma_hip_last_error(self.h).decode()))

    # ---- raw store / bitmap ----
    def raw_init(self, d):
        self._ck(self.L.gamma_hip_raw_init(self.h, d), "raw_init")
        self.raw_d = d

    def raw_append(self, vecs):
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_raw_append(self.h, vecs.shape[0], _p(vecs, _lib.f32p)), "raw_append")

    def raw_put(self, vids, vecs):
        """raw vectors sharded with their lists: the rows of THIS shard's vectors (include/gamma_hip.h)"""
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_raw_put(self.h, len(vids), _p(vids, _lib.i64p), _p(vecs, _lib.f32p)), "raw_put")

    def raw_write(self, first_vid, vecs):
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_raw_write(self.h, first_vid, vecs.shape[0], _p(vecs, _lib.f32p)), "raw_write")

    def set_deferred_replay(self, on):
        """streaming device-pointer searches: the tie replay of a call runs beside the next call's first stages; results
        of a call are complete after the next search, join() or synchronize() (include/gamma_hip.h)"""
        self._ck(self.L.gamma_hip_set_deferred_replay(self.h, 1 if on else 0), "set_deferred_replay")

    def join(self):
        self._ck(self.L.gamma_hip_join(self.h), "join")

    def raw_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.gamma_hip_raw_stats(self.h, _p(out, _lib.i64p)), "raw_stats")
        return dict(rows=int(out[0]), capacity=int(out[1]), moves=int(out[2]), in_place=bool(out[3]))

    def raw_update(self, vid, vec):
        vec = _f32(vec)
        self._ck(self.L.gamma_hip_raw_update(self.h, vid, _p(vec, _lib.f32p)), "raw_update")

    def raw_gets(self, vids):
        """VectorReader::Gets: rows of the device store by vector id"""
        vids = np.ascontiguousarray(vids, dtype=np.int64).ravel()
        out = np.empty((len(vids), self.raw_d), dtype=np.float32)
        self._ck(self.L.gamma_hip_raw_gets(self.h, len(vids), _p(vids, _lib.i64p), _p(out, _lib.f32p)), "raw_gets")
        return out

    def raw_count(self):
        return self.L.gamma_hip_raw_count(self.h)

    def bitmap_upload(self, bitmap, nbits):
        bitmap = np.ascontiguousarray(bitmap, dtype=np.uint8)
        self._ck(self.L.gamma_hip_bitmap_upload(self.h, _p(bitmap, _lib.u8p), nbits), "bitmap_upload")

    def bitmap_set(self, docids, value=1):
        docids = np.ascontiguousarray(docids, dtype=np.int64)
        self._ck(self.L.gamma_hip_bitmap_set(self.h, _p(docids, _lib.i64p), len(docids), value),
                 "bitmap_set")

    # ---- ivfpq model ----
    def ivfpq_init(self, d, nlist, M, nbits=8, metric=METRIC_L2, bucket_init_size=1000,
                   bucket_max_size=1280000):
        self._ck(self.L.gamma_hip_ivfpq_init(self.h, d, nlist, M, nbits, metric, bucket_init_size,
                                             bucket_max_size), "ivfpq_init")
        self.d, self.nlist, self.M = d, nlist, M

    def ivfpq_set_trained(self, coarse_centroids, pq_centroids, table=None):
        cc, pq = _f32(coarse_centroids), _f32(pq_centroids)
        t = _f32(table) if table is not None else None
        self._ck(self.L.gamma_hip_ivfpq_set_trained(self.h, _p(cc, _lib.f32p), _p(pq, _lib.f32p),
                                                    _p(t, _lib.f32p)), "ivfpq_set_trained")

    def use_precomputed_table(self):
        """L2 table mode of the initialised index: 1 = precomputed table resident, 0 = it would exceed
        precomputed_table_max_bytes and searches score with residual tables (include/gamma_hip.h)."""
        m = self.L.gamma_hip_ivfpq_use_precomputed_table(self.h)
        if m < 0:
            self._ck(m, "use_precomputed_table")
        return m

    def ivfpq_table(self):
        out = np.empty((self.nlist, self.M, 256), dtype=np.float32)
        self._ck(self.L.gamma_hip_ivfpq_get_precomputed_table(self.h, _p(out, _lib.f32p)), "get_table")
        return out

    def add_keys(self, list_no, vids, codes):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self._ck(self.L.gamma_hip_ivfpq_add_keys(self.h, list_no, len(vids), _p(vids, _lib.i64p),
                                                 _p(codes, _lib.u8p)), "add_keys")

    def add_keys_batch(self, list_nos, counts, vids, codes):
        list_nos = np.ascontiguousarray(list_nos, dtype=np.int32)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self._ck(self.L.gamma_hip_ivfpq_add_keys_batch(
            self.h, len(list_nos), _p(list_nos, _lib.i32p), _p(counts, _lib.i32p),
            _p(vids, _lib.i64p), _p(codes, _lib.u8p)), "add_keys_batch")

    def add_encoded(self, list_nos, codes, first_vid=0):
        """Add pre-encoded vectors (vid = first_vid + row), grouped by list in ascending list
        order like GammaIVFPQIndex::Add's std::map (gamma_index_ivfpq.cc:428-494)."""
        list_nos = np.asarray(list_nos, dtype=np.int64)
        order = np.argsort(list_nos, kind="stable")
        lists, counts = np.unique(list_nos, return_counts=True)
        self.add_keys_batch(lists, counts, first_vid + order, np.asarray(codes)[order])

    def update(self, list_no, vid, code):
        code = np.ascontiguousarray(code, dtype=np.uint8)
        self._ck(self.L.gamma_hip_ivfpq_update(self.h, list_no, vid, _p(code, _lib.u8p)), "update")

    def has_vid(self, vids):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        out = np.zeros(len(vids), dtype=np.uint8)
        self._ck(self.L.gamma_hip_ivfpq_has_vid(self.h, _p(vids, _lib.i64p), len(vids), _p(out, _lib.u8p)), "has_vid")
        return out

    def remove(self, vid):
        self._ck(self.L.gamma_hip_ivfpq_remove(self.h, int(vid)), "remove")

    def delete(self, vids):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        self._ck(self.L.gamma_hip_ivfpq_delete(self.h, _p(vids, _lib.i64p), len(vids)), "delete")

    def compact_if_need(self):
        self._ck(self.L.gamma_hip_ivfpq_compact_if_need(self.h), "compact_if_need")

    def arena_stats(self):
        out = np.zeros(4, np.int64)
        self._ck(self.L.gamma_hip_ivfpq_arena_stats(self.h, _p(out, _lib.i64p)), "arena_stats")
        return dict(zip(["cap", "used", "waste", "repacks"], [int(v) for v in out]))

    def repack_verify_stats(self):
        out = np.zeros(2, dtype=np.int64)
        self._ck(self.L.gamma_hip_ivfpq_repack_verify_stats(self.h, _p(out, _lib.i64p)), "repack_verify_stats")
        return dict(verified=int(out[0]), failures=int(out[1]))

    def arena_growth(self):
        """{moves: growths that reallocated and copied the arena, mapped: it grows in place (mapped ranges)}"""
        out = np.zeros(2, np.int64)
        self._ck(self.L.gamma_hip_ivfpq_arena_growth(self.h, _p(out, _lib.i64p)), "arena_growth")
        return {"moves": int(out[0]), "mapped": bool(out[1])}

    def set_repack_threshold(self, min_waste_entries):
        self._ck(self.L.gamma_hip_ivfpq_set_repack_threshold(self.h, min_waste_entries), "set_repack_threshold")

    def list_size(self, l):
        return self.L.gamma_hip_ivfpq_list_size(self.h, l)

    def list_capacity(self, l):
        return self.L.gamma_hip_ivfpq_list_capacity(self.h, l)

    def get_list(self, l):
        n = self.list_size(l)
        ids = np.empty(n, dtype=np.int64)
        codes = np.empty((n, self.M), dtype=np.uint8)
        self._ck(self.L.gamma_hip_ivfpq_get_list(self.h, l, _p(ids, _lib.i64p), _p(codes, _lib.u8p)),
                 "get_list")
        return ids, codes

    def set_list_mask(self, owned):
        owned = np.ascontiguousarray(owned, dtype=np.uint8) if owned is not None else None
        self._ck(self.L.gamma_hip_ivfpq_set_list_mask(self.h, _p(owned, _lib.u8p)), "set_list_mask")

    def add(self, vecs, first_vid):
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_ivfpq_add(self.h, vecs.shape[0], _p(vecs, _lib.f32p), first_vid), "add")

    def kmeans(self, x, k, niter, seed=1234, max_points_per_centroid=256):
        """faiss::Clustering::train on the device: (centroids [k, d], objective of the last assignment)"""
        x = _f32(x)
        cen = np.empty((k, x.shape[1]), dtype=np.float32)
        obj = C.c_float(0)
        self._ck(self.L.gamma_hip_kmeans(self.h, x.shape[1], x.shape[0], _p(x, _lib.f32p), k, niter, seed,
                                         max_points_per_centroid, _p(cen, _lib.f32p), C.byref(obj)), "kmeans")
        return cen, float(obj.value)

    def ivfpq_train(self, x, nlist, M):
        """IndexIVFPQ::train as GammaIVFPQIndex::Indexing runs it, on the device: (coarse centroids [nlist, d], PQ codebooks
        [M, 256, d / M]) -- the library's own training, bit for bit (gamma_hip_ivfpq_train)"""
        x = _f32(x)
        d = x.shape[1]
        cc = np.empty((nlist, d), dtype=np.float32)
        pq = np.empty((M, 256, d // M), dtype=np.float32)
        self._ck(self.L.gamma_hip_ivfpq_train(self.h, d, x.shape[0], _p(x, _lib.f32p), nlist, M, _p(cc, _lib.f32p),
                                              _p(pq, _lib.f32p)), "ivfpq_train")
        return cc, pq

    def update_batch(self, vids, vecs):
        """GammaIVFPQIndex::Update for a batch: one encode (each vector assigned as a call of its own), list updates in
        order, one publish"""
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_ivfpq_update_batch(self.h, len(vids), _p(vids, _lib.i64p), _p(vecs, _lib.f32p)),
                 "update_batch")

    def code_size(self):
        return self.L.gamma_hip_ivfpq_code_size(self.h)

    def encode(self, vecs):
        vecs = _f32(vecs)
        n = vecs.shape[0]
        lno = np.empty(n, dtype=np.int64)
        codes = np.empty((n, self.M), dtype=np.uint8)
        self._ck(self.L.gamma_hip_ivfpq_encode(self.h, n, _p(vecs, _lib.f32p), _p(lno, _lib.i64p),
                                               _p(codes, _lib.u8p)), "encode")
        return lno, codes

    # ---- search (host buffers) ----
    def ivfpq_search(self, x, k, args):
        x = _f32(x)
        nq = x.shape[0]
        D = np.empty((nq, max(k, 0)), dtype=np.float32)
        I = np.empty((nq, max(k, 0)), dtype=np.int64)
        self._ck(self.L.gamma_hip_ivfpq_search(self.h, args.ref(), nq, _p(x, _lib.f32p), k,
                                               _p(D, _lib.f32p), _p(I, _lib.i64p)), "ivfpq_search")
        return D, I

    def vid2docid_append(self, docids):
        """docid of the next vids (multi-vector documents); never called = docid == vid"""
        m = np.ascontiguousarray(docids, dtype=np.int32)
        self._ck(self.L.gamma_hip_vid2docid_append(self.h, m.size, m.ctypes.data_as(C.POINTER(C.c_int32))), "vid2docid_append")

    def vid2docid_count(self):
        return self.L.gamma_hip_vid2docid_count(self.h)

    # ---- IVFFLAT ----
    def ivfflat_init(self, d, nlist, metric=METRIC_L2, bucket_init_size=1000, bucket_max_size=1280000):
        self._ck(self.L.gamma_hip_ivfflat_init(self.h, d, nlist, metric, bucket_init_size, bucket_max_size), "ivfflat_init")
        self.d, self.nlist, self.M = d, nlist, 1

    def ivfflat_set_trained(self, coarse_centroids):
        cc = _f32(coarse_centroids)
        self._ck(self.L.gamma_hip_ivfflat_set_trained(self.h, _p(cc, _lib.f32p)), "ivfflat_set_trained")

    def ivfflat_search(self, x, k, args):
        x = _f32(x)
        nq = x.shape[0]
        D = np.empty((nq, max(k, 0)), dtype=np.float32)
        I = np.empty((nq, max(k, 0)), dtype=np.int64)
        self._ck(self.L.gamma_hip_ivfflat_search(self.h, args.ref(), nq, _p(x, _lib.f32p), k, _p(D, _lib.f32p),
                                                 _p(I, _lib.i64p)), "ivfflat_search")
        return D, I

    def ivfflat_search_device(self, d_x, nq, k, args, d_D, d_I):
        self._ck(self.L.gamma_hip_ivfflat_search_device(self.h, args.ref(), nq, d_x, k, d_D, d_I), "ivfflat_search_device")

    def last_stages(self, nq, nprobe, R):
        cd = np.empty((nq, nprobe), dtype=np.float32)
        ci = np.empty((nq, nprobe), dtype=np.int64)
        rd = np.empty((nq, R), dtype=np.float32)
        ri = np.empty((nq, R), dtype=np.int64)
        self._ck(self.L.gamma_hip_ivfpq_last_stages(self.h, _p(cd, _lib.f32p), _p(ci, _lib.i64p),
                                                    _p(rd, _lib.f32p), _p(ri, _lib.i64p)), "last_stages")
        return dict(coarse_dis=cd, coarse_idx=ci, recall_dis=rd, recall_ids=ri)

    def flat_search(self, x, k, args):
        x = _f32(x)
        nq = x.shape[0]
        D = np.empty((nq, max(k, 0)), dtype=np.float32)
        I = np.empty((nq, max(k, 0)), dtype=np.int64)
        self._ck(self.L.gamma_hip_flat_search(self.h, args.ref(), nq, _p(x, _lib.f32p), k,
                                              _p(D, _lib.f32p), _p(I, _lib.i64p)), "flat_search")
        return D, I

    # ---- search (device pointers, async on the handle's stream) ----
    def ivfpq_search_device(self, d_x, nq, k, args, d_D, d_I):
        self._ck(self.L.gamma_hip_ivfpq_search_device(self.h, args.ref(), nq, d_x, k, d_D, d_I),
                 "ivfpq_search_device")

    def ivfpq_search_device_wait(self, d_x, nq, k, args, d_D, d_I):
        """complete when it returns; concurrent callers' calls overlap (include/gamma_hip.h)"""
        self._ck(self.L.gamma_hip_ivfpq_search_device_wait(self.h, args.ref(), nq, d_x, k, d_D, d_I),
                 "ivfpq_search_device_wait")

    def flat_search_device_wait(self, d_x, nq, k, args, d_D, d_I):
        """complete when it returns; concurrent callers' calls overlap (include/gamma_hip.h)"""
        self._ck(self.L.gamma_hip_flat_search_device_wait(self.h, args.ref(), nq, d_x, k, d_D, d_I), "flat_search_device_wait")

    def flat_search_device(self, d_x, nq, k, args, d_D, d_I):
        self._ck(self.L.gamma_hip_flat_search_device(self.h, args.ref(), nq, d_x, k, d_D, d_I),
                 "flat_search_device")

    def ivfpq_search_shard(self, d_x, nq, k, args, d_rdis, d_rids):
        self._ck(self.L.gamma_hip_ivfpq_search_shard(self.h, args.ref(), nq, d_x, k, d_rdis, d_rids),
                 "ivfpq_search_shard")

    def ivfpq_coarse_device(self, d_x, nq, args, d_cdis, d_probe):
        self._ck(self.L.gamma_hip_ivfpq_coarse_device(self.h, args.ref(), nq, d_x, d_cdis, d_probe),
                 "ivfpq_coarse_device")

    def ivfpq_search_shard_preassigned(self, d_x, nq, d_cdis, d_probe, k, args, d_rdis, d_rids):
        self._ck(self.L.gamma_hip_ivfpq_search_shard_preassigned(self.h, args.ref(), nq, d_x, d_cdis,
                                                                  d_probe, k, d_rdis, d_rids),
                 "ivfpq_search_shard_preassigned")

    def ivfpq_search_shard_bounded(self, d_x, nq, d_cdis, d_probe, k, args, d_rdis, d_rids, d_bound, reduce=None):
        """two-phase shard search (include/gamma_hip.h): reduce(nq, take_max) is called ONCE from inside the call, after the
        shard's own bounds were written to d_bound [nq] floats and before its consumers read them back -- the caller's
        collective over d_bound (min / max across the shards), enqueued on the handle's stream; None = a single shard"""
        err = []

        def _cb(user, ptr, n, take_max, stream):
            try:
                reduce(n, bool(take_max))
                return 0
            except BaseException as e:   # an exception cannot cross the C frames: the call fails with EDEVICE, the error is re-raised
                err.append(e)
                return 1
        cb = _lib.BOUND_REDUCE_FN(_cb) if reduce is not None else None
        rc = self.L.gamma_hip_ivfpq_search_shard_bounded(self.h, args.ref(), nq, d_x, d_cdis, d_probe, k, d_rdis, d_rids,
                                                         d_bound, C.cast(cb, C.c_void_p) if cb is not None else None, None)
        if err:
            raise err[0]
        self._ck(rc, "ivfpq_search_shard_bounded")

    def ivfpq_merge_rerank(self, nshards, nq, d_x, k, args, d_all_dis, d_all_ids, q0, nq_local, d_D, d_I):
        self._ck(self.L.gamma_hip_ivfpq_merge_rerank(self.h, args.ref(), nshards, nq, d_x, k, d_all_dis,
                                                     d_all_ids, q0, nq_local, d_D, d_I), "merge_rerank")

    # exact ties across list shards (include/gamma_hip.h): all pointers are device addresses (ints)
    def ivfpq_shard_cut_flags(self, nq, d_flags):
        self._ck(self.L.gamma_hip_ivfpq_shard_cut_flags(self.h, nq, d_flags), "shard_cut_flags")

    def ivfpq_merge_set_shard_flags(self, d_flags):
        self._ck(self.L.gamma_hip_ivfpq_merge_set_shard_flags(self.h, d_flags), "merge_set_shard_flags")

    def ivfpq_merge_flagged(self):
        """(n_flagged, device address of the int32 list of flagged slice-local queries) after ivfpq_merge_rerank"""
        import ctypes as C
        n = C.c_int(0)
        ptr = C.c_void_p(0)
        self._ck(self.L.gamma_hip_ivfpq_merge_flagged(self.h, C.byref(n), C.byref(ptr)), "merge_flagged")
        return int(n.value), (ptr.value or 0)

    def gather_rows(self, d_src, row_words, d_list, n, d_dst):
        self._ck(self.L.gamma_hip_gather_rows(self.h, d_src, row_words, d_list, n, d_dst), "gather_rows")

    def max_list_len(self):
        return self.L.gamma_hip_ivfpq_max_list_len(self.h)

    def ivfpq_shard_export_rows(self, nf, d_probe_f, args):
        out = np.zeros(1, dtype=np.int64)
        self._ck(self.L.gamma_hip_ivfpq_shard_export_rows(self.h, args.ref(), nf, d_probe_f, _p(out, _lib.i64p)), "shard_export_rows")
        return int(out[0])

    def ivfpq_shard_export(self, nf, d_xf, d_cdis_f, d_probe_f, stride, args, d_vals, d_ids, d_off):
        self._ck(self.L.gamma_hip_ivfpq_shard_export(self.h, args.ref(), nf, d_xf, d_cdis_f, d_probe_f, stride, d_vals, d_ids,
                                                     d_off), "shard_export")

    def ivfpq_shard_exact(self, d_x, nq, d_ids, R, args, d_exact):
        self._ck(self.L.gamma_hip_ivfpq_shard_exact(self.h, args.ref(), nq, d_x, d_ids, R, d_exact), "shard_exact")

    def ivfpq_merge_rerank_exact(self, nshards, nq, d_x, k, args, d_all_dis, d_all_ids, d_all_exact, q0, nq_local, d_D, d_I):
        self._ck(self.L.gamma_hip_ivfpq_merge_rerank_exact(self.h, args.ref(), nshards, nq, d_x, k, d_all_dis, d_all_ids, d_all_exact,
                                                           q0, nq_local, d_D, d_I), "merge_rerank_exact")

    def ivfpq_shard_export_exact(self, nf, d_xf, d_vals, d_ids, d_off, stride, d_bound_f, args, d_ex):
        self._ck(self.L.gamma_hip_ivfpq_shard_export_exact(self.h, args.ref(), nf, d_xf, d_vals, d_ids, d_off, stride, d_bound_f, d_ex),
                 "shard_export_exact")

    def ivfpq_merge_replay_exact(self, nshards, nf, d_x_slice, stride, d_vals_all, d_ids_all, d_off_all, d_ex_all, k, args, d_list,
                                 d_D, d_I):
        self._ck(self.L.gamma_hip_ivfpq_merge_replay_exact(self.h, args.ref(), nshards, nf, d_x_slice, stride, d_vals_all, d_ids_all,
                                                           d_off_all, d_ex_all, k, d_list, d_D, d_I), "merge_replay_exact")

    def ivfpq_merge_replay(self, nshards, nf, d_x_slice, stride, d_vals_all, d_ids_all, d_off_all, k, args, d_list, d_D, d_I):
        self._ck(self.L.gamma_hip_ivfpq_merge_replay(self.h, args.ref(), nshards, nf, d_x_slice, stride, d_vals_all, d_ids_all,
                                                     d_off_all, k, d_list, d_D, d_I), "merge_replay")

    def debug_heap_stream(self, op, k, vals):
        """test hook: (array values, array ids, sorted values, sorted ids) of one heap fed with vals (gamma_hip.h)"""
        vals = _f32(vals).ravel()
        av, ai = np.empty(k, np.float32), np.empty(k, np.int32)
        sv, si = np.empty(k, np.float32), np.empty(k, np.int32)
        self._ck(self.L.gamma_hip_debug_heap_stream(self.h, op, k, len(vals), _p(vals, _lib.f32p), _p(av, _lib.f32p),
                                                    _p(ai, _lib.i32p), _p(sv, _lib.f32p), _p(si, _lib.i32p)), "debug_heap_stream")
        return av, ai, sv, si

    def set_scan_bound_feedback(self, on):
        self._ck(self.L.gamma_hip_set_scan_bound_feedback(self.h, 1 if on else 0), "set_scan_bound_feedback")

    def scan_bound_stats(self):
        out = np.zeros(4, np.int64)
        self._ck(self.L.gamma_hip_scan_bound_stats(self.h, _p(out, _lib.i64p)), "scan_bound_stats")
        return dict(zip(["fell_through", "queries", "backoffs", "gave_up_waiting"], [int(v) for v in out]))

    def set_dist_budget(self, nbytes):
        self._ck(self.L.gamma_hip_set_workspace_budget(self.h, int(nbytes)), "set_workspace_budget")

    def set_coarse_fused(self, on=True, list_cap=128):
        self._ck(self.L.gamma_hip_set_coarse_fused(self.h, 1 if on else 0, list_cap), "set_coarse_fused")

    def set_small_path(self, on=True):
        """True / False, or an int >= 2: on, with the two-level selection of long candidate rows forced (that many slices)."""
        self._ck(self.L.gamma_hip_set_small_path(self.h, int(on)), "set_small_path")

    def tie_stats(self, reset=False):
        out = np.zeros(3, np.int64)
        self._ck(self.L.gamma_hip_tie_stats(self.h, _p(out, _lib.i64p), 1 if reset else 0), "tie_stats")
        return dict(coarse_rows=int(out[0]), cut_ties=int(out[1]), replayed=int(out[2]))

    def ties_not_honoured(self, reset=False):
        """search calls that ran without the exact-ties mode although the handle's default asked for it (shape beyond the
        replay's range: nprobe > 1024, flat k = 4096)"""
        out = np.zeros(1, dtype=np.int64)
        self._ck(self.L.gamma_hip_ties_not_honoured(self.h, _p(out, _lib.i64p), 1 if reset else 0), "ties_not_honoured")
        return int(out[0])

    def blas_form_not_restated(self, reset=False):
        """calls whose GEMM-form coarse distances fell into a shape whose MKL kernel is not restated (ulp-level differences)"""
        out = np.zeros(1, dtype=np.int64)
        self._ck(self.L.gamma_hip_blas_form_not_restated(self.h, _p(out, _lib.i64p), 1 if reset else 0), "blas_form_not_restated")
        return int(out[0])

    def set_exact_ties(self, on=True):
        """probe exactly the lists the reference's heap keeps when coarse distances tie at the nprobe boundary"""
        self._ck(self.L.gamma_hip_set_exact_ties(self.h, 1 if on else 0), "set_exact_ties")

    # ---- numeric columns for on-device range filters ----
    _FIELD_DTYPES = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float32): 2,
                     np.dtype(np.float64): 3}

    def field_append(self, field_id, values):
        values = np.ascontiguousarray(values)
        dt = self._FIELD_DTYPES[values.dtype]
        self._ck(self.L.gamma_hip_field_append(self.h, field_id, dt, values.size, values.ctypes.data),
                 "field_append")

    def field_update(self, field_id, docid, value):
        value = np.ascontiguousarray(value)
        self._ck(self.L.gamma_hip_field_update(self.h, field_id, docid, value.ctypes.data), "field_update")

    def term_append(self, field_id, docs_items):
        """docs_items: one sequence of dictionary-encoded item ids per doc (docid = row)"""
        counts = np.ascontiguousarray([len(d) for d in docs_items], dtype=np.int32)
        flat = np.ascontiguousarray([t for d in docs_items for t in d], dtype=np.int32)
        self._ck(self.L.gamma_hip_term_append(self.h, field_id, len(counts), counts.ctypes.data_as(C.POINTER(C.c_int32)),
                                              flat.ctypes.data_as(C.POINTER(C.c_int32))), "term_append")

    def term_update(self, field_id, docid, items):
        """rewrite doc `docid`'s items (its STRING field was updated)"""
        it = np.ascontiguousarray(items, dtype=np.int32)
        self._ck(self.L.gamma_hip_term_update(self.h, field_id, docid, len(it), it.ctypes.data_as(C.POINTER(C.c_int32))),
                 "term_update")

    def term_count(self, field_id):
        return self.L.gamma_hip_term_count(self.h, field_id)

    def field_count(self, field_id):
        return self.L.gamma_hip_field_count(self.h, field_id)

    # ---- misc ----
    def stream(self):
        return self.L.gamma_hip_stream(self.h)

    def synchronize(self):
        self._ck(self.L.gamma_hip_synchronize(self.h), "synchronize")

    def total_mem_bytes(self):
        return self.L.gamma_hip_total_mem_bytes(self.h)

    def profile_enable(self, on=True):
        # on: False / 0 off, True / 1 every stage + scanned-code counter, 2 the scan stage alone (include/gamma_hip.h)
        self._ck(self.L.gamma_hip_profile_enable(self.h, int(on)), "profile_enable")

    def profile_reset(self):
        self._ck(self.L.gamma_hip_profile_reset(self.h), "profile_reset")

    def profile(self):
        out = {}
        for i, name in enumerate(_lib.STAGE_NAMES):
            ms, n = C.c_double(), C.c_int64()
            self._ck(self.L.gamma_hip_profile_get(self.h, i, C.byref(ms), C.byref(n)), "profile_get")
            out[name] = (ms.value, n.value)
        b, pr = C.c_int64(), C.c_int64()
        self._ck(self.L.gamma_hip_profile_scan_bytes(self.h, C.byref(b), C.byref(pr)), "scan_bytes")
        out["scan_bytes"] = b.value
        out["scan_pairs"] = pr.value
        return out


class GammaHipGroup:
    """Several GPUs behind one index object in ONE process (include/gamma_hip.h, gamma_hip_group_*): member i is an
    ordinary handle on devices[i] owning the lists owner(l) == i; replicated state (centroids, codebooks, raw vectors,
    bitmap) is broadcast through `members`.  The multi-process form of the same search is gamma_amd.dist."""

    def __init__(self, devices):
        self.L = _lib.load()
        devs = (C.c_int * len(devices))(*devices)
        g = C.c_void_p()
        rc = self.L.gamma_hip_group_create(devs, len(devices), C.byref(g))
        if rc != 0:
            raise GammaHipError("gamma_hip_group_create(%s): %s" % (list(devices), self.L.gamma_hip_strerror(rc).decode()))
        self.g = g
        self.members = []
        for i in range(len(devices)):
            m = GammaHip.__new__(GammaHip)       # borrowed handle: the group owns it
            m.L = self.L
            m.h = C.c_void_p(self.L.gamma_hip_group_member(g, i))
            m.d = m.M = None
            m.close = lambda: None
            self.members.append(m)

    def close(self):
        if getattr(self, "g", None):
            for m in self.members:
                m.h = None
            self.L.gamma_hip_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise GammaHipError("%s: %s (%s)" % (what, self.L.gamma_hip_strerror(rc).decode(),
                                                 self.L.gamma_hip_group_last_error(self.g).decode()))

    def each(self, fn):
        """replicated state: the same call on every member"""
        return [fn(m) for m in self.members]

    def set_owners(self, weights=None):
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.int64)
        self._ck(self.L.gamma_hip_group_set_owners(self.g, None if w is None else _p(w, _lib.i64p)), "group_set_owners")

    def owner(self, l):
        return self.L.gamma_hip_group_owner(self.g, l)

    def set_placement(self, replicate):
        """before set_owners: True = every member holds every list and a search splits the queries"""
        self._ck(self.L.gamma_hip_group_set_placement(self.g, 1 if replicate else 0), "group_set_placement")

    def add(self, vecs, first_vid):
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_group_ivfpq_add(self.g, vecs.shape[0], _p(vecs, _lib.f32p), first_vid), "group_add")

    def add_keys(self, l, vids, codes):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        self._ck(self.L.gamma_hip_group_ivfpq_add_keys(self.g, l, len(vids), _p(vids, _lib.i64p), _p(codes, _lib.u8p)),
                 "group_add_keys")

    def update(self, vids, vecs):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        vecs = _f32(vecs)
        self._ck(self.L.gamma_hip_group_ivfpq_update(self.g, len(vids), _p(vids, _lib.i64p), _p(vecs, _lib.f32p)), "group_update")

    def delete(self, vids):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        self._ck(self.L.gamma_hip_group_ivfpq_delete(self.g, _p(vids, _lib.i64p), len(vids)), "group_delete")

    def compact_if_need(self):
        self._ck(self.L.gamma_hip_group_ivfpq_compact_if_need(self.g), "group_compact_if_need")

    def list_size(self, l):
        return self.L.gamma_hip_group_ivfpq_list_size(self.g, l)

    def set_transport(self, rccl):
        self._ck(self.L.gamma_hip_group_set_transport(self.g, 1 if rccl else 0), "group_set_transport")

    def transport(self):
        """{rccl: a communicator is in use, rccl_searches: searches exchanged through it, note: why (not)}"""
        out = np.zeros(2, np.int64)
        self._ck(self.L.gamma_hip_group_transport(self.g, _p(out, _lib.i64p)), "group_transport")
        note = self.L.gamma_hip_group_transport_note(self.g)
        return {"rccl": bool(out[0]), "rccl_searches": int(out[1]), "note": (note or b"").decode()}

    def get_list(self, l, code_size):
        n = self.list_size(l)
        ids = np.empty(n, dtype=np.int64)
        codes = np.empty((n, code_size), dtype=np.uint8)
        if n:
            self._ck(self.L.gamma_hip_group_ivfpq_get_list(self.g, l, _p(ids, _lib.i64p), _p(codes, _lib.u8p)), "group_get_list")
        return ids, codes

    def ivfpq_search(self, x, k, args):
        x = _f32(x)
        nq = x.shape[0]
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        self._ck(self.L.gamma_hip_group_ivfpq_search(self.g, C.byref(args.p), nq, _p(x, _lib.f32p), k, _p(D, _lib.f32p),
                                                     _p(I, _lib.i64p)), "group_search")
        return D, I

    def ivfpq_search_device(self, d_x, nq, k, args, d_D, d_I):
        """queries / results in the memory of member 0's device (raw pointers); synchronous"""
        self._ck(self.L.gamma_hip_group_ivfpq_search_device(self.g, C.byref(args.p), nq, C.c_void_p(d_x), k, C.c_void_p(d_D),
                                                            C.c_void_p(d_I)), "group_search_device")

    def total_mem_bytes(self):
        return self.L.gamma_hip_group_total_mem_bytes(self.g)


def set_precomputed_table_max_bytes(nbytes):
    """Process-wide, like faiss::precomputed_table_max_bytes (faiss:IndexIVFPQ.cpp:379); read by ivfpq_init."""
    rc = _lib.load().gamma_hip_set_precomputed_table_max_bytes(int(nbytes))
    if rc != 0:
        raise ValueError("precomputed_table_max_bytes must be >= 0")


def get_precomputed_table_max_bytes():
    return int(_lib.load().gamma_hip_get_precomputed_table_max_bytes())


def train_ivfpq(x, nlist, M, device=0):
    """Coarse centroids + PQ codebooks the way GammaIVFPQIndex::Indexing trains them (IndexIVFPQ::train, bit for bit), on
    the device, through a handle of its own: what bench.py and the tools build their indexes with."""
    g = GammaHip(device)
    try:
        return g.ivfpq_train(x, nlist, M)
    finally:
        g.close()

