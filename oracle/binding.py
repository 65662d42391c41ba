"""ctypes bindings for the CPU oracle (oracle/libgamma_oracle.so) and, when present, the
real-faiss reference build (oracle/_ref/libgamma_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  gamma_amd/ never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libgamma_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libgamma_ref.so")

METRIC_IP = 0
METRIC_L2 = 1

_f32p = C.POINTER(C.c_float)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)


def _fp(a):
    return a.ctypes.data_as(_f32p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(_i64p) if a is not None else None


def _up(a):
    return a.ctypes.data_as(_u8p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE])


_lib = None


def usable_cpus():
    """Threads worth starting: the hardware threads this process may run on, capped by a container CPU quota (cgroup v2
    cpu.max).  The GPU boxes show 256 hardware threads under a 16-core quota; OpenMP's default of one thread per
    hardware thread then turns every barrier of the k-means loops into a spin against the throttle."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        qv, pv = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if qv != "max":
            n = min(n, max(1, int(round(float(qv) / float(pv)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cap_openmp_threads():
    if "OMP_NUM_THREADS" in os.environ:
        return
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(usable_cpus())
    except OSError:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        L = C.CDLL(ORACLE_SO)
        _cap_openmp_threads()
        L.go_fvec_L2sqr.restype = C.c_float
        L.go_fvec_L2sqr.argtypes = [_f32p, _f32p, C.c_size_t]
        L.go_fvec_inner_product.restype = C.c_float
        L.go_fvec_inner_product.argtypes = [_f32p, _f32p, C.c_size_t]
        L.go_fvec_norm_L2sqr.restype = C.c_float
        L.go_fvec_norm_L2sqr.argtypes = [_f32p, C.c_size_t]
        for n in ("go_fvec_inner_products_ny", "go_fvec_L2sqr_ny"):
            getattr(L, n).restype = None
            getattr(L, n).argtypes = [_f32p, _f32p, _f32p, C.c_size_t, C.c_size_t]
        L.go_fvec_madd.restype = None
        L.go_fvec_madd.argtypes = [C.c_size_t, _f32p, C.c_float, _f32p, _f32p]
        L.go_heap_stream.restype = None
        L.go_heap_stream.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _f32p, _i64p, _f32p, _i64p,
                                     _f32p, _i64p]
        L.go_heap_pop_push_stream.restype = None
        L.go_heap_pop_push_stream.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _f32p, _i64p,
                                              _f32p, _i64p]
        L.go_reservoir_capacity.restype = C.c_size_t
        L.go_reservoir_capacity.argtypes = [C.c_size_t]
        L.go_reservoir_stream.restype = None
        L.go_reservoir_stream.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _f32p, _i64p, _f32p, _i64p]
        L.go_knn_L2sqr.restype = None
        L.go_knn_L2sqr.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t, C.c_size_t, C.c_size_t,
                                   C.c_size_t, _f32p, _i64p]
        L.go_rand_perm.restype = None
        L.go_rand_perm.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.c_int64]
        L.go_set_kmeans_assign_mode.argtypes = [C.c_int]
        L.go_kmeans.restype = C.c_float
        L.go_kmeans.argtypes = [C.c_int, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int64, C.c_int, _f32p]
        L.go_ivfpq_train.restype = None
        L.go_ivfpq_train.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, _f32p, _f32p, _f32p]
        L.go_knn_inner_product.restype = None
        L.go_knn_inner_product.argtypes = [_f32p, _f32p, C.c_size_t, C.c_size_t, C.c_size_t,
                                           C.c_size_t, _f32p, _i64p]
        L.go_pq_inner_prod_table.restype = None
        L.go_pq_inner_prod_table.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _f32p]
        L.go_pq_compute_codes.restype = None
        L.go_pq_compute_codes.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, _f32p, _u8p, C.c_size_t]
        L.go_ivfpq_precompute_table.restype = None
        L.go_ivfpq_precompute_table.argtypes = [_f32p, C.c_int, C.c_int, _f32p, C.c_int, C.c_int,
                                                _f32p]
        L.go_ivfpq_new.restype = C.c_void_p
        L.go_ivfpq_new.argtypes = [C.c_int] * 7
        L.go_ivfpq_free.restype = None
        L.go_ivfpq_free.argtypes = [C.c_void_p]
        L.go_ivfpq_set_trained.restype = None
        L.go_ivfpq_set_trained.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        L.go_ivfpq_table.restype = _f32p
        L.go_ivfpq_table.argtypes = [C.c_void_p]
        L.go_set_precomputed_table_max_bytes.restype = None
        L.go_set_precomputed_table_max_bytes.argtypes = [C.c_size_t]
        L.go_get_precomputed_table_max_bytes.restype = C.c_size_t
        L.go_ivfpq_use_precomputed_table.restype = C.c_int
        L.go_ivfpq_use_precomputed_table.argtypes = [C.c_void_p]
        L.go_ivfpq_set_docids_bitmap.restype = None
        L.go_ivfpq_set_docids_bitmap.argtypes = [C.c_void_p, _u8p, C.c_int64]
        L.go_ivfpq_set_raw.restype = None
        L.go_ivfpq_set_raw.argtypes = [C.c_void_p, _f32p, C.c_int64]
        L.go_ivfpq_add.restype = C.c_int
        L.go_ivfpq_add.argtypes = [C.c_void_p, C.c_int64, _f32p]
        L.go_ivfpq_encode.restype = None
        L.go_ivfpq_encode.argtypes = [C.c_void_p, C.c_int64, _f32p, _i64p, _u8p]
        L.go_ivfpq_add_keys.restype = C.c_int
        L.go_ivfpq_add_keys.argtypes = [C.c_void_p, C.c_int, C.c_int, _i64p, _u8p]
        L.go_ivfpq_update.restype = C.c_int
        L.go_ivfpq_update.argtypes = [C.c_void_p, C.c_int64, _f32p]
        L.go_ivfpq_update_code.restype = C.c_int
        L.go_ivfpq_update_code.argtypes = [C.c_void_p, C.c_int, C.c_int64, _u8p]
        L.go_ivfflat_search.restype = C.c_int
        L.go_ivfflat_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int,
                                        _f32p, _i64p, _f32p, _i64p]
        L.go_ivfflat_assign.restype = None
        L.go_ivfflat_assign.argtypes = [C.c_void_p, C.c_int64, _f32p, _i64p]
        L.go_ivfpq_set_vid2docid.restype = None
        L.go_ivfpq_set_vid2docid.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int64]
        L.go_ivfpq_has_vid.restype = C.c_int
        L.go_ivfpq_has_vid.argtypes = [C.c_void_p, C.c_int64]
        L.go_ivfpq_remove.restype = C.c_int
        L.go_ivfpq_remove.argtypes = [C.c_void_p, C.c_int64]
        L.go_ivfpq_delete.restype = C.c_int
        L.go_ivfpq_delete.argtypes = [C.c_void_p, _i64p, C.c_int, _u8p]
        L.go_ivfpq_compact_if_need.restype = C.c_int
        L.go_ivfpq_compact_if_need.argtypes = [C.c_void_p, _u8p]
        L.go_ivfpq_list_size.restype = C.c_int64
        L.go_ivfpq_list_size.argtypes = [C.c_void_p, C.c_int]
        L.go_ivfpq_list_capacity.restype = C.c_int64
        L.go_ivfpq_list_capacity.argtypes = [C.c_void_p, C.c_int]
        L.go_ivfpq_get_list.restype = None
        L.go_ivfpq_get_list.argtypes = [C.c_void_p, C.c_int, _i64p, _u8p]
        L.go_ivfpq_vid_pos.restype = C.c_int64
        L.go_ivfpq_vid_pos.argtypes = [C.c_void_p, C.c_int64]
        L.go_ivfpq_search.restype = C.c_int
        L.go_ivfpq_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, _f32p, C.c_int, _f32p, _i64p, _f32p, _i64p,
                                      _f32p, _i64p]
        L.go_flat_search.restype = C.c_int
        L.go_flat_search.argtypes = [_f32p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, _f32p,
                                     C.c_int, _f32p, _i64p]
        L.go_num_threads.restype = C.c_int
        L.go_set_assign_mode.restype = None
        L.go_set_assign_mode.argtypes = [C.c_int]
        _lib = L
    return _lib


class RangeFilter(C.Structure):
    _fields_ = [("bitmap", _u8p), ("min_doc", C.c_int32), ("max_doc", C.c_int32),
                ("min_aligned", C.c_int32), ("b_not_in", C.c_int32)]


class SearchCtx(C.Structure):
    _fields_ = [("docids_bitmap", _u8p), ("docids_bitmap_bits", C.c_int64),
                ("has_range", C.c_int32), ("n_range", C.c_int32),
                ("range", C.POINTER(RangeFilter)), ("min_score", C.c_float),
                ("max_score", C.c_float), ("vid2docid", C.POINTER(C.c_int32)), ("n_vid2docid", C.c_int64)]


def make_range_filter(docids, n_total=None, b_not_in=False):
    """Build a RangeQueryResult-shaped filter (table/range_query_result.h:96-125) from a
    collection of matching docids.  Returns (RangeFilter, keepalive)."""
    docids = np.asarray(sorted(set(int(x) for x in docids)), dtype=np.int64)
    if len(docids) == 0:
        mn, mx = 0, 0
    else:
        mn, mx = int(docids[0]), int(docids[-1])
    min_aligned = (mn // 8) * 8
    max_aligned = (mx // 8 + 1) * 8 - 1
    nbits = max_aligned - min_aligned + 1
    bm = np.zeros((nbits >> 3) + 1, dtype=np.uint8)
    rel = docids - min_aligned
    np.bitwise_or.at(bm, rel >> 3, (1 << (rel & 7)).astype(np.uint8))
    rf = RangeFilter(_up(bm), mn, mx, min_aligned, 1 if b_not_in else 0)
    return rf, bm


def make_ctx(docids_bitmap=None, range_filters=None, min_score=None, max_score=None, vid2docid=None):
    """docids_bitmap: np.uint8 delete bitmap (bit set = deleted) or None.
    range_filters: None (no range_query_result) or list of (RangeFilter, keepalive)."""
    ctx = SearchCtx()
    keep = []
    if docids_bitmap is not None:
        docids_bitmap = np.ascontiguousarray(docids_bitmap, dtype=np.uint8)
        ctx.docids_bitmap = _up(docids_bitmap)
        ctx.docids_bitmap_bits = docids_bitmap.size * 8
        keep.append(docids_bitmap)
    if range_filters is not None:
        ctx.has_range = 1
        ctx.n_range = len(range_filters)
        arr = (RangeFilter * max(1, len(range_filters)))()
        for i, (rf, ka) in enumerate(range_filters):
            arr[i] = rf
            keep.append(ka)
        ctx.range = C.cast(arr, C.POINTER(RangeFilter))
        keep.append(arr)
    # GammaSearchCondition defaults (common/gamma_common_data.h:50-51)
    ctx.min_score = np.finfo(np.float32).tiny if min_score is None else min_score
    ctx.max_score = np.finfo(np.float32).max if max_score is None else max_score
    if vid2docid is not None:
        vid2docid = np.ascontiguousarray(vid2docid, dtype=np.int32)
        ctx.vid2docid = vid2docid.ctypes.data_as(C.POINTER(C.c_int32))
        ctx.n_vid2docid = vid2docid.size
        keep.append(vid2docid)
    ctx._keep = keep
    return ctx


class OracleIVFPQ:
    """Python face of go_ivfpq (mirrors GammaIVFPQIndex's role in the reference)."""

    def __init__(self, d, nlist, M, nbits=8, metric=METRIC_L2, bucket_init_size=1000,
                 bucket_max_size=1280000):
        self.L = lib()
        self.d, self.nlist, self.M, self.metric = d, nlist, M, metric
        self.ksub = 1 << nbits
        self.h = self.L.go_ivfpq_new(d, nlist, M, nbits, metric, bucket_init_size, bucket_max_size)
        if not self.h:
            raise ValueError("go_ivfpq_new failed (nbits must be 8, d % M == 0)")
        self._raw = None

    def __del__(self):
        if getattr(self, "h", None):
            self.L.go_ivfpq_free(self.h)
            self.h = None

    def set_trained(self, coarse_centroids, pq_centroids, table=None):
        cc, pq = _f32(coarse_centroids), _f32(pq_centroids)
        t = _f32(table) if table is not None else None
        self.L.go_ivfpq_set_trained(self.h, _fp(cc), _fp(pq), _fp(t))

    def use_precomputed_table(self):
        """1 after set_trained unless the table would exceed precomputed_table_max_bytes (then 0: residual tables)."""
        return self.L.go_ivfpq_use_precomputed_table(self.h)

    def table(self):
        n = self.nlist * self.M * self.ksub
        p = self.L.go_ivfpq_table(self.h)
        if not p:   # table mode 0
            return None
        return np.ctypeslib.as_array(p, shape=(n,)).reshape(self.nlist, self.M, self.ksub).copy()

    def set_docids_bitmap(self, bm):
        """bm: np.uint8 array kept alive by the caller's reference here (may be mutated later)."""
        self._bm = bm
        self.L.go_ivfpq_set_docids_bitmap(self.h, _up(bm), bm.size * 8)

    def set_vid2docid(self, m):
        self._v2d = np.ascontiguousarray(m, dtype=np.int32)
        self.L.go_ivfpq_set_vid2docid(self.h, self._v2d.ctypes.data_as(C.POINTER(C.c_int32)), self._v2d.size)

    def set_raw(self, raw):
        self._raw = _f32(raw)
        self.L.go_ivfpq_set_raw(self.h, _fp(self._raw), self._raw.shape[0])

    def add(self, x):
        x = _f32(x)
        return bool(self.L.go_ivfpq_add(self.h, x.shape[0], _fp(x)))

    def encode(self, x):
        x = _f32(x)
        n = x.shape[0]
        lno = np.empty(n, dtype=np.int64)
        codes = np.empty((n, self.M), dtype=np.uint8)
        self.L.go_ivfpq_encode(self.h, n, _fp(x), _ip(lno), _up(codes))
        return lno, codes

    def add_keys(self, list_no, keys, codes):
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        return bool(self.L.go_ivfpq_add_keys(self.h, list_no, len(keys), _ip(keys), _up(codes)))

    def update(self, vid, x):
        x = _f32(x)
        return self.L.go_ivfpq_update(self.h, vid, _fp(x))

    def update_code(self, list_no, vid, code):
        code = np.ascontiguousarray(code, dtype=np.uint8)
        return self.L.go_ivfpq_update_code(self.h, int(list_no), int(vid), _up(code))

    def has_vid(self, vids):
        return np.array([self.L.go_ivfpq_has_vid(self.h, int(v)) for v in vids], dtype=np.uint8)

    def remove(self, vid):
        return self.L.go_ivfpq_remove(self.h, int(vid))

    def delete(self, vids):
        vids = np.ascontiguousarray(vids, dtype=np.int64)
        return self.L.go_ivfpq_delete(self.h, _ip(vids), len(vids), None)

    def compact_if_need(self, docids_bitmap=None):
        bm = np.ascontiguousarray(docids_bitmap, dtype=np.uint8) if docids_bitmap is not None else None
        return self.L.go_ivfpq_compact_if_need(self.h, _up(bm))

    def list_size(self, l):
        return self.L.go_ivfpq_list_size(self.h, l)

    def list_capacity(self, l):
        return self.L.go_ivfpq_list_capacity(self.h, l)

    def get_list(self, l):
        n = self.list_size(l)
        ids = np.empty(n, dtype=np.int64)
        codes = np.empty((n, self.M), dtype=np.uint8)
        if n:
            self.L.go_ivfpq_get_list(self.h, l, _ip(ids), _up(codes))
        return ids, codes

    def vid_pos(self, vid):
        return self.L.go_ivfpq_vid_pos(self.h, vid)

    def search(self, x, k, nprobe, recall_num=100, has_rank=True, metric=None, ctx=None,
               coarse_mode=-1, want_stages=False, preassigned=None):
        """preassigned = (coarse_dis [nq, nprobe] float32, coarse_idx [nq, nprobe] int64): search_preassigned with an
        assignment computed elsewhere (coarse_mode is ignored)"""
        x = _f32(x)
        nq = x.shape[0]
        metric = self.metric if metric is None else metric
        R = max(recall_num, k)
        D = np.empty((nq, k), dtype=np.float32)
        I = np.empty((nq, k), dtype=np.int64)
        cd = ci = rd = ri = None
        if want_stages or preassigned is not None:
            cd = np.empty((nq, nprobe), dtype=np.float32)
            ci = np.empty((nq, nprobe), dtype=np.int64)
            rd = np.empty((nq, R), dtype=np.float32)
            ri = np.empty((nq, R), dtype=np.int64)
        if preassigned is not None:
            cd[:] = np.asarray(preassigned[0], dtype=np.float32).reshape(nq, nprobe)
            ci[:] = np.asarray(preassigned[1], dtype=np.int64).reshape(nq, nprobe)
            coarse_mode = 2
        rc = self.L.go_ivfpq_search(self.h, C.byref(ctx) if ctx is not None else None, metric,
                                    nprobe, recall_num, 1 if has_rank else 0, coarse_mode, nq,
                                    _fp(x), k, _fp(D), _ip(I), _fp(cd), _ip(ci), _fp(rd), _ip(ri))
        if rc != 0:
            raise RuntimeError("go_ivfpq_search rc=%d" % rc)
        if want_stages:
            return D, I, dict(coarse_dis=cd, coarse_idx=ci, recall_dis=rd, recall_ids=ri)
        return D, I


def ivfflat_search(o, x, k, nprobe, metric=METRIC_L2, ctx=None, coarse_mode=-1, want_stages=False):
    """GammaIndexIVFFlat::Search over the lists (ids) and raw store of the OracleIVFPQ `o`"""
    x = _f32(x)
    nq = x.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    cd = np.empty((nq, nprobe), dtype=np.float32)
    ci = np.empty((nq, nprobe), dtype=np.int64)
    rc = lib().go_ivfflat_search(o.h, C.byref(ctx) if ctx is not None else None, metric, nprobe, coarse_mode, nq,
                                 _fp(x), k, _fp(D), _ip(I), _fp(cd), _ip(ci))
    if rc != 0:
        raise RuntimeError("go_ivfflat_search rc=%d" % rc)
    if want_stages:
        return D, I, dict(coarse_dis=cd, coarse_idx=ci)
    return D, I


def ivfflat_assign(o, x):
    x = _f32(x)
    out = np.empty(x.shape[0], dtype=np.int64)
    lib().go_ivfflat_assign(o.h, x.shape[0], _fp(x), _ip(out))
    return out


def flat_search(raw, x, k, metric=METRIC_L2, ctx=None):
    raw, x = _f32(raw), _f32(x)
    nq = x.shape[0]
    D = np.empty((nq, k), dtype=np.float32)
    I = np.empty((nq, k), dtype=np.int64)
    rc = lib().go_flat_search(_fp(raw), raw.shape[0], raw.shape[1],
                              C.byref(ctx) if ctx is not None else None, metric, nq, _fp(x), k,
                              _fp(D), _ip(I))
    if rc != 0:
        raise RuntimeError("go_flat_search rc=%d" % rc)
    return D, I


def kmeans(x, k, niter, seed=1234, max_points_per_centroid=256, assign_mode=-1):
    """faiss::Clustering::train restated (gamma_oracle.c): (centroids [k, d], objective of the last assignment)"""
    x = _f32(x)
    cen = np.empty((k, x.shape[1]), dtype=np.float32)
    lib().go_set_kmeans_assign_mode(assign_mode)
    try:
        obj = lib().go_kmeans(x.shape[1], x.shape[0], _fp(x), k, niter, seed, max_points_per_centroid, _fp(cen))
    finally:
        lib().go_set_kmeans_assign_mode(-1)
    return cen, float(obj)


def ivfpq_train(x, nlist, M):
    """IndexIVFPQ::train as GammaIVFPQIndex::Indexing configures it: (coarse centroids, PQ codebooks)"""
    x = _f32(x)
    d = x.shape[1]
    cc = np.empty((nlist, d), dtype=np.float32)
    pq = np.empty((M, 256, d // M), dtype=np.float32)
    lib().go_ivfpq_train(d, nlist, M, x.shape[0], _fp(x), _fp(cc), _fp(pq))
    return cc, pq


def knn_L2sqr(x, y, k, mode=0):
    x, y = _f32(x), _f32(y)
    D = np.empty((x.shape[0], k), dtype=np.float32)
    I = np.empty((x.shape[0], k), dtype=np.int64)
    lib().go_knn_L2sqr(mode, _fp(x), _fp(y), x.shape[1], x.shape[0], y.shape[0], k, _fp(D), _ip(I))
    return D, I


# ---------------------------------------------------------------------------------------
# Real faiss (oracle/_ref) -- only where it has been built (this container).
# ---------------------------------------------------------------------------------------
_ref = None


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        # libmkl_rt needs GNU threading next to libgomp
        os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
        R = C.CDLL(REF_SO)
        R.ref_fvec_L2sqr.restype = C.c_float
        R.ref_fvec_L2sqr.argtypes = [_f32p, _f32p, C.c_size_t]
        R.ref_fvec_inner_product.restype = C.c_float
        R.ref_fvec_inner_product.argtypes = [_f32p, _f32p, C.c_size_t]
        R.ref_fvec_norm_L2sqr.restype = C.c_float
        R.ref_fvec_norm_L2sqr.argtypes = [_f32p, C.c_size_t]
        for n in ("ref_fvec_inner_products_ny", "ref_fvec_L2sqr_ny"):
            getattr(R, n).restype = None
            getattr(R, n).argtypes = [_f32p, _f32p, _f32p, C.c_size_t, C.c_size_t]
        R.ref_fvec_madd.restype = None
        R.ref_fvec_madd.argtypes = [C.c_size_t, _f32p, C.c_float, _f32p, _f32p]
        R.ref_set_blas_threshold.argtypes = [C.c_int]
        R.ref_rand_perm.restype = None
        R.ref_rand_perm.argtypes = [C.POINTER(C.c_int), C.c_size_t, C.c_int64]
        R.ref_kmeans.restype = C.c_float
        R.ref_kmeans.argtypes = [C.c_int, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int64, _f32p]
        R.ref_get_blas_threshold.restype = C.c_int
        for n in ("ref_flat_l2_search", "ref_flat_ip_search"):
            getattr(R, n).restype = None
            getattr(R, n).argtypes = [C.c_size_t, C.c_size_t, _f32p, C.c_size_t, _f32p, C.c_size_t,
                                      _f32p, _i64p]
        R.ref_heap_stream.restype = None
        R.ref_heap_stream.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _f32p, _i64p, _f32p, _i64p,
                                      _f32p, _i64p]
        R.ref_heap_pop_push_stream.restype = None
        R.ref_heap_pop_push_stream.argtypes = [C.c_int, C.c_size_t, C.c_size_t, _f32p, _i64p,
                                               _f32p, _i64p]
        R.ref_ivfpq_new.restype = C.c_void_p
        R.ref_ivfpq_new.argtypes = [C.c_int] * 6
        R.ref_ivfpq_free.argtypes = [C.c_void_p]
        R.ref_ivfpq_train.argtypes = [C.c_void_p, C.c_int64, _f32p]
        R.ref_ivfpq_add.argtypes = [C.c_void_p, C.c_int64, _f32p]
        R.ref_ivfpq_use_precomputed_table.restype = C.c_int
        R.ref_ivfpq_use_precomputed_table.argtypes = [C.c_void_p]
        R.ref_ivfpq_set_metric.argtypes = [C.c_void_p, C.c_int]
        if hasattr(R, "ref_set_precomputed_table_max_bytes"):   # (a prebuilt oracle/_ref from before round 6 lacks it)
            R.ref_set_precomputed_table_max_bytes.restype = None
            R.ref_set_precomputed_table_max_bytes.argtypes = [C.c_size_t]
            R.ref_get_precomputed_table_max_bytes.restype = C.c_size_t
        R.ref_ivfpq_get_coarse_centroids.argtypes = [C.c_void_p, _f32p]
        R.ref_ivfpq_get_pq_centroids.argtypes = [C.c_void_p, _f32p]
        R.ref_ivfpq_precomputed_table_size.restype = C.c_int64
        R.ref_ivfpq_precomputed_table_size.argtypes = [C.c_void_p]
        R.ref_ivfpq_get_precomputed_table.argtypes = [C.c_void_p, _f32p]
        R.ref_ivfpq_list_size.restype = C.c_int64
        R.ref_ivfpq_list_size.argtypes = [C.c_void_p, C.c_int64]
        R.ref_ivfpq_get_list.argtypes = [C.c_void_p, C.c_int64, _i64p, _u8p]
        R.ref_ivfpq_search.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int64, C.c_int, _f32p,
                                       _i64p]
        if hasattr(R, "ref_ivfpq_search_rerank"):   # a prebuilt oracle/_ref from before this entry point: bench skips it
            R.ref_ivfpq_search_rerank.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int, C.c_int, C.c_int, _f32p,
                                                  _f32p, _i64p]
        R.ref_ivfpq_coarse.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int, _f32p, _i64p]
        R.ref_ivfpq_search_preassigned.argtypes = [C.c_void_p, C.c_int64, _f32p, C.c_int64, C.c_int,
                                                   _i64p, _f32p, _f32p, _i64p]
        R.ref_ivfpq_encode.argtypes = [C.c_void_p, C.c_int64, _f32p, _i64p, _u8p]
        R.ref_ivfpq_inner_prod_table.argtypes = [C.c_void_p, _f32p, _f32p]
        _ref = R
    return _ref


class RefIVFPQ:
    """faiss::IndexIVFPQ configured the way GammaIVFPQIndex::Init configures it."""

    def __init__(self, d, nlist, M, nbits=8, metric=METRIC_L2, niter=10):
        self.R = ref()
        self.d, self.nlist, self.M, self.ksub = d, nlist, M, 1 << nbits
        self.h = self.R.ref_ivfpq_new(d, nlist, M, nbits, 1 if metric == METRIC_IP else 0, niter)

    def __del__(self):
        if getattr(self, "h", None):
            self.R.ref_ivfpq_free(self.h)
            self.h = None

    def train(self, x):
        x = _f32(x)
        self.R.ref_ivfpq_train(self.h, x.shape[0], _fp(x))

    def add(self, x):
        x = _f32(x)
        self.R.ref_ivfpq_add(self.h, x.shape[0], _fp(x))

    def use_precomputed_table(self):
        return self.R.ref_ivfpq_use_precomputed_table(self.h)

    def set_metric(self, metric):
        self.R.ref_ivfpq_set_metric(self.h, 1 if metric == METRIC_IP else 0)

    def coarse_centroids(self):
        out = np.empty((self.nlist, self.d), dtype=np.float32)
        self.R.ref_ivfpq_get_coarse_centroids(self.h, _fp(out))
        return out

    def pq_centroids(self):
        out = np.empty((self.M, self.ksub, self.d // self.M), dtype=np.float32)
        self.R.ref_ivfpq_get_pq_centroids(self.h, _fp(out))
        return out

    def precomputed_table(self):
        n = self.R.ref_ivfpq_precomputed_table_size(self.h)
        out = np.empty(n, dtype=np.float32)
        self.R.ref_ivfpq_get_precomputed_table(self.h, _fp(out))
        if n == 0:   # table mode 0: the library built none
            return out
        return out.reshape(self.nlist, self.M, self.ksub)

    def get_list(self, l):
        n = self.R.ref_ivfpq_list_size(self.h, l)
        ids = np.empty(n, dtype=np.int64)
        codes = np.empty((n, self.M), dtype=np.uint8)
        if n:
            self.R.ref_ivfpq_get_list(self.h, l, _ip(ids), _up(codes))
        return ids, codes

    def search(self, x, k, nprobe):
        x = _f32(x)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        self.R.ref_ivfpq_search(self.h, x.shape[0], _fp(x), k, nprobe, _fp(D), _ip(I))
        return D, I

    def search_rerank(self, x, k, recall_num, nprobe, raw):
        """GammaIVFPQIndex::Search with has_rank on the real library (ref_driver.cpp)"""
        x = _f32(x)
        raw = _f32(raw)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        self.R.ref_ivfpq_search_rerank(self.h, x.shape[0], _fp(x), k, recall_num, nprobe, _fp(raw), _fp(D), _ip(I))
        return D, I

    def coarse(self, x, nprobe):
        x = _f32(x)
        D = np.empty((x.shape[0], nprobe), dtype=np.float32)
        I = np.empty((x.shape[0], nprobe), dtype=np.int64)
        self.R.ref_ivfpq_coarse(self.h, x.shape[0], _fp(x), nprobe, _fp(D), _ip(I))
        return D, I

    def search_preassigned(self, x, k, keys, coarse_dis):
        x = _f32(x)
        keys = np.ascontiguousarray(keys, dtype=np.int64)
        coarse_dis = _f32(coarse_dis)
        D = np.empty((x.shape[0], k), dtype=np.float32)
        I = np.empty((x.shape[0], k), dtype=np.int64)
        self.R.ref_ivfpq_search_preassigned(self.h, x.shape[0], _fp(x), k, keys.shape[1], _ip(keys),
                                            _fp(coarse_dis), _fp(D), _ip(I))
        return D, I

    def encode(self, x):
        x = _f32(x)
        lno = np.empty(x.shape[0], dtype=np.int64)
        codes = np.empty((x.shape[0], self.M), dtype=np.uint8)
        self.R.ref_ivfpq_encode(self.h, x.shape[0], _fp(x), _ip(lno), _up(codes))
        return lno, codes

    def inner_prod_table(self, x):
        x = _f32(x)
        out = np.empty((self.M, self.ksub), dtype=np.float32)
        self.R.ref_ivfpq_inner_prod_table(self.h, _fp(x), _fp(out))
        return out
