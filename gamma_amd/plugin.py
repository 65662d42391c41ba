"""ctypes driver for libgamma_host.so: the RetrievalModel plugins (HIPIVFPQ / HIPFLAT) driven
the way Gamma's VectorManager drives a model (gamma_amd/host/harness_c_api.cc).

Test/bench convenience only; the product boundary is the C++ RetrievalModel interface in
gamma_amd/host/plugin_api.h on top of the C ABI in include/gamma_hip.h.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

HOST_LIB_PATH = os.path.join(_lib.HERE, "libgamma_host.so")
f32p, i64p = _lib.f32p, _lib.i64p

HOST_SYMBOLS = {
    "gh_host_new": (C.c_void_p, [C.c_char_p, C.c_int]),
    "gh_host_free": (None, [C.c_void_p]),
    "gh_host_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gh_host_store": (None, [C.c_void_p, C.c_int, f32p]),
    "gh_host_indexing": (C.c_int, [C.c_void_p]),
    "gh_host_add": (C.c_int, [C.c_void_p, C.c_int, f32p]),
    "gh_host_update": (C.c_int, [C.c_void_p, C.c_int64, f32p]),
    "gh_host_update_batch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), f32p]),
    "gh_host_last_perf": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gh_host_table_set": (None, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "gh_host_table_oob_reads": (C.c_long, [C.c_void_p]),
    "gh_host_delete": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "gh_host_engine_bitmap_set": (None, [C.c_void_p, i64p, C.c_int]),
    "gh_host_search_during_add": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p,
                                            C.c_int, C.c_int]),
    "gh_host_search": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                 C.c_int, f32p, C.c_int, f32p, i64p]),
    "gh_host_search_filtered": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                          C.c_int, f32p, C.c_int, f32p, i64p, C.c_int, i64p,
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gh_host_concurrent_clients": (C.c_double, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, f32p,
                                                C.c_int, C.c_int, C.c_int, f32p]),
    "gh_host_concurrent_filtered_check": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, f32p, C.c_int,
                                                    C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "gh_host_set_vid2docid": (None, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "gh_host_table_add_field": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "gh_host_table_append": (None, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "gh_host_search_scalar": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p, i64p,
                                        C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "gh_host_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gh_host_load": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gh_host_mem_bytes": (C.c_long, [C.c_void_p]),
    "gh_host_ivfpq_state": (C.c_int, [C.c_void_p, f32p, f32p]),
    "gh_host_ivfpq_set_trained": (C.c_int, [C.c_void_p, f32p, f32p]),
    "gh_parse_ivfpq_model_params": (None, [C.c_char_p, C.POINTER(C.c_int)]),
    "gh_parse_ivfpq_retrieval_params": (None, [C.c_char_p, C.POINTER(C.c_int)]),
    "gh_model_registered": (C.c_int, [C.c_char_p]),
    "gh_iwpq_write": (C.c_int, [C.c_char_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p,
                                i64p, _lib.u8p, i64p]),
    "gh_iwpq_read": (C.c_int, [C.c_char_p, i64p, f32p, f32p, i64p, _lib.u8p, i64p]),
}

_host = None
FLT_MIN = float(np.finfo(np.float32).tiny)
FLT_MAX = float(np.finfo(np.float32).max)


def load_host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise _lib.GammaHipError("libgamma_host.so not found at %s -- run `make -C gamma_amd/host`"
                                     % HOST_LIB_PATH)
        L = C.CDLL(HOST_LIB_PATH)
        for name, (res, args) in HOST_SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _host = L
    return _host


def _f(a):
    return a.ctypes.data_as(f32p)


def parse_model_params(s):
    out = (C.c_int * 10)()
    load_host().gh_parse_ivfpq_model_params(s.encode(), out)
    keys = ["rc", "ncentroids", "nsubvector", "nbits_per_idx", "nprobe", "metric", "bucket_init_size",
            "bucket_max_size", "has_hnsw", "has_opq"]
    return dict(zip(keys, list(out)))


def parse_retrieval_params(s):
    out = (C.c_int * 4)()
    load_host().gh_parse_ivfpq_retrieval_params(s.encode(), out)
    return dict(zip(["rc", "metric", "recall_num", "nprobe"], list(out)))


def iwpq_write(path, d, ntotal, metric, nprobe, cc, pq, list_sizes, list_codes, list_ids):
    """Write the reference's ivfpq.index ("IwPQ") file; metric 0 = inner product, 1 = L2."""
    cc = np.ascontiguousarray(cc, np.float32)
    pq = np.ascontiguousarray(pq, np.float32)
    sizes = np.ascontiguousarray(list_sizes, np.int64)
    codes = np.ascontiguousarray(list_codes, np.uint8)
    ids = np.ascontiguousarray(list_ids, np.int64)
    M = pq.shape[0]
    return load_host().gh_iwpq_write(path.encode(), d, ntotal, metric, cc.shape[0], nprobe, _f(cc), M, _f(pq),
                                     sizes.ctypes.data_as(i64p), codes.ctypes.data_as(_lib.u8p),
                                     ids.ctypes.data_as(i64p))


def iwpq_read(path):
    L = load_host()
    hdr = np.zeros(10, np.int64)
    rc = L.gh_iwpq_read(path.encode(), hdr.ctypes.data_as(i64p), None, None, None, None, None)
    if rc:
        raise _lib.GammaHipError("cannot read %s (%d)" % (path, rc))
    d, ntotal, metric, nlist, nprobe, M, nbits, code_size, by_res, tot = [int(v) for v in hdr]
    cc = np.empty((nlist, d), np.float32)
    pq = np.empty((M, 1 << nbits, d // M), np.float32)
    sizes = np.empty(nlist, np.int64)
    codes = np.empty((tot, code_size), np.uint8)
    ids = np.empty(tot, np.int64)
    rc = L.gh_iwpq_read(path.encode(), hdr.ctypes.data_as(i64p), _f(cc), _f(pq), sizes.ctypes.data_as(i64p),
                        codes.ctypes.data_as(_lib.u8p), ids.ctypes.data_as(i64p))
    if rc:
        raise _lib.GammaHipError("cannot read %s (%d)" % (path, rc))
    return dict(d=d, ntotal=ntotal, metric=metric, nlist=nlist, nprobe=nprobe, M=M, nbits=nbits,
                code_size=code_size, by_residual=by_res, cc=cc, pq=pq, list_sizes=sizes, list_codes=codes,
                list_ids=ids)


class PluginModel:
    """One RetrievalModel instance plus the in-memory vector store it reads."""

    def __init__(self, retrieval_type, d, retrieval_param="", indexing_size=0):
        self.L = load_host()
        self.d = d
        self.h = self.L.gh_host_new(retrieval_type.encode(), d)
        if not self.h:
            raise _lib.GammaHipError("model %r is not registered" % retrieval_type)
        rc = self.L.gh_host_init(self.h, retrieval_param.encode(), indexing_size)
        if rc:
            self.close()
            raise _lib.GammaHipError("Init(%r) returned %d" % (retrieval_param, rc))

    def close(self):
        if self.h:
            self.L.gh_host_free(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def store(self, x):
        x = np.ascontiguousarray(x, np.float32)
        self.L.gh_host_store(self.h, x.shape[0], _f(x))

    def set_vid2docid(self, first, docids):
        """documents with several vectors: the engine's VIDMgr mapping for vids [first, first + len)"""
        m = np.ascontiguousarray(docids, dtype=np.int32)
        self.L.gh_host_set_vid2docid(self.h, first, m.size, m.ctypes.data_as(C.POINTER(C.c_int)))

    def indexing(self):
        return self.L.gh_host_indexing(self.h)

    def add(self, x):
        x = np.ascontiguousarray(x, np.float32)
        return self.L.gh_host_add(self.h, x.shape[0], _f(x)) == 1

    def update(self, vid, x):
        x = np.ascontiguousarray(x, np.float32)
        return self.L.gh_host_update(self.h, vid, _f(x))

    def update_batch(self, vids, x):
        vids = np.ascontiguousarray(vids, np.int64)
        x = np.ascontiguousarray(x, np.float32)
        return self.L.gh_host_update_batch(self.h, len(vids), vids.ctypes.data_as(C.POINTER(C.c_int64)), _f(x))

    def delete(self, vids):
        v = np.ascontiguousarray(vids, np.int64)
        return self.L.gh_host_delete(self.h, v.ctypes.data_as(i64p), v.size)

    def engine_bitmap_set(self, vids):
        """set doc bits in the ENGINE's delete bitmap only (the state BitmapManager::Load restores)"""
        v = np.ascontiguousarray(vids, np.int64)
        self.L.gh_host_engine_bitmap_set(self.h, v.ctypes.data_as(i64p), v.size)

    def search_during_add(self, x, q, k, nthreads=4, batch=500, retrieval_params=""):
        """brute-force client threads running while the vectors are stored + added; returns failed calls"""
        x = np.ascontiguousarray(x, np.float32)
        q = np.ascontiguousarray(q, np.float32)
        return self.L.gh_host_search_during_add(self.h, retrieval_params.encode(), nthreads, x.shape[0], batch, _f(x),
                                                x.shape[1], _f(q), q.shape[0], k)

    def search(self, xq, k, retrieval_params="", has_rank=True, brute_force=False, min_score=FLT_MIN,
               max_score=FLT_MAX, range_filters=None):
        """range_filters: None, or a list of (matching docids, not_in) clauses (AND-ed)."""
        xq = np.ascontiguousarray(xq, np.float32)
        n = xq.shape[0]
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        if range_filters is not None:
            docs = [np.unique(np.asarray(d, np.int64)) for d, _ in range_filters]
            flat = np.ascontiguousarray(np.concatenate(docs) if docs else np.zeros(0, np.int64))
            counts = (C.c_int * max(1, len(docs)))(*[len(d) for d in docs])
            notin = (C.c_int * max(1, len(docs)))(*[int(bool(ni)) for _, ni in range_filters])
            rc = self.L.gh_host_search_filtered(
                self.h, retrieval_params.encode(), int(has_rank), int(brute_force), min_score, max_score, n,
                _f(xq), k, _f(D), I.ctypes.data_as(i64p), len(docs), flat.ctypes.data_as(i64p), counts, notin)
            if rc:
                raise _lib.GammaHipError("Search returned %d" % rc)
            return D, I
        rc = self.L.gh_host_search(self.h, retrieval_params.encode(), int(has_rank), int(brute_force),
                                   min_score, max_score, n, _f(xq), k, _f(D), I.ctypes.data_as(i64p))
        if rc:
            raise _lib.GammaHipError("Search returned %d" % rc)
        return D, I

    # ---- scalar fields + filters as the client sends them (GammaSearchCondition::range_filters / term_filters) ----
    DT = {"int": 0, "long": 1, "float": 2, "double": 3, "string": 4}
    _NP = {0: np.int32, 1: np.int64, 2: np.float32, 3: np.float64}

    def table_add_field(self, name, dtype):
        fid = self.L.gh_host_table_add_field(self.h, name.encode(), self.DT[dtype])
        self.__dict__.setdefault("_ftypes", {})[name] = (fid, self.DT[dtype])
        return fid

    def table_append(self, name, values):
        """numeric: array of the field's type; string: list of lists of items (joined with \\001 as the engine does)"""
        fid, dt = self._ftypes[name]
        if dt == 4:
            raws = [b"\x01".join(s.encode() for s in items) for items in values]
            lens = (C.c_int * len(raws))(*[len(r) for r in raws])
            blob = b"".join(raws)
            self.L.gh_host_table_append(self.h, fid, len(raws), C.c_char_p(blob), 0, lens)
        else:
            v = np.ascontiguousarray(values, dtype=self._NP[dt])
            self.L.gh_host_table_append(self.h, fid, v.size, v.ctypes.data, v.itemsize, None)

    def table_set(self, name, docid, value):
        """Table::Update of one field of one doc: a numeric value, or a list of items for a string field"""
        fid, dt = self._ftypes[name]
        raw = b"\x01".join(s.encode() for s in value) if dt == 4 else np.array([value], dtype=self._NP[dt]).tobytes()
        self.L.gh_host_table_set(self.h, fid, docid, raw, len(raw))

    def table_oob_reads(self):
        return self.L.gh_host_table_oob_reads(self.h)

    def last_perf(self):
        buf = C.create_string_buffer(4096)
        self.L.gh_host_last_perf(self.h, buf, 4096)
        return buf.value.decode()

    def search_scalar(self, xq, k, retrieval_params="", has_rank=True, brute_force=False, ranges=(), terms=()):
        """ranges: (field, lower, upper, include_lower, include_upper); terms: (field, [items], op 0 And / 1 Or / 2 Not)"""
        class HRange(C.Structure):
            _fields_ = [("field", C.c_char_p), ("lower", C.c_void_p), ("upper", C.c_void_p), ("nbytes", C.c_int),
                        ("include_lower", C.c_int), ("include_upper", C.c_int)]

        class HTerm(C.Structure):
            _fields_ = [("field", C.c_char_p), ("value", C.c_char_p), ("value_len", C.c_int), ("is_union", C.c_int)]

        keep = []
        ra = (HRange * max(1, len(ranges)))()
        for i, (f, lo, hi, il, iu) in enumerate(ranges):
            npt = self._NP[self._ftypes[f][1]]
            lo_a, hi_a = np.array([lo], dtype=npt), np.array([hi], dtype=npt)
            keep += [lo_a, hi_a]
            ra[i] = HRange(f.encode(), lo_a.ctypes.data, hi_a.ctypes.data, lo_a.itemsize, int(il), int(iu))
        ta = (HTerm * max(1, len(terms)))()
        for i, (f, items, op) in enumerate(terms):
            v = b"\x01".join(s.encode() for s in items)
            keep.append(v)
            ta[i] = HTerm(f.encode(), v, len(v), op)
        xq = np.ascontiguousarray(xq, np.float32)
        n = xq.shape[0]
        D = np.empty((n, k), np.float32)
        I = np.empty((n, k), np.int64)
        rc = self.L.gh_host_search_scalar(self.h, retrieval_params.encode(), int(has_rank), int(brute_force), n, _f(xq), k,
                                          _f(D), I.ctypes.data_as(i64p), len(ranges), C.cast(ra, C.c_void_p),
                                          len(terms), C.cast(ta, C.c_void_p))
        if rc:
            raise _lib.GammaHipError("Search returned %d" % rc)
        return D, I

    def concurrent_clients(self, pool, retrieval_params, nthreads, calls, nq_call=1, k=10, has_rank=True):
        """closed-loop client threads (C++ threads, no GIL): returns (wall seconds, latencies in us)"""
        pool = np.ascontiguousarray(pool, dtype=np.float32)
        lat = np.empty(nthreads * calls, dtype=np.float32)
        dt = self.L.gh_host_concurrent_clients(self.h, retrieval_params.encode(), int(has_rank), nthreads, calls,
                                               nq_call, _f(pool), pool.shape[0], pool.shape[1], k, _f(lat))
        if dt < 0:
            raise _lib.GammaHipError("a Search call failed")
        return dt, lat

    def concurrent_filtered_check(self, pool, retrieval_params, nthreads, calls, stride, span, k=10, has_rank=True):
        """client threads with one query and their own range filter each; returns (calls whose result differs
        from the same call made alone, wall seconds of the concurrent phase)"""
        pool = np.ascontiguousarray(pool, dtype=np.float32)
        sec = C.c_double(0)
        bad = self.L.gh_host_concurrent_filtered_check(self.h, retrieval_params.encode(), int(has_rank), nthreads, calls,
                                                       _f(pool), pool.shape[0], pool.shape[1], k, stride, span,
                                                       C.byref(sec))
        if bad < 0:
            raise _lib.GammaHipError("a Search call failed")
        return bad, sec.value

    def dump(self, d):
        return self.L.gh_host_dump(self.h, d.encode())

    def load(self, d):
        return self.L.gh_host_load(self.h, d.encode())

    def mem_bytes(self):
        return self.L.gh_host_mem_bytes(self.h)

    def set_trained(self, coarse, pq):
        coarse = np.ascontiguousarray(coarse, np.float32)
        pq = np.ascontiguousarray(pq, np.float32)
        return self.L.gh_host_ivfpq_set_trained(self.h, _f(coarse), _f(pq))

    def trained_state(self, nlist, M):
        cc = np.empty((nlist, self.d), np.float32)
        pq = np.empty((M, 256, self.d // M), np.float32)
        if self.L.gh_host_ivfpq_state(self.h, _f(cc), _f(pq)):
            return None
        return cc, pq
