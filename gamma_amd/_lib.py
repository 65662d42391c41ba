"""ctypes loader for libgamma_hip.so (the C ABI in include/gamma_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, an exception is
raised.  The oracle under oracle/ is never imported from here.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgamma_hip.so")

f32p = C.POINTER(C.c_float)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)

NUM_STAGES = 6
STAGE_NAMES = ["coarse", "tables", "scan", "select", "rerank", "flat"]


class RangeFilter(C.Structure):
    _fields_ = [("bitmap", u8p), ("bitmap_bytes", C.c_int64), ("min_doc", C.c_int32),
                ("max_doc", C.c_int32), ("min_aligned", C.c_int32), ("b_not_in", C.c_int32)]


class FieldFilter(C.Structure):
    _fields_ = [("field_id", C.c_int32), ("include_lower", C.c_int32), ("include_upper", C.c_int32),
                ("reserved", C.c_int32), ("lower_i", C.c_int64), ("upper_i", C.c_int64),
                ("lower_f", C.c_double), ("upper_f", C.c_double)]


class TermFilter(C.Structure):
    _fields_ = [("field_id", C.c_int32), ("op", C.c_int32), ("n_items", C.c_int32), ("items", C.c_int32 * 16)]


class SearchParams(C.Structure):
    _fields_ = [("metric", C.c_int32), ("nprobe", C.c_int32), ("recall_num", C.c_int32),
                ("has_rank", C.c_int32), ("min_score", C.c_float), ("max_score", C.c_float),
                ("coarse_mode", C.c_int32), ("has_range", C.c_int32), ("n_range", C.c_int32),
                ("range", C.POINTER(RangeFilter)), ("n_field", C.c_int32), ("n_term", C.c_int32),
                ("field", C.POINTER(FieldFilter)), ("term", C.POINTER(TermFilter)), ("exact_ties", C.c_int32)]


# name -> (restype, argtypes); every symbol include/gamma_hip.h declares
SYMBOLS = {
    "gamma_hip_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "gamma_hip_destroy": (C.c_int, [C.c_void_p]),
    "gamma_hip_strerror": (C.c_char_p, [C.c_int]),
    "gamma_hip_last_error": (C.c_char_p, [C.c_void_p]),
    "gamma_hip_stream": (C.c_void_p, [C.c_void_p]),
    "gamma_hip_synchronize": (C.c_int, [C.c_void_p]),
    "gamma_hip_set_workspace_budget": (C.c_int, [C.c_void_p, C.c_int64]),
    "gamma_hip_debug_heap_stream": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, f32p, f32p, i32p, f32p, i32p]),
    "gamma_hip_set_scan_bound_feedback": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_scan_bound_stats": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_set_exact_ties": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_field_append": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p]),
    "gamma_hip_field_update": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "gamma_hip_field_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "gamma_hip_term_append": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "gamma_hip_term_count": (C.c_int64, [C.c_void_p, C.c_int]),
    "gamma_hip_term_update": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]),
    "gamma_hip_raw_init": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_raw_append": (C.c_int, [C.c_void_p, C.c_int64, f32p]),
    "gamma_hip_raw_update": (C.c_int, [C.c_void_p, C.c_int64, f32p]),
    "gamma_hip_ivfpq_arena_stats": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_ivfpq_arena_growth": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_ivfpq_set_repack_threshold": (C.c_int, [C.c_void_p, C.c_int64]),
    "gamma_hip_set_small_path": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_set_coarse_fused": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "gamma_hip_tie_stats": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "gamma_hip_ties_not_honoured": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "gamma_hip_blas_form_not_restated": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "gamma_hip_raw_write": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, f32p]),
    "gamma_hip_raw_gets": (C.c_int, [C.c_void_p, C.c_int64, i64p, f32p]),
    "gamma_hip_ivfpq_repack_verify_stats": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_raw_count": (C.c_int64, [C.c_void_p]),
    "gamma_hip_raw_stats": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_bitmap_upload": (C.c_int, [C.c_void_p, u8p, C.c_int64]),
    "gamma_hip_bitmap_set": (C.c_int, [C.c_void_p, i64p, C.c_int64, C.c_int]),
    "gamma_hip_ivfpq_init": (C.c_int, [C.c_void_p] + [C.c_int] * 7),
    "gamma_hip_ivfpq_set_trained": (C.c_int, [C.c_void_p, f32p, f32p, f32p]),
    "gamma_hip_ivfpq_get_precomputed_table": (C.c_int, [C.c_void_p, f32p]),
    "gamma_hip_set_precomputed_table_max_bytes": (C.c_int, [C.c_int64]),
    "gamma_hip_get_precomputed_table_max_bytes": (C.c_int64, []),
    "gamma_hip_ivfpq_use_precomputed_table": (C.c_int, [C.c_void_p]),
    "gamma_hip_ivfpq_add_keys": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i64p, u8p]),
    "gamma_hip_ivfpq_add_keys_batch": (C.c_int, [C.c_void_p, C.c_int, i32p, i32p, i64p, u8p]),
    "gamma_hip_ivfpq_update": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, u8p]),
    "gamma_hip_vid2docid_append": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "gamma_hip_vid2docid_count": (C.c_int64, [C.c_void_p]),
    "gamma_hip_ivfflat_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "gamma_hip_ivfflat_set_trained": (C.c_int, [C.c_void_p, f32p]),
    "gamma_hip_ivfflat_search": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, f32p, C.c_int, f32p, i64p]),
    "gamma_hip_ivfflat_search_device": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_void_p, C.c_int,
                                                  C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_has_vid": (C.c_int, [C.c_void_p, i64p, C.c_int, u8p]),
    "gamma_hip_ivfpq_remove": (C.c_int, [C.c_void_p, C.c_int64]),
    "gamma_hip_ivfpq_delete": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "gamma_hip_ivfpq_compact_if_need": (C.c_int, [C.c_void_p]),
    "gamma_hip_ivfpq_list_size": (C.c_int64, [C.c_void_p, C.c_int]),
    "gamma_hip_ivfpq_list_capacity": (C.c_int64, [C.c_void_p, C.c_int]),
    "gamma_hip_ivfpq_get_list": (C.c_int, [C.c_void_p, C.c_int, i64p, u8p]),
    "gamma_hip_ivfpq_set_list_mask": (C.c_int, [C.c_void_p, u8p]),
    "gamma_hip_ivfpq_add": (C.c_int, [C.c_void_p, C.c_int64, f32p, C.c_int64]),
    "gamma_hip_ivfpq_encode": (C.c_int, [C.c_void_p, C.c_int64, f32p, i64p, u8p]),
    "gamma_hip_assign": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, f32p, C.c_int, f32p, i32p, f32p]),
    "gamma_hip_kmeans": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, f32p, C.c_int, C.c_int, C.c_int64, C.c_int, f32p,
                                   C.POINTER(C.c_float)]),
    "gamma_hip_rand_perm": (None, [i32p, C.c_int64, C.c_int64]),
    "gamma_hip_ivfpq_train": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, f32p, C.c_int, C.c_int, f32p, f32p]),
    "gamma_hip_ivfpq_search": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, f32p, C.c_int,
                                         f32p, i64p]),
    "gamma_hip_ivfpq_search_device": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                                C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_search_device_wait": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                                     C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_flat_search_device_wait": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                                    C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_last_stages": (C.c_int, [C.c_void_p, f32p, i64p, f32p, i64p]),
    "gamma_hip_ivfpq_search_shard": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_coarse_device": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_search_shard_preassigned": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                           C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_search_shard_bounded": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gamma_hip_raw_put": (C.c_int, [C.c_void_p, C.c_int64, i64p, f32p]),
    "gamma_hip_ivfpq_shard_exact": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "gamma_hip_ivfpq_merge_rerank_exact": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_shard_export_exact": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_merge_replay_exact": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                     C.c_void_p]),
    "gamma_hip_bound_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "gamma_hip_ivfpq_merge_rerank": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_int,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_flat_search": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, f32p, C.c_int,
                                        f32p, i64p]),
    "gamma_hip_flat_search_device": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int,
                                               C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "gamma_hip_raw_update_batch": (C.c_int, [C.c_void_p, C.c_int64, i64p, f32p]),
    "gamma_hip_ivfpq_dim": (C.c_int, [C.c_void_p]),
    "gamma_hip_ivfpq_nlist": (C.c_int, [C.c_void_p]),
    "gamma_hip_ivfpq_code_size": (C.c_int, [C.c_void_p]),
    "gamma_hip_ivfpq_update_batch": (C.c_int, [C.c_void_p, C.c_int, i64p, f32p]),
    "gamma_hip_ivfpq_encode_each": (C.c_int, [C.c_void_p, C.c_int64, f32p, i64p, u8p]),
    "gamma_hip_ivfpq_apply_updates": (C.c_int, [C.c_void_p, C.c_int, i32p, i64p, u8p, u8p]),
    "gamma_hip_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "gamma_hip_group_destroy": (C.c_int, [C.c_void_p]),
    "gamma_hip_group_size": (C.c_int, [C.c_void_p]),
    "gamma_hip_group_member": (C.c_void_p, [C.c_void_p, C.c_int]),
    "gamma_hip_group_last_error": (C.c_char_p, [C.c_void_p]),
    "gamma_hip_group_set_owners": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_group_owner": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_ivfpq_shard_cut_flags": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "gamma_hip_ivfpq_merge_set_shard_flags": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_merge_flagged": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "gamma_hip_gather_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "gamma_hip_ivfpq_max_list_len": (C.c_int, [C.c_void_p]),
    "gamma_hip_ivfpq_shard_export_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, i64p]),
    "gamma_hip_ivfpq_shard_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                               C.c_void_p, C.c_void_p, C.c_void_p]),
    "gamma_hip_ivfpq_merge_replay": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gamma_hip_set_deferred_replay": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_join": (C.c_int, [C.c_void_p]),
    "gamma_hip_group_set_placement": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_group_placement": (C.c_int, [C.c_void_p]),
    "gamma_hip_group_ivfpq_add": (C.c_int, [C.c_void_p, C.c_int64, f32p, C.c_int64]),
    "gamma_hip_group_ivfpq_add_keys": (C.c_int, [C.c_void_p, C.c_int, C.c_int, i64p, u8p]),
    "gamma_hip_group_ivfpq_list_size": (C.c_int64, [C.c_void_p, C.c_int]),
    "gamma_hip_group_set_transport": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_group_transport": (C.c_int, [C.c_void_p, i64p]),
    "gamma_hip_group_transport_note": (C.c_char_p, [C.c_void_p]),
    "gamma_hip_group_ivfpq_get_list": (C.c_int, [C.c_void_p, C.c_int, i64p, u8p]),
    "gamma_hip_group_ivfpq_update": (C.c_int, [C.c_void_p, C.c_int, i64p, f32p]),
    "gamma_hip_group_ivfpq_delete": (C.c_int, [C.c_void_p, i64p, C.c_int]),
    "gamma_hip_group_ivfpq_compact_if_need": (C.c_int, [C.c_void_p]),
    "gamma_hip_group_ivfpq_search": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, f32p, C.c_int, f32p, i64p]),
    "gamma_hip_group_ivfpq_search_device": (C.c_int, [C.c_void_p, C.POINTER(SearchParams), C.c_int, C.c_void_p, C.c_int,
                                                      C.c_void_p, C.c_void_p]),
    "gamma_hip_group_total_mem_bytes": (C.c_int64, [C.c_void_p]),
    "gamma_hip_total_mem_bytes": (C.c_int64, [C.c_void_p]),
    "gamma_hip_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "gamma_hip_profile_reset": (C.c_int, [C.c_void_p]),
    "gamma_hip_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double),
                                        C.POINTER(C.c_int64)]),
    "gamma_hip_profile_scan_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64),
                                               C.POINTER(C.c_int64)]),
}

_lib = None


# gamma_hip_bound_reduce_fn (include/gamma_hip.h): (user, d_bound, nq, take_max, stream) -> 0 on success
BOUND_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p)


class GammaHipError(RuntimeError):
    pass


def load():
    """Load libgamma_hip.so and bind every symbol of the C ABI.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GammaHipError(
                "libgamma_hip.so not found at %s -- build it with `python -c 'import "
                "__graft_entry__ as g; g.build()'` or `make -C gamma_amd/csrc` (no CPU fallback)"
                % LIB_PATH)
        # ONE HIP runtime per process: PyTorch ships its own libamdhip64 and the tests / bench / dist.py hand torch tensors'
        # device pointers to this library.  Whichever runtime is loaded first serves both (same SONAME), and a torch
        # imported AFTER this library has initialised the ROCm install's runtime finds no device -- so torch goes first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
