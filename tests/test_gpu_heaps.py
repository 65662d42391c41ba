"""GPU: the device's heaps, entry for entry, against the oracle's (pinned on the compiled library's heaps, faiss:utils/Heap.h:
tests/golden/heap.npz, tests/test_oracle_vs_ref.py).  gamma_hip_debug_heap_stream feeds ONE stream of keys to ONE heap through
each form of the sifts the replays use: the pipelined heap_replace_top walk (HeapWalk), ParHeap (pop / push / replace_top with
all 64 lanes: children read at once, the walk's path from two ballots and three doubling rounds), the sequential forms.
Streams with many equal keys: the ARRAY a heap ends with depends on every tie decision on the way."""
import numpy as np
import pytest

from gamma_amd import api
from oracle import binding as B

pytestmark = pytest.mark.gpu


def _oracle_replace_top(vals, k):
    n = len(vals)
    ids = np.arange(n, dtype=np.int64)
    hv, hi = np.empty(k, np.float32), np.empty(k, np.int64)
    sv, si = np.empty(k, np.float32), np.empty(k, np.int64)
    B.lib().go_heap_stream(1, k, n, B._fp(vals), B._ip(ids), B._fp(hv), B._ip(hi), B._fp(sv), B._ip(si))
    return hv, hi, sv, si


def _oracle_pop_push(vals, k):
    n = len(vals)
    ids = np.arange(n, dtype=np.int64)
    sv, si = np.empty(k, np.float32), np.empty(k, np.int64)
    B.lib().go_heap_pop_push_stream(1, k, n, B._fp(vals), B._ip(ids), B._fp(sv), B._ip(si))
    return sv, si


@pytest.mark.parametrize("k", [1, 2, 3, 10, 15, 16, 31, 63, 64, 100, 127, 128, 129, 200, 255, 256, 257, 500, 1024])
def test_every_form_of_the_sifts_leaves_the_reference_heaps(k):
    g = api.GammaHip(0)
    try:
        for seed, (n, hi) in enumerate([(0, 5), (k // 2 + 1, 3), (3 * k + 7, 4), (5000, 25), (5000, 100000), (4096, 2)]):
            rng = np.random.default_rng(100 * k + seed)
            vals = rng.integers(0, hi, size=n).astype(np.float32)
            if seed == 4:
                vals = np.sort(vals)[::-1].copy()          # descending: every key is accepted, every sift runs to a leaf
            hv, hi_, sv, si = _oracle_replace_top(vals, k)
            pv, pi = _oracle_pop_push(vals, k)
            for op in (0, 3):                               # heap_replace_top streams: the array itself, then heap_reorder
                av, ai, dv, di = g.debug_heap_stream(op, k, vals)
                assert av.tobytes() == hv.tobytes() and np.array_equal(ai.astype(np.int64), hi_), (op, k, n)
                assert dv.tobytes() == sv.tobytes() and np.array_equal(di.astype(np.int64), si), (op, k, n)
            ref_arr = None
            for op in (2, 1):                               # heap_pop + heap_push streams: sequential, then all lanes
                av, ai, dv, di = g.debug_heap_stream(op, k, vals)
                assert dv.tobytes() == pv.tobytes() and np.array_equal(di.astype(np.int64), pi), (op, k, n)
                if ref_arr is None:
                    ref_arr = (av.copy(), ai.copy())
                else:                                       # and the same array on the way
                    assert av.tobytes() == ref_arr[0].tobytes() and np.array_equal(ai, ref_arr[1]), (op, k, n)
    finally:
        g.close()


def test_reservoir_streams_of_the_compiled_library_on_the_device():
    """tests/golden/reservoir_ties.npz: streams of tied keys through the COMPILED library's ReservoirTopN (k = 100 .. 256, what
    knn_L2sqr collects the coarse assignment through from 100 probes on).  The device's replay (reservoir_dev.h: append,
    partition_fuzzy_median3 shrinks, to_result) returns the library's labels at every rank -- keep-smallest streams; and the
    oracle's restatement agrees with the device on longer streams and larger k (up to the 1024 probes the mode covers)."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reservoir_ties.npz"))
    g = api.GammaHip(0)
    try:
        seen = 0
        for ci, (ks, k, n, hi) in enumerate(z["cases"]):
            if not ks:
                continue                                  # (the device replays the L2 coarse quantizer: keep-smallest)
            keys = np.ascontiguousarray(z["keys_%d" % ci])
            _, _, sv, si = g.debug_heap_stream(4, int(k), keys)
            assert sv.tobytes() == z["D_%d" % ci].tobytes() and np.array_equal(si.astype(np.int64), z["I_%d" % ci]), (k, n, hi)
            seen += 1
        assert seen >= 10
        for seed, (k, n, hi) in enumerate([(300, 5000, 7), (512, 20000, 3), (1000, 3000, 2), (1024, 16384, 50), (100, 100, 2),
                                           (640, 641, 4)]):
            rng = np.random.default_rng(seed)
            keys = rng.integers(0, hi, size=n).astype(np.float32)
            ov, oi = np.empty(k, np.float32), np.empty(k, np.int64)
            B.lib().go_reservoir_stream(1, k, n, B._fp(keys), None, B._fp(ov), B._ip(oi))
            _, _, sv, si = g.debug_heap_stream(4, k, keys)
            assert sv.tobytes() == ov.tobytes() and np.array_equal(si.astype(np.int64), oi), (k, n, hi)
    finally:
        g.close()
