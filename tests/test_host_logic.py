"""CPU: host-side logic that does not need the device."""
import numpy as np

from gamma_amd import dist as gdist
from gamma_amd import synth
from tests.parity import compare_topk


def test_synth_is_chunk_invariant_and_sift_shaped():
    a = synth.sift_like(70000, d=16, seed=5)
    b = synth.sift_like(3000, d=16, seed=5, start=64000)
    assert np.array_equal(a[64000:67000], b)
    assert a.min() >= 0 and a.max() <= 255 and np.array_equal(a, np.rint(a))


def test_balance_lists_is_a_balanced_partition():
    rng = np.random.default_rng(0)
    sizes = (rng.pareto(1.5, size=4096) * 100).astype(np.int64)
    for w in (1, 2, 4, 8):
        owner = gdist.balance_lists(sizes, w)
        assert owner.min() >= 0 and owner.max() < w
        load = np.bincount(owner, weights=sizes, minlength=w)
        assert load.max() <= load.mean() * 1.02 + sizes.max()


def test_query_slices_cover_exactly():
    for nq in (0, 1, 7, 1024, 1025):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                q0, q1, per = gdist.query_slice(nq, r, w)
                assert q1 - q0 <= per
                got.extend(range(q0, q1))
            assert got == list(range(nq))


def test_parity_comparator():
    D = np.array([[1, 2, 2, 3]], dtype=np.float32)
    I = np.array([[5, 6, 7, 8]])
    compare_topk(D, I, D, np.array([[5, 7, 6, 8]]))          # tie swap ok
    r = compare_topk(D, I, np.array([[1, 2, 2, 3]], dtype=np.float32), np.array([[5, 6, 7, 9]]))
    assert r["n_boundary"] == 1                                # last group cut by k
    try:
        compare_topk(D, I, D, np.array([[6, 5, 7, 8]]))
        raise SystemExit("should have failed")
    except AssertionError:
        pass
