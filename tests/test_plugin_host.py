"""CPU tests of the RetrievalModel plugin layer (gamma_amd/host): the library loads, both models
are registered with the reflector under their names, and the JSON parameter rules follow
GammaIVFPQIndex (reference index/impl/gamma_index_ivfpq.h:708-851, :629-673).  No device calls."""
import ctypes as C
import os

import pytest

from gamma_amd import plugin


def test_host_library_exports_harness():
    L = plugin.load_host()
    for name in plugin.HOST_SYMBOLS:
        assert hasattr(L, name)


def test_models_registered():
    L = plugin.load_host()
    assert L.gh_model_registered(b"HIPIVFPQ") == 1
    assert L.gh_model_registered(b"HIPFLAT") == 1
    assert L.gh_model_registered(b"IVFPQ") == 0       # the CPU model lives in the reference tree


def test_model_params_defaults_and_values():
    p = plugin.parse_model_params('{"ncentroids": 256, "nsubvector": 16}')
    assert p["rc"] == 0 and p["ncentroids"] == 256 and p["nsubvector"] == 16
    assert p["nbits_per_idx"] == 8 and p["nprobe"] == 80 and p["metric"] == 0       # InnerProduct
    assert p["bucket_init_size"] == 1000 and p["bucket_max_size"] == 1280000
    p = plugin.parse_model_params('{"ncentroids": 4096, "nsubvector": 16, "nprobe": 32, "metric_type": "L2",'
                                  ' "bucket_init_size": 500, "bucket_max_size": 9000}')
    assert p["rc"] == 0 and p["nprobe"] == 32 and p["metric"] == 1
    assert p["bucket_init_size"] == 500 and p["bucket_max_size"] == 9000
    # -1 keeps the default
    p = plugin.parse_model_params('{"ncentroids": -1, "nsubvector": -1, "nprobe": -1}')
    assert p["rc"] == 0 and p["ncentroids"] == 2048 and p["nsubvector"] == 64 and p["nprobe"] == 80


@pytest.mark.parametrize("s", [
    "not json",
    '{"nsubvector": 16}',                                    # ncentroids is mandatory
    '{"ncentroids": 256}',                                   # nsubvector is mandatory
    '{"ncentroids": -5, "nsubvector": 16}',
    '{"ncentroids": 16, "nsubvector": 8, "nprobe": 32}',     # nprobe > ncentroids
    '{"ncentroids": 256, "nsubvector": 16, "metric_type": "Hamming"}',
])
def test_model_params_rejected(s):
    assert plugin.parse_model_params(s)["rc"] != 0


def test_model_params_flags_for_unsupported_quantizers():
    p = plugin.parse_model_params('{"ncentroids": 256, "nsubvector": 16, "hnsw": {"nlinks": 32}}')
    assert p["rc"] == 0 and p["has_hnsw"] == 1 and p["has_opq"] == 0
    p = plugin.parse_model_params('{"ncentroids": 256, "nsubvector": 16, "opq": {"nsubvector": 16}}')
    assert p["rc"] == 0 and p["has_opq"] == 1


def test_retrieval_params():
    r = plugin.parse_retrieval_params("")
    assert r["rc"] == 0 and r["recall_num"] == 100 and r["nprobe"] == -1
    r = plugin.parse_retrieval_params('{"metric_type": "L2", "recall_num": 200, "nprobe": 32}')
    assert r == {"rc": 0, "metric": 1, "recall_num": 200, "nprobe": 32}
    r = plugin.parse_retrieval_params('{"metric_type": "InnerProduct", "recall_num": -3}')
    assert r["metric"] == 0 and r["recall_num"] == 100
    assert plugin.parse_retrieval_params("{broken")["rc"] != 0


def test_unknown_model_name():
    L = plugin.load_host()
    assert not L.gh_host_new(b"NOPE", 8)


# ---- ivfpq.index ("IwPQ") file format, pinned on bytes written by real faiss (tests/gen_golden.py) ----
def _golden_iwpq():
    import numpy as np
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "iwpq_small.npz"))


def test_iwpq_writer_matches_faiss_bytes(tmp_path):
    z = _golden_iwpq()
    path = str(tmp_path / "ivfpq.index")
    rc = plugin.iwpq_write(path, int(z["d"]), int(z["N"]), 1, int(z["nprobe"]), z["cc"], z["pq"],
                           z["list_sizes"], z["list_codes"], z["list_ids"])
    assert rc == 0
    got = open(path, "rb").read()
    assert got == z["file_bytes"].tobytes()


def test_iwpq_reader_on_faiss_file(tmp_path):
    import numpy as np
    z = _golden_iwpq()
    path = str(tmp_path / "ref.index")
    open(path, "wb").write(z["file_bytes"].tobytes())
    f = plugin.iwpq_read(path)
    assert (f["d"], f["nlist"], f["M"], f["nbits"], f["code_size"], f["metric"], f["by_residual"]) == (
        int(z["d"]), int(z["nlist"]), int(z["M"]), 8, int(z["M"]), 1, 1)
    assert f["ntotal"] == int(z["N"]) and f["nprobe"] == int(z["nprobe"])
    assert f["cc"].tobytes() == z["cc"].tobytes() and f["pq"].tobytes() == z["pq"].tobytes()
    assert np.array_equal(f["list_sizes"], z["list_sizes"])
    assert np.array_equal(f["list_ids"], z["list_ids"]) and np.array_equal(f["list_codes"], z["list_codes"])


def test_iwpq_superseded_ids_and_bad_files(tmp_path):
    import numpy as np
    z = _golden_iwpq()
    ids = z["list_ids"].copy()
    ids[3] |= np.int64(-2 ** 63)                 # bit 63: slot superseded by an Update
    path = str(tmp_path / "moved.index")
    assert plugin.iwpq_write(path, int(z["d"]), 0, 0, 7, z["cc"], z["pq"], z["list_sizes"], z["list_codes"],
                             ids) == 0
    f = plugin.iwpq_read(path)
    assert f["ntotal"] == 0 and f["metric"] == 0 and f["nprobe"] == 7
    assert np.array_equal(f["list_ids"], ids)
    bad = str(tmp_path / "bad.index")
    open(bad, "wb").write(b"IxF2" + bytes(100))
    with pytest.raises(Exception):
        plugin.iwpq_read(bad)
    open(bad, "wb").write(z["file_bytes"].tobytes()[:5000])      # truncated
    with pytest.raises(Exception):
        plugin.iwpq_read(bad)
    with pytest.raises(Exception):
        plugin.iwpq_read(str(tmp_path / "missing.index"))
