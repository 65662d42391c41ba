"""The GEMM-form coarse path is pinned to the K-blocking of THIS build's MKL sgemm_: one k-ascending fma chain
per element up to K = 384.  MKL dispatches by CPU type, so a GPU box whose host CPU takes another kernel would compute other
low bits -- asserted here, on the box the GPU suite runs on, instead of being read off a bench line: the
device's coarse distances and probe order for a C3-shaped batch against the compiled library's IndexFlatL2::search at its
default BLAS threshold (faiss:utils/distances.cpp:215-296,303-305), and the end-to-end labels on the library's own index.
Skipped where oracle/_ref is absent (it is built from /root/reference and travels with the snapshot)."""
import ctypes

import numpy as np
import pytest

from gamma_amd import api, synth
from oracle import binding as B

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not B.have_ref(), reason="oracle/_ref not built (needs /root/reference)")]


def _mkl_description():
    """what to look at when the assertion below fires: the MKL build and the CPU it dispatches for"""
    out = []
    for name in ("libmkl_rt.so", "libmkl_rt.so.2", "libmkl_rt.so.1"):
        try:
            m = ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
            buf = ctypes.create_string_buffer(256)
            m.mkl_get_version_string(buf, 256)
            out.append(buf.value.decode(errors="replace").strip())
            try:
                m.mkl_get_cpu_clocks  # noqa: B018  (symbol probe only)
                cbwr = m.mkl_cbwr_get_auto_branch()
                out.append("mkl_cbwr_get_auto_branch = %d" % cbwr)
            except Exception:
                pass
            break
        except Exception:
            continue
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        flags = [l for l in open("/proc/cpuinfo") if l.startswith("flags")][0]
        out.append("cpu: %s; avx512f %s, avx2 %s" % (model, "avx512f" in flags, "avx2" in flags))
    except Exception:
        pass
    return " | ".join(out) or "MKL description unavailable"


def test_device_gemm_form_is_this_boxs_sgemm_at_the_c3_shape():
    R = B.ref()
    R.ref_set_blas_threshold(20)                     # the library's default: 4096 queries take the sgemm_ path
    d, nlist, M, nq, P = 128, 4096, 16, 4096, 32
    base = synth.sift_like(nlist * 40, d=d, seed=1234)
    # (the arithmetic under test is the distance matrix: any centroid table of the C3 shape will do -- every 10th training
    #  vector, integer-valued like SIFT, so exact ties among the nearest centroids occur and their ORDER is tested too)
    cc = np.ascontiguousarray(base[::10][:nlist])
    q = synth.sift_like(nq, d=d, seed=4321)
    Dr = np.empty((nq, P), np.float32)
    Ir = np.empty((nq, P), np.int64)
    R.ref_flat_l2_search(d, nlist, B._fp(cc), nq, B._fp(q), P, B._fp(Dr), B._ip(Ir))
    g = api.GammaHip(0)
    try:
        import torch
        g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2)
        g.ivfpq_set_trained(cc, np.zeros((M, 256, d // M), np.float32), None)
        dev = torch.device("cuda", 0)
        x = torch.from_numpy(q).to(dev)
        cd = torch.empty((nq, P), dtype=torch.float32, device=dev)
        pr = torch.empty((nq, P), dtype=torch.int32, device=dev)
        args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=100, coarse_mode=-1, min_score=0.0, max_score=1e30)
        g.ivfpq_coarse_device(x.data_ptr(), nq, args, cd.data_ptr(), pr.data_ptr())
        g.synchronize()
        Dg, Ig = cd.cpu().numpy(), pr.cpu().numpy().astype(np.int64)
    finally:
        g.close()
    bits = float((Dr.view(np.uint32) == Dg.view(np.uint32)).all(axis=1).mean())
    order = float((Ir == Ig).all(axis=1).mean())
    what = _mkl_description()
    print("coarse distance rows identical to the compiled library's: %.6f, probe order identical: %.6f  [%s]" % (bits, order, what))
    assert bits == 1.0 and order == 1.0, (
        "the device's GEMM-form coarse distances differ from this box's sgemm_: the K-blocking pinned in the build container "
        "(DESIGN.md 4) is not what this host's MKL dispatches -- rows identical %.6f, probe order %.6f [%s]" % (bits, order, what))
