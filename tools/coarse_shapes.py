"""Coarse quantizer over shapes: time per 8192-query call with and without the distance matrix (csrc/coarse.hip),
and -- GAMMA_HIP_COARSE_DBG=1 -- how many queries the fused path had to repair."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gamma_amd import api, synth
dev = torch.device("cuda", 0)
nq = 8192
SH = os.environ.get('CS_SHAPES')
for d, nlist, P in [tuple(int(v) for v in t.split(',')) for t in SH.split(';')] if SH else ((128, 4096, 32), (128, 8192, 32), (128, 16384, 32), (128, 16384, 16), (128, 16384, 64), (128, 32768, 32), (64, 16384, 32), (768, 16384, 64), (256, 4096, 32)):
    base = synth.sift_like(max(200000, nlist * 20), d=d, seed=1234)
    cc, pq = api.train_ivfpq(base[:nlist * 20], nlist, 16 if d % 16 == 0 else 8)
    g = api.GammaHip(0)
    g.ivfpq_init(d, nlist, 16, 8, api.METRIC_L2, 100)
    g.ivfpq_set_trained(cc, pq, None)
    q = torch.from_numpy(synth.sift_like(nq, d=d, seed=4321)).to(dev)
    cd = torch.empty((nq, P), dtype=torch.float32, device=dev)
    ci = torch.empty((nq, P), dtype=torch.int32, device=dev)
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, coarse_mode=1, min_score=-1e30, max_score=1e30)
    out = []
    for fused in (True, False):
        g.set_coarse_fused(fused, 128)
        for i in range(3):
            g.ivfpq_coarse_device(q.data_ptr(), nq, args, cd.data_ptr(), ci.data_ptr())
        g.synchronize()
        t0 = time.perf_counter()
        for i in range(10):
            g.ivfpq_coarse_device(q.data_ptr(), nq, args, cd.data_ptr(), ci.data_ptr())
        g.synchronize()
        out.append((time.perf_counter() - t0) / 10 * 1e6)
    print("d %d nlist %d nprobe %d: fused %.0f us, matrix %.0f us per %d queries" % (d, nlist, P, out[0], out[1], nq), flush=True)
    g.close()
