#!/usr/bin/env python3
"""IVFFLAT on one GPU at the C3 shape (1 M x 128, nlist 4096, nprobe 32, k 10): queries/s and the bytes the list
scan gathers (usage on the GPU box: python tools/ivfflat_bench.py [nq])."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gamma_amd import api, synth  # noqa: E402


def main():
    nq = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    N, d, nlist, P, k = 1000000, 128, 4096, 32, 10
    dev = torch.device("cuda", 0)
    base = synth.sift_like(N, d=d, seed=1234)
    q = synth.sift_like(nq, d=d, seed=4321)
    cc, _ = api.train_ivfpq(base[:nlist * 64], nlist, 16)
    g = api.GammaHip(0)
    g.ivfflat_init(d, nlist, api.METRIC_L2, bucket_init_size=max(1000, int(2.5 * N / nlist)))
    g.ivfflat_set_trained(cc)
    g.raw_init(d)
    t0 = time.time()
    for i0 in range(0, N, 1 << 16):
        xb = base[i0:i0 + (1 << 16)]
        g.raw_append(xb)
        g.add(xb, i0)
    print("add %.1f s" % (time.time() - t0))
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, min_score=0.0, max_score=1e30)
    dq = torch.from_numpy(q).to(dev)
    dD = torch.empty((nq, k), dtype=torch.float32, device=dev)
    dI = torch.empty((nq, k), dtype=torch.int64, device=dev)

    def run(n):
        for _ in range(n):
            g.ivfflat_search_device(dq.data_ptr(), nq, k, args, dD.data_ptr(), dI.data_ptr())
        g.synchronize()
    run(2)
    t0 = time.time()
    run(5)
    sec = (time.time() - t0) / 5
    # recall against the exact flat search on a sample
    fa = api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30)
    Df, If = g.flat_search(q[:200], k, fa)
    hit = np.mean([len(set(a) & set(b)) / float(k) for a, b in zip(If, dI.cpu().numpy()[:200])])
    sizes = np.array([g.list_size(l) for l in range(nlist)])
    gather = nq * P * sizes.mean() * d * 4
    print("nq %d: %.2f ms per call = %.0f queries/s, recall@10 %.3f; ~%.1f GB of rows gathered per call = %.2f TB/s"
          % (nq, sec * 1e3, nq / sec, hit, gather / 1e9, gather / sec / 1e12))
    g.close()


if __name__ == "__main__":
    main()
