"""One rank of 8 of the full-size C4 job with raw vectors SHARDED with their lists: the whole 100 M-vector stream goes through
this rank's Add (HipShardBackend.add: encode every chunk, keep entries and rows of the own lists) and the handle's device
bytes are printed -- the per-rank HBM of `bench.py --workload c4 --gpus 8 --raw-placement sharded`.
usage: python tools/exp/rank_mem_c4.py [N=1e8] [W=8]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from gamma_amd import api, synth
from gamma_amd import dist as gdist

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d, nlist, M = 128, 16384, 32
dev = torch.device("cuda", 0)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=max(200, int(1.3 * N / nlist)))
first = synth.sift_like_device(nlist * 40, d=d, seed=1234, start=0, device=dev).cpu().numpy()
cc, pq = g.ivfpq_train(first, nlist, M)
g.ivfpq_set_trained(cc, pq, None)
ns = 2000000
lno, _ = g.encode(synth.sift_like_device(ns, d=d, seed=1234, start=0, device=dev).cpu().numpy())
est = np.bincount(lno[(lno >= 0) & (lno < nlist)], minlength=nlist).astype(np.float64) * (float(N) / ns)
owner = gdist.balance_lists(np.round(est).astype(np.int64), W)
owned = (owner == 0).astype(np.uint8)
g.set_list_mask(owned)
be = gdist.HipShardBackend(g, 0, raw_sharded=True, owned=owned)
g.raw_init(d)
t0 = time.time()
CH = 1000000
for c in range(0, N, CH):
    be.add(synth.sift_like_device(min(CH, N - c), d=d, seed=1234, start=c, device=dev).cpu().numpy(), c)
mine = sum(g.list_size(l) for l in range(nlist))
print("rank 0 of %d, C4 at %d vectors, raw vectors sharded with the lists: %d vectors in %d lists, %d raw rows (%.2f GB), device bytes of "
      "the handle %.2f GB (replicated rows alone would be %.1f GB); streamed Add through this rank %.0f s" % (
          W, N, mine, int(owned.sum()), g.raw_stats()["rows"], g.raw_stats()["rows"] * d * 4 / 1e9, g.total_mem_bytes() / 1e9,
          N * d * 4 / 1e9, time.time() - t0))
