"""python tools/exp/coarse_nseg_check.py: the fused coarse quantizer (optionally with a strip count forced by a local patch)
against the distance-matrix path on C3-shaped tie-heavy data (bit for bit, exact ties on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gamma_amd import api, synth
d, nlist, P, nq = 128, 4096, 32, 16384
base = synth.sift_like(200000, d=d, seed=1234)
rng = np.random.default_rng(5)
cc = base[rng.choice(len(base), nlist, replace=False)].copy()
x = synth.sift_like(nq, d=d, seed=4321)
pq = rng.standard_normal((16, 256, 8)).astype(np.float32)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, 16, 8, api.METRIC_L2, 100)
g.ivfpq_set_trained(cc, pq, None)
dev = torch.device("cuda", 0)
dx = torch.from_numpy(x).to(dev)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, coarse_mode=1, min_score=-3e38, max_score=3e38)
def run():
    cd = torch.empty((nq, P), dtype=torch.float32, device=dev)
    ci = torch.empty((nq, P), dtype=torch.int32, device=dev)
    g.ivfpq_coarse_device(dx.data_ptr(), nq, args, cd.data_ptr(), ci.data_ptr())
    g.synchronize()
    return cd.cpu().numpy(), ci.cpu().numpy()
g.set_exact_ties(True)
g.set_coarse_fused(False)
D0, I0 = run()
g.set_coarse_fused(True, 128)
D1, I1 = run()
bad = np.nonzero((I0 != I1).any(axis=1) | (D0.view(np.uint32) != D1.view(np.uint32)).any(axis=1))[0]
print("nseg", os.environ.get("GAMMA_HIP_COARSE_NSEG"), "rows that differ:", len(bad), bad[:8])
for q in bad[:2]:
    j = np.nonzero(I0[q] != I1[q])[0]
    print(q, j, I0[q][j], I1[q][j], D0[q][j], D1[q][j])
g.close()
