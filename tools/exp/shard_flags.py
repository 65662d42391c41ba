"""how many queries does the merge flag for the cross-shard tie replay on ordinary data?  (debug)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gamma_amd import api, synth
from gamma_amd import dist as gdist
W, nq, N, d, nlist, M, P, R, k = 2, 4096, 200000, 128, 1024, 16, 32, 200, 10
base = synth.sift_like(N, d=d, seed=1234)
q = synth.sift_like(nq, d=d, seed=4321)
cc, pq = api.train_ivfpq(base[:nlist * 64], nlist, M)
grp = api.GammaHipGroup([0] * W)
for m in grp.members:
    m.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 1000)
    m.ivfpq_set_trained(cc, pq, None)
    m.raw_init(d)
    m.raw_append(base)
lno, _ = grp.members[0].encode(base)
grp.set_owners(np.bincount(lno, minlength=nlist))
grp.add(base, 0)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=0.0, max_score=1e30)
os.environ["GAMMA_HIP_GROUP_DBG"] = "1"
for m in grp.members:
    m.tie_stats(reset=True)
D, I = grp.ivfpq_search(q, k, args)
for i, m in enumerate(grp.members):
    print("member", i, m.tie_stats())
full = api.GammaHip(0)
full.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 1000)
full.ivfpq_set_trained(cc, pq, None)
full.raw_init(d); full.raw_append(base); full.add(base, 0)
full.tie_stats(reset=True)
Df, If = full.ivfpq_search(q, k, args)
print("single handle", full.tie_stats(), "equal:", np.array_equal(I, If), D.tobytes() == Df.tobytes())

# ---- by hand: the tables of the shards for member 0's slice, the conditions of k_flag_merge_cut in numpy
dev = torch.device("cuda", 0)
x = torch.from_numpy(q).to(dev)
per = nq // W
backs = [gdist.HipShardBackend(g, 0) for g in grp.members]
cd = torch.zeros((nq, P), dtype=torch.float32, device=dev)
pr = torch.full((nq, P), -1, dtype=torch.int32, device=dev)
a2 = api.SearchArgs(metric=api.METRIC_L2, nprobe=P, recall_num=R, has_rank=True, min_score=0.0, max_score=1e30, coarse_mode=1)
for s in range(W):
    backs[s].coarse(x[s * per:(s + 1) * per], a2, cd[s * per:(s + 1) * per], pr[s * per:(s + 1) * per])
    grp.members[s].synchronize()
rd, ri = [], []
for s in range(W):
    rdis = torch.zeros((nq, R), dtype=torch.float32, device=dev)
    rids = torch.full((nq, R), -1, dtype=torch.int64, device=dev)
    backs[s].search_shard(x, cd, pr, k, a2, rdis, rids)
    grp.members[s].synchronize()
    rd.append(rdis[:per].cpu().numpy()); ri.append(rids[:per].cpu().numpy())
allv = np.concatenate(rd, axis=1)
alli = np.concatenate(ri, axis=1)
allv_valid = np.where(alli >= 0, allv, np.inf)
srt = np.sort(allv_valid, axis=1)
vk = srt[:, R - 1]
vk1 = srt[:, R]
print("rows whose R-th and (R+1)-th merged values are equal:", int((vk == vk1).sum()), "of", per)
for s in range(W):
    full_tab = (ri[s] >= 0).all(axis=1)
    last = rd[s][:, R - 1]
    mx = np.where(ri[s] >= 0, rd[s], -np.inf).max(axis=1)
    print("shard", s, "full tables:", int(full_tab.sum()), " last entry == vk:", int((last == vk).sum()), " max entry == vk:", int((mx == vk).sum()),
          " sorted rows:", int((np.diff(rd[s], axis=1) >= 0).all(axis=1).sum()))
