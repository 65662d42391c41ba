"""One rank, backend nccl (= RCCL): the collectives gamma_amd.dist issues under torch.cuda.stream(ExternalStream(handle
stream)) -- all_gather_into_tensor, all_to_all_single, broadcast -- run and are ordered with the handle's kernels.  A
single-GPU box cannot run two RCCL ranks; this checks the API path the 8-GPU runs take, not the exchange itself."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gamma_amd import api, synth   # noqa: E402
from gamma_amd import dist as gdist      # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29733")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
d, nlist, M, N, nq, k = 64, 64, 8, 20000, 4096, 10
base = synth.sift_like(N, d=d, seed=1)
q = synth.sift_like(nq, d=d, seed=2)
cc, pq = api.train_ivfpq(base[:8000], nlist, M)
g = api.GammaHip(0)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, 1000)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
g.raw_append(base)
g.add(base, 0)
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=8, recall_num=60, has_rank=True, min_score=-3e38, max_score=3e38)
be = gdist.HipShardBackend(g, 0)
x = torch.from_numpy(q).cuda()
Dr, Ir = g.ivfpq_search(q, k, args)
with torch.cuda.stream(be.stream):
    t = be.empty((1024,), torch.uint8)
    t.fill_(7)
    out = be.empty((1, 1024), torch.uint8)
    dist.all_gather_into_tensor(out.view(-1), t)
    a2a = be.empty((1024,), torch.uint8)
    dist.all_to_all_single(a2a, t)
    dist.broadcast(t, 0)
torch.cuda.synchronize()
assert int(out.sum()) == 7 * 1024 and int(a2a.sum()) == 7 * 1024
D, I = gdist.replicated_search(be, x, k, args)
torch.cuda.synchronize()
assert np.array_equal(I.cpu().numpy(), Ir) and D.cpu().numpy().tobytes() == Dr.tobytes()
rs = gdist.ReplicatedStream(be, k, args)
assert rs.submit(x) is None
o1 = rs.submit(x)
o2 = rs.flush()
torch.cuda.synchronize()
for o in (o1, o2):
    assert np.array_equal(o[1].cpu().numpy(), Ir) and o[0].cpu().numpy().tobytes() == Dr.tobytes()
rs.close()
D, I = gdist.sharded_search(be, x, k, args)
torch.cuda.synchronize()
assert np.array_equal(I.cpu().numpy(), Ir)
dist.destroy_process_group()
g.close()
print("nccl stream check ok")
