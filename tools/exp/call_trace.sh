#!/bin/bash
# usage: tools/exp/call_trace.sh <nq> -- the kernel timeline of ONE C3 Search call of nq queries (rocprofv3 --kernel-trace)
root=$GRAFT_REPO_ROOT
nq=${1:-1024}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/call_tr
cat > /tmp/call_one.py <<PY
import os, sys, time
sys.path.insert(0, "$root")
import numpy as np, torch
from gamma_amd import api, synth
N, d, nlist, M = 1000000, 128, 4096, 16
base = synth.sift_like(N, d=d, seed=1234)
g = api.GammaHip(0)
cc, pq = g.ivfpq_train(base[:nlist * 64], nlist, M)
g.ivfpq_init(d, nlist, M, 8, api.METRIC_L2, bucket_init_size=1000)
g.ivfpq_set_trained(cc, pq, None)
g.raw_init(d)
for i0 in range(0, N, 250000):
    g.raw_append(base[i0:i0 + 250000]); g.add(base[i0:i0 + 250000], i0)
nq, k = $nq, 10
q = synth.sift_like(nq, d=d, seed=4321)
dq = torch.from_numpy(q).to("cuda:0")
D = torch.empty((nq, k), dtype=torch.float32, device="cuda:0"); I = torch.empty((nq, k), dtype=torch.int64, device="cuda:0")
args = api.SearchArgs(metric=api.METRIC_L2, nprobe=32, recall_num=200, has_rank=True, min_score=0.0, max_score=1e30)
for i in range(6):
    g.ivfpq_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr()); g.synchronize()
torch.zeros(1, device="cuda:0").fill_(7.0); torch.cuda.synchronize()     # marker kernel
g.ivfpq_search_device(dq.data_ptr(), nq, k, args, D.data_ptr(), I.data_ptr()); g.synchronize()
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/call_tr -o ks -- python3 /tmp/call_one.py > /tmp/call_tr.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/call_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gh::" not in r["Kernel_Name"] and "fill" in r["Kernel_Name"].lower()]
seg = [r for r in rows[idx[-1] + 1:] if True]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    print("%8.1f us  %7.1f us  grid %8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Kernel_Name"][:84]))
print("call: %.1f us of GPU timeline" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3))
PY
