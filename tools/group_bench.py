#!/usr/bin/env python3
"""The headline workload (C3) through the IN-PROCESS multi-GPU index: one process, gamma_hip_group_* over --gpus devices
(csrc/gamma_hip_group.cpp), the index sharded by IVF list, what the plugins run with "devices": "0,1,..".
`python bench.py --gpus N --in-process` runs this; the multi-process form over RCCL is bench.py's default for N > 1.
--one-gpu puts every member on device 0 (functional check on a single-GPU box, not a measurement).
Prints one JSON line in bench.py's format (value = queries/s of the whole group, weak scaling: nq per member and step)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--one-gpu", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--nq", type=int, default=16384, help="queries per member and step")
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--nlist", type=int, default=4096)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--nprobe", type=int, default=32)
    ap.add_argument("--recall-num", type=int, default=200)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--placement", default="auto", choices=["auto", "shard", "replicate"])
    a = ap.parse_args()
    import torch
    from gamma_amd import api, synth
    W, d = a.gpus, 128
    devices = [0] * W if a.one_gpu else list(range(W))
    base = synth.sift_like(a.n, d=d, seed=1234)
    gnq = a.nq * W
    nb = 2
    queries = synth.sift_like(gnq * nb, d=d, seed=4321)
    cc, pq = api.train_ivfpq(base[:min(a.n, a.nlist * 64)], a.nlist, a.m)
    replicate = a.placement == "replicate" or (a.placement == "auto" and a.n * (a.m + 12) <= (2 << 30))
    grp = api.GammaHipGroup(devices)
    grp.set_placement(replicate)
    for m in grp.members:
        m.ivfpq_init(d, a.nlist, a.m, 8, api.METRIC_L2, bucket_init_size=max(1000, int(2.5 * a.n / a.nlist / (1 if replicate else W))))
        m.ivfpq_set_trained(cc, pq, None)
        m.raw_init(d)
        for i0 in range(0, a.n, 1 << 18):
            m.raw_append(base[i0:i0 + (1 << 18)])
    lno, _ = grp.members[0].encode(base[:min(a.n, 262144)])
    grp.set_owners(np.bincount(lno, minlength=a.nlist))          # balanced by the list sizes of a sample
    t0 = time.time()
    for i0 in range(0, a.n, 100000):
        grp.add(base[i0:i0 + 100000], i0)                        # one encode per batch, AddKeys at the owners
    build_s = time.time() - t0
    args = api.SearchArgs(metric=api.METRIC_L2, nprobe=a.nprobe, recall_num=a.recall_num, has_rank=True, min_score=0.0,
                          max_score=1e30)
    dev0 = torch.device("cuda", devices[0])
    d_q = torch.from_numpy(queries).to(dev0)
    d_D = torch.empty((gnq, a.k), dtype=torch.float32, device=dev0)
    d_I = torch.empty((gnq, a.k), dtype=torch.int64, device=dev0)
    torch.cuda.synchronize()

    def step(i):
        xb = d_q[(i % nb) * gnq:(i % nb + 1) * gnq]
        grp.ivfpq_search_device(xb.data_ptr(), gnq, a.k, args, d_D.data_ptr(), d_I.data_ptr())

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # recall@10 of the last batch against the exact flat search of member 0 (raw vectors are replicated)
    nrq = 500
    xb = queries[((a.warmup + a.steps - 1) % nb) * gnq:][:nrq]
    Df, If = grp.members[0].flat_search(xb, a.k, api.SearchArgs(metric=api.METRIC_L2, min_score=0.0, max_score=1e30))
    Ig = d_I[:nrq].cpu().numpy()
    recall = sum(len(set(Ig[i].tolist()) & set(If[i].tolist())) for i in range(nrq)) / float(nrq * a.k)
    out = {"metric": "queries/sec @ recall@10>=0.95, IVFPQ nlist=%d nprobe=%d" % (a.nlist, a.nprobe),
           "value": round(gnq * a.steps / dt, 1), "unit": "queries/s", "n_gpus": W, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "C3 through the in-process group: %d members%s, %d queries per member and step, lists sharded "
                                  "by greedy sum(len)" % (W, " on ONE GPU (functional check)" if a.one_gpu else "", a.nq)
                      if not replicate else "C3 through the in-process group: %d members%s, %d queries per member and step, lists "
                      "REPLICATED, queries split" % (W, " on ONE GPU (functional check)" if a.one_gpu else "", a.nq),
                      "parallelism": "in-process, %s x%d" % ("query-parallel over replicas" if replicate else "list-sharded", W), "recall_at_10": round(recall, 4),
                      "build_s": round(build_s, 1), "device_bytes": grp.total_mem_bytes()}}
    print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
