"""C5 flat probe: inner product over 1 M x 768 embedding-shaped rows, 1024 queries, k = 100 (GAMMA_HIP_FLAT_FIRST_LOG2 sweeps the
first, exactly scored row chunk)"""
import sys, time, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gamma_amd import api, synth
dev=torch.device("cuda",0)
DD=int(os.environ.get("FLAT_D","768")); NN=int(float(os.environ.get("FLAT_N","1e6"))); g=api.GammaHip(0); g.raw_init(DD)
for c0 in range(0,NN,125000): g.raw_append(synth.embedding_like_device(125000,d=DD,seed=1234,start=c0,device=dev).cpu().numpy())
q=synth.embedding_like_device(1024,d=DD,seed=4321,device=dev)
D=torch.empty((1024,100),dtype=torch.float32,device=dev); I=torch.empty((1024,100),dtype=torch.int64,device=dev)
a=api.SearchArgs(metric=api.METRIC_IP,min_score=-1e30,max_score=1e30)
for _ in range(2): g.flat_search_device(q.data_ptr(),1024,100,a,D.data_ptr(),I.data_ptr())
g.synchronize(); t=time.perf_counter()
for _ in range(3): g.flat_search_device(q.data_ptr(),1024,100,a,D.data_ptr(),I.data_ptr())
g.synchronize(); dt=(time.perf_counter()-t)/3
print("flat %d x %d IP 1024q k100: %.2f ms, gemm-form %.1f TF/s" % (NN, DD, dt*1e3, 2*1024*NN*DD/dt/1e12))
