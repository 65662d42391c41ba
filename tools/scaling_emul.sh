#!/bin/bash
# usage (GPU box): tools/scaling_emul.sh > gpurun_out/scaling.txt
# The compute side of weak scaling at C3 without an 8-GPU node: what ONE rank of a W-GPU job runs per step of W x 16384
# queries, (a) list-sharded (tools/rank_emul.py: the lists rank 0 would own, all W x 16384 queries scanned, its slice merged
# and re-ranked; no communication) and (b) replicated (every rank answers 16384 queries on the whole index = the 1-GPU step;
# what is added is one all-gather of W x 16384 x 10 x 12 bytes).
root=$GRAFT_REPO_ROOT
echo "# 1 GPU, 16384 queries per step (bench.py, replay at the end of every call: the ranks of a dist job do not defer)"
timeout 300 python3 $root/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes --no-deferred-replay 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f  stages %s' % (j['ms_per_step'], j['config']['stage_us']))"
for W in 2 4 8; do
  echo "# list-sharded, W=$W, 16384 queries per rank and step"
  timeout 400 python3 $root/tools/rank_emul.py $W 16384 2>&1 | tail -4
done
