#!/bin/bash
# usage (GPU box, repo root): tools/prof_dist.sh <tag> <gemm launches per step> [bench args]; kernel timeline of one step of the sharded orchestration
tag=$1; per=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --force-dist --steps 20 --warmup 3 --cpu-seconds 0 --recall-queries 0 --no-extra "$@" > $out.log 2>&1
grep "host enqueue" $out.log
python3 $GRAFT_REPO_ROOT/tools/timeline_dump.py $out k_l2_gemmform $per 3
rm -rf $out
