#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# kernel-trace + stats of a short bench run; condensed summary -> gpurun_out/prof_<tag>.txt
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --recall-queries 0 --no-extra "$@" > $out.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $out $out.txt
rm -f $out/*kernel_trace.csv $out/*.db
sed -n '/kernel_trace.csv grouped/,$p' $out.txt | grep -v "k_pq_encode\|k_precompute\|blocks=128 \|blocks=512 " | head -40
