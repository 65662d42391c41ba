#!/bin/bash
# usage (GPU box, repo root): tools/prof_flat.sh <tag> ; kernel stats of tools/flat_bench.py (C2 shape)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o flat -- python3 $GRAFT_REPO_ROOT/tools/flat_bench.py > $out.log 2>&1
tail -1 $out.log
python3 $GRAFT_REPO_ROOT/tools/rocprof_summary.py $out $out.txt
rm -f $out/*kernel_trace.csv $out/*.db
sed -n '/kernel_trace.csv grouped/,$p' $out.txt | head -14
