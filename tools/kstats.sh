#!/bin/bash
# usage (GPU box, repo root): tools/kstats.sh <min calls> tools/<script>.py [args...]  -- per-kernel averages (rocprofv3)
minc=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/kstats_$$
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks -- python3 $script "$@" > $out.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("gh::", "void gh::")) and int(r["Calls"]) >= $minc:
        print("%-80s %6s %10.1f us" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -5 $out.log
rm -rf $out $out.log
