#!/bin/bash
# usage (GPU box, repo root): LAT_*=... tools/lat_prof.sh <tag>   -- kernel stats of tools/latency.py for one shape
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/latprof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o lat -- python3 $GRAFT_REPO_ROOT/tools/latency.py > $out.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r["Name"].startswith(("gh::", "void gh::")) and int(r["Calls"]) >= 300:
        print("%-70s %6s %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep "nq=" $out.log
rm -rf $out
