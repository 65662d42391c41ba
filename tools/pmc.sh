#!/bin/bash
# usage (GPU box): tools/pmc.sh <tag> "<COUNTER LIST>" <kernel-substring> [bench args]
tag=$1; ctrs=$2; kern=$3; shift 3
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-extra "$@" > $out.log 2>&1
python3 - "$out" "$kern" <<'PY'
import csv, glob, sys, collections
d, kern = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*counter_collection.csv")
if not f:
    print("no counter file", glob.glob(d + "/*")); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if kern in r["Kernel_Name"]:
        a = agg[r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in sorted(agg.items()):
    print("%-28s per-dispatch %.4g  (n=%d)" % (k, v / max(n, 1), n))
PY
rm -rf $out
