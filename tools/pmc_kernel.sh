#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <kernel-substring> <counter> [<counter> ...]
# one rocprofv3 --pmc pass of the C3 bench per counter (counters only); prints the per-launch average for the kernel
kern=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_kernel
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for ctr in "$@"; do
  rm -rf $out/$ctr
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -o pmc -- python3 $root/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes > $out/$ctr.log 2>&1
  python3 - "$out/$ctr" "$kern" "$ctr" <<'PY'
import csv, glob, sys
d, kern, ctr = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
tot, n = 0.0, 0
if f:
    for r in csv.DictReader(open(f[0])):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            tot += float(r["Counter_Value"]); n += 1
print("%-28s %-40s per launch %.4g over %d launches" % (ctr, kern, tot / n if n else float("nan"), n))
PY
  rm -rf $out/$ctr
done
