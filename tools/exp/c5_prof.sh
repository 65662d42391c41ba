root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c5p
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -o t -- python3 $root/bench.py --workload c5 --no-extra --steps 4 --warmup 2 --scale-recall-num ${1:-1200} > /tmp/c5p.log 2>&1
tail -1 /tmp/c5p.log | cut -c1-300
f=$(find /tmp/c5p -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("  %-100s calls %5s avg %9.1f us total %8.1f ms" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
