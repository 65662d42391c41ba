# kernel timeline of 1024-query device-pointer calls on the C3 index (tools/latency.py LAT_NQ=1024)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c1024
LAT_NQ=1024 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c1024 -o t -- python3 $root/tools/run_env.py LAT_NQ=1024 $root/tools/latency.py > /tmp/c1024.log 2>&1
tail -3 /tmp/c1024.log
f=$(find /tmp/c1024 -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
# the timed loops: 2 modes x 320 calls = 640 calls
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    c = int(r["Calls"])
    if c >= 600 and c <= 1400:
        print("  %-90s calls %5d avg %8.1f us" % (r["Name"][:90], c, float(r["AverageNs"]) / 1e3))
PY
