# usage: tools/exp/part_prof.sh "VAR=val ..." ...  -- kernel-trace stats of the scan kernels for each arm
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for arm in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pp_$i
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$i -o t -- python3 $root/tools/run_env.py $arm $root/bench.py --no-extra --no-shapes --no-plugin --cpu-seconds 0 --steps 10 --warmup 3 --recall-queries 0 > /tmp/pp_$i.log 2>&1
  echo "== $arm"
  f=$(find /tmp/pp_$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "scan_pair" in n or "select_final" in n or "q8" in n:
        print("  %-70s calls %5s avg %9.1f min %9.1f max %9.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
