#!/bin/bash
# per-launch durations of the flat search's kernels for the LAST call of tools/flat_bench.py (rocprofv3 --kernel-trace)
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/flat_tr
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/flat_tr -o ks -- python3 $root/tools/flat_bench.py > /tmp/flat_tr.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/flat_tr/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_flat_prep_queries" in r["Kernel_Name"]]
seg = rows[idx[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
for r in seg:
    print("%9.1f us  +%8.1f  %8.1f us  grid %8s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, 0.0, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                                   r["Grid_Size_X"], r["Kernel_Name"][:70]))
print("call: %.1f us" % ((int(seg[-1]["End_Timestamp"]) - t0) / 1e3))
PY
