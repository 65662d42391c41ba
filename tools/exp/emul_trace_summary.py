"""usage: emul_trace_summary.py <rocprofv3 kernel_trace.csv> -- per-kernel totals of the LAST emulated shard step of
tools/c4_scale.py (C4_EMUL=W): the kernels between the last two k_merge_shards launches."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_merge_shards" in r["Kernel_Name"]]
lo, hi = idx[-2] + 1, idx[-1] + 1
while hi < len(rows) and "k_coarse" not in rows[hi]["Kernel_Name"] and hi < idx[-1] + 12:
    hi += 1
seg = rows[lo:hi]
print("segment: %d kernels, wall %.3f ms" % (len(seg), (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6))
agg = collections.OrderedDict()
for r in seg:
    n = r["Kernel_Name"][:110]
    a = agg.setdefault(n, [0, 0.0, r["Grid_Size_X"], r["Workgroup_Size_X"]])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t, gx, wx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%10.1f us %3d launches  grid %10s wg %4s  %s" % (t, c, gx, wx, n))
print("in launch order:")
for r in seg:
    print("  %9.1f us  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:100]))
