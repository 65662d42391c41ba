#!/usr/bin/env python3
"""One step of a rocprofv3 kernel_trace.csv as a listing: start offset, duration, queue, kernel.
usage: timeline_dump.py <dir> <first-kernel-substring> <occurrences-per-step> [step-index-from-end]"""
import csv
import glob
import sys

d, first, per = sys.argv[1], sys.argv[2], int(sys.argv[3])
back = int(sys.argv[4]) if len(sys.argv) > 4 else 3
tr = glob.glob(d + "/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(tr)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"),
                 r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:]))
rows.sort()
starts = [i for i, r in enumerate(rows) if first in r[3]][::per]
a, b = starts[-back - 1], starts[-back]
t0 = rows[a][0]
end = t0
for s, e, q, n in rows[a:b]:
    gap = s - end
    print("%9.1f +%8.1f us  q=%-3s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n, "   <-- idle %.1f us" % (gap / 1e3) if gap > 3000 else ""))
    end = max(end, e)
print("step span %.1f us" % ((rows[b][0] - t0) / 1e3))
