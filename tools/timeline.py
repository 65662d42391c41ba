"""usage: python3 tools/timeline.py <dir with rocprofv3 --kernel-trace csv output> [step [first-kernel-of-a-step launches-of-it-per-step]]
Prints the kernels of ONE bench step (the step-th occurrence of the coarse sample kernel onwards) with start / end
relative to the step's first kernel, the queue they ran on and the idle gap in front of each on its queue: the critical
path of a Search call, launch gaps included, which --stats averages hide."""
import csv, glob, sys

d = sys.argv[1]
step = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[3] if len(sys.argv) > 3 else "k_coarse_fused<8, true>"
per = int(sys.argv[4]) if len(sys.argv) > 4 else 2
marks = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
# a step has two launches of the sample / store-all kernel: steps start at every second one
starts = marks[0::per]
lo, hi = starts[step], starts[step + 1]
t0 = int(rows[lo]["Start_Timestamp"])
last = {}
print("%-60s %6s %9s %9s %8s %7s" % ("kernel", "queue", "start us", "end us", "dur us", "gap us"))
for r in rows[lo:hi]:
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    qd = r["Queue_Id"]
    gap = (a - last[qd]) / 1e3 if qd in last else 0.0
    last[qd] = b
    print("%-60s %6s %9.1f %9.1f %8.1f %7.1f" % (r["Kernel_Name"].replace("gh::", "").replace("void ", "")[:60], qd, a / 1e3, b / 1e3,
                                             (b - a) / 1e3, gap))
print("step: %.1f us" % ((int(rows[hi]["Start_Timestamp"]) - t0) / 1e3))
