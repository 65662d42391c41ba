#!/usr/bin/env python3
"""GPU idle time between kernels from a rocprofv3 kernel_trace.csv: takes the last `nsteps` repetitions
of the step (delimited by the first kernel of a step, default the coarse GEMM) and prints, per step,
span / busy / idle and the idle gaps by (previous kernel -> next kernel)."""
import collections
import csv
import glob
import sys


def short(n):
    n = n.split("(")[0].replace("void ", "")
    return n[-48:]


def main(d, first="k_l2_gemmform", nsteps=10):
    tr = glob.glob(d + "/*kernel_trace.csv")[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tr))]
    rows.sort()
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    starts = starts[-(nsteps + 1):]
    gaps = collections.defaultdict(list)
    spans, busys = [], []
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b + 1]
        span = seg[-1][0] - seg[0][0]
        busy, cur_end = 0, seg[0][0]
        for (s, e, n), (s2, e2, n2) in zip(seg[:-1], seg[1:]):
            cur_end = max(cur_end, e)
            if s2 > cur_end:
                gaps[(n, n2)].append(s2 - cur_end)
        # union of intervals
        cur_s, cur_e = seg[0][0], seg[0][1]
        for s, e, n in seg[1:-1]:
            if s > cur_e:
                busy += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        spans.append(span)
        busys.append(busy)
    n = len(spans)
    print("steps %d: span %.1f us, busy %.1f us, idle %.1f us (avg per step)" % (
        n, sum(spans) / n / 1e3, sum(busys) / n / 1e3, (sum(spans) - sum(busys)) / n / 1e3))
    for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print("  %8.1f us/step  n/step=%.1f  %s -> %s" % (sum(v) / n / 1e3, len(v) / n, k[0], k[1]))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *(map(int, sys.argv[3:4])))
