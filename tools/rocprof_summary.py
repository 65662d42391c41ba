#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV directory into a small text summary
(our kernels only, split by launch geometry) for profiles/."""
import collections
import csv
import glob
import sys


def main(d, out=None):
    lines = []
    st = glob.glob(d + "/*kernel_stats.csv")
    if st:
        lines.append("# rocprofv3 --kernel-trace --stats : kernel_stats.csv rows of namespace gh:: "
                     "(Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs)")
        for r in csv.DictReader(open(st[0])):
            if "gh::" in r["Name"]:
                lines.append("%s | calls=%s total_ns=%s avg_ns=%.1f pct=%s min_ns=%s max_ns=%s" % (
                    r["Name"].split("(")[0].replace("void ", ""), r["Calls"], r["TotalDurationNs"],
                    float(r["AverageNs"]), r["Percentage"], r["MinNs"], r["MaxNs"]))
    tr = glob.glob(d + "/*kernel_trace.csv")
    if tr:
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            n = r["Kernel_Name"]
            if "gh::" not in n:
                continue
            key = (n.split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])),
                   r["Grid_Size_Y"], r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])
            agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        lines.append("")
        lines.append("# kernel_trace.csv grouped by (kernel, blocks_x, blocks_y, block, vgpr, lds): n, avg/min/max us")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            lines.append("%-34s blocks=%-8d y=%-5s block=%-4s vgpr=%-4s lds=%-6s n=%-4d avg=%9.1f min=%9.1f max=%9.1f" % (
                k[0], k[1], k[2], k[3], k[4], k[5], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
