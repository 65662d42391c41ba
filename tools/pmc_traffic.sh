#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag> [bench args]
# HBM traffic of the scan kernel per launch, per MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE in
# separate --pmc passes, each calibrated against a kernel with a known byte count on the same box.
tag=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/calib_$c -o pmc -- python3 $root/tools/pmc_calib.py > $out/calib_$c.log 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/bench_$c -o pmc -- python3 $root/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-extra "$@" > $out/bench_$c.log 2>&1
done
python3 - "$out" <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections, json
d = sys.argv[1]
def per_dispatch(sub, kern, ctr):
    f = glob.glob("%s/%s/**/*counter_collection.csv" % (d, sub), recursive=True)
    if not f:
        return None, 0
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f[0])):
        if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            tot += float(r["Counter_Value"]); n += 1
    return (tot / n if n else None), n
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal, ncal = per_dispatch("calib_" + c, "elementwise", c)
    res[c] = {"calib_counter_per_launch": cal, "calib_launches": ncal, "calib_true_bytes": 1 << 30}
    # FETCH_SIZE / WRITE_SIZE are in KB.  Correction of MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
    # reports half of the bytes of a wide coalesced read -> x2; WRITE_SIZE as is.  The box's own calibration
    # (a kernel that moves exactly 2^30 bytes each way) is kept next to it.
    guide = 2.0 if c == "FETCH_SIZE" else 1.0
    for kern in ("k_ivfpq_scan_pair_c8", "k_ivfpq_scan_pair<true, 16, true", "k_ivfpq_scan_pair<true, 16, false", "k_select_final", "k_rerank_topk",
                 "k_l2_gemmform_strip", "k_pq_ip_table"):
        v, n = per_dispatch("bench_" + c, kern, c)
        res[c][kern] = {"counter_per_launch": v, "launches": n,
                        "bytes_per_launch_guide": (v * 1024.0 * guide) if v is not None else None,
                        "bytes_per_launch_calibrated": (v * (1 << 30) / cal) if (v is not None and cal) else None}
print(json.dumps(res, indent=1))
PY
rm -rf $out/calib_* $out/bench_*
