#!/bin/bash
# usage (GPU box): tools/pmc_workloads.sh > gpurun_out/pmc_workloads.json
# One FETCH_SIZE and one WRITE_SIZE pass (rocprofv3 --pmc, counters only, separate passes) of the other BASELINE workloads:
# per-launch averages of the kernel that dominates each of them.  HBM-side bytes = FETCH_SIZE (KB) x 1024 x 2 (gfx950,
# MI355X_MICROARCH.md) + WRITE_SIZE (KB) x 1024; this box's own calibration of FETCH_SIZE is in profiles/r04_pmc_scan_summary.json.
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_wl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {   # name kernel-substring kernel-regex script args...
  local name=$1 kern=$2 kre=$3; shift 3
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/${name}_$ctr
    # counters only for the kernel of interest: with every kernel of the build phase counted the pass does not finish
    timeout 900 rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "$kre" --output-format csv -d $out/${name}_$ctr -o pmc -- python3 "$@" > $out/${name}_$ctr.log 2>&1
  done
  python3 - "$out" "$name" "$kern" <<'PY'
import csv, glob, json, sys
d, name, kern = sys.argv[1:4]
res = {"workload": name, "kernel": kern}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("%s/%s_%s/**/*counter_collection.csv" % (d, name, ctr), recursive=True)
    tot, n = 0.0, 0
    if f:
        for r in csv.DictReader(open(f[0])):
            if kern in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                tot += float(r["Counter_Value"]); n += 1
    res[ctr + "_kb_per_launch"] = tot / n if n else None
    res[ctr + "_launches"] = n
if res["FETCH_SIZE_kb_per_launch"] is not None and res["WRITE_SIZE_kb_per_launch"] is not None:
    res["hbm_bytes_per_launch_guide"] = res["FETCH_SIZE_kb_per_launch"] * 2048.0 + res["WRITE_SIZE_kb_per_launch"] * 1024.0
print(json.dumps(res))
PY
  rm -rf $out/${name}_FETCH_SIZE $out/${name}_WRITE_SIZE
}
run c2_flat "k_flat_filter<true, 128>" "k_flat_filter" $root/tools/flat_bench.py
run c4_shape_8m "k_ivfpq_scan_pair<true, 32, true" "k_ivfpq_scan_pair" $root/tools/c4_scale.py 8e6
run c5_shape_2m "k_ivfpq_scan_pair<false, 64, true" "k_ivfpq_scan_pair" $root/tools/c5_scale.py 2e6
run ivfflat "k_ivfflat_lm" "k_ivfflat_lm" $root/tools/ivfflat_bench.py
