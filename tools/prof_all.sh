#!/bin/bash
# usage (GPU box, repo root): tools/prof_all.sh <tag> [section ...]   sections: c3 c3default c3nocf c2 c4 c5 ivfflat single pmc (default: all but c3default)
# rocprofv3 --kernel-trace --stats of EVERY workload a BASELINE configuration times, one summary per workload under
# gpurun_out/prof_<tag>/ (copy what is to be judged to profiles/): C3 (bench.py, the headline), C2 flat, the C4 shape,
# the C5 shape, IVFFLAT, the single-query chain; then PMC passes of the C3 scan with and without the filter pass
# (FETCH_SIZE, WRITE_SIZE, LDS bank conflicts -- counters only, one per pass, as MI355X_MICROARCH.md prescribes).
tag=$1
shift
sections=${@:-c3 c3nocf c2 c4 c5 ivfflat single pmc}
want() { [[ " $sections " == *" $1 "* ]]; }
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {   # name, script, args...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_$name -o ks -- python3 "$@" > $out/${name}.log 2>&1
  python3 - "$out/raw_$name" "$out/${name}_kernel_stats.txt" "$name" <<'PY'
import csv, glob, sys
src, dst, name = sys.argv[1:4]
f = glob.glob(src + "/**/*kernel_stats.csv", recursive=True)
with open(dst, "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats, workload %s: kernels of libgamma_hip.so by total time\n" % name)
    o.write("%-96s %7s %12s %12s %7s\n" % ("kernel", "calls", "avg us", "total ms", "%"))
    if f:
        rows = [r for r in csv.DictReader(open(f[0]))]
        tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            if "gh::" in r["Name"]:
                o.write("%-96s %7s %12.1f %12.3f %7.2f\n" % (r["Name"][:96], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                          float(r["TotalDurationNs"]) / 1e6, 100.0 * float(r["TotalDurationNs"]) / tot))
PY
  # what the workload itself printed about its launches (bytes per launch, launches per call, stage times): with these
  # lines every roofline fraction quoted for the workload can be recomputed from this one file
  {
    echo "# from the workload's own output (per-launch bytes from the device-side counters of gamma_hip_profile_*; calls = kernel launches in the whole run):"
    grep -a "scan roofline\|search: \|stage ms per step\|stage avg us\|ms per call\|queries/s\|no filter:\|range filter" $out/${name}.log | grep -av "^\[rank\|latency nq" | cut -c1-400 | sed 's/^/#   /' | head -24
    tail -1 $out/${name}.log | python3 -c "
import json, sys
try:
    d = json.loads(sys.stdin.read())
    r = d['roofline']
    print('#   bench line: value %s %s, %s steps, ms_per_step %s; roofline kernel %s: algorithmic_bytes_per_launch %s, avg_launch_us %s, achieved %s GB/s, frac %s' % (
        d['value'], d['unit'], d['steps'], d['ms_per_step'], r['kernel'], r['algorithmic_bytes_per_launch'], r['avg_launch_us'], r['achieved'], r['frac']))
except Exception:
    pass
"
  } >> $out/${name}_kernel_stats.txt
  tail -2 $out/${name}.log | cut -c1-300
  rm -rf $out/raw_$name
}
want c3 && stats c3_bench $root/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes
want c3default && stats c3_bench_default_command $root/bench.py   # the driver's exact command (every leg of the default run)
want c3nocf && stats c3_bench_no_filter_pass $root/tools/run_env.py GAMMA_HIP_NO_SCAN_CF=1 $root/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes
want c2 && stats c2_flat $root/tools/flat_bench.py
want c4 && stats c4_shape_8m $root/tools/c4_scale.py 8e6
want c5 && stats c5_shape_2m $root/tools/c5_scale.py 2e6
want ivfflat && stats ivfflat $root/tools/ivfflat_bench.py
want single && stats single_query $root/tools/latency.py
pmc() {   # name, counter, script, args...
  local name=$1 ctr=$2; shift 2
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${name}_$ctr -o pmc -- python3 "$@" > $out/pmc_${name}_$ctr.log 2>&1
}
if ! want pmc; then ls $out; exit 0; fi
for ctr in FETCH_SIZE WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES; do
  pmc c8 $ctr $root/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes
  pmc cf $ctr $root/tools/run_env.py GAMMA_HIP_NO_C8=1 $root/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes
  pmc nocf $ctr $root/tools/run_env.py GAMMA_HIP_NO_SCAN_CF=1 $root/bench.py --steps 6 --warmup 2 --cpu-seconds 0 --recall-queries 0 --no-extra --no-shapes
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  pmc calib $ctr $root/tools/pmc_calib.py
done
python3 - "$out" <<'PY' | tee $out/pmc_scan_summary.json
import csv, glob, json, sys
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
res = {"note": "per launch of the bounded-scan kernel, C3, 16384 queries; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE x2 on gfx950 "
               "(MI355X_MICROARCH.md); c8 = the default (k_ivfpq_scan_pair_c8<16>: filter pass on the byte image of the query's table, "
               "first probe group of 5), cf = GAMMA_HIP_NO_C8=1 (k_ivfpq_scan_pair<true, 16, true, ..>: filter pass on the fp32 table, "
               "first group of 8: rounds 3-4), nocf = GAMMA_HIP_NO_SCAN_CF=1 (per-list tables for every probe)"}
for mode in ("c8", "cf", "nocf"):
    m = {}
    kern = "k_ivfpq_scan_pair_c8<16>" if mode == "c8" else "k_ivfpq_scan_pair<true, 16, true"
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"):
        v, n = per_dispatch("pmc_%s_%s" % (mode, ctr), kern, ctr)
        m[ctr] = {"per_launch": v, "launches": n}
    f, w = m["FETCH_SIZE"]["per_launch"], m["WRITE_SIZE"]["per_launch"]
    if f is not None and w is not None:
        m["hbm_bytes_per_launch_guide"] = f * 1024.0 * 2.0 + w * 1024.0
    res[mode] = m
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    v, n = per_dispatch("pmc_calib_%s" % ctr, "elementwise", ctr)
    res["calib_" + ctr] = {"per_launch_of_a_kernel_moving_2^30_bytes": v, "launches": n}
print(json.dumps(res, indent=1))
PY
rm -rf $out/pmc_*_FETCH_SIZE $out/pmc_*_WRITE_SIZE $out/pmc_*_SQ_LDS_BANK_CONFLICT $out/pmc_*_SQ_LDS_IDX_ACTIVE $out/pmc_*_SQ_INSTS_LDS $out/pmc_*_SQ_INSTS_VALU $out/pmc_*_SQ_BUSY_CYCLES
ls $out
