#!/bin/bash
# usage (GPU box, repo root): tools/pmc_q8.sh   -> gpurun_out/pmc_q8/summary.json
# LDS counters of the consumer side of the bounded L2 scan before / after the list-major byte-table pass (csrc/q8scan.hip), on the
# C4 shape at 20 M vectors (nlist 4096: 4 880 codes per list, M 32, nprobe 64, 8192 queries per step): one rocprofv3 --pmc pass
# per counter (counters only, as MI355X_MICROARCH.md prescribes), with the pass on (k_q8_filter + the producers'
# k_ivfpq_scan_pair) and off (GAMMA_HIP_NO_Q8=1: k_ivfpq_scan_pair does all probes).
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_q8
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in q8 noq8; do
  for ctr in SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU; do
    rm -rf $out/raw
    if [ $mode = noq8 ]; then
      timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/raw -o pmc -- python3 $root/tools/run_env.py GAMMA_HIP_NO_Q8=1 $root/bench.py --workload c4 --scale-n 2e7 --scale-nlist 4096 --no-extra --steps 2 --warmup 1 --recall-queries 0 > $out/$mode.$ctr.log 2>&1
    else
      timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/raw -o pmc -- python3 $root/bench.py --workload c4 --scale-n 2e7 --scale-nlist 4096 --no-extra --steps 2 --warmup 1 --recall-queries 0 > $out/$mode.$ctr.log 2>&1
    fi
    python3 - "$out/raw" "$mode" "$ctr" >> $out/lines.jsonl <<'PY'
import csv, glob, json, sys
d, mode, ctr = sys.argv[1:4]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = {}
if f:
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"]
        name = "k_q8_filter" if "k_q8_filter" in k else ("k_q8_exact" if "k_q8_exact" in k else ("k_ivfpq_scan_pair" if "k_ivfpq_scan_pair" in k else None))
        if name:
            a = acc.setdefault(name, [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
print(json.dumps({"mode": mode, "counter": ctr, "kernels": {k: {"total": v[0], "launches": v[1]} for k, v in acc.items()}}))
PY
    rm -rf $out/raw
  done
done
cat $out/lines.jsonl
