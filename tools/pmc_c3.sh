#!/bin/bash
# usage (GPU box, repo root): tools/pmc_c3.sh [VAR=value ...]  -> gpurun_out/pmc_c3/lines.jsonl
# SQ counters of the scan kernel of the C3 bench (k_ivfpq_scan_pair*), counters only (MI355X_MICROARCH.md), one pass per group;
# arms: the default, and the default + the variables given (e.g. GAMMA_HIP_C8=1).
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/pmc_c3
mkdir -p $out
rm -f $out/lines.jsonl
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
G2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT"
for arm in base var; do
  for g in 1 2; do
    if [ $g = 1 ]; then ctrs="$G1"; else ctrs="$G2"; fi
    rm -rf $out/raw
    if [ $arm = var ]; then
      timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw -o pmc -- python3 $root/tools/run_env.py "$@" $root/bench.py --no-extra --no-shapes --no-plugin --cpu-seconds 0 --steps 3 --warmup 2 --recall-queries 0 > $out/$arm.$g.log 2>&1
    else
      timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/raw -o pmc -- python3 $root/bench.py --no-extra --no-shapes --no-plugin --cpu-seconds 0 --steps 3 --warmup 2 --recall-queries 0 > $out/$arm.$g.log 2>&1
    fi
    python3 - "$out/raw" "$arm" >> $out/lines.jsonl <<'PY'
import csv, glob, json, sys
d, arm = sys.argv[1:3]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = {}
if f:
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if "k_ivfpq_scan_pair" not in k or "Lb1ELi16ELb0" in k or "<true, 16, false" in k:
            continue
        a = acc.setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
print(json.dumps({"arm": arm, "per_launch": {k: v[0] / max(1, v[1]) for k, v in acc.items()}, "launches": {k: v[1] for k, v in acc.items()}}))
PY
    rm -rf $out/raw
  done
done
cat $out/lines.jsonl
