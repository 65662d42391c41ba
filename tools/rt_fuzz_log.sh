# usage (GPU box, repo root): tools/rt_fuzz_log.sh -> gpurun_out/r05_rt_fuzz.txt
# The realtime-list fuzz in SIX processes sharing the GPU (the load under which unmapped-and-remapped addresses read back zeros
# in round 4): 2100 scripts with the repack forced at every opportunity (GAMMA_RT_FUZZ_REPACK=1), 2100 with the scripts' own
# thresholds.  Every script compares the lists' contents and capacities and a search with the oracle every few operations; every
# repack is verified on the device before its list tables are published (per-list checksums, csrc/gamma_hip_store.cpp).
out=$GRAFT_REPO_ROOT/gpurun_out/r05_rt_fuzz.txt
: > $out
for mode in forced own; do
  rm -f /tmp/rtlog.txt
  if [ $mode = forced ]; then export GAMMA_RT_FUZZ_REPACK=1; else unset GAMMA_RT_FUZZ_REPACK; fi
  echo "## repack threshold: $mode; GAMMA_RT_FUZZ_SEEDS=0:2100, pytest -n 6" >> $out
  GAMMA_RT_FUZZ_LOG=/tmp/rtlog.txt GAMMA_RT_FUZZ_SEEDS=0:2100 timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 6 -k realtime_script 2>&1 | tail -2 >> $out
  python3 - >> $out <<'PY'
import numpy as np
a = np.loadtxt("/tmp/rtlog.txt", dtype=np.int64).reshape(-1, 3)
print("scripts %d, scripts with at least one repack %d, repacks verified before publication %d, read-back failures %d" % (
    len(a), int((a[:, 1] > 0).sum()), int(a[:, 1].sum()), int(a[:, 2].sum())))
PY
done
cat $out
