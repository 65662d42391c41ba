cd /tmp && export TMPDIR=/tmp
for nq in 256 1024 4096; do
  FLAT_BENCH_NO_TIES=1 FLAT_BENCH_NQ=$nq timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$nq -o ks -- python3 $GRAFT_REPO_ROOT/tools/flat_bench.py > /tmp/pr_$nq.log 2>&1
  tail -1 /tmp/pr_$nq.log
  f=$(find /tmp/pr_$nq -name "*kernel_stats.csv" | head -1)
  grep "k_flat_filter\|k_pairwise_lds\|k_flat_compact" $f | cut -d, -f1-4 | cut -c1-160
done
