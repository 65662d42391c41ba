# round-end collection (GPU box, repo root): kernel stats of the driver's exact bench command, the full-size C4 / C5 runs, the
# default bench line, the whole GPU suite.  Everything under gpurun_out/; copy what is to be judged to profiles/.
cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh r05 c3 c3default c2 c4 c5 > gpurun_out/r5_prof_all.log 2>&1
{
echo "# round 5, full-size configurations on ONE MI355X through bench.py --workload c4|c5 (bench_scale.py: device streams, streamed Add)"
for mode in q8 noq8; do
  for R in 150 300; do
    echo "## C4 100M x 128, nlist 16384, M 32, nprobe 64, 8192 queries/step, recall_num $R, consumer pass: $mode"
    if [ $mode = noq8 ]; then export GAMMA_HIP_NO_Q8=1; else unset GAMMA_HIP_NO_Q8; fi
    timeout 600 python bench.py --workload c4 --no-extra --steps 8 --warmup 4 --scale-recall-num $R 2> gpurun_out/r5_c4_${mode}_$R.err | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps({k:j[k] for k in ('value','ms_per_step','roofline')}), json.dumps({'recall_at_10':j['config']['recall_at_10'],'build':j['config']['build'],'per_rank':j['config']['per_rank']}))"
  done
done
unset GAMMA_HIP_NO_Q8
for R in 1000 1200; do
  echo "## C5 10M x 768 IP, nlist 4096, M 64, nprobe 64, 4096 queries/step, recall_num $R (+ range filters, + searches under a 10 k vec/s insert stream)"
  timeout 900 python bench.py --workload c5 --steps 8 --warmup 4 --scale-recall-num $R 2> gpurun_out/r5_c5_$R.err | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=j['config']; print(json.dumps({k:j[k] for k in ('value','ms_per_step','roofline')}), json.dumps({'recall_at_10':c['recall_at_10'],'build':c['build'],'per_rank':c['per_rank'],'range_filter':c.get('range_filter'),'search_during_inserts':c.get('search_during_inserts')}))"
done
echo "## one rank of W list shards emulated on this GPU, C4 shape at 20 M (tools/c4_scale.py 2e7, C4_EMUL=2,8), byte-table pass on / off"
for mode in q8 noq8; do
  if [ $mode = noq8 ]; then export GAMMA_HIP_NO_Q8=1; else unset GAMMA_HIP_NO_Q8; fi
  echo "# consumer pass: $mode"
  C4_EMUL=2,8 timeout 600 python tools/c4_scale.py 2e7 2>/dev/null | grep "search:\|emulated rank" | cut -c1-340
done
unset GAMMA_HIP_NO_Q8
} > gpurun_out/r05_scale_runs.txt 2>&1
python bench.py > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
tail -c 600 gpurun_out/r05_bench.json
timeout 2400 python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r5_suite_final.log 2>&1
tail -12 gpurun_out/r5_suite_final.log
