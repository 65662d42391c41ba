# round 6 round-end collection (GPU box, repo root): kernel stats of the C3 bench and the C4 shape, HBM-traffic counters of the
# scan kernel (FETCH_SIZE / WRITE_SIZE in passes of their own + the box's calibration), the full-size C4 / C5 lines, the
# emulated-rank matrices of the two-phase shard scan, the default bench line.  Everything under gpurun_out/; what is to be
# judged is copied to profiles/ afterwards.
cd $GRAFT_REPO_ROOT
bash tools/prof_all.sh r06 c3 c4 > gpurun_out/r6_prof_all.log 2>&1
bash tools/pmc_traffic.sh r06 --no-shapes > gpurun_out/r6_pmc_traffic.log 2>&1
{
echo "# round 6, full-size configurations on ONE MI355X through bench.py --workload c4|c5 (bench_scale.py: device streams, streamed Add)"
for R in 150 300; do
  echo "## C4 100M x 128, nlist 16384, M 32, nprobe 64, 8192 queries/step, recall_num $R"
  timeout 600 python bench.py --workload c4 --no-extra --steps 8 --warmup 4 --scale-recall-num $R 2> gpurun_out/r6_c4_$R.err | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps({k:j[k] for k in ('value','ms_per_step','roofline')}), json.dumps({'recall_at_10':j['config']['recall_at_10'],'build':j['config']['build'],'per_rank':j['config']['per_rank']}))"
done
echo "## C5 10M x 768 IP, nlist 4096, M 64, nprobe 64, 4096 queries/step, recall_num 1000 (+ range filters, + searches under a 10 k vec/s insert stream)"
timeout 900 python bench.py --workload c5 --steps 8 --warmup 4 --scale-recall-num 1000 2> gpurun_out/r6_c5_1000.err | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=j['config']; print(json.dumps({k:j[k] for k in ('value','ms_per_step','roofline')}), json.dumps({'recall_at_10':c['recall_at_10'],'build':c['build'],'per_rank':c['per_rank'],'range_filter':c.get('range_filter'),'search_during_inserts':c.get('search_during_inserts')}))"
} > gpurun_out/r06_scale_runs.txt 2>&1
G1=2 timeout 600 python tools/shard_two_phase.py 2e7 2,4,8 100 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_emul_20m_final.txt
G1=2 timeout 900 python tools/shard_two_phase.py 1e8 2,4,8 150 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_emul_100m_final.txt
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 400 gpurun_out/r06_bench.json
