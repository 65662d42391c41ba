cd $GRAFT_REPO_ROOT
for R in 1000 1200; do
  echo "## C5 10M x 768 IP, nlist 4096, M 64, nprobe 64, 4096 queries/step, recall_num $R (+ range filters, + searches under a 10 k vec/s insert stream)"
  timeout 900 python bench.py --workload c5 --steps 8 --warmup 4 --scale-recall-num $R 2> gpurun_out/r5_c5_$R.err | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=j['config']; print(json.dumps({k:j[k] for k in ('value','ms_per_step','roofline')}), json.dumps({'recall_at_10':c['recall_at_10'],'build':c['build'],'per_rank':c['per_rank'],'range_filter':c.get('range_filter'),'search_during_inserts':c.get('search_during_inserts')}))"
done
