# usage: tools/exp/c8_ab.sh "VAR=val VAR2=val" ...   -- one C3 bench run per argument (variables of that arm), prints value / step / scan frac / stages
i=0
for arm in "$@"; do
  i=$((i+1))
  env $arm $DBG timeout 300 python bench.py --no-extra --no-shapes --no-plugin --cpu-seconds 0 --steps 20 --warmup 5 > gpurun_out/ab_$i.json 2> gpurun_out/ab_$i.err
  echo "== $arm"
  grep "scan bound" gpurun_out/ab_$i.err | tail -1 | cut -c1-160
  python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'], j['roofline']['frac'], j['config'].get('stage_us'), j['config'].get('recall_at_10'))" gpurun_out/ab_$i.json
done
