# C8 query-major pass against the list-major q8 pass at list lengths around the switch (M = 16, nlist 4096, nprobe 32)
for n in 2000000 3000000 4000000 6000000; do
  for arm in "GAMMA_HIP_NO_Q8=1" "GAMMA_HIP_Q8_MINLEN=0"; do
    echo "== n $n  $arm"
    env $arm timeout 400 python bench.py --n $n --no-extra --no-shapes --no-plugin --cpu-seconds 0 --steps 10 --warmup 3 --recall-queries 0 2>/dev/null | python -c "import json,sys; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['ms_per_step'], j['config'].get('stage_us'))"
  done
done
