cd /root/repo
for v in 0 1; do
timeout 300 python bench.py --cpu-seconds 0 --steps 40 --no-extra --no-shapes --no-plugin --recall-queries 0 2>/dev/null | grep "^{" | python -c "
import json,sys
z=json.loads(sys.stdin.read()); print(z['value'], z['ms_per_step'], z['roofline']['frac'], z['config']['stage_us'])"
done
