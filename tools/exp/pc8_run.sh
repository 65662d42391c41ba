cd /root/repo
timeout 1500 python -m pytest -q -x -m gpu tests/test_gpu_more.py tests/test_gpu_ties.py tests/test_gpu_concurrent.py -k "not fullsize" 2>&1 | tail -3
for v in 0 1; do
timeout 300 python bench.py --cpu-seconds 0 --steps 40 --no-shapes --no-plugin --recall-queries 0 2>/dev/null | grep "^{" | python -c "
import json,sys
z=json.loads(sys.stdin.read()); e=z['config']['exact_ties']; print(z['value'], z['ms_per_step'], z['roofline']['frac'], z['config']['stage_us'], e['caller_threads_each_call_complete_on_return']['2']['qps'], e['qps_with_ties_off'])"
done
