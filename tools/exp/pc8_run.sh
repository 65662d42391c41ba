cd /root/repo
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_more.py tests/test_gpu_ties.py -k "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or ivfpq_exact_ties or cut_ties" 2>&1 | tail -3
for v in 0 1; do
timeout 300 python bench.py --cpu-seconds 0 --steps 40 2>/dev/null | grep "^{" | python -c "
import json,sys
z=json.loads(sys.stdin.read()); print(z['value'], z['ms_per_step'], z['roofline']['frac'], z['config']['stage_us'])"
done
