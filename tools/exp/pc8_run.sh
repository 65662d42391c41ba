cd /root/repo
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_more.py tests/test_gpu_ties.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -k "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch or ivfpq_exact_ties or cut_ties or filter_pass or c4 or c5" 2>&1 | tail -4
for v in 0 1; do
timeout 600 python bench.py --cpu-seconds 0 2>/dev/null | grep "^{" > gpurun_out/bench_wc.json
python - <<'PY'
import json
z=json.load(open("gpurun_out/bench_wc.json"))
print(z["value"], z["ms_per_step"], z["roofline"]["frac"], z["config"]["stage_us"])
c=z["config"]["c4_shape_8m"]; print("c4_8m", c["qps"], c["ms_per_call"], c["roofline"]["frac"], c.get("qps_with_10pct_filter"))
c=z["config"]["c5_shape_2m"]; print("c5_2m", c["qps"], c.get("qps_with_10pct_range_filter"))
PY
done
