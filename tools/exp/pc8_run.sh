cd /root/repo
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_more.py tests/test_gpu_ties.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -k "scan_bound_parity_at_batch_size or c3_headline or bounded_scan or large_batch or ivfpq_exact_ties or cut_ties or list_major or q8 or c4 or c5" 2>&1 | tail -3
timeout 600 python bench.py --cpu-seconds 0 2>/dev/null | grep "^{" > gpurun_out/bench_wc.json
python - <<'PY'
import json
z=json.load(open("gpurun_out/bench_wc.json"))
print(z["value"], z["ms_per_step"], z["roofline"]["frac"], z["config"]["stage_us"])
for k,v in z["config"].items():
    if isinstance(v,dict) and ("qps" in v or "scan_frac" in str(v)): print(k, json.dumps(v)[:400])
PY
timeout 900 python tools/c4_scale.py 20000000 2>&1 | tail -6
