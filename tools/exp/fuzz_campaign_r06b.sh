# round 6, second campaign: fresh seeds on the tree with the re-compiled filter loops (two copies by validity predicates, lists
# pipelined per wave; q8scan kernels by template) -- the families that reach them, with and without predicates, the list-major
# pass forced on short lists, and the opt-in one-workgroup-per-query path (GPU box, repo root).
export GAMMA_FUZZ_SEEDS=70000:70500 GAMMA_LARGE_FUZZ_SEEDS=7000:7300 GAMMA_PLUGIN_FUZZ_SEEDS=17000:17200 GAMMA_RT_FUZZ_SEEDS=17000:17150
echo "## default paths: random_configuration 500, large_batch 300, plugin_script 200, realtime_script 150"
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k "random_configuration or large_batch or plugin_script or realtime_script" 2>&1 | tail -3
unset GAMMA_FUZZ_SEEDS GAMMA_PLUGIN_FUZZ_SEEDS GAMMA_RT_FUZZ_SEEDS
export GAMMA_LARGE_FUZZ_SEEDS=7300:7500
echo "## GAMMA_HIP_Q8_MINLEN=0 (list-major pass on short lists): large_batch 200"
GAMMA_HIP_Q8_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k large_batch 2>&1 | tail -3
echo "## GAMMA_HIP_PROD_C8=1 (one workgroup per query): large_batch 200"
GAMMA_HIP_PROD_C8=1 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k large_batch 2>&1 | tail -3
echo "## GAMMA_HIP_NO_C8=1 (fp32 table pass): large_batch 200"
GAMMA_HIP_NO_C8=1 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k large_batch 2>&1 | tail -3
