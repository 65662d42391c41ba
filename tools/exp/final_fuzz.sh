cd $GRAFT_REPO_ROOT
export GAMMA_FUZZ_SEEDS=95000:95400 GAMMA_LARGE_FUZZ_SEEDS=9500:9800 GAMMA_FLAT_FUZZ_SEEDS=19500:19800 GAMMA_IVFFLAT_FUZZ_SEEDS=19500:19800
export GAMMA_SHARD_FUZZ_SEEDS=19500:19800 GAMMA_GROUP_FUZZ_SEEDS=19500:19700 GAMMA_RT_FUZZ_SEEDS=19500:19600 GAMMA_PLUGIN_FUZZ_SEEDS=19500:19700
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 > gpurun_out/r06d_fuzz.txt 2>&1
unset GAMMA_FUZZ_SEEDS GAMMA_FLAT_FUZZ_SEEDS GAMMA_IVFFLAT_FUZZ_SEEDS GAMMA_SHARD_FUZZ_SEEDS GAMMA_GROUP_FUZZ_SEEDS GAMMA_RT_FUZZ_SEEDS GAMMA_PLUGIN_FUZZ_SEEDS
export GAMMA_LARGE_FUZZ_SEEDS=9800:10000
GAMMA_HIP_Q8_MINLEN=0 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k large_batch > gpurun_out/r06d_q8.txt 2>&1
GAMMA_HIP_PROD_C8=1 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k large_batch > gpurun_out/r06d_pc8.txt 2>&1
GAMMA_HIP_NO_C8=1 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k large_batch > gpurun_out/r06d_noc8.txt 2>&1
grep -a "passed\|failed" gpurun_out/r06d_*.txt
