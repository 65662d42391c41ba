# round 6, third campaign: after the 64-item wave sorts moved to DPP stages (select.hip: every selection kernel), fresh seeds over every
# family, then the whole GPU suite (GPU box, repo root); summaries under gpurun_out/.
export GAMMA_FUZZ_SEEDS=80000:80300 GAMMA_LARGE_FUZZ_SEEDS=8000:8200 GAMMA_FLAT_FUZZ_SEEDS=18000:18300 GAMMA_IVFFLAT_FUZZ_SEEDS=18000:18300
export GAMMA_SHARD_FUZZ_SEEDS=18000:18300 GAMMA_GROUP_FUZZ_SEEDS=18000:18200 GAMMA_RT_FUZZ_SEEDS=18000:18100 GAMMA_PLUGIN_FUZZ_SEEDS=18000:18150
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 2>&1 | tail -3 > gpurun_out/r06c_fuzz.txt
unset GAMMA_FUZZ_SEEDS GAMMA_LARGE_FUZZ_SEEDS GAMMA_FLAT_FUZZ_SEEDS GAMMA_IVFFLAT_FUZZ_SEEDS GAMMA_SHARD_FUZZ_SEEDS GAMMA_GROUP_FUZZ_SEEDS GAMMA_RT_FUZZ_SEEDS GAMMA_PLUGIN_FUZZ_SEEDS
timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r06c_suite.txt
cat gpurun_out/r06c_fuzz.txt gpurun_out/r06c_suite.txt
