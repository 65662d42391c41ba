# round 6, third campaign: after the 64-item wave sorts moved to DPP stages (select.hip: every selection kernel) and the re-rank
# kernel's top-(k + 1) path, fresh seeds over every family, then the whole GPU suite (GPU box, repo root); summaries under gpurun_out/.
export GAMMA_FUZZ_SEEDS=${S0:-80000}:$((${S0:-80000} + 300)) GAMMA_LARGE_FUZZ_SEEDS=${S1:-8000}:$((${S1:-8000} + 200)) GAMMA_FLAT_FUZZ_SEEDS=${S2:-18000}:$((${S2:-18000} + 300)) GAMMA_IVFFLAT_FUZZ_SEEDS=${S2:-18000}:$((${S2:-18000} + 300))
export GAMMA_SHARD_FUZZ_SEEDS=${S2:-18000}:$((${S2:-18000} + 300)) GAMMA_GROUP_FUZZ_SEEDS=${S2:-18000}:$((${S2:-18000} + 200)) GAMMA_RT_FUZZ_SEEDS=${S2:-18000}:$((${S2:-18000} + 100)) GAMMA_PLUGIN_FUZZ_SEEDS=${S2:-18000}:$((${S2:-18000} + 150))
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 > gpurun_out/r06c_fuzz_full.txt 2>&1
unset GAMMA_FUZZ_SEEDS GAMMA_LARGE_FUZZ_SEEDS GAMMA_FLAT_FUZZ_SEEDS GAMMA_IVFFLAT_FUZZ_SEEDS GAMMA_SHARD_FUZZ_SEEDS GAMMA_GROUP_FUZZ_SEEDS GAMMA_RT_FUZZ_SEEDS GAMMA_PLUGIN_FUZZ_SEEDS
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r06c_suite_full.txt 2>&1
grep -a "passed\|failed" gpurun_out/r06c_fuzz_full.txt gpurun_out/r06c_suite_full.txt
