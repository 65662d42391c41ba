# round 6: randomised parity campaign with fresh seeds over every fuzz family, on the final tree (GPU box, repo root).
# The list-shard family runs the TWO-PHASE shard scan (tests/shard_emul.py default), once more with one producer probe
# (GAMMA_HIP_SHARD_G1=1) and once single-phase; the in-process group runs it with the barrier + peer-copy reduction.
export GAMMA_FUZZ_SEEDS=60000:60500 GAMMA_LARGE_FUZZ_SEEDS=6000:6200 GAMMA_FLAT_FUZZ_SEEDS=16000:16300 GAMMA_IVFFLAT_FUZZ_SEEDS=16000:16300
export GAMMA_SHARD_FUZZ_SEEDS=16000:16600 GAMMA_GROUP_FUZZ_SEEDS=16000:16300 GAMMA_RT_FUZZ_SEEDS=16000:16200 GAMMA_PLUGIN_FUZZ_SEEDS=16000:16200
timeout 2400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 2>&1 | tail -4
unset GAMMA_FUZZ_SEEDS GAMMA_LARGE_FUZZ_SEEDS GAMMA_FLAT_FUZZ_SEEDS GAMMA_IVFFLAT_FUZZ_SEEDS GAMMA_GROUP_FUZZ_SEEDS GAMMA_RT_FUZZ_SEEDS GAMMA_PLUGIN_FUZZ_SEEDS
export GAMMA_SHARD_FUZZ_SEEDS=17000:17400
GAMMA_HIP_SHARD_G1=1 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k list_shard 2>&1 | tail -3
GAMMA_TEST_TWO_PHASE=0 timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -n 5 -k list_shard 2>&1 | tail -3
