for mode in 1 2; do
export GAMMA_HIP_C8=$mode
timeout 1200 python -m pytest tests/test_gpu_more.py tests/test_gpu_ties.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -x -q -n 4 -k "scan_bound or c3_headline or bounded_scan or large_batch or ivfpq_exact_ties or cut_ties or c4_shape or list_length_regime_m32" 2>&1 | tail -3
done
unset GAMMA_HIP_C8
bash tools/exp/c8_ab.sh "GAMMA_HIP_SCAN_G=4 GAMMA_HIP_C8=1" "GAMMA_HIP_SCAN_G=3 GAMMA_HIP_C8=1" "GAMMA_HIP_SCAN_G=2 GAMMA_HIP_C8=1" "GAMMA_HIP_SCAN_G=5 GAMMA_HIP_C8=1"
