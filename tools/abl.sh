#!/bin/bash
# ablation timing of the qscan kernel: GAMMA_HIP_QSCAN_DBG bit flags
for f in 0 1 2 4 8 3 7 15; do
  GAMMA_HIP_QSCAN_DBG=$f python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --recall-queries 0 2>&1 | grep "stage avg" | sed "s/^/dbg=$f /" | cut -c1-160
done
