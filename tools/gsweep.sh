#!/bin/bash
for g in 1 2 4 8 16 32; do
  GAMMA_HIP_SCAN_G=$g python3 bench.py --steps 20 --warmup 3 --cpu-seconds 0 --recall-queries 0 2>&1 | grep "stage avg" | sed "s/^/G=$g /" | cut -c1-170
done
