#!/bin/bash
# usage (GPU box): tools/ab.sh "<ENV=..>" ["<ENV2=..>" ...]   -- quick A/B of bench stage times
for e in "$@"; do
  echo "== $e"
  env $e python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 --recall-queries 0 --no-extra 2>&1 | grep -E "stage avg|value" | sed -e 's/"config".*"roofline"/"roofline"/' | cut -c1-420
done
