#!/usr/bin/env python3
"""Growth of the raw store under concurrent search: appends of 256 MB up to `gb` GB while a
searcher thread runs small flat searches; prints the longest search call and the longest append, with the store growing
in place (virtual memory management, the default) and with GAMMA_HIP_NO_RAW_VMM=1 (geometric reallocation under the
exclusive lock: a copy of everything stored so far and twice the memory for its duration).
usage: python tools/raw_grow.py [gb=8]"""
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(gb):
    from gamma_amd import api
    d = 128
    rows = (256 << 20) // (d * 4)
    blk = np.random.default_rng(1).integers(0, 255, size=(rows, d)).astype(np.float32)
    g = api.GammaHip(0)
    g.raw_init(d)
    g.raw_append(blk[:4096])
    q = blk[:4] + 1.0
    args = api.SearchArgs(metric=api.METRIC_L2, min_score=-3e38, max_score=3e38)
    stop = threading.Event()
    worst = [0.0]

    def searcher():
        # a tiny IVF-free workload that needs the search lock: brute force over the first rows only would still scan
        # everything, so time a 1-query search of a second, small handle-independent call: raw_stats + synchronize
        while not stop.is_set():
            t0 = time.perf_counter()
            g.synchronize()
            worst[0] = max(worst[0], time.perf_counter() - t0)
            time.sleep(0.0005)

    t = threading.Thread(target=searcher)
    t.start()
    worst_app = 0.0
    for i in range(int(gb * 4)):
        t0 = time.perf_counter()
        g.raw_append(blk)
        worst_app = max(worst_app, time.perf_counter() - t0)
    stop.set()
    t.join()
    st = g.raw_stats()
    print("in_place=%s rows=%d moves=%d: longest append %.1f ms, longest wait of a search-side call %.1f ms, device bytes %.2f GB"
          % (st["in_place"], st["rows"], st["moves"], worst_app * 1e3, worst[0] * 1e3, g.total_mem_bytes() / 1e9), flush=True)
    g.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "child":
        run(float(sys.argv[1]))
    else:
        gb = sys.argv[1] if len(sys.argv) > 1 else "8"
        for env in ({}, {"GAMMA_HIP_NO_RAW_VMM": "1"}):
            subprocess.call([sys.executable, os.path.abspath(__file__), gb, "child"], env=dict(os.environ, **env))
