"""usage: shard_fuzz_debug.py <seed> [nobudget] -- one configuration of tests/test_gpu_fuzz.py::test_random_list_shard_configuration,
with the budget decision printed and optionally disabled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tests.test_gpu_fuzz as T
from gamma_amd import api
seed = int(sys.argv[1])
nobudget = len(sys.argv) > 2 and sys.argv[2] == "nobudget"
orig = api.GammaHip.set_dist_budget
def patched(self, n):
    print("set_dist_budget", n, "(skipped)" if nobudget else "")
    if not nobudget:
        orig(self, n)
api.GammaHip.set_dist_budget = patched
try:
    T.test_random_list_shard_configuration(seed)
    print("seed", seed, "PASS")
except AssertionError as e:
    print("seed", seed, "FAIL", str(e)[:300])
