import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("MKL_THREADING_LAYER", "GNU")
# passive waiting: the oracle's k-means has thousands of short parallel regions; spinning at their barriers under a
# container CPU quota (the GPU boxes: 256 hardware threads, 16 cores' worth of quota) is pathologically slow
os.environ.setdefault("OMP_WAIT_POLICY", "passive")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
