"""The plugins' threaded host code under the sanitizers the CPU build has: tests/sanitize/ builds the
RetrievalModel plugins (gamma_amd/host) + the harness on a CPU stub of the C ABI (stub_abi.cpp, backed by the oracle) with
-fsanitize=thread and with -fsanitize=address,undefined, and plugin_stress.cc drives the engine's threading contract
(SURVEY 8b): four client threads searching, the indexing thread growing the store / Add / Update, an API thread deleting.
Any report fails the test.  stress_group_tsan links the REAL csrc/gamma_hip_group.cpp (member threads, barriers, snapshots;
replicate placement) over three stub members, its HIP runtime calls answered by tests/sanitize/fakehip (host memory, copies on
the calling thread).  (The combining queue and the store locks of the device library live in translation units full of kernel
launches: they are exercised by the GPU suites, tests/test_gpu_concurrent.py; GPU sanitizers do not exist on this pool.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = os.path.join(ROOT, "tests", "sanitize")


@pytest.fixture(scope="module")
def built():
    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run(["make", "-s", "-C", SAN], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return os.path.join(SAN, "_build")


@pytest.mark.parametrize("binary,marks", [("stress_tsan", ("ThreadSanitizer",)),
                                          ("stress_asan", ("AddressSanitizer", "runtime error", "LeakSanitizer")),
                                          # the REAL in-process group (csrc/gamma_hip_group.cpp: a thread per member, barriers, go / no-go
                                          # snapshots) behind the plugin's "devices" key, on three stub members (fakehip/: its HIP calls)
                                          ("stress_group_tsan", ("ThreadSanitizer",))])
def test_plugins_under_sanitizer(built, binary, marks):
    env = dict(os.environ, OMP_NUM_THREADS="1", TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1",
               ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([os.path.join(built, binary)], capture_output=True, text=True, timeout=600, env=env)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    for m in marks:
        assert m not in out, out[-6000:]
    assert "HIPIVFPQ:" in out and " 0 failures" in out and " 0 searches" not in out
    if "group" not in binary:
        assert "HIPFLAT:" in out
