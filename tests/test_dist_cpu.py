"""CPU: the multi-GPU orchestration (gamma_amd/dist.py) under world_size 2 with gloo."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_search_world2_gloo():
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout


def test_sub_batch_plan_keeps_results_contiguous():
    """plan_sub_batches: contiguous cover of [0, nq); every sub-batch but the last holds a multiple of the
    world size (its gathered rows carry no padding, so the result table stays one [nq, k] block)."""
    sys.path.insert(0, ROOT)
    from gamma_amd.dist import plan_sub_batches, query_slice
    for nq in (0, 1, 7, 33, 61, 4099, 16384, 65536):
        for world in (1, 2, 3, 8):
            for nsub in (1, 2, 3):
                plan = plan_sub_batches(nq, world, nsub)
                assert plan[0][0] == 0 and plan[-1][1] == nq
                assert all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
                assert all((e - s) % world == 0 and e > s for s, e in plan[:-1])
                row = 0
                for s, e in plan:       # row offset of a sub-batch in the gathered table == its first query
                    assert row == s or (s, e) == plan[-1] and row == s
                    per = max(1, -(-(e - s) // world))
                    covered = sum(query_slice(e - s, r, world)[1] - query_slice(e - s, r, world)[0] for r in range(world))
                    assert covered == e - s
                    row += world * per if (s, e) != plan[-1] else 0
