"""bench.py plumbing that needs no GPU: `--gpus N` starts its own ranks (a child torch.distributed.run) before the
process touches the GPU, and a rank without a GPU refuses to run (no CPU fallback)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_without_a_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)


def test_bench_gpus_n_launches_its_own_ranks():
    """No RANK/WORLD_SIZE in the environment: the parent must spawn 2 ranks; here both fail ("needs a GPU"), which
    shows they were started with rank environments, and the parent passes the failure code on."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert (r.stderr + r.stdout).count("needs a GPU") >= 2


def test_csrc_fingerprint_is_stable():
    sys.path.insert(0, ROOT)
    import bench
    a, b = bench.csrc_fingerprint(), bench.csrc_fingerprint()
    assert a == b and len(a) == 16
