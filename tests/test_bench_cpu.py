"""bench.py's launcher logic (no GPU needed: every check here runs before the first HIP call)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_n_refuses_loudly_when_fewer_devices_are_visible():
    """`--gpus 2` must never run silently as one rank: with fewer than 2 devices it exits 2 and says why."""
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 2" in r.stderr and "GPU(s)" in r.stderr
    assert r.stdout.strip() == ""                 # no JSON line


def test_world_size_must_match_gpus_flag():
    """Launched by torch.distributed.run with another world size than --gpus says: exit 2 before any GPU call."""
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0"], RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "WORLD_SIZE=1" in r.stderr
