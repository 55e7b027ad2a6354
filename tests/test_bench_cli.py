"""bench.py's command line: the N > 1 bench must never silently fall back to the one-GPU workload."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_gpus_n_without_n_devices_fails_loudly():
    # (this container has no GPU, the GPU boxes have one: --gpus 64 cannot be satisfied on either)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "device" in r.stderr
    assert r.stdout.strip() == "", "no JSON line may be printed for a run that did not happen"


def test_gpus_must_match_world_size():
    env = dict(__import__("os").environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    assert r.stdout.strip() == ""
