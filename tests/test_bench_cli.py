"""CPU-only: bench.py's launcher contract (no GPU needed for these paths): `--gpus N` never degrades to fewer ranks."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                           "BENCH_SHARE_GPU")}
    env.update(extra)
    return env


def test_more_ranks_than_devices_is_refused_without_a_line():
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(have + 3), "--steps", "2"], capture_output=True, text=True,
                         timeout=300, env=_clean_env(), cwd=ROOT)
    assert out.returncode == 2 and "refusing" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_world_size_that_contradicts_gpus_is_an_error():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2"], capture_output=True, text=True, timeout=300,
                         env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_zero_gpus_is_an_error():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "0"], capture_output=True, text=True, timeout=300, env=_clean_env(), cwd=ROOT)
    assert out.returncode != 0
