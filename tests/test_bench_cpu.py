"""bench.py's launch contract on the CPU: `python bench.py --gpus N` without a launcher must start N ranks itself
(torch.distributed.run on 127.0.0.1) before touching the GPU, and rank 0 must print ONE JSON line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_n_without_launcher_spawns_n_ranks():
    out = _run(["--gpus", "2", "--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    assert doc == {"dry_run": True, "world": 2, "ranks_seen": 2}


def test_single_gpu_default_needs_no_launcher():
    out = _run(["--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(out.stdout.strip().splitlines()[-1]) == {"dry_run": True, "world": 1, "ranks_seen": 1}


def test_mismatched_world_size_fails_loudly():
    out = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in (out.stderr + out.stdout)


def test_failing_rank_propagates_a_nonzero_exit():
    out = _run(["--gpus", "2", "--dry-run", "--steps", "not-a-number"])
    assert out.returncode != 0
