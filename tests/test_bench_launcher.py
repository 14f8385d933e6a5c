"""bench.py --gpus N starts its own ranks: the launcher, the rendezvous on 127.0.0.1,
the barrier and the max-over-ranks reduction, exercised on CPU with --dry-run (gloo, no alignment)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_launcher_spawns_n_ranks_and_rank0_prints_one_line():
    r = _run(["--gpus", "3", "--dry-run"])
    assert r["n_gpus"] == 3 and r["dry_run"] is True
    assert r["max_over_ranks"] == 3.0  # every rank contributed rank + 1


def test_single_rank_spawns_nothing():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r["n_gpus"] == 1 and r["max_over_ranks"] == 1.0


def test_under_an_external_launcher_the_process_is_a_rank():
    # WORLD_SIZE already set (torch.distributed.run): no children, the environment's world size is reported
    r = _run(["--gpus", "1", "--dry-run"], env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                                                      "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert r["n_gpus"] == 1
