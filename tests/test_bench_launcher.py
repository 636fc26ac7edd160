"""`python bench.py --gpus N` must use N ranks by itself (the driver runs it plainly): with N > 1 and no launcher environment it
starts N fresh child ranks before touching a GPU, relays rank 0's line and fails when a rank fails. Exercised here on CPU with
the stand-in step over gloo (`--stand-in`: the launcher, rendezvous, barrier, max-over-ranks timing and all-gather plumbing of
the real path; nothing of the product runs)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=timeout)


def test_gpus_2_starts_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stand-in"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # ONE line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["parallelism"].endswith("x2")


def test_a_failing_rank_fails_the_run():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--stand-in"], {"BENCH_STANDIN_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "rank 1 of 2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_world_size_must_match_gpus():
    """under a launcher (RANK / WORLD_SIZE in the environment) a mismatch between --gpus and the ranks the group has is an
    error, not a silently wrong n_gpus"""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--stand-in"],
             {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29517"})
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout)


def test_single_rank_stand_in():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--stand-in"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1
