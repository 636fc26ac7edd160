"""RCCL on hardware, one rank: bench.py's distributed leg (init_process_group("nccl") = RCCL, the all-gather on the
library's stream) driven in a FRESH child process that has not touched the GPU before the rendezvous — the only
multi-GPU readiness evidence obtainable on a one-GPU box (the 8-GPU runs are the driver's)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-extra"]
    r = subprocess.run(cmd, env=env, cwd=str(ROOT), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_under_rccl_world_size_one():
    plain = _bench({})
    dist = _bench({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                   "MASTER_PORT": str(_free_port()), "TORCHELASTIC_RUN_ID": "rccl-one-rank"})
    assert dist["n_gpus"] == 1 and "RCCL all-gather" in dist["config"]["step"]
    assert "RCCL all-gather" not in plain["config"]["step"]
    assert dist["value"] > 0.9 * plain["value"], (dist["value"], plain["value"])
