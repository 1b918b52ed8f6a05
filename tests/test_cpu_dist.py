"""N > 1 plumbing of bench.py on CPU: two gloo ranks, independent streams, barrier + MAX-reduced time, no data-path
collective (the path shards by stream: SURVEY.md 8e)."""
import json
import os
import socket
import subprocess
import sys

from tests import helpers


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_dry_run():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames-per-step", "32", "--dry-run", "--backend", "gloo"]
    p = subprocess.run(cmd, cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout           # only rank 0 prints
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["steps"] == 3
    # the slower rank (rank 1 sleeps 20 ms per step) sets the time: MAX over ranks
    assert out["ms_per_step"] >= 19.0
    assert abs(out["value"] - 2 * 3 * 32 / (out["ms_per_step"] * 3 / 1e3)) / out["value"] < 0.02


def test_plain_gpus_2_spawns_two_ranks_itself():
    """the driver's SCALE command shape is `python bench.py --gpus N ...` with no launcher around it: the script must start N ranks
    itself (one process per GPU) and print n_gpus = N with N ranks listed - never a single rank under an N-GPU label"""
    cmd = [sys.executable, os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", "16", "--dry-run"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(cmd, cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and len(out["ranks"]) == 2
    assert sorted(r["rank"] for r in out["ranks"]) == [0, 1] and out["ranks"][0]["pid"] != out["ranks"][1]["pid"]


def test_world_size_must_match_gpus():
    """a launcher that starts another number of ranks than --gpus says is refused (the line's n_gpus would label the wrong thing)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(helpers.ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0", "--frames-per-step", "8", "--dry-run"]
    p = subprocess.run(cmd, cwd=helpers.ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
