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


def test_ranks_of_different_gpus_pin_to_different_cores(tmp_path):
    """bench.py keeps each rank on the cores next to its GPU (gpu_local_cpus: <sysfs>/bus/pci/devices/<bus id>/local_cpulist).  On a fake sysfs tree of an 8-GPU node -
    two sockets, four GPUs each, the kernel's cpulist format with SMT siblings in a second range - no two GPUs of different NUMA domains share a core, GPUs of one domain get the
    same list, malformed or missing entries give (None, None) instead of a wrong set, and a real rank process pins itself to exactly that set."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(helpers.ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11}
    for bad in ("", "3-1", "a-b", "1,,2"):
        try:
            bench.parse_cpulist(bad)
            raise AssertionError("accepted %r" % bad)
        except ValueError:
            pass
    root = tmp_path / "sys"
    lists = {}
    for g in range(8):
        bus = "0000:%02x:00.0" % (0x05 + 0x10 * g)
        dom = g // 2                                    # four NUMA domains of 32 cores (+ their SMT siblings), two GPUs each
        lists[bus] = "%d-%d,%d-%d" % (32 * dom, 32 * dom + 31, 128 + 32 * dom, 128 + 32 * dom + 31)
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "local_cpulist").write_text(lists[bus] + "\n")
    sets = {bus: bench.gpu_local_cpus(bus.upper(), str(root)) for bus in lists}      # (the runtime reports upper-case hex digits on some systems)
    for bus, (spec_str, cpus) in sets.items():
        assert spec_str == lists[bus] and len(cpus) == 64
    buses = sorted(lists)
    for i, a in enumerate(buses):
        for b in buses[i + 1:]:
            if lists[a] != lists[b]:
                assert not (sets[a][1] & sets[b][1]), (a, b)
    assert len({frozenset(v[1]) for v in sets.values()}) == 4
    assert bench.gpu_local_cpus("0000:ff:00.0", str(root)) == (None, None)
    (root / "bus" / "pci" / "devices" / "0000:05:00.0" / "local_cpulist").write_text("garbage\n")
    assert bench.gpu_local_cpus("0000:05:00.0", str(root)) == (None, None)

    # a rank process pins itself: a stand-in for the library that reports a bus id, a cpulist made of cores this process really has
    class FakeLib:
        def __init__(self, bus):
            self.bus = bus

        def rd_device_pci_bus_id(self, dev, buf, n):
            buf.value = self.bus.encode()
            return 0
    have = sorted(os.sched_getaffinity(0))
    if len(have) >= 4:
        mine = have[1:3]
        bus = "0000:15:00.0"
        (root / "bus" / "pci" / "devices" / bus / "local_cpulist").write_text("%d-%d\n" % (mine[0], mine[1]) if mine[1] == mine[0] + 1 else "%d,%d\n" % tuple(mine))
        before = os.sched_getaffinity(0)
        try:
            used = bench.pin_to_gpu_cores(FakeLib(bus), 0, str(root))
            assert used is not None and os.sched_getaffinity(0) == set(mine)
        finally:
            os.sched_setaffinity(0, before)
