#!/usr/bin/env python3
"""bench.py - 1920x1080 frames/s of the rect/vidrect per-frame path on N MI355X (one synthetic stream per GPU).

One "step" = one batch of FRAMES_PER_STEP consecutive frames of a synthetic 1920x1080 vidrect stream (BASELINE.json
configs[4]: independent streams, one per GPU) pushed through the whole hot path: device stages on gfx950, read-back of
segments + probes, host post-process -> rectangle lists.  `value`: frames resident in HBM before the timed region starts (the
bench contract); `value_host_frames`: the same run with host buffers handed over, i.e. SURVEY.md 8(d)'s unit of work with the
PCIe upload inside the timed region.  `configs`: the other single-GPU configurations of BASELINE.json (1280x720 x 300 frames,
3840x2160 x 16 frames) measured by the same process, outside the timed region.

Prints ONE JSON line on rank 0.  Multi-GPU: launched by torch.distributed.run, one rank per GPU, no data-path
collective (frames are independent); only a barrier, a MAX-reduce of the elapsed time and a gather of a few bytes per rank,
over gloo (host side: no RCCL communicator and no extra stream on the device the detector is saturating).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

IW, IH = 1920, 1080
B_ALG_PER_PIXEL = 829          # algorithmic bytes per pixel per frame of the fused dataflow
HBM_PEAK = 8.0e12              # MI355X_MICROARCH.md: 8.0 TB/s spec
TAN_AOV = float(np.tan(72.0 / 2 / 180.0 * np.pi))


def _cpu_baseline_worker(args):
    """one process = one instance of the CPU path on `n` consecutive frames of stream `seed` (the first one untimed: page faults, LUTs)"""
    seed, n, use_ref = args
    from tests import helpers
    import rectdetect_amd as ra
    from rectdetect_amd import synth
    frames = [synth.frame(synth.SEED0 + seed, IW, IH, t) for t in range(n + 1)]
    if use_ref:
        r = helpers.RefRect(IW, IH)
        run = lambda f: r.execute_once(f, TAN_AOV)
    else:
        o = helpers.OracleRect(IW, IH)

        def run(f):
            o.frame(f)
            ra.postprocess_planes(o.segments(), o.plane("boundary"), o.plane("table"), IW, IH, TAN_AOV)
    run(frames[0])
    t0 = time.time()
    for f in frames[1:]:
        run(f)
    return time.time() - t0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline():
    """The reference itself (oracle/_ref: its kernels + host C on the serial OpenCL shim) when that build travelled with the repo, else
    our C restatement + the product's host post-process, on the GPU box's host cores: one thread (4 frames), and all streams a
    host would run side by side - one single-threaded instance per core on up to 32 cores, 2 frames each (frames of different
    streams are independent, SURVEY.md 8e: this is what `nproc` threads buy the CPU path).  A bounded sample, ~25 s."""
    import multiprocessing as mp
    from tests import helpers
    use_ref = helpers.have_ref()
    ncpu = len(os.sched_getaffinity(0))
    t1 = _cpu_baseline_worker((0, 4, use_ref))
    P = max(1, min(ncpu, 32))
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(P) as pool:
        per = pool.map(_cpu_baseline_worker, [(100 + k, 2, use_ref) for k in range(P)])
    wall = time.time() - t0
    many = sum(2 / t for t in per)        # aggregate rate while all P instances were running (start-up of the workers excluded)
    return {"value": round(many, 3), "unit": "frames/s", "cores": P, "kind": "reference" if use_ref else "port", "cpu_model": cpu_model(), "host_cores": ncpu,
            "single_thread": {"value": round(4 / t1, 4), "cores": 1, "sample": "4 consecutive 1920x1080 frames after one untimed, %.1f s" % t1},
            "sample": "%d single-threaded instances side by side, one per core, 2 consecutive 1920x1080 frames each after one untimed (%.1f s wall incl. start-up); "
                      "the reference's kernels run one work-item at a time on the serial OpenCL shim" % (P, wall)}


TRAFFIC_FILE = os.path.join("profiles", "r06_traffic.json")


def traffic_per_frame():
    """HBM-side bytes per frame from rocprofv3 PMC passes OF THIS COMMAND LINE's configuration (tools/gpu_pmc.sh: FETCH_SIZE and
    WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes; they cannot be collected from inside this process),
    committed under profiles/; null when that file is absent.  Returns (bytes, provenance)."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            j = json.load(f)
        return int(j["hbm_bytes_per_frame"]), {"file": TRAFFIC_FILE, "measured_at_commit": j.get("commit"), "command": j.get("command")}
    except (OSError, KeyError, ValueError):
        return None, None


def device_bus_id(L, dev):
    import ctypes
    buf = ctypes.create_string_buffer(64)
    try:
        return buf.value.decode().lower() if L.rd_device_pci_bus_id(dev, buf, 64) == 0 and buf.value else None
    except AttributeError:
        return None


def parse_cpulist(spec):
    """'0-15,128-143' -> {0..15, 128..143} (the kernel's cpulist format); ValueError on anything else"""
    cpus = set()
    for part in spec.strip().split(","):
        lo, _, hi = part.partition("-")
        lo, hi = int(lo), int(hi or lo)
        if lo < 0 or hi < lo:
            raise ValueError(spec)
        cpus.update(range(lo, hi + 1))
    return cpus


def gpu_local_cpus(bus_id, sysfs_root="/sys"):
    """the cores next to the GPU with PCI bus id `bus_id`: (the kernel's cpulist string, the set) from <sysfs>/bus/pci/devices/<bus id>/local_cpulist, or (None, None)"""
    try:
        with open(os.path.join(sysfs_root, "bus", "pci", "devices", bus_id.lower(), "local_cpulist")) as f:
            spec = f.read().strip()
        return spec, parse_cpulist(spec)
    except (OSError, ValueError):
        return None, None


def pin_to_gpu_cores(L, dev, sysfs_root="/sys"):
    """One process per GPU runs the enqueue / poll loop plus one post-process worker thread per frame slot; on the 8-GPU node (256 host cores, several NUMA nodes) keep
    them on the cores next to this rank's GPU.  Best effort: returns the cpulist used, or None when the topology is not visible (or leaves fewer than two of the cores
    this process may use)."""
    bus = device_bus_id(L, dev)
    if not bus:
        return None
    spec, cpus = gpu_local_cpus(bus, sysfs_root)
    if not cpus:
        return None
    cpus &= os.sched_getaffinity(0)
    if len(cpus) < 2:
        return None
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    return spec


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher (N > 1): start N ranks of this script - one process per GPU - through
    torch.distributed.run on 127.0.0.1, the way the driver's own multi-GPU command does, and hand their exit code back.  More ranks than
    the node has devices is an error, never a silent N = 1 (a dry run needs no device)."""
    import socket
    import subprocess
    if not args.dry_run:
        import rectdetect_amd as ra
        have = ra.lib().rd_device_count()
        if have < 1:
            raise SystemExit("bench.py: no HIP device - the product has no CPU fallback")
        if have < args.gpus and not args.share_gpus:      # (--share-gpus: tests on a one-GPU box, ranks share devices on purpose)
            raise SystemExit("bench.py: --gpus %d but this node has %d HIP device(s); one rank per GPU, no oversubscription" % (args.gpus, have))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def reference_opencl_baseline(timeout_s=150):
    """The reference ITSELF on this box's OpenCL device - on the GPU box that is the MI355X the HIP path runs on, through ROCm's OpenCL: oracle/_ref/librdref_ocl.so
    (the reference's unchanged host C linked against the system's OpenCL loader; its .cl sources are compiled at run time by the device's own compiler).  A same-box,
    same-device baseline beside the CPU one; checker-side code (oracle/), in a process of its own with a time limit - None where there is no OpenCL device or library."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "_ref", "librdref_ocl.so")
    if not os.path.exists(so):
        return None
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_on_opencl.py"), "bench"], cwd=ROOT, capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        return json.loads(lines[-1]) if p.returncode == 0 and lines else {"value": None, "error": (p.stderr or p.stdout)[-300:]}
    except Exception as e:      # (no device, no loader, time limit: a baseline that is missing must not take the line with it)
        return {"value": None, "error": str(e)[-300:]}


def side_config(ra, L, label, iw, ih, seed, nframes, slots, dev, min_seconds=1.5):
    """one of the other single-GPU configurations of BASELINE.json, the way the headline is measured: `nframes` consecutive frames of the
    synthetic stream resident in HBM, `slots` frames in flight, one untimed pass (graph capture, round budget), then whole passes over
    the stream until `min_seconds` have gone by.  Returns the line's `configs` entry."""
    from rectdetect_amd import synth
    N = iw * ih
    dframes = []
    a = np.zeros((ih, iw, 3), np.uint8)
    for t in range(nframes):
        L.rd_synth_frame(a.ctypes.data, iw, ih, iw * 3, synth.SEED0 + seed, t, 1)
        p = L.rd_device_alloc(a.nbytes)
        L.rd_upload(p, a.ctypes.data, a.nbytes)
        dframes.append(p)
    det = ra.Detector(iw, ih, device=dev, nslots=slots, nworkers=1)
    nrect = 0

    def run(npasses):
        """`npasses` times over the stream without a pause (the stream repeats: a camera that never stops), `slots` frames in flight throughout;
        everything is collected and drained before the clock is read"""
        nonlocal nrect
        inflight = 0
        for _ in range(npasses):
            for p in dframes:
                if inflight == slots:
                    nrect += len(det.poll(TAN_AOV)); inflight -= 1
                det.enqueue(p, ws=iw * 3, on_device=True)
                inflight += 1
        while inflight:
            nrect += len(det.poll(TAN_AOV)); inflight -= 1
        det.drain()

    run(max(1, (2 * slots + nframes - 1) // nframes))      # untimed: graph capture, round budget, polyline mode
    nrect = 0
    t0 = time.perf_counter()
    run(1)
    passes = max(2, int(min_seconds / max(1e-3, time.perf_counter() - t0)))
    nrect = 0
    t0 = time.perf_counter()
    run(passes)
    dt = time.perf_counter() - t0
    fps = passes * nframes / dt
    out = {"workload": label, "frame": "%dx%d" % (iw, ih), "stream_frames": nframes, "passes": passes, "frames_in_flight": slots, "frames_per_launch": det.frames_per_launch(), "value": round(fps, 2), "unit": "frames/s",
           "gpixel_per_s": round(fps * N / 1e9, 3), "roofline_frac": round(fps * B_ALG_PER_PIXEL * N / HBM_PEAK, 4), "rectangles_per_frame": round(nrect / (passes * nframes), 2),
           "frames_repeated": {"round_budget": det.region_round_budget()[1], "polyline_overflow": det.redone_frames(), "absorption_slow_path": det.absorption()[2]}}
    det.close()
    for p in dframes:
        L.rd_device_free(p)
    return out


def reference_api_config(ra, frames, dev, min_seconds=1.5):
    """The path the reference's own programs take, through the reference's own C API on host buffers (upload inside): vidrect.cpp:159-205 keeps TWO
    frames in flight with oclrect_enqueueTask / oclrect_pollTask; rect.cpp:105 calls oclrect_executeOnce per frame.  The headline `value` needs the
    rectdetect_hip.h detector with 64 frames in flight - an application that keeps the reference's call sequence gets these rates instead.  Each shape twice: frames in
    pageable memory (what cv::Mat hands over: copied by the caller's thread, as the reference does) and in page-locked memory from the reference's own allocatePinnedMemory
    (oclhelper.h; read in place by the copy engine, the call still returns only when the buffer may be reused)."""
    ctx = ra.Context(dev)
    det = ra.RectDetector(ctx, IW, IH)
    rows = []
    for memory in ("pageable", "pinned"):
        fr = frames[:16] if memory == "pageable" else [ctx.pinned_copy(f) for f in frames[:16]]
        n = len(fr)
        pinned0 = ra.lib().rd_detector_counter(_api_detector(det), 18)
        for k in range(8):      # (graph capture, round budget)
            det.execute_once(fr[k % n], TAN_AOV)
        lat, t_end, k = [], time.perf_counter() + min_seconds, 0
        while time.perf_counter() < t_end or k < 16:
            t0 = time.perf_counter()
            det.execute_once(fr[k % n], TAN_AOV)
            lat.append(time.perf_counter() - t0)
            k += 1
        lat.sort()
        once = {"workload": "oclrect_executeOnce per 1920x1080 frame, %s host buffers (rect.cpp:105)" % memory, "frame": "%dx%d" % (IW, IH), "frames": k, "frames_in_flight": 1, "host_memory": memory,
                "value": round(k / sum(lat), 2), "unit": "frames/s", "latency_ms_median": round(1e3 * lat[len(lat) // 2], 3), "latency_ms_p90": round(1e3 * lat[(len(lat) * 9) // 10], 3),
                "roofline_frac": round(k / sum(lat) * B_ALG_PER_PIXEL * IW * IH / HBM_PEAK, 4)}
        det.enqueue(fr[0])
        k, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < min_seconds or k < 32:
            det.enqueue(fr[(k + 1) % n])      # vidrect.cpp:159-172: the next frame is handed over, then the one before it is polled
            det.poll(TAN_AOV)
            k += 1
        dt = time.perf_counter() - t0
        det.poll(TAN_AOV)
        two = {"workload": "oclrect_enqueueTask / oclrect_pollTask, two 1920x1080 frames in flight, %s host buffers (what vidrect.cpp:159-205 gets)" % memory, "frame": "%dx%d" % (IW, IH), "host_memory": memory,
               "frames": k, "frames_in_flight": 2, "value": round(k / dt, 2), "unit": "frames/s", "roofline_frac": round(k / dt * B_ALG_PER_PIXEL * IW * IH / HBM_PEAK, 4)}
        took = ra.lib().rd_detector_counter(_api_detector(det), 18) - pinned0
        two["frames_read_in_place_by_the_copy_engine"] = once["frames_read_in_place_by_the_copy_engine"] = bool(took > 0)
        rows += [two, once]
        if memory == "pinned":
            for f in fr:
                ctx.free_pinned(f)
    det.close()
    ctx.close()
    return rows


def _api_detector(rect_detector):
    """the rd_detector behind an oclrect_t (struct oclrect_t { uint32_t magic; rd_detector *det; ... }: rd_api.hip) - for its counters"""
    import ctypes
    return ctypes.cast(rect_detector.h + 8, ctypes.POINTER(ctypes.c_void_p))[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=512, help="frames of the stream resident in HBM and handed over per step (512 x 6.2 MB = 3.2 GB; 20 steps then run about 3.5 s)")
    ap.add_argument("--slots", type=int, default=64, help="frames in flight per GPU (from three on: one stream per frame on four shared streams; from 6 / 12 / 32 on: groups of 2 / 4 / 8 frames per set of launches; 64 = two groups queued on each of the four streams, so that a stream never waits for the host to collect a group and hand over the next; 48 / 64 / 96 / 128 measured 2706-2717 / 2764-2782 / 2753-2754 / 2783-2810 frames/s on one box in round 5)")
    ap.add_argument("--host-frames", action="store_true", help="hand over host buffers (PCIe upload inside the timed region); not the headline value")
    ap.add_argument("--host-mode", default="pageable", choices=["pageable", "pinned"], help="with --host-frames (profiling runs): pageable buffers, copied by the caller's thread first, or pinned ones, read in place")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="profiling runs only: skip the sequential verification pass and the host-frames pass (the line then says outputs_verified: null)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercises sharding/aggregation only (CPU tests)")
    ap.add_argument("--no-configs", action="store_true", help="skip the side measurements of the 1280x720 and 3840x2160 configurations")
    ap.add_argument("--share-gpus", action="store_true", help="tests on a one-GPU box only: ranks beyond the device count share devices (rank mod count); without it more ranks than devices is an error")
    ap.add_argument("--frame", default=None, help="profiling runs only: another frame size than the headline's, e.g. 1280x720 or 3840x2160 (the line's metric then names that size); "
                                                  "the driver's line is always 1920x1080")
    ap.add_argument("--stream-seed", type=int, default=None, help="with --frame: seed offset of the synthetic stream (BASELINE.json configs[2]: 1, configs[3]: 4)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend of the control plane (default gloo: barrier + MAX-reduce of a double need no device)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.frame:
        global IW, IH
        IW, IH = (int(v) for v in args.frame.lower().split("x"))
        args.no_configs = True
        if IW * IH > 1920 * 1088:
            args.slots = min(args.slots, 16)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))      # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE); refusing to label one as the other" % (args.gpus, world))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = args.backend or "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)

    F = args.frames_per_step
    seed_stream = rank + (args.stream_seed or 0)      # stream id = rank: independent 1080p streams, one per GPU (BASELINE.json configs[4])

    import rectdetect_amd as ra
    from rectdetect_amd import synth

    if args.dry_run:
        frames, det, dframes = None, None, None
    else:
        L = ra.lib()
        if L.rd_device_count() <= 0:
            raise SystemExit("bench.py: no HIP device - the product has no CPU fallback")
        frames = []
        for t in range(F):
            a = np.zeros((IH, IW, 3), np.uint8)
            L.rd_synth_frame(a.ctypes.data, IW, IH, IW * 3, synth.SEED0 + seed_stream, t, 1)
            frames.append(a)
        ndev = L.rd_device_count()
        if local >= ndev and not args.share_gpus:
            raise SystemExit("bench.py: rank %d has no GPU of its own (%d device(s) on this node); one rank per GPU" % (rank, ndev))
        dev = local % ndev                    # identity on a full node (--share-gpus: two ranks on the only GPU of a test box)
        pinned = pin_to_gpu_cores(L, dev)
        det = ra.Detector(IW, IH, device=dev, nslots=args.slots, nworkers=1)
        dframes = []
        for a in frames:
            p = L.rd_device_alloc(a.nbytes)
            L.rd_upload(p, a.ctypes.data, a.nbytes)
            dframes.append(p)

    import zlib
    pframes = []                 # the frames once more in pinned host memory (allocated for the host-frames pass only)
    if det is not None and args.host_frames and args.host_mode == "pinned":
        import ctypes
        for a in frames:
            p = L.rd_host_alloc(a.nbytes)
            ctypes.memmove(p, a.ctypes.data, a.nbytes)
            pframes.append(p)
    results = []                 # rectangle lists in stream order (numpy arrays returned by poll)
    inflight = 0
    use_host = [("pinned" if args.host_mode == "pinned" else True) if args.host_frames else False]

    def step(d=None, slots=None):
        """one pass over the batch of F frames: every frame is handed to the detector; it is a stream, so the results of the
        last frames in flight are collected at the beginning of the next step (or by sync() at the end of the timed region)"""
        nonlocal inflight
        if args.dry_run:
            time.sleep(0.01 * (1 + rank))
            return
        d = d or det
        slots = slots or args.slots
        for i in range(F):
            if inflight == slots:
                results.append(d.poll(TAN_AOV))
                inflight -= 1
            if use_host[0] == "pinned":
                d.enqueue(pframes[i], ws=IW * 3, pinned=True)
            elif use_host[0]:
                d.enqueue(frames[i])
            else:
                d.enqueue(dframes[i], ws=IW * 3, on_device=True)
            inflight += 1

    def sync(d=None):
        """all frames handed over so far are finished, post-processed and their rectangles collected"""
        nonlocal inflight
        if not args.dry_run:
            d = d or det
            while inflight:
                results.append(d.poll(TAN_AOV))
                inflight -= 1
            d.drain()
            if dist is not None:
                import torch
                if torch.cuda.is_available():      # (the contract's torch.cuda.synchronize(): the detector's own drain() above has already waited for its streams)
                    torch.cuda.synchronize(dev)

    def timed(nsteps, d=None, slots=None):
        t0 = time.perf_counter()
        for _ in range(nsteps):
            step(d, slots)
        sync(d)
        return time.perf_counter() - t0

    for _ in range(args.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    dev0 = det.device_time() if det is not None else (0, 0)
    enq0 = ra.lib().rd_detector_counter(det.h, 3) if det is not None else 0
    post0 = [ra.lib().rd_detector_counter(det.h, k) for k in (11, 12, 13)] if det is not None else [0, 0, 0]
    elapsed = timed(args.steps)
    post1 = [ra.lib().rd_detector_counter(det.h, k) for k in (11, 12, 13)] if det is not None else [0, 0, 0]
    own_elapsed = elapsed
    dev1 = det.device_time() if det is not None else (0, 0)
    enq1 = ra.lib().rd_detector_counter(det.h, 3) if det is not None else 0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    # (1) EVERY rank checks its timed run's outputs against a plain sequential pass over its own stream (one frame in flight, no worker threads,
    #     post-process inline), outside the timed region: a run that returned garbage on any GPU cannot print a valid line
    digest = lambda lists: zlib.crc32(b"".join(np.ascontiguousarray(r).tobytes() for r in lists)) & 0xFFFFFFFF
    verify = None
    rho = None
    if det is not None and not args.no_verify:
        timed_lists = results[args.warmup * F:]
        nrect = int(sum(len(r) for r in timed_lists))
        chk = ra.Detector(IW, IH, device=dev, nslots=1, nworkers=0)
        seq = []
        for k in range((args.warmup + args.steps) * F):
            chk.enqueue(dframes[k % F], ws=IW * 3, on_device=True)
            seq.append(chk.poll(TAN_AOV))
        ctr = chk.plane("polyctr", np.int32, 64)
        rho = {"chain_pixels": int(ctr[0]), "chains": int(ctr[1]), "live_pixels": int(ctr[24]), "edge_density": round(float(ctr[0]) / (IW * IH), 5)}
        chk.close()
        seq_digest = digest(seq[args.warmup * F:])
        verify = {"outputs_verified": bool(len(timed_lists) == args.steps * F and digest(timed_lists) == seq_digest),
                  "rectangles_in_timed_frames": nrect, "rect_list_crc32": "%08x" % digest(timed_lists),
                  "against": "sequential pass of the same %d-frame stream, 1 frame in flight, no worker threads, outside the timed region" % len(seq)}

    # who did what: every rank reports its device, stream seed, frames, time and the outcome of its own check (host-side gather of a few bytes, not a data-path collective)
    mine = {"rank": rank, "outputs_verified": verify["outputs_verified"] if verify else None, "rect_list_crc32": verify["rect_list_crc32"] if verify else None, "device": None if args.dry_run else dev, "stream_seed": seed_stream, "frames": args.steps * F, "frames_per_s": round(args.steps * F / own_elapsed, 2),
            "rectangles": int(sum(len(r) for r in results[args.warmup * F:])), "own_elapsed_s": round(own_elapsed, 4), "pinned_cpus": None if args.dry_run else pinned,
            "pci_bus_id": None if args.dry_run else device_bus_id(ra.lib(), dev), "pid": os.getpid()}
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        devs = [r["pci_bus_id"] or r["device"] for r in per_rank]
        if not args.dry_run and not args.share_gpus and len(set(devs)) != world:
            raise SystemExit("bench.py: %d ranks on %d distinct devices (%s) - not a %d-GPU measurement" % (world, len(set(devs)), devs, world))
        total_frames = world * args.steps * F
        fps = total_frames / elapsed
        N = IW * IH
        achieved = fps / world * B_ALG_PER_PIXEL * N      # bytes/s per GPU
        host_rate = None
        if verify is not None:
            if world > 1:      # the line is valid only if every rank's lists passed its own check
                verify = dict(verify, outputs_verified=all(r["outputs_verified"] is True for r in per_rank), rectangles_in_timed_frames=int(sum(r["rectangles"] for r in per_rank)),
                              against=verify["against"] + " - on every rank, each against its own stream")
            # (2) the same work with host buffers handed over (memcpy into pinned memory + PCIe upload inside the timed region):
            #     SURVEY.md 8(d)'s "upload -> ... -> post-process" unit; reported beside the HBM-resident headline, never as `value`
            #     - from the caller's PINNED memory, read in place by the copy engine (rd_detector_enqueue(..., RD_FRAME_HOST_PINNED): what a capture loop that allocates its
            #       frames with allocatePinnedMemory / rd_host_alloc gets), and from pageable memory, copied by the caller's thread into the detector's staging pages first
            #       (the reference's own oclrect.c:1256).  `value_host_frames` is the pinned one; both are printed, and the lists of both passes are checked.
            if world == 1 and not args.host_frames:
                import ctypes
                host_passes = {}
                for mode in ("pinned", "pageable"):
                    if mode == "pinned":
                        for a in frames:
                            p = L.rd_host_alloc(a.nbytes)
                            ctypes.memmove(p, a.ctypes.data, a.nbytes)
                            pframes.append(p)
                    results.clear()
                    use_host[0] = "pinned" if mode == "pinned" else True
                    c0 = [ra.lib().rd_detector_counter(det.h, k) for k in (18, 19)]
                    timed(1)
                    th = timed(args.steps)
                    c1 = [ra.lib().rd_detector_counter(det.h, k) for k in (18, 19)]
                    use_host[0] = False
                    rate = round(args.steps * F / th, 2)
                    host_passes[mode] = {"value": rate, "unit": "frames/s", "roofline_frac": round(rate * B_ALG_PER_PIXEL * N / HBM_PEAK, 4),
                                         "frames_read_in_place_by_the_copy_engine": c1[0] - c0[0], "frames_copied_by_the_callers_thread_first": c1[1] - c0[1],
                                         "outputs_verified": bool(len(results) == (1 + args.steps) * F and digest(results[F:]) == seq_digest)}
                    if mode == "pinned":
                        for p in pframes:
                            L.rd_host_free(p)
                        pframes.clear()
                host_rate = host_passes["pinned"]["value"]
                verify = dict(verify, outputs_verified=bool(verify["outputs_verified"] and all(h["outputs_verified"] for h in host_passes.values())))
        traffic, traffic_src = traffic_per_frame()
        out = {
            "metric": "%dx%d frames/sec" % (IW, IH), "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32/f32 (bit-exact integer + IEEE f32 stencil path)", "data": "synthetic",
            "config": {"workload": "vidrect 1920x1080 synthetic stream per GPU (BASELINE.json configs[4]; configs[1] is one frame of it)" if not args.frame else "vidrect %dx%d synthetic stream (profiling run, --frame)" % (IW, IH),
                       "frames_per_step": F, "frames_in_flight": args.slots, "frames_per_launch": None if args.dry_run else det.frames_per_launch(), "input": "BGR u8 host buffers (PCIe upload timed)" if args.host_frames else "BGR u8 frames resident in HBM", "parallelism": "independent streams, one per GPU, no collective"},
            "ranks": per_rank,
            "value_definition": "frames resident in HBM when the timed region starts (bench contract); value_host_frames = the same work with host BGR buffers handed over, "
                                "PCIe upload inside the timed region (SURVEY.md 8(d)'s unit of work), same process, N=1 only: from the caller's pinned memory (read in place by the copy engine); "
                                "host_frames.pageable = from pageable memory, copied by the caller's thread into pinned staging pages first, as the reference does",
            "value_host_frames": host_rate,
            "roofline_frac_host_frames": round(host_rate * B_ALG_PER_PIXEL * N / HBM_PEAK, 4) if host_rate else None,
            "host_frames": None if not host_rate else dict(host_passes, value_host_frames_is="pinned"),
            "roofline": {"bound": "hbm", "kernel": "whole per-frame device pipeline (all stages, one stream per frame slot)",
                         "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                         "algorithmic_bytes_per_frame": B_ALG_PER_PIXEL * N,
                         # how to read `frac`: SURVEY.md 8(d) prices the polyline stages S12-S18 as dense sweeps (697 of the 829 B/px); this
                         # implementation runs them on compacted chain pixels (edge density below), so the 132 B/px of the stages it does run
                         # densely are the part of the contract figure that costs HBM time here
                         "algorithmic_bytes_dense_stages": 132 * N, "algorithmic_bytes_polyline_stages_priced_dense": 697 * N,
                         "frac_dense_stages_only": round(fps / world * 132 * N / HBM_PEAK, 4), "stream_statistics": rho,
                         "traffic": traffic, "traffic_unit": "bytes/frame", "traffic_source": traffic_src,
                         # "rocprof-reported achieved HBM GB/s against the gfx950 peak" (north_star): counter bytes per frame x this run's frames/s (per GPU); 6.29 TB/s = what a
                         # streaming copy reaches on this part (MI355X_MICROARCH.md)
                         "hbm_measured": None if not traffic else {"GB_per_s": round(traffic * fps / world / 1e9, 1), "frac_of_peak_8.0TBps": round(traffic * fps / world / HBM_PEAK, 4),
                                                                   "frac_of_achievable_6.29TBps": round(traffic * fps / world / 6.29e12, 4)},
                         # HIP events on each frame's own stream over the timed region (rank 0): first kernel start -> last copy end.
                         # With several frames in flight these intervals overlap, which is why `achieved` is taken from the aggregate rate.
                         "frame_device_us_avg": round((dev1[0] - dev0[0]) / max(1, dev1[1] - dev0[1]), 1), "frames_in_flight": args.slots, "frames_per_launch": None if args.dry_run else det.frames_per_launch(),
                         "host_enqueue_us_avg": round((enq1 - enq0) / max(1, dev1[1] - dev0[1]), 1),
                         # rectangles from segments + probes: on the host's worker threads (one per frame slot, CPU time per frame) or - RD_DEVICE_POST=1 - on the device
                         "postprocess": {"frames_on_device": post1[0] - post0[0], "frames_on_host": post1[1] - post0[1],
                                         "host_cpu_us_per_frame": round((post1[2] - post0[2]) / max(1, post1[1] - post0[1]), 1)},
                         "region_round_budget": det.region_round_budget()[0] if det is not None else None,
                         "frames_per_launch_need": {str(k): ra.lib().rd_detector_counter(det.h, 40 + k) for k in range(21) if ra.lib().rd_detector_counter(det.h, 40 + k)} if det is not None else None,
                         "frames_per_launch_budget": {str(8 + 2 * k): ra.lib().rd_detector_counter(det.h, 20 + k) for k in range(7)} if det is not None else None,
                         "frames_repeated": {"round_budget": det.region_round_budget()[1], "polyline_overflow": det.redone_frames(), "absorption_slow_path": det.absorption()[2]} if det is not None else None},
        }
        if det is not None and world == 1 and not args.no_configs and not args.no_verify:
            # BASELINE.json configs[2] and configs[3] (configs[1], the 1920x1080 still, is a frame of the headline stream), outside the timed region - each
            # configuration alone on the device, as it would run: the headline's detector (its 64 slots of planes, its worker threads) is closed first
            det.close()
            det = None
            for p in dframes:
                L.rd_device_free(p)
            dframes = []
            out["configs"] = [side_config(ra, L, "vidrect 1280x720 synthetic 300-frame stream (BASELINE.json configs[2])", 1280, 720, 1, 300, args.slots, dev),
                              side_config(ra, L, "vidrect 3840x2160 synthetic stream, 16 frames resident (BASELINE.json configs[3])", 3840, 2160, 4, 16, min(args.slots, 16), dev)]
            out["configs"] += reference_api_config(ra, frames, dev)
        out.update(verify or {"outputs_verified": None})
        if not args.dry_run and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
            out["reference_opencl_same_gpu"] = reference_opencl_baseline()
        if args.dry_run:
            out["dry_run"] = True
        print(json.dumps(out), flush=True)
        if verify and not verify["outputs_verified"]:
            raise SystemExit("bench.py: the timed run's rectangle lists differ from the sequential pass - the line above is INVALID")

    if dist is not None:
        dist.barrier()       # (rank 0 verifies its outputs while the others wait)
    if det is not None:
        det.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
