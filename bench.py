#!/usr/bin/env python3
"""bench.py - 1920x1080 frames/s of the rect/vidrect per-frame path on N MI355X (one synthetic stream per GPU).

One "step" = one batch of FRAMES_PER_STEP consecutive frames of a synthetic 1920x1080 vidrect stream (BASELINE.json
configs[4]: independent streams, one per GPU) pushed through the whole hot path: device stages on gfx950, read-back of
segments + probes, host post-process -> rectangle lists.  Frames are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0.  Multi-GPU: launched by torch.distributed.run, one rank per GPU, no data-path
collective (frames are independent); only a barrier and a MAX-reduce of the elapsed time.
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


def cpu_baseline(frames):
    """The reference itself (oracle/_ref: its kernels + host C on the serial OpenCL shim) when that build travelled
    with the repo, else our C restatement + the product's host post-process, timed on a bounded sample."""
    from tests import helpers
    import rectdetect_amd as ra
    sample = frames[:4]     # ~13 s of the reference on one core
    t0 = time.time()
    if helpers.have_ref():
        r = helpers.RefRect(IW, IH)
        for f in sample:
            r.execute_once(f, TAN_AOV)
        r.close()
        kind = "reference"
    else:
        o = helpers.OracleRect(IW, IH)
        for f in sample:
            o.frame(f)
            ra.postprocess_planes(o.segments(), o.plane("boundary"), o.plane("table"), IW, IH, TAN_AOV)
        o.close()
        kind = "port"
    dt = time.time() - t0
    return {"value": round(len(sample) / dt, 4), "unit": "frames/s", "cores": 1, "kind": kind,
            "sample": "%d consecutive 1920x1080 frames of the bench stream, single thread, %.1f s" % (len(sample), dt)}


def traffic_per_frame():
    """HBM-side bytes per frame from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE and WRITE_SIZE cannot be
    collected in the same pass, nor from inside this process); null when that file is absent."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return int(json.load(f)["hbm_bytes_per_frame"])
    except (OSError, KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames-per-step", type=int, default=64)
    ap.add_argument("--slots", type=int, default=8, help="frames in flight per GPU (from three on: one stream per frame on four shared streams)")
    ap.add_argument("--host-frames", action="store_true", help="hand over host buffers (PCIe upload inside the timed region); not the headline value")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercises sharding/aggregation only (CPU tests)")
    ap.add_argument("--backend", default=None)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist_mod
        dist = dist_mod
        backend = args.backend or ("gloo" if args.dry_run else "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)

    F = args.frames_per_step
    seed_stream = rank          # stream id = rank: independent 1080p streams, one per GPU (BASELINE.json configs[4])

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
        dev = local % L.rd_device_count()     # identity on a full node; lets two ranks share the only GPU of a test box (gloo)
        det = ra.Detector(IW, IH, device=dev, nslots=args.slots, nworkers=1)
        dframes = []
        for a in frames:
            p = L.rd_device_alloc(a.nbytes)
            L.rd_upload(p, a.ctypes.data, a.nbytes)
            dframes.append(p)

    nrect = 0
    inflight = 0

    def step():
        """one pass over the batch of F frames: every frame is handed to the detector; it is a stream, so the results of the
        last frames in flight are collected at the beginning of the next step (or by sync() at the end of the timed region)"""
        nonlocal nrect, inflight
        if args.dry_run:
            time.sleep(0.01 * (1 + rank))
            return
        for i in range(F):
            if inflight == args.slots:
                nrect += len(det.poll(TAN_AOV))
                inflight -= 1
            if args.host_frames:
                det.enqueue(frames[i])
            else:
                det.enqueue(dframes[i], ws=IW * 3, on_device=True)
            inflight += 1

    def sync():
        """all frames handed over so far are finished, post-processed and their rectangles collected"""
        nonlocal nrect, inflight
        if not args.dry_run:
            while inflight:
                nrect += len(det.poll(TAN_AOV))
                inflight -= 1
            det.drain()
            if dist is not None and dist.get_backend() == "nccl":
                import torch
                torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    dev0 = det.device_time() if det is not None else (0, 0)
    enq0 = ra.lib().rd_detector_counter(det.h, 3) if det is not None else 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
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

    if rank == 0:
        total_frames = world * args.steps * F
        fps = total_frames / elapsed
        N = IW * IH
        achieved = fps / world * B_ALG_PER_PIXEL * N      # bytes/s per GPU
        out = {
            "metric": "1920x1080 frames/sec", "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/i32/f32 (bit-exact integer + IEEE f32 stencil path)", "data": "synthetic",
            "config": {"workload": "vidrect 1920x1080 synthetic stream per GPU (BASELINE.json configs[4]; configs[1] is one frame of it)",
                       "frames_per_step": F, "frames_in_flight": args.slots, "input": "BGR u8 host buffers (PCIe upload timed)" if args.host_frames else "BGR u8 frames resident in HBM", "parallelism": "independent streams, one per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": "whole per-frame device pipeline (all stages, one stream per frame slot)",
                         "achieved": round(achieved / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(achieved / HBM_PEAK, 4),
                         "algorithmic_bytes_per_frame": B_ALG_PER_PIXEL * N, "traffic": traffic_per_frame(), "traffic_unit": "bytes/frame",
                         # HIP events on each frame's own stream over the timed region (rank 0): first kernel start -> last copy end.
                         # With several frames in flight these intervals overlap, which is why `achieved` is taken from the aggregate rate.
                         "frame_device_us_avg": round((dev1[0] - dev0[0]) / max(1, dev1[1] - dev0[1]), 1), "frames_in_flight": args.slots,
                         "host_enqueue_us_avg": round((enq1 - enq0) / max(1, dev1[1] - dev0[1]), 1),
                         "region_round_budget": det.region_round_budget()[0] if det is not None else None,
                         "frames_per_budget_8_12_16_20": [ra.lib().rd_detector_counter(det.h, 6 + k) for k in range(4)] if det is not None else None,
                         "frames_repeated": {"round_budget": det.region_round_budget()[1], "polyline_overflow": det.redone_frames()} if det is not None else None},
        }
        if not args.dry_run and not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames)
        if args.dry_run:
            out["dry_run"] = True
        print(json.dumps(out))

    if det is not None:
        det.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
