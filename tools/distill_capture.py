#!/usr/bin/env python3
"""Turn what tools/gpu_capture.sh <tag> brought back (gpurun_out/cap<tag>/) into the files committed under profiles/.
usage: distill_capture.py <tag> <round prefix, e.g. r02_a>"""
import sys, os, json, shutil, sqlite3, subprocess, collections

tag, pre = sys.argv[1], sys.argv[2]
ROUND = pre.split("_")[0]      # r03_a -> r03
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "gpurun_out", "cap" + tag)
P = os.path.join(R, "profiles")


def run(*a):
    return subprocess.run([sys.executable] + list(a), capture_output=True, text=True, check=True).stdout


for src, dst in (("bench_default.json", "bench_default.json"), ("bench_slots1.json", "bench_slots1.json"), ("bench_slots2.json", "bench_slots2.json")):
    shutil.copy(os.path.join(C, src), os.path.join(P, pre + "_" + dst))
with open(os.path.join(P, pre + "_size_sweep.txt"), "w") as f:
    f.write("# tools/size_sweep.py, SLOTS=32 (3840x2160: 16)\n" + open(os.path.join(C, "size_sweep.txt")).read())

# kernel traces: frames = launches of the one kernel that every frame launches on its own, grouped or not (the strong mask: it
# starts from the previous frame's)
for name, out in (("trace_default", "kernel_stats_1080p_default.txt"), ("trace_slots1", "kernel_stats_1080p_slots1.txt")):
    db = os.path.join(C, name, "t_results.db")
    con = sqlite3.connect(db)
    frames = 256 if name == "trace_default" else 16      # (steps + warmup) x frames per step of the command below
    head = "# rocprofv3 --kernel-trace --stats -- python bench.py %s (%d frames)\n" % (
        "--steps 3 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify --no-configs" if name == "trace_default" else "--steps 1 --warmup 1 --slots 1 --frames-per-step 8 --no-cpu-baseline --no-verify --no-configs", frames)
    with open(os.path.join(P, pre + "_" + out), "w") as f:
        f.write(head + run(os.path.join(R, "tools", "prof_summary.py"), db, str(frames)))

for name, out, cmd in (("trace_720p", "kernel_stats_720p_default.txt", "--frame 1280x720 --stream-seed 1 --steps 3 --warmup 1 --frames-per-step 64 --no-cpu-baseline --no-verify"),
                       ("trace_4k", "kernel_stats_4k_default.txt", "--frame 3840x2160 --stream-seed 4 --steps 3 --warmup 1 --frames-per-step 16 --no-cpu-baseline --no-verify")):
    db = os.path.join(C, name, "t_results.db")
    if not os.path.exists(db):
        continue
    frames = 256 if name == "trace_720p" else 64      # (steps + warmup) x frames per step
    with open(os.path.join(P, pre + "_" + out), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python bench.py %s (%d frames)\n" % (cmd, frames) + run(os.path.join(R, "tools", "prof_summary.py"), db, str(frames)))

# counters (RD_NO_GRAPH=1, --slots 1 --frames-per-step 4: 8 frames)
sq = run(os.path.join(R, "tools", "pmc_summary.py"), os.path.join(C, "pmc_sq", "results.db"), "8", "40")
with open(os.path.join(P, pre + "_pmc_sq_1080p.txt"), "w") as f:
    f.write("# rocprofv3 --pmc SQ_* --kernel-trace (own pass; RD_NO_GRAPH=1, bench.py --steps 1 --warmup 1 --slots 1 --frames-per-step 4)\n" + sq)


def total(db, counter, like=None):
    """sum of a counter over the capture; without `like`: over the FRAME PATH's kernels - the runtime's fill / copy kernels and the table set-up run once per
    detector (vote tables, ring planes of 64 slots: 55 MB per frame of a 48-frame capture) and are no part of a frame in the steady state"""
    con = sqlite3.connect(db)
    q = "select sum(value) from counters_collection where counter_name=?"
    a = [counter]
    if like:
        q += " and kernel_name like ?"; a.append(like)
    else:
        q += " and kernel_name not like '%__amd_rocclr_%' and kernel_name not like '%k_quant24_lut%'"
    return con.execute(q, a).fetchone()[0] or 0.0


FRAMES = {"pmc_rd": 48, "pmc_wr": 48, "pmc_rd_720p": 48, "pmc_wr_720p": 48, "pmc_rd_4k": 24, "pmc_wr_4k": 24}      # (steps + warmup) x frames per step of tools/gpu_pmc.sh's commands


def frames_in(db):
    return FRAMES[os.path.basename(os.path.dirname(db))]


TRAFFIC_CMD = "python bench.py --steps 2 --warmup 1 --frames-per-step 16 --no-cpu-baseline --no-verify --no-configs"
nf_rd, nf_wr = frames_in(os.path.join(C, "pmc_rd", "results.db")), frames_in(os.path.join(C, "pmc_wr", "results.db"))
rd = total(os.path.join(C, "pmc_rd", "results.db"), "FETCH_SIZE") / nf_rd
wr = total(os.path.join(C, "pmc_wr", "results.db"), "WRITE_SIZE") / nf_wr
cal_rd = total(os.path.join(C, "pmc_calrd", "results.db"), "FETCH_SIZE", "%k_copy_i%") / 3
cal_wr = total(os.path.join(C, "pmc_calwr", "results.db"), "WRITE_SIZE", "%k_copy_i%") / 3
gib_kib = (1 << 30) / 1024
corr_rd = round(gib_kib / cal_rd, 3)
corr_wr = round(gib_kib / cal_wr, 3)
with open(os.path.join(P, pre + "_pmc_traffic_1080p.txt"), "w") as f:
    f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB per frame; the benchmarked configuration: %s; %d frames)\n" % (TRAFFIC_CMD, nf_rd))
    f.write(run(os.path.join(R, "tools", "pmc_summary.py"), os.path.join(C, "pmc_rd", "results.db"), str(nf_rd), "40"))
    f.write(run(os.path.join(R, "tools", "pmc_summary.py"), os.path.join(C, "pmc_wr", "results.db"), str(nf_wr), "40"))
    f.write("# calibration (tools/pmc_calibrate.py: k_copy_i moves 1 GiB = %d KiB each way per call): FETCH_SIZE reports %.0f KiB, WRITE_SIZE %.0f KiB per call\n" % (gib_kib, cal_rd, cal_wr))
    f.write("# corrections: fetch x %.3f, write x %.3f\n" % (corr_rd, corr_wr))
hbm = int((rd * corr_rd + wr * corr_wr) * 1024)
commit = subprocess.run(["git", "-C", R, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
captured = open(os.path.join(C, "commit.txt")).read().strip() if os.path.exists(os.path.join(C, "commit.txt")) else None
dirty = subprocess.run(["git", "-C", R, "status", "--porcelain", "--", "rectdetect_amd", "bench.py"], capture_output=True, text=True).stdout.strip()
# (the measured code - library, bench, capture scripts - must be what HEAD holds; commits that touch nothing of it, e.g. this script, may lie between)
MEASURED = ["rectdetect_amd", "include", "bench.py", "tools/gpu_capture.sh", "tools/gpu_pmc.sh", "tools/pmc_calibrate.py", "tools/size_sweep.py"]
moved = captured is None or subprocess.run(["git", "-C", R, "diff", "--quiet", captured, "HEAD", "--"] + MEASURED).returncode != 0
if moved or dirty:
    sys.exit("distill_capture: the capture ran at %s, HEAD is %s%s - profiles must describe the code that is benchmarked: capture again (tools/capture_round.sh)" % (captured, commit, " with uncommitted changes" if dirty else ""))
other = {}
for key, label, npix in (("720p", "1280x720 (BASELINE.json configs[2])", 1280 * 720), ("4k", "3840x2160 (BASELINE.json configs[3])", 3840 * 2160)):
    dr, dw = os.path.join(C, "pmc_rd_" + key, "results.db"), os.path.join(C, "pmc_wr_" + key, "results.db")
    if os.path.exists(dr) and os.path.exists(dw) and frames_in(dr) and frames_in(dw):
        r2, w2 = total(dr, "FETCH_SIZE") / frames_in(dr), total(dw, "WRITE_SIZE") / frames_in(dw)
        b = int((r2 * corr_rd + w2 * corr_wr) * 1024)
        other[label] = {"frames": frames_in(dr), "fetch_size_kib_per_frame": int(r2), "write_size_kib_per_frame": int(w2), "hbm_bytes_per_frame": b,
                        "hbm_bytes_per_pixel": round(b / npix, 1), "algorithmic_bytes_per_frame": 829 * npix}
        with open(os.path.join(P, pre + "_pmc_traffic_%s.txt" % key), "w") as f:
            f.write("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB per frame; python bench.py --frame ... --steps 2 --warmup 1; %d frames); corrections as for 1080p: fetch x %.3f, write x %.3f\n" % (frames_in(dr), corr_rd, corr_wr))
            f.write(run(os.path.join(R, "tools", "pmc_summary.py"), dr, str(frames_in(dr)), "40"))
            f.write(run(os.path.join(R, "tools", "pmc_summary.py"), dw, str(frames_in(dw)), "40"))
with open(os.path.join(P, ROUND + "_traffic.json"), "w") as f:
    json.dump({"commit": captured, "command": TRAFFIC_CMD, "frames": nf_rd, "other_configurations": other,
               "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes over the benchmarked configuration (default slots, graphs on; profiles/%s_pmc_traffic_1080p.txt); "
               "each corrected by the factor the calibration copy of 3 x 1 GiB (4 B/lane coalesced, tools/pmc_calibrate.py) yields in the same capture "
               "(FETCH_SIZE reports half of the bytes read on gfx950, as MI355X_MICROARCH.md describes); the runtime's fill / copy kernels of detector set-up are left out (steady state)" % pre,
               "fetch_size_kib_per_frame": int(rd), "write_size_kib_per_frame": int(wr), "fetch_correction": corr_rd, "write_correction": corr_wr,
               "hbm_bytes_per_frame": hbm}, f, indent=1)
for src, dst in (("ref_opencl.json", pre + "_ref_opencl.json"), ("pytest_gpu.log", pre + "_pytest_gpu.txt")):
    if os.path.exists(os.path.join(C, src)):
        shutil.copy(os.path.join(C, src), os.path.join(P, dst))
print("fetch %.0f KiB x %.3f, write %.0f KiB x %.3f -> %.3f GB/frame" % (rd, corr_rd, wr, corr_wr, hbm / 1e9))
