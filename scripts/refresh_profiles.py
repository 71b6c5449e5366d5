#!/usr/bin/env python3
"""Raw output of profiles/tools/run_profiles.sh (gpurun_out/prof, gpurun_out/pmc) -> the files committed under
profiles/: <tag>_kernel_stats.md/.csv, <tag>_pmc_traffic.md, <tag>_sq_counters.md, bench lines, per-wave phases, and
the machine-readable pmc_traffic.json / sq_counters.json that bench.py quotes (keyed by the kernel build id).
Runs ON the GPU box at the end of run_profiles.sh (the raw traces are too big to travel):
    python scripts/refresh_profiles.py <tag> <out_dir>
then, here:  cp gpurun_out/prof/final/* profiles/"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
os.chdir(ROOT)
tag, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
P = "gpurun_out/prof"
line = json.loads([l for l in open(P + "/bench_trace.json") if l.startswith("{")][-1])
KERNEL = line["roofline"].get("kernel", "trex_step_kernel<false, false>")     # trex_step_pair_kernel since round 4 (even resident batches)
build = line["roofline"]["kernel_build"]

# ---- kernel trace
subprocess.check_call([sys.executable, "scripts/summarize_rocprof.py", P + "/trace", "%s/%s_kernel_stats.md" % (out, tag), "200",
                       P + "/bench_trace.json"], stdout=subprocess.DEVNULL)
shutil.copy(max(glob.glob(P + "/trace/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime),
            "%s/%s_kernel_stats.csv" % (out, tag))

# ---- HBM traffic (separate FETCH_SIZE / WRITE_SIZE passes, calibrated in the same run)
def load(d):
    f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    o = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        o[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return o

timed = 40
res, L = {}, ["# HBM traffic of `%s` from PMC counters (%s, kernel build %s)" % (KERNEL, tag, build), "",
              "separate passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps %d --warmup 30`;" % timed,
              "calibration `profiles/tools/pmc_calib.hip` (1 GiB dword-per-lane copy) under the same counters.", ""]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    b, k = load("gpurun_out/pmc/bench_%s" % c), load("gpurun_out/pmc/calib_%s" % c)
    cal = [v for n, v in k.items() if "calib_copy_dword" in n][0]
    factor = (1 << 20) / (sum(cal) / len(cal))          # true KiB / counted KiB
    step = [v for n, v in b.items() if KERNEL in n][0][-timed:]
    raw = sum(step) / len(step)
    res[c] = raw * 1024 * factor
    L.append("* %s: calibration counts %.0f KiB for 1048576 KiB -> factor %.3f; step kernel raw %.0f KiB/launch -> **%.2f MB/launch**"
             % (c, sum(cal) / len(cal), factor, raw, res[c] / 1e6))
total = res["FETCH_SIZE"] + res["WRITE_SIZE"]
alg = 912 * 4096
L += ["", "HBM bytes per launch (4096 envs): **%.2f MB** vs algorithmic %.2f MB (x%.1f)." % (total / 1e6, alg / 1e6, total / alg)]
open("%s/%s_pmc_traffic.md" % (out, tag), "w").write("\n".join(L) + "\n")
json.dump({"build_id": build, "tag": tag, "hbm_bytes_per_launch": total, "fetch_bytes": res["FETCH_SIZE"], "write_bytes": res["WRITE_SIZE"],
           "envs": 4096, "source": "profiles/%s_pmc_traffic.md" % tag}, open(out + "/pmc_traffic.json", "w"))

# ---- SQ counters
v = {}
for d in (P + "/sq", P + "/sq2"):
    f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, dd in per.items():
        vals = [dd[k] for k in sorted(dd, key=int)][-40:]
        v[c] = sum(vals) / len(vals)
dur, w = v["GRBM_GUI_ACTIVE"] / 8, v["SQ_WAVES"]
util = v["SQ_INSTS_VALU"] * 2 / (1024 * dur)
life = 4 * v["SQ_WAVE_CYCLES"] / w
S = ["# SQ counters of `%s` (%s, kernel build %s), bench.py scenario, 4096 envs, per launch (mean of the 40 timed launches)" % (KERNEL, tag, build), "",
     "two passes, `rocprofv3 --kernel-trace --pmc ... -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline` (profiles/tools/run_profiles.sh)", ""]
S += ["* %s = %.4g" % (k, v[k]) for k in sorted(v)]
S += ["", "derived:",
      "* kernel duration = GRBM_GUI_ACTIVE/8 = %.3g cycles" % dur,
      "* waves = %d (%.1f per SIMD on 1024 SIMDs: one env per wave, all resident at once)" % (w, w / 1024.0),
      "* instructions per wave: VALU %d, SALU %d, LDS %d" % (v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w),
      "* mean wave lifetime = 4*SQ_WAVE_CYCLES/waves = %.3g cycles = %.0f %% of the kernel duration" % (life, 100 * life / dur),
      "* cycles per instruction and wave while alive = %.2f" % (life / ((v["SQ_INSTS_VALU"] + v["SQ_INSTS_SALU"] + v["SQ_INSTS_LDS"]) / w)),
      "* VALU issue utilisation = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * duration) = %.1f %%" % (100 * util),
      "* wave cycles: waiting on counters (SQ_WAIT_ANY) %.0f %%, issue stalls (SQ_WAIT_INST_ANY) %.0f %%" % (
          100 * v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"], 100 * v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"])]
open("%s/%s_sq_counters.md" % (out, tag), "w").write("\n".join(S) + "\n")
alone = None
if os.path.exists(P + "/bench_256.json"):
    alone = json.loads([l for l in open(P + "/bench_256.json") if l.startswith("{")][-1])["roofline"]["kernel_ms"]
    S.append("* a 256-env launch of the same build (one wave per four SIMDs: every env runs alone, the launch is its slowest env) "
             "takes %.4f ms: no launch of any size is shorter" % alone)
    open("%s/%s_sq_counters.md" % (out, tag), "w").write("\n".join(S) + "\n")
json.dump({"build_id": build, "tag": tag, "envs": 4096, "valu_insts_per_launch": v["SQ_INSTS_VALU"], "waves": w, "alone_kernel_ms": alone,
           "waves_per_simd": w / 1024.0, "duration_cycles": dur, "mean_wave_lifetime_frac": life / dur,
           "valu_issue_utilisation": util, "source": "profiles/%s_sq_counters.md" % tag}, open(out + "/sq_counters.json", "w"))

# ---- the rest is copied under the tag
for src, dst in (("bench_300.json", "bench_line.json"), ("bench_256.json", "bench_line_256envs.json"), ("bench_20.json", "bench_line_20steps.json"), ("bench_trace.json", "bench_line_profiled.json"),
                 ("bench_300_cycle16.json", "bench_line_cycle16.json"), ("bench_20_cycle16.json", "bench_line_cycle16_20steps.json"),
                 ("parity_stats.txt", "parity_stats_head.txt"), ("ppo_rate.txt", "ppo_rate.txt"), ("other_configs.md", "other_configs.md"),
                 ("soak.txt", "soak.txt"), ("ppo_learning_curve.txt", "ppo_learning_curve.txt"),
                 ("bench_300_spl10.json", "bench_line_steps_per_launch10.json"), ("bench_300_spl30.json", "bench_line_steps_per_launch30.json"),
                 ("wave_phases_4096.txt", "wave_phases_4096.txt"), ("wave_phases_256.txt", "wave_phases_256.txt"),
                 ("wave_phases_pair_4096.txt", "wave_phases_pair_4096.txt"), ("wave_phases_pair_256.txt", "wave_phases_pair_256.txt"),
                 ("learn_phases.txt", "learn_phases.txt"), ("ppo_kernels.md", "ppo_kernels.md"), ("launch_floor.txt", "launch_floor.txt"),
                 ("row_bench.txt", "row_bench.txt"), ("census.txt", "census.txt")):
    if os.path.exists(P + "/" + src):
        shutil.copy(P + "/" + src, "%s/%s_%s" % (out, tag, dst))
print("\n".join(S[-9:]))
print("\n".join(L[-4:]))
