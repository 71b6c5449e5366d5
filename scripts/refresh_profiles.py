#!/usr/bin/env python3
"""gpurun_out/ (profiles/tools/run_profiles.sh + bench/stamps/wave-balance outputs) -> profiles/<tag>_*.
usage: scripts/refresh_profiles.py r01_v12 [old_tag_to_remove]"""
import collections
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
os.chdir(ROOT)
tag = sys.argv[1]
if len(sys.argv) > 2:
    for f in glob.glob("profiles/%s_*" % sys.argv[2]):
        os.remove(f)
subprocess.check_call([sys.executable, "scripts/summarize_rocprof.py", "gpurun_out/prof/trace", "profiles/%s_kernel_stats.md" % tag, "200", "gpurun_out/prof/bench_trace.json"])
subprocess.check_call([sys.executable, "scripts/summarize_pmc.py", "gpurun_out/pmc", tag, "40"])
v = {}
for d in ("gpurun_out/prof/sq", "gpurun_out/prof/sq2"):
    f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "trex_step_kernel<false, false>" in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, dd in per.items():
        vals = list(dd.values())[-40:]
        v[c] = sum(vals) / len(vals)
dur, w = v["GRBM_GUI_ACTIVE"] / 8, v["SQ_WAVES"]
L = ["# SQ counters of `trex_step_kernel<false, false>` (%s), bench.py scenario, 4096 envs, per launch (mean of the 40 timed launches)" % tag, "",
     "two passes, `rocprofv3 --kernel-trace --pmc ... -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline` (profiles/tools/run_profiles.sh)", ""]
L += ["* %s = %.4g" % (k, v[k]) for k in sorted(v)]
L += ["", "derived:",
      "* kernel duration = GRBM_GUI_ACTIVE/8 = %.3g cycles (dv-form kernel v8: 2.9e6)" % dur,
      "* waves = %d (%.1f per SIMD on 1024 SIMDs)" % (w, w / 1024.0),
      "* VALU instructions per wave = %d; SALU %d; LDS %d (v8: 137 500 / 7 265 / 8 335)" % (v["SQ_INSTS_VALU"] / w, v["SQ_INSTS_SALU"] / w, v["SQ_INSTS_LDS"] / w),
      "* mean wave lifetime = 4*SQ_WAVE_CYCLES/waves = %.3g cycles = %.0f %% of the kernel duration (v8: 38 %%)" % (4 * v["SQ_WAVE_CYCLES"] / w, 100 * 4 * v["SQ_WAVE_CYCLES"] / w / dur),
      "* VALU issue utilisation = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * duration) = %.1f %% (v8: 19 %%)" % (100 * v["SQ_INSTS_VALU"] * 2 / (1024 * dur)),
      "* per-wave durations of one launch: `%s_wave_balance.txt` (diagnostic build)" % tag]
open("profiles/%s_sq_counters.md" % tag, "w").write("\n".join(L) + "\n")
ver = tag.split("_")[-1]
for src, dst in (("gpurun_out/bench_%s.json" % ver, "bench_line.json"), ("gpurun_out/prof/bench_trace.json", "bench_line_profiled.json"),
                 ("gpurun_out/stamps_%s_4096.txt" % ver, "phase_stamps.txt"), ("gpurun_out/wave_balance_%s.txt" % ver, "wave_balance.txt")):
    if os.path.exists(src):
        shutil.copy(src, "profiles/%s_%s" % (tag, dst))
shutil.copy(max(glob.glob("gpurun_out/prof/trace/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime),
            "profiles/%s_kernel_stats.csv" % tag)
# machine-readable twins that bench.py quotes - only for the kernel build they were collected on
import json
line = json.loads([l for l in open("gpurun_out/prof/bench_trace.json") if l.startswith("{")][-1])
build = line["roofline"]["kernel_build"]
json.dump({"build_id": build, "tag": tag, "envs": 4096, "valu_insts_per_launch": v["SQ_INSTS_VALU"], "waves": w,
           "waves_per_simd": w / 1024.0, "duration_cycles": dur,
           "mean_wave_lifetime_frac": 4 * v["SQ_WAVE_CYCLES"] / w / dur,
           "valu_issue_utilisation": v["SQ_INSTS_VALU"] * 2 / (1024 * dur),
           "source": "profiles/%s_sq_counters.md" % tag}, open("profiles/sq_counters.json", "w"))
pj = json.load(open("profiles/pmc_traffic.json"))
pj["build_id"] = build
json.dump(pj, open("profiles/pmc_traffic.json", "w"))
print("\n".join(L[-6:]))
print(open("profiles/%s_kernel_stats.md" % tag).read().split("\n\n")[-2])
print(open("profiles/pmc_traffic.json").read())
