#!/usr/bin/env python3
"""rocprofv3 --pmc output dir(s) -> per-launch means of every counter for trex_step_kernel<false, false>
(last N dispatches) and the derived issue figures. usage: scripts/sq_summary.py N dir [dir ...]"""
import collections
import csv
import glob
import os
import sys

n = int(sys.argv[1])
v = {}
meta = {}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "trex_step_kernel<false, false>" in r["Kernel_Name"] or "trex_step_pair_kernel" in r["Kernel_Name"]:
                per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
                for k in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size", "Accum_VGPR_Count"):
                    if k in r:
                        meta[k] = r[k]
        for c, dd in per.items():
            vals = [dd[k] for k in sorted(dd, key=int)][-n:]
            v[c] = sum(vals) / len(vals)
print("dispatch:", meta)
for k in sorted(v):
    print("%-24s %.4g" % (k, v[k]))
if "GRBM_GUI_ACTIVE" in v and "SQ_INSTS_VALU" in v:
    dur = v["GRBM_GUI_ACTIVE"] / 8
    print("duration cycles %.4g; VALU issue utilisation %.1f %%" % (dur, 100 * v["SQ_INSTS_VALU"] * 2 / (1024 * dur)))
    if "SQ_WAVES" in v and "SQ_WAVE_CYCLES" in v:
        w = v["SQ_WAVES"]
        print("waves %d; VALU/wave %d; mean wave lifetime %.3g cycles = %.0f %% of the launch" % (
            w, v["SQ_INSTS_VALU"] / w, 4 * v["SQ_WAVE_CYCLES"] / w, 100 * 4 * v["SQ_WAVE_CYCLES"] / w / dur))
