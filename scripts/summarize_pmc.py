#!/usr/bin/env python3
"""profiles/tools/run_pmc.sh output -> profiles/<tag>_pmc_traffic.md + profiles/pmc_traffic.json.
Applies the calibration measured in the same run (dword-per-lane copy of a known 1 GiB) - on gfx950
FETCH_SIZE reads 1/2 of the bytes, WRITE_SIZE reads them exactly (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys


def load(d):
    f = max(glob.glob(d + "/runc/*_counter_collection.csv"), key=os.path.getmtime)
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    src, tag, timed = sys.argv[1], sys.argv[2], int(sys.argv[3])
    res, lines = {}, ["# HBM traffic of `trex_step_kernel<false, false>` from PMC counters (%s)" % tag, "",
                      "separate passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --steps %d --warmup 30`;" % timed,
                      "calibration `profiles/tools/pmc_calib.hip` (1 GiB dword-per-lane copy) under the same counters.", ""]
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        b, k = load("%s/bench_%s" % (src, c)), load("%s/calib_%s" % (src, c))
        cal = [v for n, v in k.items() if "calib_copy_dword" in n][0]
        factor = (1 << 20) / (sum(cal) / len(cal))          # true KiB / counted KiB
        step = [v for n, v in b.items() if "trex_step_kernel<false, false>" in n][0][-timed:]
        raw = sum(step) / len(step)
        res[c] = raw * 1024 * factor
        lines.append("* %s: calibration counts %.0f KiB for 1048576 KiB -> factor %.3f; step kernel raw %.0f KiB/launch -> **%.1f MB/launch**"
                     % (c, sum(cal) / len(cal), factor, raw, res[c] / 1e6))
    total = res["FETCH_SIZE"] + res["WRITE_SIZE"]
    alg = 912 * 4096
    lines += ["", "HBM bytes per launch (4096 envs): **%.1f MB** vs algorithmic %.2f MB (x%.0f)." % (total / 1e6, alg / 1e6, total / alg)]
    open("profiles/%s_pmc_traffic.md" % tag, "w").write("\n".join(lines) + "\n")
    json.dump({"hbm_bytes_per_launch": total, "fetch_bytes": res["FETCH_SIZE"], "write_bytes": res["WRITE_SIZE"],
               "envs": 4096, "source": "profiles/%s_pmc_traffic.md" % tag}, open("profiles/pmc_traffic.json", "w"))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
