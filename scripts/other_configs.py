#!/usr/bin/env python3
"""BASELINE configs other than the headline, on one GPU: batch-size sweep (config 4'), domain randomisation (config 5),
collision primitives, and the on-device PPO loop (config 3). Writes a markdown table.
    python scripts/other_configs.py <out.md>"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
os.chdir(ROOT)
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/other_configs.md"


def bench(*extra):
    cmd = [sys.executable, "bench.py", "--steps", "200", "--warmup", "30", "--no-cpu-baseline"] + list(extra)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not lines:
        return None, r.stderr[-400:]
    return json.loads(lines[-1]), ""


rows = ["| config | envs | env-steps/s | ms/step | step launch ms | mean contact points |", "|---|---|---|---|---|---|"]
cases = [("random actions (4')", n, ["--envs-per-gpu", str(n)]) for n in (256, 1024, 2048, 4096, 8192, 16384, 32768)]
cases += [("5 domain randomisation", 4096, ["--domain-rand"]), ("collision primitives", 4096, ["--collision", "primitives"])]
for name, n, extra in cases:
    d, err = bench(*extra)
    if d is None:
        rows.append("| %s | %d | FAILED %s | | | |" % (name, n, err.replace("\n", " ")))
        continue
    rows.append("| %s | %d | %.3f M | %.4f | %.4f | %.2f |" % (name, n, d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernel_ms"],
                                                             d["state_mix"]["mean_contacts"][-1]))
    print(rows[-1], flush=True)

# config 3 (PPO-driven, learner included): scripts/ppo_rate.py -> <tag>_ppo_rate.txt
open(out, "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
