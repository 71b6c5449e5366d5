#!/bin/bash
# rocprofv3 per-kernel summary of the PPO loop (4 rollouts of 32 steps + one epoch of 32 minibatch steps; 4096 envs):
# the step kernel beside the trainer-side kernels of include/trex_policy.h. Run on the GPU box from the repo root:
#   bash scripts/profile_ppo_kernels.sh <out.md>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=${1:-gpurun_out/ppo_kernels.md}
T=gpurun_out/ppo_trace; rm -rf $T; mkdir -p $T
rocprofv3 --kernel-trace --stats --output-format csv -d $T -- python scripts/ppo_profile_workload.py > $T/log 2>&1
f=$(find $T -name "*kernel_stats.csv" | head -1)
python - "$f" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
keep = [r for r in rows if any(k in r["Name"] for k in ("trex_step_kernel", "trex_step_pair_kernel", "observe_kernel", "act_kernel", "learn_", "gae_kernel", "adv_stats", "adam_kernel"))]
L = ["# PPO loop, per kernel (rocprofv3 --kernel-trace --stats; scripts/ppo_profile_workload.py: 4 rollouts x 32 steps + 32 minibatch steps, 4096 envs)", "",
     "| kernel | calls | avg us | total ms |", "|---|---|---|---|"]
for r in keep:
    import re
    name = re.search(r"(trex_step_kernel<[a-z, ]+>|trex_step_pair_kernel|trex_step_many_kernel|observe_kernel|act_kernel|learn_grad_kernel|learn_reduce_kernel|learn_adam_kernel|learn_apply_kernel|gae_kernel|adv_stats_kernel|adam_kernel)", r["Name"]).group(1)
    L.append("| `%s` | %s | %.1f | %.2f |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
other = sum(float(r["TotalDurationNs"]) for r in rows if r not in keep) / 1e6
L += ["", "everything else (PyTorch: noise, permutation, orthogonal initialisation ...): %.2f ms in total" % other]
open(sys.argv[2], "w").write("\n".join(L) + "\n")
print("\n".join(L))
PY
rm -rf $T
