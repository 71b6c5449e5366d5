set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3d; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python scripts/_prof_ppo.py > $O/prof.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:25]:
    print("%-90s calls %6s  avg %9.1f us  total %8.2f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
rm -rf $O/trace
