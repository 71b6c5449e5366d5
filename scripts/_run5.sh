set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/b300.json 2> $O/b300.err
python -c "
import json
d=json.loads([l for l in open('$O/b300.json') if l.startswith('{')][-1]); print('b300', round(d['value']/1e6,3), 'M', d['ms_per_step'], d['roofline']['kernel_ms'])"
bash profiles/tools/run_pmc.sh > $O/pmc.log 2>&1
python - <<'PY'
import collections, csv, glob, os
def load(d):
    f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    o = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        o[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return o
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    b, k = load("gpurun_out/pmc/bench_%s" % c), load("gpurun_out/pmc/calib_%s" % c)
    cal = [v for n, v in k.items() if "calib_copy_dword" in n][0]
    factor = (1 << 20) / (sum(cal) / len(cal))
    step = [v for n, v in b.items() if "trex_step_kernel<false, false>" in n][0][-40:]
    res[c] = sum(step) / len(step) * 1024 * factor
    print(c, "factor %.3f" % factor, "%.2f MB/launch" % (res[c] / 1e6))
t = res["FETCH_SIZE"] + res["WRITE_SIZE"]
print("total %.2f MB = x%.2f algorithmic" % (t / 1e6, t / (912 * 4096)))
PY
rm -rf gpurun_out/pmc
