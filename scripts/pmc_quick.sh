cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --pmc ${TREX_PMC:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES} --output-format csv -d gpurun_out/ic -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > gpurun_out/bench_ic.json 2> gpurun_out/bench_ic.err || echo FAILED
python - <<'PY'
import csv,glob,collections
f=max(glob.glob("gpurun_out/ic/**/*_counter_collection.csv", recursive=True))
per=collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if ("trex_step_kernel<false, false>" in r["Kernel_Name"] or "trex_step_pair_kernel" in r["Kernel_Name"]):
        per[r["Counter_Name"]][r["Dispatch_Id"]]+=float(r["Counter_Value"])
for c,dd in per.items():
    vals=[dd[k] for k in sorted(dd,key=int)][-40:]
    print(c, sum(vals)/len(vals))
PY
rm -rf gpurun_out/ic
