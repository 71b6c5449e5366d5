cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3g; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/ic -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $O/b.json 2> $O/b.err
rocprofv3 --kernel-trace --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_TC_STALL SQ_BUSY_CYCLES --output-format csv -d $O/ic2 -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $O/b2.json 2> $O/b2.err
python - <<'PY'
import csv, glob, collections, os
for d in ("gpurun_out/r3g/ic", "gpurun_out/r3g/ic2"):
    fs = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)
    if not fs: print("no csv in", d); continue
    f = max(fs, key=os.path.getmtime)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "trex_step_kernel<false, false>" in r["Kernel_Name"]:
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, dd in sorted(per.items()):
        vals = [dd[k] for k in sorted(dd, key=int)][-40:]
        print("%-32s %.4g" % (c, sum(vals) / len(vals)))
PY
rm -rf $O/ic $O/ic2
