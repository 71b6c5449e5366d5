cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TREX_BENCH_FORCE_GATHER=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29513 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rg -- python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/b_rg.json 2> gpurun_out/b_rg.err
f=$(ls gpurun_out/rg/*/*_kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])), key=lambda r:int(r["Start_Timestamp"]))
steps=[i for i,r in enumerate(rows) if ("trex_step_kernel<false, false>" in r["Kernel_Name"] or "trex_step_pair_kernel" in r["Kernel_Name"])]
i0=steps[-30]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0+12]:
    print("%9.1f %8.1f  q%s  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Queue_Id","?"), r["Kernel_Name"][:60]))
PY
rm -rf gpurun_out/rg
