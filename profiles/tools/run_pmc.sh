#!/bin/bash
# HBM traffic of the step kernel from PMC counters: separate --pmc passes (FETCH_SIZE and WRITE_SIZE
# do not fit one pass on gfx950), kernel-trace only, plus the dword-copy calibration under the same
# counters. Run on the GPU box from the repo root: bash profiles/tools/run_pmc.sh
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc
rm -rf $OUT && mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 -o $OUT/pmc_calib profiles/tools/pmc_calib.hip
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/bench_$C -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $OUT/bench_$C.json 2> $OUT/bench_$C.err
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/calib_$C -- $OUT/pmc_calib > $OUT/calib_$C.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head
