#!/bin/bash
# Round profile set for bench.py (run on the GPU box from the repo root):
#   1. kernel trace + stats   2. HBM traffic PMC passes (+ calibration)   3. SQ instruction/occupancy PMC pass
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf $OUT gpurun_out/pmc && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
bash profiles/tools/run_pmc.sh > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $OUT/bench_sq.json 2> $OUT/bench_sq.err || echo "SQ pass failed"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $OUT/bench_sq2.json 2> $OUT/bench_sq2.err || echo "SQ2 pass failed"
ls $OUT
