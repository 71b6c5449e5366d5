cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sqx && mkdir -p gpurun_out/sqx
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sqx -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > gpurun_out/sqx/bench.json 2> gpurun_out/sqx/err.txt
