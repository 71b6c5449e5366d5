#!/bin/bash
# Round profile set for bench.py (run on the GPU box from the repo root):
#   1. kernel trace + stats   2. HBM traffic PMC passes (+ calibration)   3. SQ instruction/occupancy PMC passes
#   4. per-wave phase cycles of the diagnostic build   5. the plain bench lines (300 and 20 timed steps)
# scripts/refresh_profiles.py <tag> then turns gpurun_out/ into profiles/<tag>_* (+ sq_counters.json, pmc_traffic.json)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof
rm -rf $OUT gpurun_out/pmc && mkdir -p $OUT
python bench.py --steps 300 --warmup 30 > $OUT/bench_300.json 2> $OUT/bench_300.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err      # the driver's call, CPU baseline included
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --action-cycle 16 > $OUT/bench_300_cycle16.json 2> $OUT/bench_300_cycle16.err   # rounds 1-2's input (NOT the headline)
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --action-cycle 16 > $OUT/bench_20_cycle16.json 2> $OUT/bench_20_cycle16.err
python bench.py --steps 200 --warmup 30 --no-cpu-baseline --envs-per-gpu 256 > $OUT/bench_256.json 2> $OUT/bench_256.err   # one wave per four SIMDs: the serial floor of a launch
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 200 --warmup 30 --no-cpu-baseline > $OUT/bench_trace.json 2> $OUT/bench_trace.err
bash profiles/tools/run_pmc.sh > $OUT/pmc.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $OUT/bench_sq.json 2> $OUT/bench_sq.err || echo "SQ pass failed"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq2 -- python bench.py --steps 40 --warmup 30 --no-cpu-baseline > $OUT/bench_sq2.json 2> $OUT/bench_sq2.err || echo "SQ2 pass failed"
TREX_LIB=$PWD/trex-gym_amd/trex_gym/libtrex_hip_stamps.so python scripts/wave_phases.py 4096 1300 0 2>&1 | grep -v amdgpu.ids > $OUT/wave_phases_4096.txt
TREX_LIB=$PWD/trex-gym_amd/trex_gym/libtrex_hip_stamps.so python scripts/wave_phases.py 256 1300 0 2>&1 | grep -v amdgpu.ids > $OUT/wave_phases_256.txt
# the pair launch (two envs per workgroup, the product's step launch up to 4096 envs) and the learner, stamped (diagnostic builds)
TREX_LIB=$PWD/trex-gym_amd/trex_gym/libtrex_hip_pstamps.so python scripts/wave_phases_pair.py 4096 1300 2>&1 | grep -v amdgpu.ids > $OUT/wave_phases_pair_4096.txt || echo "pair stamps failed"
TREX_LIB=$PWD/trex-gym_amd/trex_gym/libtrex_hip_pstamps.so python scripts/wave_phases_pair.py 256 1300 2>&1 | grep -v amdgpu.ids > $OUT/wave_phases_pair_256.txt || true
TREX_LIB=$PWD/trex-gym_amd/trex_gym/libtrex_hip_lstamps.so python scripts/learn_phases.py 2>&1 | grep -v amdgpu.ids > $OUT/learn_phases.txt || echo "learner stamps failed"
bash scripts/profile_ppo_kernels.sh $OUT/ppo_kernels.md > /dev/null 2>&1 || echo "ppo kernel table failed"
./profiles/tools/launch_floor_bench > $OUT/launch_floor.txt 2>&1 || true
python scripts/parity_stats.py 2>&1 | grep -v amdgpu.ids > $OUT/parity_stats.txt
python scripts/ppo_rate.py 4 2>&1 | grep PPO > $OUT/ppo_rate.txt
python scripts/ppo_rate.py 32 2>&1 | grep PPO >> $OUT/ppo_rate.txt
python scripts/ppo_signal.py 4096 32 150 32 1 32 2>&1 | grep -v amdgpu.ids > $OUT/ppo_learning_curve.txt || echo "ppo_signal failed"
for S in 10 30; do python bench.py --steps 300 --warmup 30 --no-cpu-baseline --steps-per-launch $S > $OUT/bench_300_spl$S.json 2> $OUT/bench_300_spl$S.err; done
python scripts/other_configs.py $OUT/other_configs.md > $OUT/other_configs.log 2>&1 || echo "other_configs failed"
python scripts/soak.py 20000 2>&1 | grep -v amdgpu.ids > $OUT/soak.txt || echo "soak failed"
./profiles/tools/row_bench > $OUT/row_bench.txt 2>&1 || true
./profiles/tools/census 4096 > $OUT/census.txt 2>&1 || true
# summarise on the box (the raw traces are too big to travel), keep only gpurun_out/prof/final
python scripts/refresh_profiles.py ${1:-r03} $OUT/final
# the two headline lines once more, now that the traffic of THIS build is known (bench.py reads profiles/pmc_traffic.json)
cp $OUT/final/pmc_traffic.json $OUT/final/sq_counters.json profiles/
python bench.py --steps 300 --warmup 30 > $OUT/bench_300.json 2> $OUT/bench_300.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_20.json 2> $OUT/bench_20.err
python scripts/refresh_profiles.py ${1:-r03} $OUT/final
rm -rf $OUT/trace $OUT/sq $OUT/sq2 gpurun_out/pmc
ls $OUT/final
