set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3b; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/b20.json 2> $O/b20.err
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $O/b300.json 2> $O/b300.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20n.json 2> $O/b20n.err
for v in abl_math abl_quat abl_both; do TREX_LIB=$PWD/trex-gym_amd/trex_gym/libtrex_hip_$v.so python scripts/parity_stats.py > $O/parity_$v.txt 2>&1; done
python scripts/parity_stats.py > $O/parity_head.txt 2>&1
for f in b20 b300 b20n; do python -c "
import json,sys
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', round(d['value']/1e6,3), 'M', d['ms_per_step'], d['roofline']['kernel_ms'], d['state_mix']['mean_contacts'], d.get('gather'))"; done
for v in head abl_math abl_quat abl_both; do echo $v; grep -E "dqd|dtau" $O/parity_$v.txt; done
export TREX_BENCH_BACKEND=gloo TREX_BENCH_SHARE_DEVICE=1
OMP_NUM_THREADS=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/g2_omp1.json 2> $O/g2_omp1.err; echo "rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 5 > $O/g2_torchrun.json 2> $O/g2_torchrun.err; echo "rc=$?"
timeout -k 10 300 python3 bench.py --gpus 2 --steps 20 --warmup 5 --action-cycle 16 > $O/g2_cycle.json 2> $O/g2_cycle.err; echo "rc=$?"
for f in g2_omp1 g2_torchrun g2_cycle; do python -c "
import json,sys
d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', round(d['value']/1e6,3), 'M', d['ms_per_step'], d.get('gather'))"; done
