set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3c; mkdir -p $O
python -m pytest tests/test_gpu_policy.py -x -q -s > $O/policy_tests.log 2>&1 || { tail -60 $O/policy_tests.log; exit 1; }
tail -3 $O/policy_tests.log
python -m pytest tests -m gpu -x -q -s --deselect tests/test_gpu_policy.py > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
python scripts/ppo_rate.py 4 > $O/ppo_rate.txt 2>&1 || { tail -30 $O/ppo_rate.txt; exit 1; }
grep PPO $O/ppo_rate.txt
