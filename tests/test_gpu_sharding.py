"""The N > 1 path on real kernels (one-GPU box: two rank processes share the GPU, gloo carries the collective - RCCL
refuses two ranks per device): env-id sharding + the in-place pipelined all-gather of the [obs | reward | done] row block
(sharding.PipelinedGather, the form bench.py --gpus N times) against a single-process run of the same global batch.
Bitwise: envs are independent and the actions are keyed by the GLOBAL env id."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ASSET_URDF, ROOT

pytestmark = pytest.mark.gpu
N_GLOBAL, STEPS = 16, 7


def _actions(model_lo, model_hi, ids, t, dev):
    from trex_gym import sharding
    return sharding.synthetic_actions(ids, t, model_lo, model_hi, seed=0, device=dev)


def _rank(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trex_gym.vec_env import TrexVecEnv
    dev = torch.device("cuda", 0)
    env = TrexVecEnv(N_GLOBAL, urdf_path=ASSET_URDF, device=dev, rank=rank, world_size=world, max_episode_steps=4, row_buffers=2)
    env.reset_tensor()
    ids = torch.arange(env.env_lo, env.env_hi, device=dev)
    got = []
    for t in range(STEPS):
        env.step_tensor(_actions(env.model.lower, env.model.upper, ids, t, dev))
        prev = env.all_gather_rows_pipelined(wait=False)       # bench.py's call: in place, no consumer-side wait
        got.append(None if prev is None else prev.clone().cpu())
    got.append(env._pipe.flush().clone().cpu())
    blocking = env.all_gather_rows().clone().cpu()              # the blocking form: this step's rows
    # the same exchange by peer copies (sharding.CopyGather: IPC-mapped buffers, copy engines, no collective kernel)
    got_copy = []
    for t in range(STEPS, STEPS + 5):
        env.step_tensor(_actions(env.model.lower, env.model.upper, ids, t, dev))
        prev = env.all_gather_rows_copy()
        got_copy.append(None if prev is None else prev.clone().cpu())
    got_copy.append(env._copy_pipe.flush().clone().cpu())
    if rank == 1:
        ret["got"], ret["blocking"], ret["got_copy"] = got, blocking, got_copy
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_gpu_gather_what_one_process_computes():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
    from trex_gym.vec_env import TrexVecEnv
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rank, args=(2, port, ret), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    ref = TrexVecEnv(N_GLOBAL, urdf_path=ASSET_URDF, device=dev, max_episode_steps=4)
    ref.reset_tensor()
    ids = torch.arange(N_GLOBAL, device=dev)
    rows = []
    for t in range(STEPS + 5):
        ref.step_tensor(_actions(ref.model.lower, ref.model.upper, ids, t, dev))
        rows.append(ref.rows.clone().cpu())
    got = ret["got"]
    assert got[0] is None
    for t in range(1, STEPS + 1):                 # call t returns the rows of step t - 1, of BOTH shards
        assert torch.equal(got[t], rows[t - 1]), t
    assert torch.equal(ret["blocking"], rows[STEPS - 1])
    gc = ret["got_copy"]
    assert gc[0] is None
    for i in range(1, 6):                         # copy exchange: call i returns the rows of its previous step
        assert torch.equal(gc[i], rows[STEPS + i - 1]), i
    assert float(rows[3][:, 76].sum()) == N_GLOBAL     # the episode limit fired inside the launches, on every env


NCCL_CHILD = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(pkg)r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%(port)r)
from trex_gym import sharding
from trex_gym.vec_env import TrexVecEnv
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)          # "nccl" IS RCCL on ROCm
assert dist.get_backend() == "nccl"
for join in ("host", "stream"):
    env = TrexVecEnv(64, urdf_path=%(urdf)r, device=dev, max_episode_steps=5, row_buffers=2)
    env.reset_tensor()
    ids = torch.arange(64, device=dev)
    pipe = sharding.PipelinedGather(env.num_envs, env.rows.shape[1], 1, env.rows.dtype, dev)
    prev_rows = None
    for t in range(12):
        env.step_tensor(sharding.synthetic_actions(ids, t, env.model.lower, env.model.upper, seed=0, device=dev))
        got = pipe.push(env.rows, copy=False, wait=False, join=join)          # bench.py's call, through ProcessGroupNCCL
        if t == 0:
            assert got is None
        else:
            torch.cuda.synchronize()
            assert torch.equal(got, prev_rows), (join, t)                     # the rows of the PREVIOUS step, bitwise
        prev_rows = env.rows.clone()
        w = pipe.last_work()
        assert w is None or w.is_completed() in (True, False)                 # the completion query answers on a real NCCL work
    last = pipe.flush()
    torch.cuda.synchronize()
    assert torch.equal(last, prev_rows), join
    assert float(prev_rows[:, 76].sum()) >= 0.0
# the blocking form and the staged form, too
env = TrexVecEnv(32, urdf_path=%(urdf)r, device=dev)
env.reset_tensor()
out = sharding.all_gather_rows(env.rows, 32, 1)
assert out is env.rows
dist.barrier()
dist.destroy_process_group()
print("NCCL_WORLD_OF_ONE_OK")
"""


def test_pipelined_gather_over_rccl_with_a_world_of_one():
    """the first test that touches ProcessGroupNCCL (= RCCL). One rank, one GPU: the in-place pipelined
    gather exactly as bench.py --gpus N runs it (copy=False, wait=False), with the host join and with the stream join,
    returns the previous step's rows bitwise, and the host join's completion query terminates on a real NCCL work handle.
    (More than one rank per device RCCL refuses; the two-rank test above runs over gloo.)"""
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    code = NCCL_CHILD % dict(pkg=os.path.join(ROOT, "trex-gym_amd"), port=str(port), urdf=ASSET_URDF)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "NCCL_WORLD_OF_ONE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert r.returncode == 0
