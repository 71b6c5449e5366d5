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


# ---------------------------------------------------------------- data-parallel trainer
DP_D, DP_A, DP_N, DP_MB, DP_NMB = 75, 25, 2048, 512, 4


def _dp_data(rank, theta0):
    """a rank's share of a rollout: seeded by the rank, drawn around the common policy theta0"""
    from trex_gym import _capi
    from trex_gym.ppo import MlpPolicy
    dev = torch.device("cuda", 0)
    kern = _capi.Policy(64, DP_D, DP_A, 64, 0)
    pol = MlpPolicy(kern.layout, kern.param_count, dev)
    with torch.no_grad():
        pol.theta.copy_(theta0.to(dev))
    g = torch.Generator(device=dev).manual_seed(100 + rank)
    obs = torch.randn(DP_N, DP_D, device=dev, generator=g).clamp(-10, 10)
    with torch.no_grad():
        d = pol.dist(obs)
        act = d.loc + d.scale * torch.randn(DP_N, DP_A, device=dev, generator=g)
        logp0 = d.log_prob(act).sum(-1) + 0.3 * torch.randn(DP_N, device=dev, generator=g)
        val0 = pol.value(obs) + 0.3 * torch.randn(DP_N, device=dev, generator=g)
    adv = 2 * torch.randn(DP_N, device=dev, generator=g) + 0.5 + rank
    ret = val0 + torch.randn(DP_N, device=dev, generator=g)
    perm = torch.randperm(DP_N, device=dev, generator=g)
    return kern, pol, (obs, act, logp0, val0, adv, ret), perm


def _dp_rank(rank, world, port, theta0, ret_dict):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trex_gym import ppo
    kern, pol, srcs, perm = _dp_data(rank, theta0)
    dev = pol.theta.device
    m, v = torch.zeros_like(pol.theta), torch.zeros_like(pol.theta)
    stats = torch.zeros(DP_NMB, 2, device=dev)
    sums = torch.zeros(2, device=dev)
    ppo.dp_minibatch_stats(srcs[4], perm, DP_NMB, DP_MB, world, stats)
    for i in range(DP_NMB):                 # one epoch of data-parallel minibatch steps
        ppo.dp_minibatch_step(kern, pol.theta, pol.grad, m, v, srcs, perm, i * DP_MB, DP_MB, stats[i], world, None,
                              0.2, 0.01, 0.5, 3e-4, 0.5, sums)
    dist.all_reduce(sums)
    ret_dict[rank] = (pol.theta.detach().cpu(), stats.cpu(), sums.cpu())
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_minibatch_steps_equal_the_single_process_update():
    """Two rank processes (sharing the GPU; gloo carries the all-reduce) each hold their own share of a rollout and their own
    permutation. One epoch of data-parallel minibatch steps - gradient / ranks, all-reduce, clip + Adam - leaves BOTH ranks with
    the same parameters, and those are the parameters a single process gets from the concatenated minibatches (rank 0's
    samples followed by rank 1's) through trex_policy_minibatch_step: to 1e-3 of a step where the gradient is not negligible,
    2 % of a step everywhere (Adam's first steps divide by |g| + eps: the tolerance of the single-GPU test)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
    from trex_gym import _capi
    from trex_gym.ppo import MlpPolicy
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    k0 = _capi.Policy(64, DP_D, DP_A, 64, 0)
    p0 = MlpPolicy(k0.layout, k0.param_count, dev)
    with torch.no_grad():
        p0.theta.add_(0.05 * torch.randn_like(p0.theta))
    theta0 = p0.theta.detach().cpu().clone()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_rank, args=(2, port, theta0, ret), nprocs=2, join=True)
    th0, st0, sums0 = ret[0]
    th1, st1, sums1 = ret[1]
    assert torch.equal(th0, th1) and torch.equal(st0, st1)          # every rank applied the same update
    # the single process: the same samples, concatenated minibatch by minibatch
    parts = [_dp_data(r, theta0) for r in range(2)]
    cat = [torch.cat([parts[0][2][j], parts[1][2][j]]) for j in range(6)]
    idx = []
    for i in range(DP_NMB):
        idx += [parts[0][3][i * DP_MB:(i + 1) * DP_MB], DP_N + parts[1][3][i * DP_MB:(i + 1) * DP_MB]]
    perm = torch.cat(idx)
    kern, pol = parts[0][0], parts[0][1]
    m, v = torch.zeros_like(pol.theta), torch.zeros_like(pol.theta)
    stats = torch.zeros(DP_NMB, 2, device=dev)
    sums = torch.zeros(2, device=dev)
    kern.minibatch_stats(cat[4], perm, DP_NMB, 2 * DP_MB, stats)
    torch.testing.assert_close(st0.to(dev), stats, rtol=2e-6, atol=2e-6)
    gbuf = torch.zeros_like(pol.theta)
    for i in range(DP_NMB):
        kern.minibatch_step(pol.theta, gbuf, m, v, *cat, perm, i * 2 * DP_MB, 2 * DP_MB, stats[i], cliprange=0.2, ent_coef=0.01,
                            vf_coef=0.5, lr=3e-4, eps=1e-5, max_grad_norm=0.5, loss_sums=sums)
    want = pol.theta.detach().cpu().double()
    got = th0.double()
    step = (want - theta0.double())
    assert float(step.abs().max()) > 0.5 * 3e-4 * DP_NMB * 0.5          # the parameters really moved
    err = (got - want).abs()
    assert float(err.max()) <= 0.02 * 3e-4 * DP_NMB
    big = gbuf.cpu().abs() > 1e-3 * float(gbuf.abs().max())
    assert float(err[big].max()) <= 1e-3 * 3e-4 * DP_NMB + 1e-7
    torch.testing.assert_close(sums0.to(dev), sums, rtol=1e-4, atol=1e-5)


def _dp_ppo_rank(rank, world, port, ret_dict):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from trex_gym.ppo import PPO
    from trex_gym.vec_env import TrexVecEnv
    env = TrexVecEnv(128, urdf_path=ASSET_URDF, device=torch.device("cuda", 0), rank=rank, world_size=world, max_episode_steps=16)
    agent = PPO(env, nsteps=8, nminibatches=2, noptepochs=2, seed=3)
    hist = agent.learn(3 * 8 * 128, log=None)
    st = agent.kern.get_stats()
    ret_dict[rank] = (agent.policy.theta.detach().cpu(), st["obs_mean"].copy(), st["obs_var"].copy(), float(st["obs_count"]),
                      float(st["ret_var"]), [h["policy_loss"] for h in hist], agent.total_env_steps,
                      agent.b_act[0, :2].cpu())
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_ppo_keeps_its_replicas_identical():
    """trex_gym.ppo.PPO on an env-sharded TrexVecEnv (two ranks x 64 envs, sharing the GPU over gloo): after three updates
    both ranks hold the SAME parameters and the SAME VecNormalize statistics (gradient all-reduce per minibatch step,
    moments merged per rollout), report the same losses, count the global env-steps - and explored with different noise."""
    import numpy as np
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_ppo_rank, args=(2, port, ret), nprocs=2, join=True)
    a, b = ret[0], ret[1]
    assert torch.equal(a[0], b[0]) and torch.isfinite(a[0]).all()
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3] and a[4] == b[4]
    assert abs(a[3] - (1e-4 + 128 * (1 + 3 * 8))) < 1e-6           # the reset's observation + three rollouts, of ALL 128 envs
    assert a[5] == b[5] and len(a[5]) == 3 and a[6] == b[6] == 3 * 8 * 128
    assert not torch.equal(a[7], b[7])                             # different exploration noise per rank
