"""Env-id sharding across the GPUs of one node and the one collective of the path.

Envs are independent (the reference runs a single env, trex_train.py:44), so the batch shards by
contiguous env-id ranges with no data-path collective. The only exchange is the all-gather of the
[obs | reward | done] row block when a centralised consumer wants the whole batch: `all_gather_rows`
(torch.distributed.all_gather_into_tensor: RCCL over xGMI with backend "nccl" on ROCm, gloo in the
CPU tests). Device-agnostic on purpose: nothing here touches HIP.
"""
import time

import numpy as np
import torch
import torch.distributed as dist


def shard_range(num_envs, rank, world_size):
    """Contiguous [lo, hi) of global env ids owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of %d" % (rank, world_size))
    base, extra = divmod(int(num_envs), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def synthetic_actions(env_ids, step, low, high, seed=0, device="cpu"):
    """Uniform actions in [low, high) keyed by (seed, GLOBAL env id, step): any sharding of the env
    ids produces the same rows. Counter-based: a hash of the key, no RNG state."""
    ids = torch.as_tensor(env_ids, dtype=torch.int64, device=device).reshape(-1, 1)
    low = torch.as_tensor(low, dtype=torch.float32, device=device).reshape(1, -1)
    high = torch.as_tensor(high, dtype=torch.float32, device=device).reshape(1, -1)
    j = torch.arange(low.shape[1], dtype=torch.int64, device=device).reshape(1, -1)
    x = (ids * 1000003 + int(step)) * 1000033 + j * 7919 + int(seed) * 104729 + 12345
    # splitmix-style integer mix in 31-bit arithmetic (exact on every backend)
    m = (1 << 31) - 1
    x = x & m
    for mul, sh in ((1103515245, 15), (214013, 13), (69069, 16)):
        x = (x * mul + 12345) & m
        x = x ^ (x >> sh)
    u = (x & ((1 << 24) - 1)).to(torch.float32) / float(1 << 24)
    return low + (high - low) * u


def pack_rows(obs, reward, done, out=None):
    """[n, 3J] obs, [n] reward, [n] done (any dtype) -> the [n, 3J+2] f32 row block the exchange carries
    (obs + reward + done in one message). The HIP step writes this layout directly
    (trex_batch_step_rows); this is the host-side / CPU-test equivalent."""
    n, c = obs.shape
    if out is None:
        out = torch.empty(n, c + 2, dtype=torch.float32, device=obs.device)
    out[:, :c] = obs
    out[:, c] = reward
    out[:, c + 1] = done.to(torch.float32)
    return out


def split_rows(rows, obs_cols=None):
    """[N, 3J+2] row block -> (obs [N, 3J], reward [N], done [N] bool) views. obs_cols = 3J for wider rows (the
    GPU env's 3J+5 form also carries the three penalties behind done: rows[:, 3J+2:3J+5])."""
    c = rows.shape[1] - 2 if obs_cols is None else int(obs_cols)
    return rows[:, :c], rows[:, c], rows[:, c + 1] != 0


def all_gather_rows(local, global_rows, world_size, group=None, out=None):
    """Concatenate per-rank row blocks [n_r, C] into [global_rows, C] on every rank. Shards may
    differ by one row, so blocks are padded to the largest shard for the collective."""
    if world_size == 1:
        return local
    cols = local.shape[1:]
    per = -(-int(global_rows) // int(world_size))
    send = local
    if local.shape[0] != per:
        send = torch.zeros((per,) + tuple(cols), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    even = world_size * per == global_rows
    if even and out is not None and tuple(out.shape) == (global_rows,) + tuple(cols):
        buf = out   # gather straight into the caller's buffer: no allocation, no copy per step
    else:
        buf = torch.empty((world_size * per,) + tuple(cols), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, send.contiguous(), group=group)
    if even:
        return buf
    pieces = []
    for r in range(world_size):
        lo, hi = shard_range(global_rows, r, world_size)
        pieces.append(buf[r * per: r * per + (hi - lo)])
    return torch.cat(pieces, 0)


class PipelinedGather:
    """All-gather of the per-rank observation rows that overlaps the NEXT env step: the collective is launched
    asynchronously on the communicator's stream and the call returns the block gathered ONE call earlier. A
    centralised consumer therefore sees observations one step late, and no step waits for xGMI.
    Equal shards only (the bench / trainer case); ragged shards go through all_gather_rows.

    Two forms. copy=True: the rows are first copied to one of two staging blocks on the caller's stream; the caller
    may overwrite its rows at once. copy=False: the rows are gathered IN PLACE from a producer that writes two row
    blocks in turn (TrexVecEnv(row_buffers=2)): the gather G(t) of step t's block is still reading it while step
    t+1 fills the other one, and step t+2 rewrites it. ORDERING RULE of the in-place form: push(t+1) - which runs
    between the launches of step t+1 and step t+2 - makes the caller's stream wait for G(t), always, whatever
    `wait` says. That is the one cross-stream wait per step, and it is the one that orders step t+2 behind the last
    reader of the block it rewrites. (Round 2 waited for G(t) at the top of push(t+2), AFTER step t+2 had been
    enqueued: a write-after-read race whenever a gather outlives the following step.)

    join="host": the same ordering, enforced by the HOST - push(t+1) returns only when gather t has completed
    (the work handle's completion query), so the caller's stream gets no wait of its own. On this ROCm build a
    hipStreamWaitEvent on the stream that carries the step launches costs the chain of launches far more than the
    wait itself (scripts/sync_cost_probe.py, one MI355X: back-to-back step launches 0.360 ms per step; with the
    fork to a side stream 0.383; with the compute stream ALSO waiting for an event of the side stream 0.523,
    whichever earlier step's event it is; with the host waiting instead 0.364). The host then runs at most two
    steps ahead of the GPU, which a 0.35 ms step launch does not notice."""

    def __init__(self, rows_local, cols, world_size, dtype, device, group=None):
        self.world, self.group = int(world_size), group
        self.stage = [torch.empty(rows_local, cols, dtype=dtype, device=device) for _ in range(2)]
        self.out = [torch.empty(rows_local * self.world, cols, dtype=dtype, device=device) for _ in range(2)]
        self.work = [None, None]
        self.k = 0

    @staticmethod
    def _host_wait(work):
        while not work.is_completed():       # (an event query; the gather it waits for ended a step launch ago)
            time.sleep(0)

    def push(self, local, copy=True, wait=True, join="stream"):
        """Launch the gather of `local`; returns the previous call's gathered rows (None on the first).
        join: "stream" - waits are stream waits on the caller's stream (work.wait()); "host" - the host waits for the
        completion of the gather instead (class note). The ordering guarantees are the same.
        copy=False: `local` is gathered in place; the caller must not write it before the NEXT push() has returned
        (that push orders the caller's stream behind this gather) - a producer alternating between two row blocks
        and pushing after every step satisfies that by construction.
        wait=False (staged form only): the returned rows are NOT ordered before the caller's stream; a consumer on
        another stream orders itself behind `last_work()`. The in-place form always orders them (see the class note)."""
        k, prev = self.k, 1 - self.k
        src = local
        if copy:
            # stage[k] was last read by the gather of two calls ago (complete unless wait=False skipped its wait)
            if self.work[k] is not None:
                self._host_wait(self.work[k]) if join == "host" else self.work[k].wait()
            self.stage[k].copy_(local)
            src = self.stage[k]
        # (out[k], last written by the gather of two calls ago, is rewritten behind it: the communicator's stream is
        # in order; a wait=False consumer still reading out[k] holds that gather's work handle and must be done with it)
        self.work[k] = dist.all_gather_into_tensor(self.out[k], src, group=self.group, async_op=True)
        self.k = prev
        if self.work[prev] is None:
            return None
        if wait or not copy:
            # in place: the block the producer's NEXT step rewrites is the one work[prev] is reading
            self._host_wait(self.work[prev]) if join == "host" else self.work[prev].wait()
        return self.out[prev]

    def last_work(self):
        """Work handle of the gather whose rows the last push() returned (None before the second push)."""
        return self.work[self.k]

    def flush(self):
        """Wait for everything in flight; returns the most recent gathered rows."""
        last = None
        for k in (self.k, 1 - self.k):
            if self.work[k] is not None:
                self.work[k].wait()
                last = self.out[k]
        return last
