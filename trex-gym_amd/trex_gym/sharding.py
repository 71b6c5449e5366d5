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

    HOST_JOIN_TIMEOUT_S = 120.0     # a gather takes microseconds; a peer that never arrives must not hang the loop for ever

    @classmethod
    def _host_wait(cls, work):
        """Spin on the work handle's completion query (an event query; the gather it waits for ended a step launch ago)
        - with a deadline, and treating a work that completed WITH AN ERROR as an error: `is_completed()` of a c10d work
        also turns true once an exception is set, and a failed or aborted gather must not hand out (or let the next step
        overwrite) rows that never arrived."""
        t0 = None
        while not work.is_completed():
            now = time.monotonic()
            if t0 is None:
                t0 = now
            elif now - t0 > cls.HOST_JOIN_TIMEOUT_S:
                raise TimeoutError("PipelinedGather: the all-gather did not complete within %.0f s (a peer rank is stalled or gone)"
                                   % cls.HOST_JOIN_TIMEOUT_S)
            time.sleep(0)
        # (Work.exception() cannot be converted to Python on this torch build - "Unregistered type exception_ptr";
        # is_success() is the query that works: gloo implements it, ProcessGroupNCCL may refuse it - there the watchdog
        # aborts the process on an asynchronous error, so a refusal to answer is not taken as a failure)
        try:
            ok = bool(work.is_success())
        except Exception:  # noqa: BLE001
            ok = True
        if not ok:
            raise RuntimeError("PipelinedGather: the all-gather completed with an error (peer failure or aborted communicator)")

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


class CopyGather:
    """The same exchange WITHOUT a collective kernel: every rank copies its row block straight into every peer's
    gathered buffer (peer memory mapped through HIP IPC; xGMI is point-to-point, so an all-gather IS world-1 peer
    writes per rank) with plain device-to-device copies on a side stream - copy-engine work, no workgroup beside the
    step launch. Why: the step launch of 4096 envs holds every CU's wave slots and most of its LDS; a collective
    KERNEL on another hardware queue slowed the step launches by 9 - 14 % in the one-GPU rehearsals with RCCL
    (DESIGN.md 7), a device-to-device copy on a side stream by 1.6 - 1.9 % (scripts/overlap_probe.py). RCCL stays the
    default the north star names; this is the variant `bench.py` runs under TREX_BENCH_GATHER=copy.

    In-place, pipelined like PipelinedGather(copy=False): the producer alternates between two row blocks and calls
    push() after every step; push(t) returns the rows of step t-1 (None on the first call).
      * producer side: the copies of step t read block t & 1 while step t+1 fills the other; step t+2 rewrites it, so
        push(t+1) returns only when the HOST has seen the copies of step t complete (an event query - no wait on the
        stream that carries the step launches, same reasoning as PipelinedGather's host join);
      * across ranks: a rank's out[k] holds step t of EVERY rank once every rank's copies of step t are complete.
        sync="barrier": push(t+1) then passes a barrier of `signal_group` (a CPU-side gloo group; the GPU is a step
        ahead and does not notice) before it returns out[k]; sync="none": no cross-rank signal - for a caller that
        only wants the traffic (the bench's timed region: nothing consumes the rows there).
      The rows returned by push(t+1) are rewritten by the peers' copies of step t+2, which start when THEIR step t+2
      has ended: consume them (or copy them out) before the next push.
    Equal shards only. Needs one process per rank with peer access between the devices (one node)."""

    def __init__(self, rows_local, cols, world_size, rank, dtype, device, group=None, signal_group=None, sync="barrier"):
        from torch.multiprocessing.reductions import reduce_tensor
        self.world, self.rank, self.n = int(world_size), int(rank), int(rows_local)
        self.sync, self.signal_group = sync, (signal_group if signal_group is not None else group)
        dev = torch.device(device)
        self.out = [torch.zeros(self.n * self.world, cols, dtype=dtype, device=dev) for _ in range(2)]
        mine = [reduce_tensor(o) for o in self.out]          # (rebuild function, IPC handle + geometry): picklable
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.signal_group)
        self.peer_out = {}
        for p in range(self.world):
            if p != self.rank:
                self.peer_out[p] = [fn(*a) for fn, a in everyone[p]]
        self.side = torch.cuda.Stream(device=dev)
        self.ev_step = [torch.cuda.Event(), torch.cuda.Event()]
        self.ev_copy = [torch.cuda.Event(), torch.cuda.Event()]
        self.pushed = [False, False]
        self.k = 0
        dist.barrier(group=self.signal_group)       # nobody writes a peer's buffer before every rank has mapped them all

    HOST_JOIN_TIMEOUT_S = PipelinedGather.HOST_JOIN_TIMEOUT_S

    def _host_wait(self, ev):
        t0 = None
        while not ev.query():
            now = time.monotonic()
            if t0 is None:
                t0 = now
            elif now - t0 > self.HOST_JOIN_TIMEOUT_S:
                raise TimeoutError("CopyGather: the peer copies did not complete within %.0f s" % self.HOST_JOIN_TIMEOUT_S)
            time.sleep(0)

    def push(self, local):
        k, prev = self.k, 1 - self.k
        lo, hi = self.rank * self.n, (self.rank + 1) * self.n
        cur = torch.cuda.current_stream(local.device)
        self.ev_step[k].record(cur)                  # the step that filled `local` ends here
        self.side.wait_event(self.ev_step[k])        # (a wait on the SIDE stream: the step launches' stream gets none)
        with torch.cuda.stream(self.side):
            self.out[k][lo:hi].copy_(local, non_blocking=True)
            for d in range(1, self.world):           # every rank starts with a different peer: no two writers on one link at once
                p = (self.rank + d) % self.world
                self.peer_out[p][k][lo:hi].copy_(local, non_blocking=True)
            self.ev_copy[k].record(self.side)
        self.pushed[k] = True
        self.k = prev
        if not self.pushed[prev]:
            return None
        self._host_wait(self.ev_copy[prev])          # before the producer's next step rewrites block `prev`
        if self.sync == "barrier":
            dist.barrier(group=self.signal_group)    # every rank's copies of that step are complete: out[prev] is whole
        return self.out[prev]

    def flush(self):
        """Wait for the copies in flight on every rank; returns the most recent gathered rows."""
        torch.cuda.current_stream().synchronize()
        self.side.synchronize()
        dist.barrier(group=self.signal_group)
        return self.out[1 - self.k] if self.pushed[1 - self.k] else None
