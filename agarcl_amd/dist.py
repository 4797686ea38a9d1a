"""Multi-GPU sharding: arenas are independent, so they shard contiguously across ranks with NO
data-path collective inside the tick.  The only exchange is gathering per-step results (reward +
done, 8 bytes per arena) to rank 0 -- RCCL over xGMI on GPUs ("nccl" backend), gloo in CPU tests.

There is no counterpart in the reference (single process, one arena); the closest thing is its thread
pool running independent engines (/root/reference/agario/bots/benchmark.cpp:146-168).
"""


def shard_bounds(total, world_size, rank):
    """Contiguous block [lo, hi) of `total` arenas owned by `rank` (first ranks take the remainder)."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def arena_seeds(base_seed, lo, hi):
    """Per-arena seeds depend on the GLOBAL arena index only, so results are independent of how many
    GPUs the job is sharded over."""
    import numpy as np
    return (base_seed + np.arange(lo, hi, dtype=np.int64)).astype(np.uint32)


class ResultGatherer:
    """Double-buffered asynchronous gather of (reward, done) rows to rank 0.

    pack(k, rewards, dones) copies this step's results into slot k % depth on the current stream and
    starts a gather; the collective overlaps the next step's kernel.  Buffers are float32 [rows, 2].
    Works with any torch.distributed backend (nccl == RCCL on ROCm, gloo on CPU).

    Unequal shards (shard_bounds() hands the first ranks one arena more when total % world != 0): dist.gather needs equally sized
    contributions, so every rank sends `n_max` rows -- the largest shard -- and rank 0 drops the padding rows again in gathered().  A rank
    whose shard is the largest sends zero-copy; a smaller one goes through its padded staging buffer (one small device copy)."""

    def __init__(self, n_local, device, depth=2, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.depth = depth
        self.n_local = int(n_local)
        sizes = [None] * self.world
        dist.all_gather_object(sizes, self.n_local, group=group)
        self.sizes = [int(x) for x in sizes]
        self.n_max = max(self.sizes)
        self.send = [torch.zeros((self.n_max, 2), dtype=torch.float32, device=device) for _ in range(depth)]
        self.recv = None
        if self.rank == 0:
            self.recv = [[torch.zeros((self.n_max, 2), dtype=torch.float32, device=device) for _ in range(self.world)]
                         for _ in range(depth)]
        self.work = [None] * depth
        self.reset_stats()

    def reset_stats(self):
        """diagnostics: host seconds spent waiting for collectives, collectives started, payload bytes this rank sent"""
        self.wait_s, self.calls, self.bytes_sent = 0.0, 0, 0

    def _wait(self, s):
        import time
        t = time.perf_counter()
        self.work[s].wait()
        self.wait_s += time.perf_counter() - t
        self.work[s] = None

    def pack(self, k, rewards, dones):
        s = k % self.depth
        if self.work[s] is not None:
            self._wait(s)
        buf = self.send[s]
        buf[:self.n_local, 0].copy_(rewards.reshape(-1))
        buf[:self.n_local, 1].copy_(dones.reshape(-1))
        self.work[s] = self.dist.gather(buf, self.recv[s] if self.rank == 0 else None, dst=0, group=self.group,
                                        async_op=True)
        self.calls += 1; self.bytes_sent += self.n_local * 2 * buf.element_size()
        return s

    def gather_packed(self, slot, packed):
        """Zero-copy variant: `packed` is the engine's own ping-pong buffer of step parity `slot` ([n_local, 2] f32,
        agarcl_packed_dev).  wait_slot(slot) must be called before the engine overwrites that parity again.  A gather still
        in flight on this buffer is waited for first.  (A shard smaller than the largest one is staged into its padded buffer.)"""
        self.wait_slot(slot)
        if packed.shape[0] != self.n_max:
            self.send[slot][:packed.shape[0]].copy_(packed)
            packed = self.send[slot]
        self.work[slot] = self.dist.gather(packed, self.recv[slot] if self.rank == 0 else None, dst=0, group=self.group,
                                           async_op=True)
        self.calls += 1; self.bytes_sent += self.n_local * 2 * packed.element_size()

    def wait_slot(self, slot):
        if self.work[slot] is not None:
            self._wait(slot)

    def wait_all(self):
        for i, w in enumerate(self.work):
            if w is not None:
                self._wait(i)

    def gathered(self, slot):
        """rank 0: float32 [sum of the shards' rows, 2] of the given slot (after its work completed), padding rows dropped."""
        if self.rank != 0:
            return None
        return self.torch.cat([r[:n] for r, n in zip(self.recv[slot], self.sizes)], dim=0)


class TensorGatherer:
    """Per-step gather of one tensor per rank to rank 0 -- the observation path of a learner that lives on
    rank 0 (uint8 screen frames are 21 KB per arena; the int32 grid tensor, 512 KB per arena, is better left
    on the producing GPU).  One collective in flight: gather(t) starts it, wait() completes it before `t` is reused.
    `shape` is THIS rank's shape, first axis = its arenas: with unequal shards (total % world != 0) every rank sends the largest shard's rows
    (a smaller one through a padded staging buffer) and rank 0 drops the padding again, as ResultGatherer does."""

    def __init__(self, shape, dtype, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        shape = tuple(int(x) for x in shape)
        sizes = [None] * self.world
        dist.all_gather_object(sizes, shape[0], group=group)
        self.sizes = [int(x) for x in sizes]
        self.n_local, self.n_max = shape[0], max(self.sizes)
        padded = (self.n_max,) + shape[1:]
        self.stage = torch.zeros(padded, dtype=dtype, device=device) if self.n_local != self.n_max else None
        self.recv = [torch.empty(padded, dtype=dtype, device=device) for _ in range(self.world)] if self.rank == 0 else None
        self.work = None
        self.reset_stats()

    def reset_stats(self):
        self.wait_s, self.calls, self.bytes_sent = 0.0, 0, 0

    def gather(self, tensor):
        self.wait()
        self.calls += 1; self.bytes_sent += tensor.numel() * tensor.element_size()
        if self.stage is not None:
            self.stage[:self.n_local].copy_(tensor)
            tensor = self.stage
        self.work = self.dist.gather(tensor, self.recv, dst=0, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            import time
            t = time.perf_counter()
            self.work.wait()
            self.wait_s += time.perf_counter() - t
            self.work = None

    def gathered(self):
        """rank 0: [sum of the shards' rows, ...] after wait(), padding rows dropped"""
        if self.rank != 0:
            return None
        if self.n_max * self.world == sum(self.sizes):
            return self.torch.cat(self.recv, dim=0)
        return self.torch.cat([r[:n] for r, n in zip(self.recv, self.sizes)], dim=0)
