"""N > 1 path on CPU: arena sharding + the per-step (reward, done) gather to rank 0, world_size 2, gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from agarcl_amd import dist as agdist


def test_shard_bounds_cover_exactly():
    for total in (1, 7, 4096, 32768, 50001):
        for world in (1, 2, 3, 8):
            spans = [agdist.shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_seeds_independent_of_world_size():
    full = agdist.arena_seeds(10000, 0, 64)
    parts = np.concatenate([agdist.arena_seeds(10000, *agdist.shard_bounds(64, 4, r)) for r in range(4)])
    assert np.array_equal(full, parts)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_local = 6
    g = agdist.ResultGatherer(n_local, torch.device("cpu"), depth=2)
    ok = True
    for k in range(5):
        lo = rank * n_local
        rewards = torch.arange(lo, lo + n_local, dtype=torch.float64).reshape(n_local, 1) + 100.0 * k
        dones = ((torch.arange(lo, lo + n_local) + k) % 3 == 0).to(torch.uint8).reshape(n_local, 1)
        slot = g.pack(k, rewards, dones)
        g.wait_all()
        if rank == 0:
            got = g.gathered(slot)
            exp_r = torch.arange(0, world * n_local, dtype=torch.float32) + 100.0 * k
            exp_d = ((torch.arange(0, world * n_local) + k) % 3 == 0).to(torch.float32)
            ok = ok and torch.equal(got[:, 0], exp_r) and torch.equal(got[:, 1], exp_d)
    # zero-copy variant used by bench.py: gather straight from the (engine's) ping-pong buffers
    packed = [torch.zeros((n_local, 2), dtype=torch.float32) for _ in range(2)]
    for k in range(6):
        g.wait_slot(k & 1)
        lo = rank * n_local
        packed[k & 1][:, 0] = torch.arange(lo, lo + n_local, dtype=torch.float32) * 2 + k
        packed[k & 1][:, 1] = float(k % 2)
        g.gather_packed(k & 1, packed[k & 1])
        if k >= 1:
            g.wait_slot((k - 1) & 1)
            if rank == 0:
                got = g.gathered((k - 1) & 1)
                ok = ok and torch.equal(got[:, 0], torch.arange(0, world * n_local, dtype=torch.float32) * 2 + (k - 1))
                ok = ok and bool((got[:, 1] == float((k - 1) % 2)).all())
    g.wait_all()
    # bench.py's pattern: the engine's 64-slot result ring, one asynchronous gather per block of BLK steps into one of two
    # gather buffers, and a flush of the partial last block (BLK = 32: two blocks = the two ring halves; BLK = 8: eight blocks
    # share the two buffers)
    for SLOTS, BLK, A, STEPS in ((64, 32, 5, 75), (64, 8, 3, 101)):
        ring = torch.zeros((SLOTS, A, 2), dtype=torch.float32)
        gb = agdist.ResultGatherer(BLK * A, torch.device("cpu"), depth=2)
        seen = {}
        last = SLOTS - 1
        for step in range(STEPS):                    # not a multiple of the block
            nxt = (last + 1) % SLOTS
            if nxt % BLK == 0:
                b = (nxt // BLK) & 1
                gb.wait_slot(b)                      # the gather that last used this buffer has left (and, BLK = 32, this ring half)
                if rank == 0 and b in seen:
                    blk0 = seen.pop(b)
                    got = gb.gathered(b).reshape(world, BLK, A, 2)
                    for r in range(world):
                        for j in range(BLK):
                            ok = ok and bool((got[r, j, :, 0] == float(1000 * r + blk0 + j)).all())
            ring[nxt, :, 0] = float(1000 * rank + step); ring[nxt, :, 1] = 0.0   # "the engine" writes step `step`
            last = nxt
            if last % BLK == BLK - 1:
                h = last // BLK
                if rank == 0 and (h & 1) in seen:    # (BLK < 32: the buffer's previous block is checked before it is reused)
                    gb.wait_slot(h & 1); blk0 = seen.pop(h & 1)
                    got = gb.gathered(h & 1).reshape(world, BLK, A, 2)
                    for r in range(world):
                        for j in range(BLK):
                            ok = ok and bool((got[r, j, :, 0] == float(1000 * r + blk0 + j)).all())
                gb.gather_packed(h & 1, ring[h * BLK:(h + 1) * BLK].reshape(-1, 2))
                seen[h & 1] = step - BLK + 1
        if last % BLK != BLK - 1:                    # flush
            h = last // BLK
            gb.wait_slot(h & 1); gb.gather_packed(h & 1, ring[h * BLK:(h + 1) * BLK].reshape(-1, 2))
            gb.wait_slot(h & 1)
            if rank == 0:
                got = gb.gathered(h & 1).reshape(world, BLK, A, 2)
                first = STEPS - (last % BLK + 1)
                for r in range(world):
                    for j in range(last % BLK + 1):
                        ok = ok and bool((got[r, j, :, 0] == float(1000 * r + first + j)).all())
        gb.wait_all()
    SLOTS, A = 64, 5
    ring = torch.zeros((SLOTS, A, 2), dtype=torch.float32)
    # bench.py --gather step: one collective per step straight from the ring slot the step wrote, parity-buffered
    gs = agdist.ResultGatherer(A, torch.device("cpu"), depth=2)
    for step in range(7):
        slot = step % SLOTS
        gs.wait_slot(slot & 1)
        ring[slot, :, 0] = float(5000 * rank + step); ring[slot, :, 1] = float(step & 1)
        gs.gather_packed(slot & 1, ring[slot])
        gs.wait_slot(slot & 1)
        if rank == 0:
            got = gs.gathered(slot & 1).reshape(world, A, 2)
            for r in range(world):
                ok = ok and bool((got[r, :, 0] == float(5000 * r + step)).all()) and bool((got[r, :, 1] == float(step & 1)).all())
    # bench.py --gather-obs screen: every step's uint8 frames, one collective in flight
    tg = agdist.TensorGatherer((A, 6, 6, 3), torch.uint8, torch.device("cpu"))
    frames = torch.zeros((A, 6, 6, 3), dtype=torch.uint8)
    for step in range(4):
        tg.wait()
        frames[:] = 10 * rank + step
        tg.gather(frames)
        tg.wait()
        if rank == 0:
            got = tg.gathered().reshape(world, A, 6, 6, 3)
            for r in range(world):
                ok = ok and bool((got[r] == 10 * r + step).all())
    # unequal shards (shard_bounds hands the first ranks one arena more): padded to the largest one, the padding dropped on rank 0
    total = 2 * 4 + 1
    lo, hi = agdist.shard_bounds(total, world, rank)
    gu = agdist.ResultGatherer(hi - lo, torch.device("cpu"), depth=2)
    ok = ok and gu.n_max == 5 and gu.sizes == [5, 4]
    for k in range(4):
        rewards = torch.arange(lo, hi, dtype=torch.float64) * 3 + k
        slot = gu.pack(k, rewards, torch.zeros(hi - lo, dtype=torch.uint8))
        gu.wait_all()
        if rank == 0:
            got = gu.gathered(slot)
            ok = ok and tuple(got.shape) == (total, 2) and torch.equal(got[:, 0], torch.arange(0, total, dtype=torch.float32) * 3 + k)
        own = torch.stack([torch.arange(lo, hi, dtype=torch.float32) + 100 * k, torch.ones(hi - lo)], dim=1)     # the zero-copy entry point
        gu.gather_packed(k & 1, own); gu.wait_slot(k & 1)
        if rank == 0:
            got = gu.gathered(k & 1)
            ok = ok and torch.equal(got[:, 0], torch.arange(0, total, dtype=torch.float32) + 100 * k) and bool((got[:, 1] == 1).all())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put(ok)


def test_result_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() is True
