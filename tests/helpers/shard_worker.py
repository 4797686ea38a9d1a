"""One rank of the shard-equivalence tests (world size 2, 3, 8) (TEST INFRASTRUCTURE; started as a child process).

    python shard_worker.py RANK WORLD PORT TOTAL_ARENAS STEPS LIB OUTDIR        LIB = "hip" | "emu"

Rank r owns the contiguous arena block agarcl_amd.dist.shard_bounds gives it, seeds it with arena_seeds (a function of
the GLOBAL arena index), steps it with the slice of one global action stream, sends every step's (reward, done) to rank
0 through ResultGatherer over gloo -- pack() and the zero-copy gather_packed() -- and every step's masses through TensorGatherer (both pad
unequal shards), and writes its arenas' final state blobs to OUTDIR/rank<r>.npz."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CFG = dict(arena_size=300, num_pellets=300, num_viruses=6, mode=6)


def actions(t, total):
    rng = np.random.RandomState(4000 + t // 4)
    return rng.uniform(-1, 1, size=(total, 1, 2)).astype(np.float32), rng.randint(0, 3, size=(total, 1)).astype(np.int32)


def make_engine(lib, n):
    from agarcl_amd import _capi
    if lib == "emu":
        L = _capi.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "_build", "libagarcl_emu.so")))
        return _capi.BatchedEngine(n, lib=L, **CFG)
    return _capi.BatchedEngine(n, device=0, **CFG)      # both ranks share GPU 0 on a 1-GPU box


def run_shard(lib, lo, hi, total, steps, on_step=None):
    from agarcl_amd import dist as agdist
    eng = make_engine(lib, hi - lo)
    eng.seed(agdist.arena_seeds(900, lo, hi)); eng.reset(reset_ids=True)
    for t in range(steps):
        dxdy, act = actions(t, total)
        eng.set_actions(dxdy[lo:hi], act[lo:hi]); eng.step()
        if on_step:
            on_step(t, eng.rewards(), eng.dones(), eng.masses())
    assert not eng.flags().any()
    blobs = [eng.dump(a) for a in range(hi - lo)]
    eng.close()
    return blobs


def main():
    rank, world, port, total, steps = (int(x) for x in sys.argv[1:6])
    lib, outdir = sys.argv[6], sys.argv[7]
    import torch
    import torch.distributed as dist
    from agarcl_amd import dist as agdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = agdist.shard_bounds(total, world, rank)
    g = agdist.ResultGatherer(hi - lo, torch.device("cpu"), depth=2)
    gz = agdist.ResultGatherer(hi - lo, torch.device("cpu"), depth=2)         # the zero-copy entry point bench.py uses
    tg = agdist.TensorGatherer((hi - lo, 1), torch.int32, torch.device("cpu"))
    got, got_z, got_m = [], [], []

    def on_step(t, rewards, dones, masses):
        slot = g.pack(t, torch.from_numpy(rewards), torch.from_numpy(dones.astype(np.uint8)))
        g.wait_slot(slot)
        packed = torch.stack([torch.from_numpy(rewards[:, 0].astype(np.float32)), torch.from_numpy(dones[:, 0].astype(np.float32))], dim=1)
        gz.gather_packed(t & 1, packed); gz.wait_slot(t & 1)
        tg.gather(torch.from_numpy(masses.astype(np.int32).reshape(-1, 1))); tg.wait()
        if rank == 0:
            got.append(g.gathered(slot).numpy().copy()); got_z.append(gz.gathered(t & 1).numpy().copy()); got_m.append(tg.gathered().numpy().copy())
    blobs = run_shard(lib, lo, hi, total, steps, on_step)
    g.wait_all()
    dist.barrier(); dist.destroy_process_group()
    out = {"lo": lo, "hi": hi, "n": len(blobs)}
    for i, b in enumerate(blobs):
        out["blob_%d" % i] = b
    if rank == 0:
        out["gathered"] = np.stack(got)      # [steps][total][2]
        out["gathered_packed"] = np.stack(got_z); out["gathered_masses"] = np.stack(got_m)
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)


if __name__ == "__main__":
    main()
