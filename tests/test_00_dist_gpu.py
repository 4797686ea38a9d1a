"""Multi-GPU path, shard equivalence (VERDICT r1 #1b): two rank processes, each owning the arena block
agarcl_amd.dist.shard_bounds gives it and seeding it with arena_seeds, produce exactly the arenas -- state blobs and the
per-step (reward, done) rows gathered to rank 0 -- that ONE process stepping all arenas produces.  gloo, both ranks on
GPU 0.  This file sorts first on purpose: the rank children are started BEFORE this process makes any GPU call."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "helpers", "shard_worker.py")


def run_ranks(lib, total, steps, world=2):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tempfile.mkdtemp(prefix="agar_shard_")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), str(total), str(steps), lib, out]) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(os.path.join(out, "rank%d.npz" % r)) for r in range(world)]


def run_two_ranks(lib, total, steps):
    return run_ranks(lib, total, steps, 2)


def check_against_single_process(lib, parts, total, steps):
    sys.path.insert(0, os.path.join(HERE, "helpers"))
    import shard_worker
    rows = []
    masses = []
    blobs = shard_worker.run_shard(lib, 0, total, total, steps, lambda t, r, d, m: (rows.append(np.stack([r[:, 0].astype(np.float32), d[:, 0].astype(np.float32)], axis=1)),
                                                                                    masses.append(m.astype(np.int32).reshape(-1, 1))))
    assert parts[0]["lo"] == 0 and parts[-1]["hi"] == total and all(parts[i]["hi"] == parts[i + 1]["lo"] for i in range(len(parts) - 1))
    k = 0
    for part in parts:
        for i in range(int(part["n"])):
            assert np.array_equal(part["blob_%d" % i], blobs[k]), "arena %d differs between the sharded and the single-process run" % k
            k += 1
    assert k == total
    assert np.array_equal(parts[0]["gathered"], np.stack(rows))       # what rank 0 received == the single process's results, every step
    assert np.array_equal(parts[0]["gathered_packed"], np.stack(rows))        # ... through the zero-copy entry point too
    assert np.array_equal(parts[0]["gathered_masses"], np.stack(masses))      # ... and a tensor through TensorGatherer


@pytest.mark.gpu
def test_two_rank_shards_equal_one_rank_run_gpu():
    from agarcl_amd import _capi, build as hip_build
    if not os.path.exists(_capi.HIP_SO):              # (checked by path only: loading the library is left to the children)
        hip_build.build()
    total, steps = 64, 120
    parts = run_two_ranks("hip", total, steps)        # children first: this process has not touched the GPU yet
    check_against_single_process("hip", parts, total, steps)


@pytest.mark.gpu
def test_rccl_branch_on_one_gpu():
    """the "nccl" (= RCCL) backend with a process group of ONE rank on GPU 0: ResultGatherer.pack / gather_packed, TensorGatherer, the all-reduce and
    the object gather bench.py makes -- the calls the driver's multi-GPU runs take, as far as a 1-GPU box can exercise them (a child process: this
    process has not touched the GPU yet)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    p = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "nccl_one_rank.py"), str(port)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "nccl one-rank ok" in p.stdout, (p.stdout[-500:], p.stderr[-1500:])


def test_two_rank_shards_equal_one_rank_run_emulated(emu_lib):
    """the same on the CPU with the test-only host build of the kernel source (world size 2, gloo)"""
    total, steps = 10, 60
    parts = run_two_ranks("emu", total, steps)
    check_against_single_process("emu", parts, total, steps)


@pytest.mark.parametrize("world,total", [(3, 11), (8, 11)])
def test_unequal_shards_world_3_and_8_emulated(emu_lib, world, total):
    """total % world != 0 at world size 3 and 8: shard_bounds hands the first ranks one arena more; ResultGatherer.pack /
    gather_packed and TensorGatherer pad to the largest shard and rank 0 drops the padding -- every arena and every gathered row equals the
    single-process run"""
    steps = 24
    parts = run_ranks("emu", total, steps, world)
    assert sorted(int(p["n"]) for p in parts) == sorted([total // world + (1 if r < total % world else 0) for r in range(world)])
    check_against_single_process("emu", parts, total, steps)
