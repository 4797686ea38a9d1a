"""Sub-batch pipelining (include/agarcl_batch.h agarcl_pipe_*): k contiguous arena ranges of one job, each a complete engine on a HIP stream
of its own -- the reference's vectorised runner lets every engine run ahead on its own pool thread and waits once at the end
(/root/reference/agario/bots/benchmark.cpp:149-167).  Arenas never interact and seeds go by the global arena index, so the pipe must compute,
arena for arena and bit for bit, what ONE engine over all arenas computes in lock-step -- and what the oracle computes."""
import numpy as np
import pytest

from lockstep import policy
from oracle import blob


def _actions(A, n, t, sticky=1):
    dxdy = np.zeros((A, n, 2), np.float32); act = np.zeros((A, n), np.int32)
    for a in range(A):
        dxdy[a], act[a] = policy(1 + 7919 * a, t, n, True, sticky)
    return dxdy, act


def _pipe_vs_single(lib, cfg, A, k, steps, seeds=None, base=0):
    from agarcl_amd import _capi
    one = _capi.BatchedEngine(A, lib=lib, **cfg)
    pipe = _capi.PipelinedEngine(A, k, lib=lib, **cfg)
    assert pipe.sub_batches == k and sum(n for _, n in pipe.ranges) == A and pipe.ranges[0][0] == 0
    assert all(pipe.ranges[j][0] + pipe.ranges[j][1] == pipe.ranges[j + 1][0] for j in range(k - 1))
    # as created (default seeds 5489 + global index, one reset): already the same arenas
    for j, (lo, n) in enumerate(pipe.ranges):
        for a in (0, n - 1):
            assert np.array_equal(pipe.parts[j].dump(a), one.dump(lo + a)), "as created: sub-batch %d arena %d" % (j, a)
    if seeds is not None or base:
        one.seed(seeds, base); pipe.seed(seeds, base)
    one.reset(reset_ids=True); pipe.reset(reset_ids=True)
    n_ag = cfg.get("num_agents", 1)
    for t in range(steps):
        dxdy, act = _actions(A, n_ag, t, sticky=4)
        one.set_actions(dxdy, act); one.step()
        for j, (lo, n) in enumerate(pipe.ranges):        # every sub-batch's step is enqueued before any of them is waited for
            pipe.parts[j].set_actions(dxdy[lo:lo + n], act[lo:lo + n]); pipe.parts[j].step()
    pipe.sync(); one.sync()
    return one, pipe


@pytest.mark.parametrize("k,A", [(2, 9), (3, 10), (4, 4)])
def test_pipe_equals_one_engine_on_the_emulation(emu_lib, k, A):
    cfg = dict(arena_size=200, num_pellets=300, num_viruses=4, mode=6)
    seeds = (np.arange(A, dtype=np.uint32) * 13 + 5).astype(np.uint32) if k == 3 else None
    one, pipe = _pipe_vs_single(emu_lib, cfg, A, k, 25, seeds=seeds, base=0 if k == 3 else 4242)
    for j, (lo, n) in enumerate(pipe.ranges):
        for a in range(n):
            assert np.array_equal(pipe.parts[j].dump(a), one.dump(lo + a)), "sub-batch %d arena %d" % (j, a)
        assert np.array_equal(pipe.parts[j].rewards(), one.rewards()[lo:lo + n])
    assert pipe.concurrent == k
    pipe.close(); one.close()


def test_pipe_argument_errors(emu_lib):
    from agarcl_amd import _capi
    for A, k in ((4, 0), (4, 5), (100, 65)):
        with pytest.raises(_capi.AgarclError):
            _capi.PipelinedEngine(A, k, lib=emu_lib, arena_size=200, num_pellets=100)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 3])
def test_pipe_streams_really_overlap(hip_engine_cls, k):
    """the sub-batch streams were verified at creation to execute concurrently (distinct hardware queues): with the default 4 hardware
    queues of the HIP runtime, 3 sub-batches beside the default stream always fit"""
    from agarcl_amd import _capi
    import torch
    torch.zeros(1, device="cuda")          # (the default stream exists, as in any torch program)
    pipe = _capi.PipelinedEngine(512, k, arena_size=300, num_pellets=300, num_viruses=3, mode=6)
    assert pipe.concurrent == k, "only %d of %d sub-batch streams run concurrently" % (pipe.concurrent, k)
    assert len({p.stream() for p in pipe.parts}) == k
    pipe.close()


@pytest.mark.gpu
def test_pipe_every_arena_of_4096_equals_lockstep_and_oracle(hip_engine_cls, oracle_lib):
    """BASELINE configs[2] (the full rule set, mode 6) at 4096 arenas as two sub-batches: after 40 steps EVERY arena is bit-equal to the
    lock-step engine's, and sampled arenas (incl. the range boundary) equal the oracle driven with the same seeds and actions"""
    cfg = dict(arena_size=1000, num_pellets=1000, num_viruses=25, mode=6)
    A, k, steps = 4096, 2, 40
    one, pipe = _pipe_vs_single(None, cfg, A, k, steps, base=10000)
    diff = []
    for j, (lo, n) in enumerate(pipe.ranges):
        for a in range(n):
            if not np.array_equal(pipe.parts[j].dump(a), one.dump(lo + a)):
                diff.append(lo + a)
    assert not diff, "%d arenas differ from the lock-step run, first %s" % (len(diff), diff[:8])
    assert not one.flags().any()
    for a in (0, 1, 2047, 2048, 2049, 3000, 4095):
        o = oracle_lib.OraEnv(**cfg); o.seed(10000 + a); o.reset(True)
        for t in range(steps):
            dd, aa = policy(1 + 7919 * a, t, 1, True, 4)
            o.take_actions(dd, aa); o.step()
        j = 0 if a < pipe.ranges[1][0] else 1
        d = blob.diff(o.dump(), pipe.parts[j].dump(a - pipe.ranges[j][0]), 0.0)
        assert not d, "arena %d vs oracle: %s" % (a, d)
        o.close()
    pipe.close(); one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0, dt=1.0 / 60),      # C1's population
                                 dict(arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)], ids=["C1", "C2"])           # the quiet path (k_fused)
def test_pipe_equals_lockstep_other_workloads(hip_engine_cls, cfg):
    A, k = 1000, 3                                         # ragged: 334 + 333 + 333
    one, pipe = _pipe_vs_single(None, cfg, A, k, 30, base=77)
    for j, (lo, n) in enumerate(pipe.ranges):
        for a in range(n):
            assert np.array_equal(pipe.parts[j].dump(a), one.dump(lo + a)), "arena %d" % (lo + a)
    pipe.close(); one.close()


@pytest.mark.gpu
def test_pipelined_vec_environment_send_recv(hip_engine_cls):
    """PipelinedVecEnvironment: full-batch step == per-sub-batch send / recv == the lock-step VecEnvironment, with CUDA action tensors that
    are produced on torch's current stream right before they are used (the ordering is the pipe's job: agarcl_stream_wait / _signal)"""
    import torch
    from agarcl_amd.vec_env import PipelinedVecEnvironment, VecEnvironment
    cfg = dict(arena_size=300, num_pellets=300, num_viruses=5, mode_number=6)
    A, k = 600, 2
    one = VecEnvironment(A, **cfg); one.seed(base_seed=9); one.reset()
    pipe = PipelinedVecEnvironment(A, k, **cfg); pipe.seed(base_seed=9); pipe.reset()
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for t in range(24):
        dxdy = torch.rand((A, 1, 2), generator=g, device="cuda") * 2 - 1       # fresh tensors on the current stream, every step
        act = torch.randint(0, 3, (A, 1), generator=g, device="cuda", dtype=torch.int32)
        one.take_actions(dxdy, act); one.step()
        if t % 2 == 0:
            parts = pipe.step(dxdy, act)
        else:
            for j in (1, 0):                                 # any order
                lo, n = pipe.ranges[j]
                pipe.send(j, dxdy[lo:lo + n], act[lo:lo + n])
            parts = [pipe.recv(j) for j in range(k)]
        got = torch.cat([p.rewards for p in parts])          # (on the current stream: recv ordered it after the sub-batches)
        assert torch.equal(got, one.rewards), "step %d" % t
        assert torch.equal(torch.cat([p.masses for p in parts]), one.masses)
    grids = [pipe.parts[j].grid_obs(32) for j in range(k)]     # each on its sub-batch's own stream ...
    for j in range(k):
        pipe.recv(j)                                         # ... which the current stream now waits for
    assert torch.equal(torch.cat(grids), one.grid_obs(32))
    pipe.close(); one.close()
