"""include/agarcl_vec.h: agarcl_vec_step = step + episode bookkeeping + same-step auto-reset + observation in one host call.  Its bookkeeping
is what /root/reference/gym_agario/AgarioEnv.py:105-132 does per env on the host (done = engine flag or the episodic cut-off compared before
the step is counted, step counter, the reset a user performs after a finished episode); here it is checked against exactly that, written
out in Python around the plain C ABI (set_actions / step / rewards / dones / masked reset) on a second engine -- on the CPU through the
test-only emulation of the kernel source, on the GPU through the product library."""
import ctypes as C

import numpy as np
import pytest

CASES = {
    # one agent, cut-off every 5 steps (every arena ends together: the common case of the paper's episodic tasks)
    "cutoff": (dict(num_agents=1, arena_size=120, num_pellets=200, num_viruses=2, mode=0), dict(number_steps=5, episodic=1), 40),
    # continuing env: the cut-off never applies
    "continuing": (dict(num_agents=1, arena_size=120, num_pellets=200, num_viruses=2, mode=4), dict(number_steps=3, episodic=0), 12),
    # two agents and an aggressive bot, modes 7-10: the engine sets done for agent 0 only when somebody dies -> agent 1's row is TRUNCATED
    "two-agents-bot": (dict(num_agents=2, arena_size=60, num_pellets=150, num_viruses=0, num_bots=1, mode=9), dict(number_steps=100000, episodic=1), 260),
    # mode 3: done at the mass threshold (not reached here) + cut-off with two agents: both rows done, none truncated
    "two-agents-cutoff": (dict(num_agents=2, arena_size=100, num_pellets=150, num_viruses=0, mode=3), dict(number_steps=4, episodic=1), 18),
}


def _run(lib, cfg, spec_kw, steps, A, alloc, to_host, obs_check=None):
    """alloc(shape, dtype) -> (device array object, pointer); to_host(obj) -> numpy"""
    from agarcl_amd import _capi
    n = cfg["num_agents"]
    eng = _capi.BatchedEngine(A, lib=lib, **cfg)
    ref = _capi.BatchedEngine(A, lib=lib, **cfg)
    seeds = np.arange(700, 700 + A, dtype=np.uint32)
    eng.seed(seeds); ref.seed(seeds)
    names = [("steps", (A,), np.int32), ("reward", (A, n), np.float32), ("done", (A, n), np.uint8), ("truncated", (A, n), np.uint8), ("ended", (A,), np.uint8),
             ("ep_return", (A, n), np.float32), ("final_return", (A, n), np.float32), ("final_length", (A,), np.int32)]
    bufs = {k: alloc(shape, dt) for k, shape, dt in names}
    spec = _capi.VecSpec(spec_kw["number_steps"], spec_kw["episodic"], 0, _capi.OBS_NONE, (C.c_int32 * 6)(0, 0, 0, 0, 0, 0), 0)
    vb = _capi.VecBuffers(*([bufs[k][1] for k, _, _ in names] + [None]))
    assert lib.agarcl_vec_reset(eng.h, C.byref(spec), C.byref(vb)) == 0, lib.agarcl_last_error()
    ref.reset()
    played = np.zeros(A, np.int64); ret = np.zeros((A, n), np.float32); fin_ret = np.zeros((A, n), np.float32); fin_len = np.zeros(A, np.int64)
    rng = np.random.RandomState(5)
    engine_dones = truncs = resets = 0
    for t in range(steps):
        move = rng.uniform(-1, 1, size=(A, n, 2)).astype(np.float32); kind = rng.randint(0, 3, size=(A, n)).astype(np.int32)
        dm, dk = alloc((A, n, 2), np.float32, move), alloc((A, n), np.int32, kind)
        fl = C.c_uint32(0)
        assert lib.agarcl_vec_step(eng.h, C.byref(spec), C.byref(vb), dm[1], dk[1], C.byref(fl)) == 0, lib.agarcl_last_error()
        # the same step as a user of the plain surface performs it (AgarioEnv.py:99-132)
        ref.set_actions(move, kind); ref.step()
        r = ref.rewards().astype(np.float32); d = ref.dones()
        timeout = (played >= spec_kw["number_steps"]) & bool(spec_kw["episodic"])
        done = d | timeout[:, None]
        ended = done.any(axis=1)
        trunc = ended[:, None] & ~done
        ret = (ret + r).astype(np.float32)
        played += 1
        fin_ret[ended] = ret[ended]; fin_len[ended] = played[ended]
        ret[ended] = 0; played[ended] = 0
        engine_dones += int((d.any(axis=1) & ~timeout).sum()); truncs += int(trunc.sum()); resets += int(ended.sum())
        if ended.any():
            ref.reset(ended.astype(np.uint8))
        got = {k: to_host(bufs[k][0]) for k, _, _ in names}
        assert np.array_equal(got["reward"], r), t
        assert np.array_equal(got["done"].astype(bool), done) and np.array_equal(got["truncated"].astype(bool), trunc) and np.array_equal(got["ended"].astype(bool), ended), t
        assert np.array_equal(got["steps"], played) and np.array_equal(got["ep_return"], ret), t
        assert np.array_equal(got["final_return"], fin_ret) and np.array_equal(got["final_length"], fin_len), t
        if ended.any() or t % 8 == 0 or t == steps - 1:
            for a in range(A):
                assert np.array_equal(eng.dump(a), ref.dump(a)), "step %d arena %d: state after the auto-reset differs" % (t, a)
    eng.close(); ref.close()
    return engine_dones, truncs, resets


@pytest.mark.parametrize("case", sorted(CASES))
def test_vec_step_bookkeeping_on_the_emulation(emu_lib, case):
    cfg, spec_kw, steps = CASES[case]
    keep = []

    def alloc(shape, dt, init=None):
        a = np.zeros(shape, dtype=dt) if init is None else np.ascontiguousarray(init, dtype=dt)
        keep.append(a)
        return a, C.c_void_p(a.ctypes.data)
    engine_dones, truncs, resets = _run(emu_lib, cfg, spec_kw, steps, 6, alloc, lambda a: a.copy())
    if case == "cutoff":
        assert resets == 6 * (steps // 6)              # an episode lasts number_steps + 1 steps (the comparison precedes the count)
    if case == "continuing":
        assert resets == 0
    if case == "two-agents-bot":
        assert engine_dones > 0 and truncs > 0, "the truncated-survivor path did not run (dones %d, truncations %d)" % (engine_dones, truncs)
    if case == "two-agents-cutoff":
        assert resets > 0 and truncs == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_vec_step_bookkeeping_hip(hip_engine_cls, case):
    import torch
    from agarcl_amd import _capi
    cfg, spec_kw, steps = CASES[case]
    keep = []
    tdt = {np.int32: torch.int32, np.float32: torch.float32, np.uint8: torch.uint8}

    def alloc(shape, dt, init=None):
        t = torch.zeros(shape, dtype=tdt[dt], device="cuda") if init is None else torch.as_tensor(np.ascontiguousarray(init, dtype=dt), device="cuda")
        torch.cuda.synchronize()      # (the engines run on streams of their own)
        keep.append(t)
        return t, C.c_void_p(t.data_ptr())

    def to_host(t):
        return t.cpu().numpy()
    engine_dones, truncs, resets = _run(_capi.hip_lib(), cfg, spec_kw, steps, 48, alloc, _sync_then(to_host, keep))
    assert resets > 0 or case == "continuing"
    if case == "two-agents-bot":
        assert engine_dones > 0 and truncs > 0


def _sync_then(fn, keep):
    import torch

    def f(t):
        torch.cuda.synchronize()
        return fn(t)
    return f


def test_vec_step_resets_flagged_arenas_on_request(emu_lib):
    """agarcl_vec_spec.reset_flagged: an arena that raised a capacity flag (here: two food slots, an agent that keeps ejecting) is cut like an ended
    episode -- reset in the same step, its row truncated and not done, the flag gone -- instead of running on diverged; without the switch it
    keeps its flag and is never reset"""
    from agarcl_amd import _capi
    cfg = dict(num_agents=1, arena_size=120, num_pellets=100, num_viruses=0, mode=6, cap_foods=2)
    A = 4
    for switch in (1, 0):
        eng = _capi.BatchedEngine(A, lib=emu_lib, **cfg)
        eng.seed(np.arange(40, 40 + A, dtype=np.uint32))
        names = [("steps", (A,), np.int32), ("reward", (A, 1), np.float32), ("done", (A, 1), np.uint8), ("truncated", (A, 1), np.uint8), ("ended", (A,), np.uint8),
                 ("ep_return", (A, 1), np.float32), ("final_return", (A, 1), np.float32), ("final_length", (A,), np.int32)]
        bufs = {k: np.zeros(shape, dt) for k, shape, dt in names}
        spec = _capi.VecSpec(1000, 0, 0, _capi.OBS_NONE, (C.c_int32 * 6)(0, 0, 0, 0, 0, 0), 0, switch)
        vb = _capi.VecBuffers(*([C.c_void_p(bufs[k].ctypes.data) for k, _, _ in names] + [None]))
        assert emu_lib.agarcl_vec_reset(eng.h, C.byref(spec), C.byref(vb)) == 0
        move = np.zeros((A, 1, 2), np.float32); move[:, 0, 0] = 1.0
        kind = np.ones((A, 1), np.int32)                       # feed, every step
        cut = 0
        for t in range(40):
            fl = C.c_uint32(0)
            assert emu_lib.agarcl_vec_step(eng.h, C.byref(spec), C.byref(vb), C.c_void_p(move.ctypes.data), C.c_void_p(kind.ctypes.data), C.byref(fl)) == 0
            if switch:
                assert not eng.flags().any(), "a flagged arena survived the step"
                assert np.array_equal(bufs["truncated"][:, 0], bufs["ended"]) and not bufs["done"].any()
                assert (bufs["steps"][bufs["ended"] != 0] == 0).all()
                cut += int(bufs["ended"].sum())
            else:
                assert not bufs["ended"].any()
        if switch:
            assert cut >= 2, "no arena was cut (%d)" % cut
        else:
            assert (eng.flags() & 2).any(), "no arena overflowed its two food slots: the switch = 1 half proved nothing"
        eng.close()
