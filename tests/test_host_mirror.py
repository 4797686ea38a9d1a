"""Host-side mirrors of the reference surface (agarcl module classes, gym wrapper): same names, return shapes and
error behaviour.  CPU runs use the test-only emulation build through the documented test seam; the GPU variant
uses the HIP library."""
import numpy as np
import pytest


def _exercise(agarcl, gym_agario, oracle_lib):
    env = agarcl.GridEnvironment(1, 4, 300, True, 300, 5, 0, 1, 0, 0)
    with pytest.raises(RuntimeError):
        env.observation_shape()                       # "GridObservation was not configured." GridEnvironment.hpp:72-88
    env.configure_observation({"grid_size": 32, "observe_others": False})
    assert env.observation_shape() == (6, 32, 32)
    env.seed(7); env.reset()
    with pytest.raises(RuntimeError):
        env.take_actions([(0.1, 0.2, 0), (0.0, 0.0, 0)])  # BaseEnvironment.hpp:142-144
    o = oracle_lib.OraEnv(num_agents=1, ticks_per_step=4, arena_size=300, num_pellets=300, num_viruses=5, mode=0)
    o.seed(7); o.reset(False)
    for t in range(30):
        a = (0.5, -0.25, t % 3)
        env.take_actions([a]); r = env.step()
        o.take_actions(np.array([[a[0], a[1]]], np.float32), np.array([a[2]], np.int32)); ro = o.step()
        assert isinstance(r, list) and isinstance(r[0], float) and r[0] == ro[0]
        assert env.dones() == [bool(o.dones()[0])]
    st = env.get_state()
    assert len(st) == 1 and st[0].dtype == np.int32 and st[0].shape == (6, 32, 32)
    assert np.array_equal(st[0], o.grid_obs(0, 32, True, False, True, True))
    env.close()
    with pytest.raises(RuntimeError):
        agarcl.GridEnvironment(1, 4, 300, True, 300, 5, 0, 1, 0, 42)   # invalid mode, Engine.hpp:413-414
    assert agarcl.has_screen_env is True          # provided by the rule-based HIP rasteriser (csrc/agar_screen.inl)
    # gym wrapper
    g = gym_agario.AgarioEnv(obs_type="grid", difficulty="trivial", grid_size=16, number_steps=5)
    with pytest.raises(AssertionError):
        g.step(((0.0, 0.0), 0))
    g.seed(3)
    obs, info = g.reset()
    assert obs.shape == (16, 16, 8) and info == {}
    done = False
    for k in range(7):
        obs, rew, done, trunc, info = g.step(((0.3, 0.1), 0))
        assert obs.shape == (16, 16, 8) and isinstance(rew, float) and trunc is False and info["steps"] == k + 1
        assert done == (k >= 5)                        # episodic cut-off, AgarioEnv.py:111-112
    with pytest.raises(ValueError):
        g.step(((2.0, 0.0), 0))
    with pytest.raises(ValueError):
        gym_agario.AgarioEnv(obs_type="nope")
    g.close()


def test_host_mirror_cpu(emu_lib, oracle_lib, monkeypatch):
    from agarcl_amd import agarcl, gym_agario
    monkeypatch.setattr(agarcl, "_LIB", emu_lib)
    monkeypatch.setattr(gym_agario, "agarcl", agarcl)      # (the wrapper would otherwise pick the compiled module: HIP only)
    _exercise(agarcl, gym_agario, oracle_lib)


@pytest.mark.gpu
def test_host_mirror_gpu(oracle_lib, monkeypatch):
    """the ctypes mirror (agarcl_amd/agarcl.py) on the HIP library"""
    from agarcl_amd import agarcl, gym_agario
    assert agarcl._LIB is None
    monkeypatch.setattr(gym_agario, "agarcl", agarcl)
    _exercise(agarcl, gym_agario, oracle_lib)


@pytest.mark.gpu
def test_compiled_agarcl_module_gpu(oracle_lib, monkeypatch):
    """`import agarcl`: the compiled pybind11 module over the C ABI (agarcl_amd/csrc/agarcl_pybind.cpp), driven through the
    reference's own usage sequence -- reset -> seed -> take_actions -> step -> get_state -> dones -- and under the gym wrapper."""
    from agarcl_amd import build as hip_build, gym_agario
    hip_build.build_pybind()
    import agarcl
    assert agarcl.__file__.endswith(".so") and agarcl.has_screen_env is True
    monkeypatch.setattr(gym_agario, "agarcl", agarcl)
    _exercise(agarcl, gym_agario, oracle_lib)
    # multi-frame observation: the frame slots of GridObservation (GridEnvironment.hpp:91-123)
    env = agarcl.GridEnvironment(1, 4, 300, True, 300, 5, 0, 1, 0, 0)
    env.configure_observation({"grid_size": 16, "num_frames": 4})
    assert env.observation_shape() == (32, 16, 16)
    env.seed(1); env.reset(); env.take_actions([(0.1, 0.1, 0)]); env.step()
    st = env.get_state()[0]
    assert st.shape == (32, 16, 16) and not st[:24].any() and st[24:].any()          # the state after the step fills the last slot
    env.configure_observation({"grid_size": 16, "num_frames": 4, "literal_frame_index": True})   # frame_index = 0 - (4 - 4) = 0
    lit = env.get_state()[0]
    assert np.array_equal(lit[:8], st[24:]) and not lit[8:].any()
    env.configure_observation({"grid_size": 16, "num_frames": 1, "literal_frame_index": True})   # frame_index = -3: nothing stored
    assert not env.get_state()[0].any()
    env.close()
    # snapshots through the compiled module: written in the reference's JSON format, loaded back, stepping continues
    import json, os, tempfile
    a = agarcl.ScreenEnvironment(1, 4, 300, True, 200, 3, 0, True, 0, 6, False, 32, 32, False)
    a.seed(11); a.reset()
    for t in range(5):
        a.take_actions([(0.3, 0.2, t % 3)]); a.step()
    path = os.path.join(tempfile.mkdtemp(), "snap.json")
    a.save_env_state(path)
    snap = json.load(open(path))
    assert {"players", "pellets", "viruses", "foods", "seed"} <= set(snap)
    b = agarcl.ScreenEnvironment(1, 4, 300, True, 200, 3, 0, True, 0, 6, True, 32, 32, False)
    b.load_env_state(path)
    b.take_actions([(0.0, 0.0, 0)]); r = b.step()
    assert isinstance(r, list) and len(b.get_state()) == 1 and b.get_state()[0].shape == (1, 32, 32, 3)
    with pytest.raises(RuntimeError):
        b.load_env_state("/nonexistent/dir/x.json")
    a.close(); b.close()
    # GoBigger: value classes come with the module; get_state follows bindings.cpp:28-47
    g = agarcl.GoBiggerEnvironment(512, 512, 1000, 1, 4, 300, True, 400, 6, 2, True)
    g.seed(5); g.reset()
    for t in range(10):
        g.take_actions([(0.4, 0.3, t % 3)]); g.step()
    st = g.get_state()
    assert set(st[0]) == {"global_state", "player_states"} and isinstance(st[0]["global_state"], agarcl.GlobalState)
    assert len(st[0]["player_states"].get_all_player_states()) == 3 and g.observation_shape() == (11, 512, 512)
    g.close()


@pytest.mark.gpu
def test_gym_wrapper_all_three_observation_types():
    """gym_agario.AgarioEnv as gym.make("agario-{grid,screen,gobigger}-v0") would build it (AgarioEnv.py:202-268)"""
    import gym_agario  # noqa: F401  (registers the ids when gymnasium is installed)
    from gym_agario.AgarioEnv import AgarioEnv
    for obs_type in ("grid", "screen", "gobigger"):
        g = AgarioEnv(obs_type=obs_type, difficulty="normal", num_viruses=5, number_steps=3)
        g.seed(4)
        obs, info = g.reset()
        for k in range(4):
            obs, rew, done, trunc, info = g.step(((0.5, 0.5), k % 3))
            assert isinstance(rew, float) and trunc is False and info["steps"] == k + 1 and done == (k >= 3)
        if obs_type == "grid":
            assert obs.shape == (128, 128, 8) and obs.dtype == np.int32
        elif obs_type == "screen":
            assert obs.shape == (1, 84, 84, 3) and obs.dtype == np.uint8
        else:
            assert set(obs) == {"global_state", "player_states"}
        g.close()
    # "ram": accepted by the reference's first check and rejected when the env is built (AgarioEnv.py:52,211); here it is an extension -- a flat
    # float32 vector per agent (include/agarcl_batch.h agarcl_ram_obs)
    g = AgarioEnv(obs_type="ram", num_viruses=5, k_pellets=8)
    g.seed(3); obs, _ = g.reset()
    assert obs.shape == (4 + 48 + 16 + 24 + 48,) and obs.dtype == np.float32 and obs[2] == 25 and obs[3] == 1
    obs, rew, done, trunc, info = g.step(((0.5, 0.5), 0))
    assert obs.shape == (140,) and isinstance(rew, float)
    g.close()
    with pytest.raises(ValueError):
        AgarioEnv(obs_type="pixels")


def test_registration_with_a_stand_in_gymnasium(monkeypatch):
    """gymnasium is not installed in this image: a stand-in module records what `import gym_agario` registers
    (gym_agario/__init__.py:9-23: three ids, one entry point) and the entry point resolves to the wrapper class."""
    import importlib, sys, types
    calls = []
    gymn = types.ModuleType("gymnasium"); gymn.Env = object
    spaces = types.ModuleType("gymnasium.spaces")
    for n in ("Box", "Tuple", "Discrete"):
        setattr(spaces, n, lambda *a, **k: ("space", a, k))
    reg = types.ModuleType("gymnasium.envs.registration"); reg.register = lambda **kw: calls.append(kw)
    envs = types.ModuleType("gymnasium.envs"); envs.registration = reg
    gymn.spaces, gymn.envs = spaces, envs
    for name, mod in (("gymnasium", gymn), ("gymnasium.spaces", spaces), ("gymnasium.envs", envs), ("gymnasium.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, name, mod)
    for name in ("gym_agario", "gym_agario.AgarioEnv", "agarcl_amd.gym_agario"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    import gym_agario
    assert gym_agario.registered is True
    assert [c["id"] for c in calls] == ["agario-grid-v0", "agario-screen-v0", "agario-gobigger-v0"]
    assert [c["kwargs"]["obs_type"] for c in calls] == ["grid", "screen", "gobigger"]
    for c in calls:
        mod, cls = c["entry_point"].split(":")
        klass = getattr(importlib.import_module(mod), cls)
        assert isinstance(klass, type) and klass.__name__ == "AgarioEnv" and hasattr(klass, "step") and hasattr(klass, "reset")
    for name in ("gym_agario", "gym_agario.AgarioEnv", "agarcl_amd.gym_agario"):      # leave no stand-in-derived classes behind
        sys.modules.pop(name, None)


def test_gobigger_object_view(emu_lib, monkeypatch):
    """SURVEY 8f N3 (object view derived from the kernel's tensors, parity unpinned): structure and invariants."""
    from agarcl_amd import agarcl
    monkeypatch.setattr(agarcl, "_LIB", emu_lib)
    env = agarcl.GoBiggerEnvironment(512, 512, 1000, 1, 4, 300, True, 400, 6, 2, True, 0, 0)
    env.seed(5); env.reset()
    for t in range(40):
        env.take_actions([(0.4, 0.3, t % 3)]); env.step()
    st = env.get_state()
    assert isinstance(st, list) and set(st[0]) == {"global_state", "player_states"}
    gs, ps = st[0]["global_state"], st[0]["player_states"]
    assert gs.get_map_width() == 512 and gs.get_frame_limit() == 1000 and gs.get_team_num() == 1
    states = ps.get_all_player_states()
    assert len(states) == 3                                     # the agent and both bots, keyed by pid
    from oracle import blob
    d = blob.parse(env._engine.dump(0))
    for p in d["players"]:
        s = states[p["pid"]]
        assert s.get_team_name() == "dummy" and s.canEject() and s.canSplit()
        if p["n_cells"]:
            assert s.get_score() == float(p["cell_mass"].sum())
            clones = s.get_clone_infos()
            assert 1 <= len(clones) <= p["n_cells"] and all(c.owner == p["pid"] and c.teamId == 0 for c in clones)
            assert sum(c.score for c in clones) <= p["cell_mass"].sum()
            # one-cell players sit at their own centre
            if p["n_cells"] == 1:
                assert abs(clones[0].get_position_x()) < 1e-3 and abs(clones[0].get_position_y()) < 1e-3
            assert all(f.score == 1 and abs(f.radius - 0.5641896) < 1e-5 for f in s.get_food_infos())
            view = min(max(2.0 * p["cell_mass"].sum(), 100.0), 300.0)
            assert all(abs(f.get_position_x()) <= view / 2 + 1e-3 and abs(f.get_position_y()) <= view / 2 + 1e-3 for f in s.get_food_infos())
    assert env.observation_shape()[1:] == (512, 512) and env.observation_shape()[0] == 41     # one frame per reset / step
    env.close()


def test_gobigger_observation_class_is_a_module_attribute():
    """bindings.cpp:297-318: GoBiggerObservation(map_width, map_height, frame_limit, last_frame, team_num) with update_global_state,
    update_player_state (two keyword spellings), get_global_state, get_player_states -- on both forms of the module (CPU only: no engine)"""
    import importlib
    from agarcl_amd import agarcl as ct
    mods = [ct]
    try:
        mods.append(importlib.import_module("agarcl"))       # the compiled pybind11 module, when built
    except ImportError:
        pass
    for m in mods:
        for name in ("FoodInfo", "VirusInfo", "SporeInfo", "CloneInfo", "GlobalState", "PlayerState", "PlayerStates", "GoBiggerObservation"):
            assert hasattr(m, name), (m.__name__, name)
        o = m.GoBiggerObservation(map_width=512, map_height=256, frame_limit=1000, last_frame=0, team_num=2)
        assert (o.get_global_state().get_map_width(), o.get_global_state().get_map_height(), o.get_global_state().get_frame_limit(), o.get_global_state().get_team_num()) == (512, 256, 1000, 2)
        assert o.get_player_states().get_all_player_states() == {}
        o.update_global_state(7)
        o.update_player_state(1, [], [], [], [], "t", 5.0, True, False)
        o.update_player_state(player_id=0, food_positions=[], thorn_positions=[], spore_positions=[], clone_positions=[], team_name="u", score=2.0, can_eject=False, can_split=True)
        ps = o.get_player_states()
        assert sorted(ps.get_all_player_states()) == [0, 1] and ps.get_player_state(1).get_score() == 5.0 and ps.get_player_state(0).canSplit() and not ps.get_player_state(1).canSplit()
