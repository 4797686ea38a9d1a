"""The reference's own Python smoke tests of the gym surface (/root/reference/tests/grid_env_test.py:27-115, screen_env_test.py:9-64, configuration
tests/__init__.py:3-19), restated as known answers for the `gym_agario.AgarioEnv` this repository ships -- together with the grid-env gtests
(tests/test_grid_env_known_answers.py) the only evidence the reference itself holds for rows H1 / O1 / O2.  They are properties, not values:
the observation VALUES stay compared with this repository's own restatements (parity unpinned).

    grid_env_test.py  :27-32 creation | :34-40 reset returns a valid state | :42-58 action space | :60-72 1024 null-action steps: reward float,
                      done bool, info dict, state valid (:116-137: int32, the space's shape, min >= -1, max < 1000, not one single value, writable)
                      | :74-115 the observation space's shape follows the configuration
    screen_env_test.py :9-20 1024 x 1024 screen, 25 bots, 25 viruses, 100 pellets, one tick per step | :28-30 creation | :32-47 one step, :49-64 ten steps:
                      frame is an ndarray of the space's shape, sum > 0, not all 255; reward float, done bool, info dict

Where the reference's files are stale against its own code they are restated as the CODE has them:
  * `step` returns five values (AgarioEnv.py:132), the tests unpack four (grid :67, screen :36, :54) -- they cannot run against the reference;
  * `env.reset()` returns (observation, info) (AgarioEnv.py:140-141), grid :38-39 validates the tuple as if it were the state;
  * grid :47-53 expects actions with |x| up to 10 inside Box(-1, 1): restated as "inside [-1, 1]^2 x {0, 1, 2} is inside, kinds -1, -2, 3, 4, 5 and
    moves beyond 1 are outside";
  * grid :78-80 includes zero frames and a zero grid size: agarcl_grid_obs refuses a grid smaller than 1 (AGARCL_E_INVALID), so the loops start at 1;
  * channels per frame as GridObservation::channels_per_frame has them (GridEnvironment.hpp:188-196: 1 + cells + 2 others + 2 viruses + 2 pellets),
    not one per flag (:103) -- the same correction as in tests/test_grid_env_known_answers.py.
gymnasium is not installed here: `gym.make(id, **config)` is `AgarioEnv(obs_type=..., **config)` (gym_agario/__init__.py:9-23 registers exactly
that entry point), and `env.observation_space` / `env.action_space` are agarcl_amd/spaces.py's stand-ins with gymnasium's attribute names.
"""
import numpy as np
import pytest

# /root/reference/tests/__init__.py:3-19
NULL_ACTION = (np.zeros(2), 0)
DEFAULT_CONFIG = dict(ticks_per_step=4, num_frames=1, arena_size=1000, num_pellets=1000, num_viruses=25, num_bots=25, pellet_regen=True, grid_size=128,
                      observe_cells=True, observe_others=True, observe_viruses=True, observe_pellets=True)
# /root/reference/tests/screen_env_test.py:12-21
SCREEN_CONFIG = dict(ticks_per_step=1, arena_size=1000, pellet_regen=True, num_pellets=100, num_viruses=25, num_bots=25, screen_len=1024, c_death=0)


def _valid_state(env, state):                       # grid_env_test.py:116-137
    assert isinstance(state, np.ndarray) and state.dtype == np.int32
    assert state.shape == env.observation_space.shape and env.observation_space.contains(state)
    assert state.min() >= -1 and state.max() < 1000
    assert state.min() < state.max()                # "this one is really important": not all just one value
    state.fill(0)                                   # the array is the caller's own, writable memory


def check_grid(gym_agario, steps):
    env = gym_agario.AgarioEnv(obs_type="grid", **DEFAULT_CONFIG)                       # :27-32
    assert env.observation_space.shape == (128, 128, 8) and env.observation_space.dtype == np.int32
    state, info = env.reset()                                                          # :34-40
    assert info == {}
    _valid_state(env, state)
    space = env.action_space                                                           # :42-58
    for x in np.linspace(-1, 1, 9):
        for y in np.linspace(-1, 1, 9):
            for a in (0, 1, 2):
                assert space.contains((np.array([x, y], dtype=np.float32), a))
            for a in (-1, -2, 3, 4, 5):
                assert not space.contains((np.array([x, y], dtype=np.float32), a))
    assert not space.contains((np.array([10.0, 0.0], dtype=np.float32), 0)) and not space.contains((np.array([0.0, -1.5], dtype=np.float32), 1))
    for _ in range(steps):                                                             # :60-72
        state, reward, done, truncated, info = env.step(NULL_ACTION)
        assert isinstance(reward, float) and isinstance(done, bool) and truncated is False and isinstance(info, dict)
        _valid_state(env, state)
    env.close()


def check_grid_shapes(gym_agario):                                                     # :74-115
    for ticks_per_step in (1, 2, 3):
        for grid_size in (1, 10, 100):
            for o in range(16):
                flags = dict(observe_cells=bool(o & 1), observe_others=bool(o & 2), observe_viruses=bool(o & 4), observe_pellets=bool(o & 8))
                env = gym_agario.AgarioEnv(obs_type="grid", ticks_per_step=ticks_per_step, num_frames=1, arena_size=100, num_pellets=50, num_viruses=5, num_bots=5,
                                           pellet_regen=True, grid_size=grid_size, **flags)
                channels = 1 + int(flags["observe_cells"]) + 2 * (int(flags["observe_others"]) + int(flags["observe_viruses"]) + int(flags["observe_pellets"]))
                shape = env.observation_space.shape
                assert isinstance(shape, tuple) and len(shape) == 3 and shape == (grid_size, grid_size, channels)
                s, _ = env.reset()
                assert s.shape == shape and env.observation_space.contains(s)
                if o in (0, 15) or grid_size == 10:
                    for _ in range(10):
                        s, _r, done, *_ = env.step(NULL_ACTION)
                        if done:
                            break
                        assert s.shape == shape and env.observation_space.contains(s)
                env.close()


def check_screen(gym_agario, steps):
    env = gym_agario.AgarioEnv(obs_type="screen", **SCREEN_CONFIG)                      # :28-30
    assert env.observation_space.dtype == np.uint8 and tuple(env.observation_space.shape)[-3:-1] == (1024, 1024)
    env.reset()
    rng = np.random.RandomState(0)
    for _ in range(steps):                                                             # :32-47 (one step), :49-64 (ten)
        move = rng.rand(2) * 2 - 1
        frame, reward, done, truncated, info = env.step(((move[0], move[1]), 0))
        assert isinstance(frame, np.ndarray) and isinstance(reward, float) and isinstance(done, bool) and isinstance(info, dict)
        assert frame.shape == env.observation_space.shape
        assert np.sum(frame) > 0 and not np.all(frame == 255)
    env.close()


def test_grid_known_answers_on_the_emulation(emu_lib, monkeypatch):
    """the grid env's properties on the CPU build of the kernel source (64 of the 1024 steps: the emulation is slow; the GPU test plays them all)"""
    from agarcl_amd import agarcl, gym_agario
    monkeypatch.setattr(agarcl, "_LIB", emu_lib)
    monkeypatch.setattr(gym_agario, "agarcl", agarcl)
    check_grid(gym_agario, 64)
    check_grid_shapes(gym_agario)


@pytest.mark.gpu
@pytest.mark.parametrize("binding", ["pybind", "ctypes"])
def test_grid_known_answers_gpu(hip_engine_cls, monkeypatch, binding):
    """1024 null-action steps with 25 bots, as grid_env_test.py:60-72 plays them, through the compiled `agarcl` module and through the ctypes mirror"""
    from agarcl_amd import agarcl as mirror, gym_agario
    if binding == "pybind":
        import agarcl as compiled
        monkeypatch.setattr(gym_agario, "agarcl", compiled)
    else:
        monkeypatch.setattr(gym_agario, "agarcl", mirror)
    check_grid(gym_agario, 1024)
    check_grid_shapes(gym_agario)


@pytest.mark.gpu
@pytest.mark.parametrize("binding", ["pybind", "ctypes"])
def test_screen_known_answers_gpu(hip_engine_cls, monkeypatch, binding):
    """the 1024 x 1024 screen with 25 bots (screen_env_test.py): one step and ten steps"""
    from agarcl_amd import agarcl as mirror, gym_agario
    if binding == "pybind":
        import agarcl as compiled
        monkeypatch.setattr(gym_agario, "agarcl", compiled)
    else:
        monkeypatch.setattr(gym_agario, "agarcl", mirror)
    check_screen(gym_agario, 1)
    check_screen(gym_agario, 10)
