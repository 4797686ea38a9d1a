"""The gym-facing environment class of this repository: `AgarioEnv`, the N = 1 user surface over the batched HIP engine.

It keeps the *interface* a user of the reference's gym wrapper relies on (/root/reference/gym_agario/AgarioEnv.py:46-404 is the
counterpart): constructor keywords (a "difficulty" preset plus per-option overrides), `reset() -> (obs, {})`,
`step(a) -> (obs, reward, done, False, {"steps", "untransformed_rewards"})`, single-agent unwrapping, the episodic cut-off after
`number_steps` steps, `seed`, `render`, the video recorder, `save_env_state` / `load_env_state`, and the three gymnasium ids.
The implementation is this repository's own: options live in one table, every observation kind is a small builder, actions are
validated in one vectorised check, and recorded agent-view frames are coloured by a palette look-up on a per-pixel class index.

gymnasium is optional (it is not installed in the build image): when importable the class derives from gymnasium.Env and carries
gymnasium's spaces; otherwise `action_space` / `observation_space` are the same-shaped stand-ins of agarcl_amd/spaces.py.  There is no OpenGL window on a GPU server: render mode "human" shows nothing and returns None; "rgb_array" returns
the screen observation, or the engine's 512 x 512 frame for the grid / GoBigger observations.
"""
import os

import numpy as np


def _binding():
    """The `agarcl` module this wrapper drives: the compiled pybind11 module over the C ABI (repo root, built by
    agarcl_amd/build.py:build_pybind) when it is importable, else the ctypes mirror agarcl_amd/agarcl.py -- both are the same
    HIP engine; AGARCL_BINDING=ctypes|pybind pins the choice."""
    want = os.environ.get("AGARCL_BINDING", "")
    if want != "ctypes":
        try:
            import agarcl as compiled
            return compiled
        except ImportError:
            if want == "pybind":
                raise
    from . import agarcl as mirror
    return mirror


agarcl = _binding()

try:  # optional
    import gymnasium as _gym
    _Base = _gym.Env
except Exception:  # pragma: no cover - gymnasium is not installed in the build image
    _gym = None
    _Base = object
from . import spaces as _spaces   # gymnasium's spaces when importable, same-shaped stand-ins otherwise

OBS_TYPES = ("ram", "screen", "grid", "gobigger")

# ---- options --------------------------------------------------------------------------------------------------------------
# keyword -> default, per preset.  "normal" is the base row; the other presets only list what differs.  Everything a caller passes
# overrides the preset (the counterpart's defaults: AgarioEnv.py:313-349).
_PRESETS = {
    "normal": dict(ticks_per_step=4, num_frames=1, arena_size=1000, num_pellets=1000, num_viruses=0, num_bots=0, pellet_regen=True,
                   allow_respawn=True, reward_type=1),
    "empty": {},
    "trivial": dict(arena_size=50, num_pellets=200, num_viruses=0, num_bots=0),
}
# options that are not part of a preset: keyword -> default
_PLAIN = dict(grid_size=128, num_agents=1, c_death=0, mode=0, load_env_snapshot=False, agent_view=False, add_noise=True,
              number_steps=500, env_type=0, multi_agent=False)
# the engine constructor's positional order (environment/bindings.cpp:102,146; the screen / GoBigger classes append to it)
_CTOR_ORDER = ("num_agents", "ticks_per_step", "arena_size", "pellet_regen", "num_pellets", "num_viruses", "num_bots", "reward_type",
               "c_death", "mode", "load_env_snapshot")
_GRID_KEYS = ("num_frames", "grid_size", "observe_cells", "observe_others", "observe_viruses", "observe_pellets")
_RAM_KEYS = ("k_cells", "k_pellets", "k_viruses", "k_others")


def _resolve_options(kwargs):
    """preset + overrides -> {option: value}; raises ValueError for an unknown preset or a bad ticks_per_step"""
    preset = str(kwargs.get("difficulty", "normal")).lower()
    if preset not in _PRESETS:
        raise ValueError("unknown difficulty preset %r (choose from %s)" % (preset, ", ".join(sorted(_PRESETS))))
    opts = dict(_PRESETS["normal"]); opts.update(_PRESETS[preset]); opts.update(_PLAIN)
    opts.update({k: v for k, v in kwargs.items() if k in opts})
    opts["multi_agent"] = bool(opts["multi_agent"]) or opts["num_agents"] > 1
    t = opts["ticks_per_step"]
    if isinstance(t, bool) or not isinstance(t, int) or t < 1:
        raise ValueError("ticks_per_step must be an integer >= 1, got %r" % (t,))
    return opts


# ---- one builder per observation kind: (options, raw kwargs) -> (engine object, observation shape) -------------------------------
def _build_grid(o, kw):
    # (the counterpart hands an undefined name to the constructor here, AgarioEnv.py:226; the ten engine arguments are what is meant)
    env = agarcl.GridEnvironment(*[o[k] for k in _CTOR_ORDER[:10]])
    cfg = dict(num_frames=1, grid_size=128, observe_cells=True, observe_others=True, observe_viruses=True, observe_pellets=True)
    cfg.update({k: kw[k] for k in _GRID_KEYS + ("literal_frame_index",) if k in kw})
    env.configure_observation(cfg)
    c, w, h = env.observation_shape()
    return env, (w, h, c)          # channel-last, as _observe() hands the frames out


def _build_screen(o, kw):
    if not agarcl.has_screen_env:
        raise ValueError("this agarcl module was built without a ScreenEnvironment")
    side = kw.get("screen_len", 84)
    env = agarcl.ScreenEnvironment(*([o[k] for k in _CTOR_ORDER] + [side, side, o["agent_view"]]))
    return env, tuple(env.observation_shape())


def _build_gobigger(o, kw):
    head = [kw.get("map_width", 512), kw.get("map_height", 512), kw.get("frame_limit", 1000)]
    env = agarcl.GoBiggerEnvironment(*(head + [o[k] for k in _CTOR_ORDER] + [o["agent_view"]]))
    return env, tuple(env.observation_shape())


def _build_ram(o, kw):
    # an extension: the counterpart names "ram" and then refuses to build it (AgarioEnv.py:52,211; BASELINE configs[0]).  Here it is a
    # flat float32 vector per agent (include/agarcl_batch.h agarcl_ram_obs); always through the ctypes classes -- the compiled module
    # keeps exactly the reference's class list
    from . import agarcl as mirror
    env = mirror.RamEnvironment(*[o[k] for k in _CTOR_ORDER[:10]])
    env.configure_observation({k: kw[k] for k in _RAM_KEYS if k in kw})
    return env, tuple(env.observation_shape())


_BUILDERS = {"grid": _build_grid, "screen": _build_screen, "gobigger": _build_gobigger, "ram": _build_ram}

# ---- recorded agent-view frames: class index per pixel -> colour ------------------------------------------------------------
# classes of the 4-channel agent-view frame (channel tests of the counterpart's video painter, AgarioEnv.py:159-181), listed from the
# one that wins to the one that loses; whatever matches none is background
_VIDEO_PALETTE = np.array([(255, 0, 0),       # 0 background
                           (255, 255, 255),   # 1 pellet
                           (153, 51, 204),    # 2 other player / bot
                           (0, 255, 0),       # 3 virus
                           (0, 0, 255),       # 4 the agent itself
                           (26, 0, 0)],       # 5 grid line
                          dtype=np.uint8)


def _agent_view_classes(frame):
    """frame: uint8 [W][H][4] (pellets, others, viruses, alpha) -> class index per pixel (see _VIDEO_PALETTE)"""
    pel, oth, vir, alpha = (frame[..., k] for k in range(4))
    return np.select([alpha <= 30, alpha <= 230, vir == 255, oth == 255, pel != 255], [5, 4, 3, 2, 1], default=0)


class AgarioEnv(_Base):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 60}

    def __init__(self, obs_type="grid", render_mode=None, **kwargs):
        if _gym is not None:
            super().__init__()
        if obs_type not in OBS_TYPES:
            raise ValueError("obs_type must be one of %s, got %r" % (OBS_TYPES, obs_type))
        opts = _resolve_options(kwargs)
        for name, value in opts.items():      # every option is also an attribute (num_agents, arena_size, mode, agent_view, ...)
            setattr(self, name, value)
        self.number_of_steps = opts["number_steps"]
        self.obs_type, self.render_mode = obs_type, render_mode
        self._env, self.observation_shape = _BUILDERS[obs_type](opts, kwargs)
        self.steps = None                     # None until the first reset()
        self.video_recorder, self.video_recorder_enabled = [], False
        self._seed = None
        # the counterpart's spaces (AgarioEnv.py:55-62, 232-264): always present -- gymnasium's objects when it is importable
        self.action_space = _spaces.single_action_space()
        self.observation_space = _spaces.observation_space(obs_type, self.observation_shape)

    # ---- actions ------------------------------------------------------------------------------------------------------------
    def _checked_actions(self, actions):
        """one ((dx, dy), a) pair, or a list of them with several agents -> [(dx, dy, a)] for the engine; ValueError when the count or
        a value is outside the action space (the sampled noise of the counterpart is never applied: AgarioEnv.py:282-295)"""
        if not self.multi_agent and not isinstance(actions, list):
            actions = [actions]
        if not isinstance(actions, list):
            raise ValueError("with several agents the actions come as a list of ((dx, dy), a) pairs")
        if len(actions) != self.num_agents:
            raise ValueError("%d actions for %d agents" % (len(actions), self.num_agents))
        try:
            move = np.array([[a[0][0], a[0][1]] for a in actions], dtype=np.float64).reshape(len(actions), 2)
            kind = np.array([int(a[1]) for a in actions], dtype=np.int64)
        except (TypeError, IndexError) as e:
            raise ValueError("an action is a ((dx, dy), a) pair: %s" % e)
        if not (np.all(np.abs(move) <= 1.0) and np.all((kind >= 0) & (kind <= 2))):    # (NaN fails the first test)
            raise ValueError("action outside the action space [-1, 1]^2 x {0, 1, 2}: %r" % (actions,))
        return [(float(m[0]), float(m[1]), int(k)) for m, k in zip(move, kind)]

    # ---- observations -------------------------------------------------------------------------------------------------------
    def _observe(self):
        per_agent = self._env.get_state()
        assert len(per_agent) == self.num_agents
        if self.obs_type == "grid":            # the engine's frames are channel-first; users get (grid, grid, channels)
            per_agent = [frame.transpose(1, 2, 0) for frame in per_agent]
        return per_agent

    def _unwrap(self, per_agent):
        return per_agent if self.multi_agent else per_agent[0]

    # ---- gym surface --------------------------------------------------------------------------------------------------------
    def reset(self, **kwargs):
        self._env.reset()
        self.steps = 0
        return self._unwrap(self._observe()), {}

    def step(self, actions):
        assert self.steps is not None, "reset() must be called before the first step()"
        self._env.take_actions(self._checked_actions(actions))
        rewards = self._env.step()
        assert len(rewards) == self.num_agents
        per_agent = self._observe()
        if self.video_recorder_enabled:       # one agent's view is recorded
            self.video_recorder.append(self._make_video_observation(per_agent[0]))
        # an episodic env (env_type 0) ends after number_steps steps whatever the engine says; the comparison is made before this
        # step is counted, so the first `number_steps` steps of an episode do not end it
        out_of_time = self.env_type == 0 and self.steps >= self.number_of_steps
        dones = [True if out_of_time else bool(d) for d in self._env.dones()]
        self.steps += 1
        self.observations = self._unwrap(per_agent)
        reward = self._unwrap(rewards)
        info = {"steps": self.steps, "untransformed_rewards": reward}
        return self.observations, reward, self._unwrap(dones), self._unwrap([False] * len(dones)), info

    def seed(self, seed=None):
        if seed is None:
            return None
        self._seed = seed
        self._env.seed(seed)
        return [seed]

    def render(self):
        if self.render_mode == "human":
            self._env.render()                # nothing to show without a display
        elif self.render_mode == "rgb_array":
            if self.obs_type == "screen":
                return getattr(self, "observations", None)
            if self.obs_type in ("grid", "gobigger"):
                return self._env.get_frame()
        return None

    # ---- video recorder -----------------------------------------------------------------------------------------------------
    def _make_video_observation(self, observation):
        """the frame one recorded step contributes: the engine's 512 x 512 picture for grid / GoBigger, the screen observation as it is,
        or -- for the 4-channel agent view -- a colour picture made from the channels"""
        if self.obs_type in ("grid", "gobigger"):
            return self._env.get_frame()[0]
        if not self.agent_view:
            return observation
        return _VIDEO_PALETTE[_agent_view_classes(observation[0])]

    def enable_video_recorder(self):
        self.video_recorder_enabled = True

    def disable_video_recorder(self):
        self.video_recorder_enabled = False

    def generate_video(self, path, video_name):
        """writes the recorded frames as a Motion-JPEG AVI at 60 frames per second; returns the file's path, or None when there is
        nothing to write"""
        if not self.video_recorder_enabled:
            print("generate_video: the recorder is off (enable_video_recorder() first)")
            return None
        if not self.video_recorder:
            print("generate_video: no frame has been recorded yet")
            return None
        frames = []
        for k, frame in enumerate(self.video_recorder):
            if not isinstance(frame, np.ndarray):
                raise TypeError("recorded frame %d is a %s, not a numpy array" % (k, type(frame).__name__))
            frames.append(frame[0] if frame.ndim == 4 else frame)     # the plain screen observation carries a leading axis of 1
        os.makedirs(path, exist_ok=True)
        target = os.path.join(path, video_name)
        from .video import write_mjpeg_avi
        write_mjpeg_avi(target, frames, fps=60.0)
        return target

    # ---- pass-throughs ------------------------------------------------------------------------------------------------------
    def close(self):
        self._env.close()

    def save_env_state(self, filename):
        self._env.save_env_state(filename)

    def load_env_state(self, filename):
        self._env.load_env_state(filename)


def register():
    """the three gymnasium ids (gym_agario/__init__.py:9-23 of the counterpart); False without gymnasium"""
    if _gym is None:
        return False
    from gymnasium.envs.registration import register as _reg
    kinds = ["grid"] + (["screen"] if agarcl.has_screen_env else []) + ["gobigger"]
    for kind in kinds:
        _reg(id="agario-%s-v0" % kind, entry_point="gym_agario.AgarioEnv:AgarioEnv", kwargs={"obs_type": kind})
    return True
