"""gym-style wrapper with the surface of the reference's gym_agario.AgarioEnv
(/root/reference/gym_agario/AgarioEnv.py:46-404) on top of agarcl_amd.agarcl.

Same constructor keywords ("difficulty" presets + overrides, AgarioEnv.py:298-363), same return shapes:
reset() -> (obs, {}), step(a) -> (obs, reward, done, truncated=False, {'steps', 'untransformed_rewards'}),
single-agent unwrapping (AgarioEnv.py:114-118), episodic cut-off after `number_steps` (AgarioEnv.py:111-112).
gymnasium is optional: when it is importable the class derives from gymnasium.Env and exposes real spaces.
render() and the video recorder (AgarioEnv.py:134-181, 366-404) are provided on top of the engine's rule-based rasteriser: "rgb_array"
returns the screen observation or a 512 x 512 frame (get_frame), recorded frames are painted as the reference paints them and
generate_video writes the same Motion-JPEG container (cv2 when importable, else agarcl_amd/video.py).  There is no OpenGL window: render
mode "human" has nothing to show on a GPU server and returns None.
"""
import os

import numpy as np

from .agar_utils import Color, get_color_array


def _binding():
    """The `agarcl` module this wrapper drives: the compiled pybind11 module over the C ABI (repo root, built by
    agarcl_amd/build.py:build_pybind) when it is importable, else the ctypes mirror agarcl_amd/agarcl.py -- both are the same
    HIP engine; AGARCL_BINDING=ctypes|pybind pins the choice."""
    import os
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
    from gymnasium import spaces as _spaces
    _Base = _gym.Env
except Exception:  # pragma: no cover - gymnasium is not installed in the build image
    _gym = None
    _spaces = None
    _Base = object


class AgarioEnv(_Base):
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 60}

    def __init__(self, obs_type="grid", render_mode=None, **kwargs):
        if _gym is not None:
            super().__init__()
        if obs_type not in ("ram", "screen", "grid", "gobigger"):
            raise ValueError(obs_type)
        self._env, self.observation_shape = self._make_environment(obs_type, kwargs)
        self.steps = None
        self.obs_type = obs_type
        self.render_mode = render_mode
        self.video_recorder = []
        self.video_recorder_enabled = False
        self.agent_view = kwargs.get("agent_view", False)
        self.add_noise = kwargs.get("add_noise", True)
        self.number_of_steps = kwargs.get("number_steps", 500)
        self.mode = kwargs.get("mode", 0)
        self.env_type = kwargs.get("env_type", 0)  # 0 episodic, 1 continuing
        self._seed = None
        if _spaces is not None:   # AgarioEnv.py:226-268
            self.action_space = _spaces.Tuple((_spaces.Box(low=-1, high=1, shape=(2,)), _spaces.Discrete(3)))
            if obs_type == "grid":
                self.observation_space = _spaces.Box(-1, np.iinfo(np.int32).max, self.observation_shape, dtype=np.int32)
            elif obs_type == "screen":
                self.observation_space = _spaces.Box(low=0, high=255, shape=self.observation_shape, dtype=np.uint8)
            elif obs_type == "ram":
                self.observation_space = _spaces.Box(low=-np.inf, high=np.inf, shape=self.observation_shape, dtype=np.float32)
            else:
                self.observation_space = _spaces.Box(low=0, high=255, shape=self.observation_shape, dtype=np.float32)

    # -- AgarioEnv.py:298-363 -------------------------------------------------------------------------
    def _get_env_args(self, kwargs):
        difficulty = kwargs.get("difficulty", "normal").lower()
        if difficulty not in ["normal", "empty", "trivial"]:
            raise ValueError("Unrecognized difficulty: %s" % difficulty)
        d = dict(ticks_per_step=4, num_frames=1, arena_size=1000, num_pellets=1000, num_viruses=0, num_bots=0,
                 pellet_regen=True, allow_respawn=True, reward_type=1)
        if difficulty == "trivial":
            d.update(arena_size=50, num_pellets=200, num_viruses=0, num_bots=0)
        self.grid_size = kwargs.get("grid_size", 128)
        self.multi_agent = kwargs.get("multi_agent", False)
        self.num_agents = kwargs.get("num_agents", 1)
        for k in ("ticks_per_step", "num_frames", "arena_size", "num_pellets", "num_viruses", "num_bots", "pellet_regen",
                  "allow_respawn", "reward_type"):
            setattr(self, k, kwargs.get(k, d[k]))
        self.c_death = kwargs.get("c_death", 0)
        self.mode = kwargs.get("mode", 0)
        self.load_env_snapshot = kwargs.get("load_env_snapshot", False)
        self.multi_agent = self.multi_agent or self.num_agents > 1
        if type(self.ticks_per_step) is not int or self.ticks_per_step <= 0:
            raise ValueError("ticks_per_step must be a positive integer")
        return (self.num_agents, self.ticks_per_step, self.arena_size, self.pellet_regen, self.num_pellets,
                self.num_viruses, self.num_bots, self.reward_type, self.c_death, self.mode, self.load_env_snapshot)

    def _make_environment(self, obs_type, kwargs):
        base_args = self._get_env_args(kwargs)
        if obs_type == "grid":
            # (the reference passes an undefined name here, AgarioEnv.py:226; the evident intent is the ten constructor arguments)
            env = agarcl.GridEnvironment(*base_args[:10])
            cfg = dict(num_frames=1, grid_size=128, observe_cells=True, observe_others=True, observe_viruses=True, observe_pellets=True)
            cfg.update({k: kwargs[k] for k in list(cfg) + ["literal_frame_index"] if k in kwargs})
            env.configure_observation(cfg)
            channels, width, height = env.observation_shape()
            return env, (width, height, channels)
        if obs_type == "screen":             # AgarioEnv.py:235-250
            if not agarcl.has_screen_env:
                raise ValueError("agarcl was not compiled to include ScreenEnvironment")
            screen_len = kwargs.get("screen_len", 84)
            self.agent_view = kwargs.get("agent_view", False)
            env = agarcl.ScreenEnvironment(*(base_args + (screen_len, screen_len, self.agent_view)))
            return env, tuple(env.observation_shape())
        if obs_type == "gobigger":           # AgarioEnv.py:251-264
            full_args = (kwargs.get("map_width", 512), kwargs.get("map_height", 512), kwargs.get("frame_limit", 1000)) + base_args + (kwargs.get("agent_view", False),)
            env = agarcl.GoBiggerEnvironment(*full_args)
            return env, tuple(env.observation_shape())
        if obs_type == "ram":
            # an extension: the reference names "ram" (AgarioEnv.py:52, BASELINE configs[0]) and then raises for it (AgarioEnv.py:211).
            # Here it is a flat float32 vector per agent (include/agarcl_batch.h agarcl_ram_obs); always through the ctypes classes,
            # the compiled module keeps exactly the reference's class list
            from . import agarcl as mirror
            env = mirror.RamEnvironment(*base_args[:10])
            env.configure_observation({k: kwargs[k] for k in ("k_cells", "k_pellets", "k_viruses", "k_others") if k in kwargs})
            return env, tuple(env.observation_shape())
        raise ValueError(obs_type)

    # -- AgarioEnv.py:270-296 (validation; the reference samples noise and then discards it) --------------
    def _sanitize_actions(self, actions):
        if not self.multi_agent and type(actions) is not list:
            actions = [actions]
        if type(actions) is not list:
            raise ValueError("Action list must be a list of two-element tuples")
        if len(actions) != self.num_agents:
            raise ValueError("Number of actions %d does not match number of agents %d" % (len(actions), self.num_agents))
        out = []
        for tgt, a in actions:
            tx, ty = float(tgt[0]), float(tgt[1])
            if not (-1.0 <= tx <= 1.0 and -1.0 <= ty <= 1.0 and int(a) in (0, 1, 2)):
                raise ValueError("action %r not in action space" % ((tgt, a),))
            out.append((tx, ty, int(a)))
        return out

    def _make_observations(self):
        states = self._env.get_state()
        assert len(states) == self.num_agents
        if self.obs_type == "grid":
            return [np.transpose(s, [1, 2, 0]) for s in states]  # NCHW -> NHWC, AgarioEnv.py:192-194
        return states                                            # screen: the (1, W, H, 3) frame as is, AgarioEnv.py:196-197

    def step(self, actions):
        assert self.steps is not None, "Cannot call step() before calling reset()"
        self._env.take_actions(self._sanitize_actions(actions))
        rewards = self._env.step()
        assert len(rewards) == self.num_agents
        self.observations = self._make_observations()
        if self.video_recorder_enabled:   # (one agent, as in the reference: AgarioEnv.py:99-101)
            self.video_recorder.append(self._make_video_observation(self.observations[0]))
        dones = self._env.dones()
        truncations = [False] * len(dones)
        if self.steps >= self.number_of_steps and self.env_type == 0:
            dones = [True] * len(dones)
        if not self.multi_agent:
            self.observations, rewards, dones, truncations = self.observations[0], rewards[0], dones[0], truncations[0]
        self.steps += 1
        return self.observations, rewards, dones, truncations, {"steps": self.steps, "untransformed_rewards": rewards}

    def reset(self, **kwargs):
        self.steps = 0
        self._env.reset()
        obs = self._make_observations()
        return (obs if self.multi_agent else obs[0]), {}

    def seed(self, seed=None):
        if seed is not None:
            self._seed = seed
            self._env.seed(seed)
            return [self._seed]

    def render(self):                       # AgarioEnv.py:134-147
        if self.render_mode == "human":
            self._env.render()              # (no window exists on a GPU server: a no-op, like the reference built without a display)
        if self.render_mode == "rgb_array":
            if self.obs_type == "screen":
                return getattr(self, "observations", None)
            if self.obs_type in ("grid", "gobigger"):
                return self._env.get_frame()
        return None

    def _make_video_observation(self, observation):   # AgarioEnv.py:159-181
        if self.obs_type in ("grid", "gobigger"):
            return self._env.get_frame()[0]
        if not self.agent_view:
            return observation
        observation = observation[0]
        rgb = np.zeros_like(observation[..., :3])
        rgb[..., 0].fill(255)
        pellets_mask = observation[..., 0] != 255
        bots_mask = observation[..., 1] == 255
        virus_mask = observation[..., 2] == 255
        main_agent_mask = (observation[..., 3] <= 230) & (observation[..., 3] > 30)
        grid_lines_mask = observation[..., 3] <= 30
        rgb[pellets_mask] = get_color_array(Color.WHITE)
        rgb[bots_mask] = get_color_array(Color.PURPLE)
        rgb[virus_mask] = get_color_array(Color.GREEN)
        rgb[main_agent_mask] = get_color_array(Color.BLUE)
        rgb[grid_lines_mask] = [26, 0, 0]
        return rgb

    def enable_video_recorder(self):        # AgarioEnv.py:372-376
        self.video_recorder_enabled = True

    def disable_video_recorder(self):
        self.video_recorder_enabled = False

    def generate_video(self, path, video_name):   # AgarioEnv.py:379-404: Motion-JPEG, 60 frames per second
        if not os.path.exists(path):
            os.makedirs(path, exist_ok=True)
        full_path = os.path.join(path, video_name)
        if not self.video_recorder_enabled:
            print("Video recorder is not enabled. Please enable it before generating video")
            return None
        if len(self.video_recorder) == 0:
            print("No frames to generate video")
            return None
        from .video import write_mjpeg_avi
        frames = []
        for frame in self.video_recorder:
            if not isinstance(frame, np.ndarray):
                raise TypeError("Error: A frame is not a numpy array.")
            frames.append(frame[0] if frame.ndim == 4 else frame)   # the plain screen observation is (1, W, H, 3)
        write_mjpeg_avi(full_path, frames, fps=60.0)
        return full_path

    def close(self):
        self._env.close()

    def save_env_state(self, filename):
        self._env.save_env_state(filename)

    def load_env_state(self, filename):
        self._env.load_env_state(filename)


def register():
    """gymnasium ids of the reference (gym_agario/__init__.py:9-23); no-op without gymnasium."""
    if _gym is None:
        return False
    from gymnasium.envs.registration import register as _reg
    _reg(id="agario-grid-v0", entry_point="gym_agario.AgarioEnv:AgarioEnv", kwargs={"obs_type": "grid"})
    if agarcl.has_screen_env:  # only register the screen environment if it is available
        _reg(id="agario-screen-v0", entry_point="gym_agario.AgarioEnv:AgarioEnv", kwargs={"obs_type": "screen"})
    _reg(id="agario-gobigger-v0", entry_point="gym_agario.AgarioEnv:AgarioEnv", kwargs={"obs_type": "gobigger"})
    return True
