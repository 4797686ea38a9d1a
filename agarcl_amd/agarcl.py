"""Single-arena classes with the names, positional constructor signatures and method names of the
reference's pybind11 module `agarcl` (/root/reference/environment/bindings.cpp:94-376), backed by the
HIP engine through the C ABI (num_arenas == 1).  `from agarcl_amd import agarcl` stands in for
`import agarcl` in gym_agario-style code.

Errors follow the reference: every failure is a RuntimeError (AgarclError subclasses it).
"""
import numpy as np

from . import _capi, snapshot

# ScreenEnvironment is provided through a rule-based HIP rasteriser (csrc/agar_screen.inl), not OpenGL
has_screen_env = True

# Test seam only: CPU tests point this at the test-only wave-emulation build of the kernel source.  The default
# (None) is the HIP library, and there is no automatic fallback: without libagarcl_hip.so / a GPU construction raises.
_LIB = None


class _Environment:
    def __init__(self, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots,
                 reward_type, c_death=0, mode_number=0, device=0, screen_respawn=False):
        self._num_agents = int(num_agents)
        self._cfg = dict(num_agents=int(num_agents), ticks_per_step=int(ticks_per_step), arena_size=int(arena_size), num_bots=int(num_bots),
                         reward_type=bool(reward_type), c_death=int(c_death), pellet_regen=bool(pellet_regen),
                         mode_number=0)   # Engine::mode_number as the reference's writer sees it: see snapshot.blob_to_json
        self._loaded = False             # BaseEnvironment::is_loading_env_state
        self._names = None               # player names in iteration order once a snapshot has been loaded
        self._engine = _capi.BatchedEngine(1, int(num_agents), int(ticks_per_step), int(arena_size), bool(pellet_regen),
                                           int(num_pellets), int(num_viruses), int(num_bots), int(bool(reward_type)) if isinstance(reward_type, bool) else int(reward_type),
                                           int(c_death), int(mode_number), device=device, screen_respawn=screen_respawn, lib=_LIB)

    def seed(self, s):                       # bindings.cpp:103
        self._engine.seed(np.asarray([int(s) & 0xFFFFFFFF], dtype=np.uint32))

    def reset(self):                         # bindings.cpp:130
        if self._loaded:                     # BaseEnvironment.hpp:180-181: reset() is a no-op once a snapshot was loaded
            return
        self._engine.reset(reset_ids=False)

    def take_actions(self, actions):         # bindings.cpp:50-64,117-119
        actions = list(actions)
        if len(actions) != self._num_agents:  # BaseEnvironment.hpp:142-144
            raise RuntimeError("Number of actions (%d) does not match number of agents (%d)" % (len(actions), self._num_agents))
        dxdy = np.array([[float(a[0]), float(a[1])] for a in actions], dtype=np.float32)
        act = np.array([int(a[2]) for a in actions], dtype=np.int32)
        self._engine.set_actions(dxdy, act)

    def step(self):                          # bindings.cpp:132 -> list[float]
        self._engine.step(0)
        return [float(r) for r in self._engine.rewards()[0]]

    def dones(self):                         # bindings.cpp:116 -> list[bool]
        return [bool(d) for d in self._engine.dones()[0]]

    def render(self):
        pass

    def close(self):
        self._engine.close()

    def save_env_state(self, path):          # bindings.cpp:131 -> BaseEnvironment.hpp:213-310: the reference's JSON
        snap = snapshot.save_arena(self._engine, 0, self._cfg, names=self._names)
        try:
            with open(path, "w") as f:
                f.write(snapshot.dumps(snap))
        except OSError:
            raise RuntimeError("Failed to open %s for writing" % path)   # BaseEnvironment.hpp:304-306

    def load_env_state(self, path):          # bindings.cpp:170,372 -> BaseEnvironment.hpp:312-343, Engine.hpp:247-348
        import json
        try:
            with open(path) as f:
                snap = json.load(f)
        except OSError:
            raise RuntimeError("Failed to open %s for reading" % path)   # Engine.hpp:255-257
        # ids continue from the arena's counter, like the reference's process-global counter does
        self._names = snapshot.load_arena(self._engine, 0, snap, reset_ids=False)
        self._cfg["mode_number"] = int(snap["mode_number"])              # Engine.hpp:263
        self._loaded = True


class RamEnvironment(_Environment):
    """NOT in the reference's module (obs_type "ram" is named by gym_agario/AgarioEnv.py:52 and by BASELINE configs[0], but the reference
    has no ram environment: AgarioEnv.py:211 raises, environment/test/ram-env-test.hpp is empty).  Same surface as the other classes;
    get_state() returns one float32 vector per agent in the layout of include/agarcl_batch.h agarcl_ram_obs."""

    def __init__(self, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots,
                 reward_type=0, c_death=0, mode_number=0):
        super().__init__(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, c_death, mode_number)
        self._k = dict(k_cells=16, k_pellets=16, k_viruses=8, k_others=16)

    def configure_observation(self, config):
        for k in self._k:
            if k in config:
                self._k[k] = int(config[k])

    def observation_shape(self):
        return (4 + 3 * self._k["k_cells"] + 2 * self._k["k_pellets"] + 3 * self._k["k_viruses"] + 3 * self._k["k_others"],)

    def get_state(self):
        obs = self._engine.ram_obs(**self._k)
        return [obs[0, i].copy() for i in range(self._num_agents)]


class GridEnvironment(_Environment):
    """agarcl.GridEnvironment (bindings.cpp:99-135)."""

    def __init__(self, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots,
                 reward_type=0, c_death=0, mode_number=0):
        # GridEnvironment forwards c_death = 0 to its base (GridEnvironment.hpp:369-372)
        super().__init__(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots,
                         reward_type, 0, mode_number)
        self._obs_cfg = None
        self._ticks_per_step = int(ticks_per_step)

    def configure_observation(self, config):  # bindings.cpp:104-114
        self._obs_cfg = dict(num_frames=int(config.get("num_frames", 1)), grid_size=int(config.get("grid_size", 128)),
                             observe_cells=bool(config.get("observe_cells", True)), observe_others=bool(config.get("observe_others", True)),
                             observe_viruses=bool(config.get("observe_viruses", True)), observe_pellets=bool(config.get("observe_pellets", True)),
                             # extra key (the reference ignores unknown keys): put the frame exactly where the reference's arithmetic does
                             literal_frame_index=bool(config.get("literal_frame_index", False)))
        if self._obs_cfg["num_frames"] < 1:
            raise RuntimeError("num_frames must be positive")

    def _channels(self):
        c = self._obs_cfg
        return c["num_frames"] * (1 + c["observe_cells"] + 2 * c["observe_others"] + 2 * c["observe_viruses"] + 2 * c["observe_pellets"])

    def observation_shape(self):             # bindings.cpp:115
        if self._obs_cfg is None:
            raise RuntimeError("GridObservation was not configured.")  # GridEnvironment.hpp:72-88
        g = self._obs_cfg["grid_size"]
        return (self._channels(), g, g)

    def get_state(self):                     # bindings.cpp:67-91,133: one owned int32 (C,G,G) array per agent
        if self._obs_cfg is None:
            raise RuntimeError("GridObservation was not configured.")
        c = self._obs_cfg
        obs = self._engine.grid_obs(c["grid_size"], c["observe_cells"], c["observe_others"], c["observe_viruses"], c["observe_pellets"])
        # The observation holds num_frames frame slots, cleared at every step (GridEnvironment.hpp:91-123,405-410).  The reference
        # calls _partial_observation(agent, tick_index = 0) once per step and stores the frame at frame_index =
        # 0 - (ticks_per_step - num_frames) if that is >= 0 (:417-431): with the default arguments (1 frame, 4 ticks) NO frame is
        # ever stored and the observation is all zeros.  Default here = the evident intent: the state after the step in the LAST
        # slot; literal_frame_index=True reproduces the reference's output exactly.
        nf, C = c["num_frames"], obs.shape[2]
        slot = nf - self._ticks_per_step if c["literal_frame_index"] else nf - 1
        out = []
        for i in range(self._num_agents):
            a = np.zeros((nf * C,) + obs.shape[3:], dtype=np.int32)
            if 0 <= slot < nf:
                a[slot * C:(slot + 1) * C] = obs[0, i]
            out.append(a)
        return out


class ScreenEnvironment(_Environment):
    """agarcl.ScreenEnvironment (bindings.cpp:142-171; environment/envs/ScreenEnvironment.hpp:130-245): RGB frames from
    the agent's perspective, drawn by the rule-based rasteriser of csrc/agar_screen.inl instead of OpenGL, plus this
    class's respawn hook (a dead agent is respawned right after the ticks in every mode, ScreenEnvironment.hpp:233-243)."""

    def __init__(self, num_agents, frames_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots,
                 reward_type, c_death, mode_number, load_env_snapshot, screen_width, screen_height, agent_view):
        self._agent_view = bool(agent_view)
        super().__init__(num_agents, frames_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots,
                         reward_type, c_death, mode_number, screen_respawn=True)
        self._w, self._h = int(screen_width), int(screen_height)
        self._loaded = bool(load_env_snapshot)   # BaseEnvironment(..., load_env_snapshot): reset() is a no-op from the start

    def observation_shape(self):             # bindings.cpp:156 -> (num_frames, width, height, channels)
        return (1, self._w, self._h, 4 if self._agent_view else 3)

    def get_state(self):                     # bindings.cpp:157-168: the frame buffer's bytes viewed as uint8 (1, W, H, 3)
        frames = self._engine.screen_obs(self._w, self._h, agent_view=self._agent_view)   # [1][n_agents][H][W][3|4], rows bottom-up
        # the reference keeps ONE frame buffer that every agent's render overwrites in turn: the last agent's frame remains.
        # The binding returns a py::list holding that one array (bindings.cpp:157-168), so gym_agario's obs[0] is (1, W, H, C).
        return [frames[0, self._num_agents - 1].reshape(1, self._w, self._h, 4 if self._agent_view else 3)]


class GoBiggerEnvironment(_Environment):
    """agarcl.GoBiggerEnvironment (bindings.cpp:321-375; environment/envs/GoBiggerEnvironment.hpp:551-737): the engine
    with the GoBigger-style structured observation (object view, agarcl_amd/gobigger.py).  get_frame (an OpenGL
    512x512 frame) is served by the rule-based rasteriser."""

    def __init__(self, map_width, map_height, frame_limit, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets,
                 num_viruses, num_bots, reward_type, c_death=0, mode_number=0, load_env_snapshot=False, agent_view=False):
        from . import gobigger
        # _partial_observation zeroes c_death_ on every call (GoBiggerEnvironment.hpp:624): death is never penalised here
        super().__init__(num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, 0, mode_number)
        self._gb = gobigger
        self._global = gobigger.GlobalState(map_width, map_height, frame_limit, 0, num_agents)
        self._states = gobigger.PlayerStates()
        self._grid_size = 128
        self._no_frames = 0
        self._loaded = bool(load_env_snapshot)   # (nothing is observed before the first reset(): the base constructor's reset runs the base hook)

    def configure_observation(self, config):  # bindings.cpp:341-352
        self._grid_size = int(config.get("grid_size", 128))

    def _observe(self):                       # _partial_observation per agent -> GoBiggerObservation::add_frame (:519-541, 618-636)
        # one kernel launch lists every player's entities (padded tensors); the reference calls add_frame once per agent,
        # each call refreshing ALL players from the same state, so one refresh is equivalent
        tensors = self._engine.gobigger_obs(self._grid_size)
        self._gb.add_frame(self._states, tensors, 0)
        self._no_frames += self._num_agents
        self._global.update_last_frame_count(0)

    def reset(self):
        super().reset()
        if not self._loaded:
            self._observe()

    def step(self):
        r = super().step()
        self._observe()
        return r

    def observation_shape(self):              # GoBiggerObservation::shape (:411-416): (frames added so far, map_height, map_width)
        return (self._no_frames, self._global.get_map_height(), self._global.get_map_width())

    def get_state(self):                      # bindings.cpp:28-47
        return [{"global_state": self._global, "player_states": self._states}]

    def get_frame(self):                      # bindings.cpp:354-363: uint8 (1, 512, 512, 3) from the last agent's perspective
        f = self._engine.screen_obs(512, 512)
        return f[0, self._num_agents - 1].reshape(1, 512, 512, 3)


# the GoBigger value classes and GoBiggerObservation as module attributes (bindings.cpp:184-318), as the compiled module carries them
from .gobigger import (CloneInfo, FoodInfo, GlobalState, GoBiggerObservation, PlayerState, PlayerStates, SporeInfo,  # noqa: E402,F401
                       VirusInfo)
