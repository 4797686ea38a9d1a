"""AgarioVectorEnv: the batched user surface -- `num_envs` arenas behind ONE `reset()` / `step()` pair, shaped like
gymnasium.vector.VectorEnv (batched observations / rewards / terminated / truncated, auto-reset of finished episodes), with every tensor
resident in HBM and nothing inside `step()` that waits for the GPU.

`gym.make("agario-*-v0")` (agarcl_amd/gym_agario.py, the counterpart of /root/reference/gym_agario/AgarioEnv.py:85-132) is the N = 1 case:
one arena per object, Python lists and host arrays.  This class keeps its meaning per arena --

    * the same constructor keywords (difficulty preset + overrides, obs_type, number_steps, env_type, ...);
    * step = take_actions + ticks_per_step engine ticks + observation; reward per agent as BaseEnvironment::step computes it;
    * done = the engine's done flag, or -- for an episodic env (env_type 0) -- `number_steps` steps played (AgarioEnv.py:111-112: the cut-off
      sets done, never truncated; the comparison happens before the step is counted);
    * a finished arena is reset as `env.reset()` resets it (BaseEnvironment::reset: the arena's own random stream continues)

-- and runs all arenas in one launch per call:

    obs, info = venv.reset(seed=123)
    obs, reward, terminated, truncated, info = venv.step((move, kind))     # move f32 [N, 2] in [-1, 1], kind int [N] in {0, 1, 2}

Auto-reset is "same step" (gymnasium.vector.AutoresetMode.SAME_STEP): when an arena's episode ends in a step, that arena is reset inside the
same `step()` on the device (agarcl_reset_device with the done mask: no host round trip) and the returned observation row is the FIRST
observation of its next episode; reward and terminated of that row belong to the episode that ended.  The last observation of the ended
episode is not kept (it would double the observation traffic of every step).

`info` carries the episode statistics gymnasium's RecordEpisodeStatistics wrapper would keep, as device tensors: "episode_steps" and
"episode_return" of the running episodes, "ended" (the arenas whose episode ended in this step) and, valid for those rows, "final_return" /
"final_length" of the episode that ended.

Returned tensors are CUDA tensors owned by the environment and rewritten in place by the next `step()` -- copy what must outlive it (a
rollout buffer does that anyway).  With one agent per arena the agent axis is dropped: obs [N, ...], reward [N]; with several agents
[N, num_agents, ...].
"""
import ctypes
import os

import numpy as np

from . import gym_agario as _single
from .vec_env import VecEnvironment


def _load_vecpost():
    """agarcl_amd/libagarcl_vec.so (include/agarcl_vec.h): the episode bookkeeping as ONE launch per step.  There is no torch fall-back: the
    library is built in-tree by agarcl_amd/build.py (build_vecpost) and a missing one is an error."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libagarcl_vec.so")
    if not os.path.exists(path):
        from . import build as _build
        _build.build_vecpost()
    lib = ctypes.CDLL(path)
    f = lib.agarcl_vec_post
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_void_p] * 7
    return f


class AgarioVectorEnv:
    metadata = {"render_modes": [], "autoreset_mode": "same_step"}

    def __init__(self, num_envs, obs_type="grid", device=0, channels_last=False, **kwargs):
        """num_envs arenas; obs_type "grid" | "screen" | "ram" | "gobigger" | "none"; channels_last: grid observations as a [.., G, G, C]
        VIEW (what AgarioEnv hands out per arena) instead of the engine's channel-first layout; every other keyword as AgarioEnv takes it
        (difficulty, ticks_per_step, arena_size, num_pellets, num_viruses, num_bots, pellet_regen, reward_type, c_death, mode, num_agents,
        number_steps, env_type, grid_size, observe_*, screen_len, agent_view, k_cells / k_pellets / k_viruses / k_others)."""
        import torch
        if obs_type not in _single.OBS_TYPES + ("none",):
            raise ValueError("obs_type must be one of %s, got %r" % (_single.OBS_TYPES + ("none",), obs_type))
        self.torch = torch
        o = _single._resolve_options(kwargs)
        self.options, self.obs_type, self.num_envs = o, obs_type, int(num_envs)
        self.num_agents, self.multi_agent = o["num_agents"], o["multi_agent"]
        self.number_of_steps, self.env_type = o["number_steps"], o["env_type"]
        self.channels_last = bool(channels_last)
        self.env = VecEnvironment(self.num_envs, num_agents=o["num_agents"], ticks_per_step=o["ticks_per_step"], arena_size=o["arena_size"],
                                  pellet_regen=o["pellet_regen"], num_pellets=o["num_pellets"], num_viruses=o["num_viruses"],
                                  num_bots=o["num_bots"], reward_type=o["reward_type"], c_death=o["c_death"], mode_number=o["mode"],
                                  device=device, **{k: kwargs[k] for k in ("strict_flags", "dt") if k in kwargs})
        self.device = self.env.device
        if obs_type == "grid":
            self._obs_args = dict(grid_size=kwargs.get("grid_size", 128), cells=kwargs.get("observe_cells", True), others=kwargs.get("observe_others", True),
                                  viruses=kwargs.get("observe_viruses", True), pellets=kwargs.get("observe_pellets", True))
        elif obs_type == "screen":
            side = kwargs.get("screen_len", 84)
            self._obs_args = dict(width=side, height=side, agent_view=o["agent_view"])
        elif obs_type == "ram":
            self._obs_args = {k: kwargs[k] for k in ("k_cells", "k_pellets", "k_viruses", "k_others") if k in kwargs}
        elif obs_type == "gobigger":
            self._obs_args = {k: kwargs[k] for k in ("grid_size", "cap_food", "cap_virus", "cap_spore", "cap_clone") if k in kwargs}
        else:
            self._obs_args = {}
        N, n = self.num_envs, self.num_agents
        self._steps = torch.zeros(N, dtype=torch.int32, device=self.device)       # steps played in the current episode, per arena
        # episode statistics, kept on the device (what gymnasium's RecordEpisodeStatistics wrapper keeps per env on the host): the running
        # return of the current episode, and -- rewritten only for the arenas whose episode ended in a step -- the return and length of it
        self._ep_return = torch.zeros((N, n), dtype=torch.float32, device=self.device)
        self._final_return = torch.zeros((N, n), dtype=torch.float32, device=self.device)
        self._final_length = torch.zeros(N, dtype=torch.int32, device=self.device)
        self._reward = torch.zeros((N, n), dtype=torch.float32, device=self.device)
        self._done = torch.zeros((N, n), dtype=torch.bool, device=self.device)
        self._trunc = torch.zeros((N, n), dtype=torch.bool, device=self.device)   # never set: the cut-off is a `done` (see the module text)
        self._mask = torch.zeros(N, dtype=torch.uint8, device=self.device)
        self._started = False
        self.single_observation_shape = None      # set by the first observation
        self._vec_post = _load_vecpost()
        self._post_args = None                    # (pointers of the tensors above: they never move)

    # ---- helpers ------------------------------------------------------------------------------------------------------------------
    def _agents(self, t):
        """[N, num_agents, ...] -> [N, ...] for single-agent envs"""
        return t if self.multi_agent else t[:, 0]

    def _observe(self):
        k = self.obs_type
        if k == "none":
            return None
        if k == "gobigger":
            return self.env.gobigger_obs(**self._obs_args)        # rows per PLAYER (agents and bots), see VecEnvironment.gobigger_obs
        if k == "grid":
            t = self.env.grid_obs(**self._obs_args)
            t = self._agents(t)
            if self.channels_last:
                t = t.movedim(-3, -1)
        elif k == "screen":
            t = self._agents(self.env.screen_obs(**self._obs_args))
        else:
            t = self._agents(self.env.ram_obs(**self._obs_args))
        self.single_observation_shape = tuple(t.shape[1:])
        return t

    def _as_device(self, x, dtype, shape):
        torch = self.torch
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), device=self.device)
        if x.numel() != int(np.prod(shape)):
            raise ValueError("actions for %d arenas x %d agents expected, got a tensor of shape %s" % (self.num_envs, self.num_agents, tuple(x.shape)))
        return x.to(device=self.device, dtype=dtype).reshape(shape)

    # ---- the vector-env surface -------------------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """all arenas start a new episode.  seed: None (the arenas' random streams continue), an int s (arena a gets seed s + a, as
        num_envs separate `env.seed(s + a)` calls would give), or a sequence of num_envs seeds."""
        if seed is not None:
            if np.isscalar(seed):
                self.env.seed(base_seed=int(seed))
            else:
                self.env.seed(np.asarray(seed, dtype=np.uint32))
        self.env.reset()
        self._steps.zero_(); self._ep_return.zero_()
        self._started = True
        return self._observe(), {}

    def step(self, actions):
        """actions = (move, kind): move f32 [N, 2] (or [N, num_agents, 2]) in [-1, 1]^2, kind int [N] (or [N, num_agents]) in {0 none, 1 feed,
        2 split}; CUDA tensors are used in place, host arrays are uploaded.  Everything below is enqueued on the current CUDA stream; nothing
        waits for it.  (Values are not range-checked here -- that would need them on the host; the engine clamps nothing either, as the
        reference's take_action: BaseEnvironment.hpp:162-176.)"""
        assert self._started, "reset() must be called before the first step()"
        torch = self.torch
        N, n = self.num_envs, self.num_agents
        move, kind = actions
        # (CUDA tensors of the right type are handed to the engine as they are -- the step below reads them in stream order; anything else
        # is converted / uploaded first)
        self.env.take_actions(self._as_device(move, torch.float32, (N, n, 2)), self._as_device(kind, torch.int32, (N, n)))
        self.env.step()
        # episode bookkeeping on the device, one launch (include/agarcl_vec.h): done = the engine's flag or the episodic cut-off -- compared
        # BEFORE this step is counted, AgarioEnv.py:111-112 --, the step counters, f64 -> f32 rewards, the episode statistics and the mask
        # of the arenas that start their next episode now
        if self._post_args is None:
            e = self.env
            self._post_args = (e.dones_u8.data_ptr(), e.rewards.data_ptr(), N, n, int(self.number_of_steps), 1 if self.env_type == 0 else 0,
                               self._steps.data_ptr(), self._reward.data_ptr(), self._done.data_ptr(), self._mask.data_ptr(),
                               self._ep_return.data_ptr(), self._final_return.data_ptr(), self._final_length.data_ptr())
        rc = self._vec_post(self.env.stream_handle, *self._post_args)   # (the stream the engine was bound to: the launch sits between its step and its reset)
        if rc != 0:
            raise RuntimeError("agarcl_vec_post failed (%d)" % rc)
        # same-step auto-reset: arenas whose episode ended (any agent) start the next one now, on the device
        self.env.reset(mask=self._mask)
        ended = self._mask.view(torch.bool)
        obs = self._observe()
        # "ended": the arenas whose episode ended in this step (and were reset); for those rows "final_return" / "final_length" are the
        # finished episode's return and length (they keep their last value otherwise); "episode_return" / "episode_steps" run with the
        # current episode.  All CUDA tensors owned by the env, like the observations.
        info = {"episode_steps": self._steps, "episode_return": self._agents(self._ep_return), "ended": ended,
                "final_return": self._agents(self._final_return), "final_length": self._final_length}
        return obs, self._agents(self._reward), self._agents(self._done), self._agents(self._trunc), info

    def close(self):
        self.env.close()
