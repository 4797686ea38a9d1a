"""AgarioVectorEnv: the batched user surface -- `num_envs` arenas behind ONE `reset()` / `step()` pair, shaped like
gymnasium.vector.VectorEnv (batched observations / rewards / terminated / truncated, spaces, auto-reset of finished episodes), with every
tensor resident in HBM, nothing inside `step()` that waits for the GPU, and ONE host call into the library per step and sub-batch.

`gym.make("agario-*-v0")` (agarcl_amd/gym_agario.py, the counterpart of /root/reference/gym_agario/AgarioEnv.py:85-132) is the N = 1 case:
one arena per object, Python lists and host arrays.  This class keeps its meaning per arena --

    * the same constructor keywords (difficulty preset + overrides, obs_type, number_steps, env_type, ...);
    * step = take_actions + ticks_per_step engine ticks + observation; reward per agent as BaseEnvironment::step computes it;
    * done = the engine's done flag, or -- for an episodic env (env_type 0) -- `number_steps` steps played (AgarioEnv.py:111-112: the cut-off
      sets done, never truncated; the comparison happens before the step is counted);
    * a finished arena is reset as `env.reset()` resets it (BaseEnvironment::reset: the arena's own random stream continues)

-- and runs all arenas in one call of `agarcl_vec_step` (include/agarcl_vec.h: step kernel, one bookkeeping + masked-reset launch, the
observation kernel):

    obs, info = venv.reset(seed=123)
    obs, reward, terminated, truncated, info = venv.step((move, kind))     # move f32 [N, 2] in [-1, 1], kind int [N] in {0, 1, 2}

Auto-reset is "same step" (gymnasium.vector.AutoresetMode.SAME_STEP): when an arena's episode ends in a step, that arena is reset inside the
same `step()` on the device and the returned observation row is the FIRST observation of its next episode; reward and terminated of that
row belong to the episode that ended.  The last observation of the ended episode is not kept (it would double the observation traffic of
every step).  With several agents per arena the arena ends -- and is reset -- as soon as ANY agent is done: the rows of the agents that were
not done get truncated = True for that step (their episode was cut by the reset; a learner must not bootstrap across it as if nothing
happened), every row of the arena gets its "final_return".

Spaces: `single_observation_space`, `single_action_space`, `observation_space`, `action_space` (gymnasium.spaces objects when gymnasium is
importable, agarcl_amd/spaces.py's equivalents otherwise) and `num_envs`, as gymnasium.vector.VectorEnv carries them; the reference's
per-env spaces are AgarioEnv.py:55-62 and :232-264.

`info` carries the episode statistics gymnasium's RecordEpisodeStatistics wrapper would keep, as device tensors: "episode_steps" and
"episode_return" of the running episodes, "ended" (the arenas whose episode ended in this step) and, valid for those rows, "final_return" /
"final_length" of the episode that ended.

Sub-batches (`sub_batches=k`): the arenas are split into k contiguous ranges that run on HIP streams of their own (include/agarcl_batch.h
agarcl_pipe_*), so a range never waits for the slowest arena of another and one range's observation kernel runs under another's step --
the reference's vectorised runner works that way (every engine on its own pool thread, /root/reference/agario/bots/benchmark.cpp:149-167).
`step()` still takes and returns full-batch tensors (each range writes its rows of them).  The two halves are also available on their own, for
double-buffered sampling (the policy works on range j's observations while the other ranges step):

    venv.async_reset(seed=0)
    for j in cycle(range(k)):
        with torch.cuda.stream(venv.stream(j)):                       # range j's policy on range j's own stream: nothing to order across streams
            obs_j, reward_j, term_j, trunc_j, info_j = venv.recv(j)   # rows of range j (views of the full tensors)
            venv.send(policy(obs_j), j)

(The full-batch step() of a pipelined env orders every range against the caller's stream twice per step -- two HIP calls of ~7 us and a
device-side hand-over each way -- and is slower than the lock-step env; the halves on their own streams are what sub-batches are for.)

Returned tensors are CUDA tensors owned by the environment and rewritten in place by the next `step()` -- copy what must outlive it (a
rollout buffer does that anyway).  With one agent per arena the agent axis is dropped: obs [N, ...], reward [N]; with several agents
[N, num_agents, ...].
"""
import ctypes

import numpy as np

from . import _capi
from . import gym_agario as _single
from . import spaces as _spaces
from .vec_env import PipelinedVecEnvironment, VecEnvironment, default_sub_batches


class AgarioVectorEnv:
    metadata = {"render_modes": [], "autoreset_mode": "same_step"}

    def __init__(self, num_envs, obs_type="grid", device=0, channels_last=False, sub_batches="auto", halves=False, on_capacity_flag="raise", **kwargs):
        """num_envs arenas; obs_type "grid" | "screen" | "ram" | "gobigger" | "none"; channels_last: grid observations as a [.., G, G, C]
        VIEW (what AgarioEnv hands out per arena) instead of the engine's channel-first layout; sub_batches: an int, or "auto" (the default): ONE range for the
        full-batch step(); with halves=True (sampling through recv(j) / send(.., j) on the ranges' own streams) vec_env.default_sub_batches -- 4
        where the general engine handles most arena-steps (bots / several agents / modes 5 and 6), 1 for quiet batches -- see the module text
        and the measurement beside the choice below; on_capacity_flag: what happens to an arena that raises a capacity flag
        (README "capacity flags": nearly always the reference's own arithmetic leaving its domain, e.g. the anti-team decay factor going
        negative in a tiny arena full of viruses) -- "raise" (default: the next step() raises AgarclError -6 until reset()), "reset" (the arena is
        reset inside the same step like an ended episode, its rows truncated = True: a diverged arena never feeds a learner) or "ignore"; every other
        keyword as AgarioEnv takes it (difficulty, ticks_per_step, arena_size, num_pellets, num_viruses, num_bots, pellet_regen, reward_type,
        c_death, mode, num_agents, number_steps, env_type, grid_size, observe_*, screen_len, agent_view, k_cells / k_pellets / k_viruses /
        k_others)."""
        import torch
        if on_capacity_flag not in ("raise", "reset", "ignore"):
            raise ValueError("on_capacity_flag must be 'raise', 'reset' or 'ignore'")
        if on_capacity_flag != "raise":
            kwargs = dict(kwargs, strict_flags=False)
        self.on_capacity_flag = on_capacity_flag
        if obs_type not in _single.OBS_TYPES + ("none",):
            raise ValueError("obs_type must be one of %s, got %r" % (_single.OBS_TYPES + ("none",), obs_type))
        self.torch = torch
        o = _single._resolve_options(kwargs)
        self.options, self.obs_type, self.num_envs = o, obs_type, int(num_envs)
        self.num_agents, self.multi_agent = o["num_agents"], o["multi_agent"]
        self.number_of_steps, self.env_type = o["number_steps"], o["env_type"]
        self.channels_last = bool(channels_last)
        if sub_batches == "auto":
            # The full-batch step() orders every range against the caller's stream twice per step (fork / join) and a learner's policy sits between
            # two steps, so the ranges restart together every step: nothing is staggered.  Measured at 4096 arenas, k = 1 / 2 / 4
            # (scripts/gpu_vec_pipe_ab.py, profiles/r06_vec_pipe_ab.txt), with the flag-word fork / join of round 6 (HIP events in brackets): mode 6
            # 366 / 370 / 554 us per vector step (411 / 731), mode 6 + 84 x 84 screen 450 / 591 / 666 (599 / 919), task 6 with its 128 x 128 frame
            # 548 / 625 / 611 (760 / 900).  So "auto" is ONE range for step(); ranges pay where nothing orders them against each other -- the recv(j) /
            # send(.., j) halves on the ranges' own streams (halves=True: "auto" is then vec_env.default_sub_batches, 4 for bots / several agents /
            # modes 5 and 6) and free-running engine loops (PipelinedVecEnvironment; bench.py's <workload>/pipe4 rows)
            sub_batches = default_sub_batches(self.num_envs, o["num_agents"], o["num_bots"], o["mode"]) if (halves and obs_type != "gobigger") else 1
        self.sub_batches = int(sub_batches)
        if self.sub_batches < 1 or self.sub_batches > self.num_envs:
            raise ValueError("sub_batches must be in [1, num_envs]")
        if obs_type == "gobigger" and self.sub_batches > 1:
            raise ValueError("the GoBigger observation (five padded tensors per call) is available with sub_batches=1 only")
        eng = dict(num_agents=o["num_agents"], ticks_per_step=o["ticks_per_step"], arena_size=o["arena_size"], pellet_regen=o["pellet_regen"],
                   num_pellets=o["num_pellets"], num_viruses=o["num_viruses"], num_bots=o["num_bots"], reward_type=o["reward_type"], c_death=o["c_death"],
                   mode_number=o["mode"], device=device, **{k: kwargs[k] for k in ("strict_flags", "dt", "cap_foods", "cap_viruses") if k in kwargs})
        if obs_type == "screen":
            # the screen env respawns a dead agent right after the ticks of a step in every mode and adds c_death to that step's reward
            # (/root/reference/environment/envs/ScreenEnvironment.hpp:233-243): what agarcl.ScreenEnvironment -- the N = 1 case -- does too
            eng["screen_respawn"] = True
        if self.sub_batches == 1:
            self.env = VecEnvironment(self.num_envs, **eng)
            self.pipe, self._parts, self._ranges = None, [self.env], [(0, self.num_envs)]
        else:
            self.pipe = PipelinedVecEnvironment(self.num_envs, self.sub_batches, **eng)
            self.env, self._parts, self._ranges = None, self.pipe.parts, self.pipe.ranges
        self.concurrent_sub_batches = self.pipe.concurrent if self.pipe is not None else 1
        self.strict_flags = bool(kwargs.get("strict_flags", True))
        self.device = self._parts[0].device
        N, n = self.num_envs, self.num_agents
        # ---- the observation: its configuration, per-agent shape, and ONE tensor for all arenas (each sub-batch writes its rows) -------------
        spec_kind, arg, shape, dtype = _capi.OBS_NONE, [0] * 6, None, None
        if obs_type == "grid":
            a = dict(grid_size=int(kwargs.get("grid_size", 128)), cells=bool(kwargs.get("observe_cells", True)), others=bool(kwargs.get("observe_others", True)),
                     viruses=bool(kwargs.get("observe_viruses", True)), pellets=bool(kwargs.get("observe_pellets", True)))
            C = 1 + int(a["cells"]) + 2 * (int(a["others"]) + int(a["viruses"]) + int(a["pellets"]))
            spec_kind, arg[:5], shape, dtype = _capi.OBS_GRID, [a["grid_size"], a["cells"], a["others"], a["viruses"], a["pellets"]], (C, a["grid_size"], a["grid_size"]), torch.int32
            self._obs_args = a
        elif obs_type == "screen":
            side = int(kwargs.get("screen_len", 84))
            spec_kind, arg[:3], shape, dtype = _capi.OBS_SCREEN, [side, side, int(bool(o["agent_view"]))], (side, side, 4 if o["agent_view"] else 3), torch.uint8
            self._obs_args = dict(width=side, height=side, agent_view=o["agent_view"])
        elif obs_type == "ram":
            k = [int(kwargs.get(key, dflt)) for key, dflt in (("k_cells", 16), ("k_pellets", 16), ("k_viruses", 8), ("k_others", 16))]
            spec_kind, arg[:4], shape, dtype = _capi.OBS_RAM, k, (4 + 3 * k[0] + 2 * k[1] + 3 * k[2] + 3 * k[3],), torch.float32
            self._obs_args = dict(zip(("k_cells", "k_pellets", "k_viruses", "k_others"), k))
        else:
            self._obs_args = {k: kwargs[k] for k in ("grid_size", "cap_food", "cap_virus", "cap_spore", "cap_clone") if k in kwargs}
        self._obs = torch.zeros((N, n) + shape, dtype=dtype, device=self.device) if shape is not None else None
        self._obs_row_bytes = (int(np.prod(shape)) * n * self._obs.element_size()) if shape is not None else 0
        # ---- bookkeeping tensors (include/agarcl_vec.h agarcl_vec_buffers), one allocation each for all arenas -------------------------------
        self._steps = torch.zeros(N, dtype=torch.int32, device=self.device)       # steps played in the current episode, per arena
        self._ep_return = torch.zeros((N, n), dtype=torch.float32, device=self.device)
        self._final_return = torch.zeros((N, n), dtype=torch.float32, device=self.device)
        self._final_length = torch.zeros(N, dtype=torch.int32, device=self.device)
        self._reward = torch.zeros((N, n), dtype=torch.float32, device=self.device)
        self._done = torch.zeros((N, n), dtype=torch.bool, device=self.device)
        self._trunc = torch.zeros((N, n), dtype=torch.bool, device=self.device)   # several agents only: reset under an agent that was not done
        self._mask = torch.zeros(N, dtype=torch.uint8, device=self.device)        # "ended"
        self._started = False
        self._L = self._parts[0].engine.L
        self._spec = _capi.VecSpec(int(self.number_of_steps), 1 if self.env_type == 0 else 0, 0, spec_kind, (ctypes.c_int32 * 6)(*[int(x) for x in arg]), 0,
                                   1 if on_capacity_flag == "reset" else 0)
        self._bufs = []
        for lo, cnt in self._ranges:
            off = lambda t, per_row: t.data_ptr() + lo * per_row
            self._bufs.append(_capi.VecBuffers(off(self._steps, 4), off(self._reward, 4 * n), off(self._done, n), off(self._trunc, n), off(self._mask, 1),
                                               off(self._ep_return, 4 * n), off(self._final_return, 4 * n), off(self._final_length, 4),
                                               (self._obs.data_ptr() + lo * self._obs_row_bytes) if self._obs is not None else None))
        self._flags = ctypes.c_uint32(0)
        self._act_keep = [None] * self.sub_batches
        # what step() / recv(j) hand out: views of the tensors above, built ONCE (slicing a tensor costs microseconds, and step() is a per-step path)
        self._ret_full = None if obs_type == "gobigger" else self._make_ret(0, None)
        self._ret_part = [None if obs_type == "gobigger" else self._make_ret(lo, lo + cnt) for lo, cnt in self._ranges]
        self._byref = [(ctypes.byref(self._spec), ctypes.byref(b)) for b in self._bufs]
        self._flags_ref = ctypes.byref(self._flags)
        self._vec_step = self._L.agarcl_vec_step
        self._handles = [p.engine.h for p in self._parts]
        # ---- spaces -----------------------------------------------------------------------------------------------------------------------------
        self.single_observation_shape = self._single_obs_shape(shape)
        self.single_action_space = _spaces.single_action_space(n, self.multi_agent)
        self.action_space = _spaces.batched_action_space(N, n, self.multi_agent)
        if self.single_observation_shape is not None and obs_type in _spaces.OBS_BOUNDS:
            self.single_observation_space = _spaces.observation_space(obs_type, self.single_observation_shape)
            self.observation_space = _spaces.observation_space(obs_type, (N,) + self.single_observation_shape)
        else:       # "none", and the GoBigger observation (a dict of padded tensors per player: no single Box describes it)
            self.single_observation_space = self.observation_space = None

    # ---- helpers ------------------------------------------------------------------------------------------------------------------
    def _single_obs_shape(self, shape):
        if shape is None:
            return None
        if self.obs_type == "grid" and self.channels_last:
            shape = (shape[1], shape[2], shape[0])
        return ((self.num_agents,) + tuple(shape)) if self.multi_agent else tuple(shape)

    def _agents(self, t):
        """[N, num_agents, ...] -> [N, ...] for single-agent envs"""
        return t if self.multi_agent else t[:, 0]

    def _obs_view(self, lo=0, hi=None):
        if self.obs_type == "none":
            return None
        if self.obs_type == "gobigger":
            return self.env.gobigger_obs(**self._obs_args)        # rows per PLAYER (agents and bots), see VecEnvironment.gobigger_obs
        t = self._agents(self._obs[lo:hi])
        if self.obs_type == "grid" and self.channels_last:
            t = t.movedim(-3, -1)
        return t

    def _as_device(self, x, dtype, shape):
        torch = self.torch
        if isinstance(x, torch.Tensor) and x.dtype == dtype and x.is_cuda and x.is_contiguous() and x.numel() == int(np.prod(shape)):
            return x                                              # the policy's own tensor, handed through
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(np.asarray(x), device=self.device)
        if x.numel() != int(np.prod(shape)):
            raise ValueError("actions for %d arenas x %d agents expected, got a tensor of shape %s" % (shape[0], self.num_agents, tuple(x.shape)))
        return x.to(device=self.device, dtype=dtype).reshape(shape).contiguous()

    def _chk(self, rc):
        if rc != 0:
            raise _capi.AgarclError(rc, (self._L.agarcl_last_error() or b"").decode())

    def _check_flags(self):
        if self.strict_flags and self._flags.value:
            raise _capi.AgarclError(-6, "capacity flags 0x%x were raised in at least one arena (it has diverged from the reference's unbounded "
                                        "containers): read engine.flags() and reset those arenas" % self._flags.value)

    def _info(self, lo=0, hi=None):
        # "ended": the arenas whose episode ended in this step (and were reset); for those rows "final_return" / "final_length" are the finished
        # episode's return and length (they keep their last value otherwise); "episode_return" / "episode_steps" run with the current episode
        return {"episode_steps": self._steps[lo:hi], "episode_return": self._agents(self._ep_return[lo:hi]), "ended": self._mask[lo:hi].view(self.torch.bool),
                "final_return": self._agents(self._final_return[lo:hi]), "final_length": self._final_length[lo:hi]}

    def _seed(self, seed):
        if seed is None:
            return
        if self.pipe is not None:
            self.pipe.seed(None, int(seed)) if np.isscalar(seed) else self.pipe.seed(np.asarray(seed, dtype=np.uint32))
        elif np.isscalar(seed):
            self.env.seed(base_seed=int(seed))
        else:
            self.env.seed(np.asarray(seed, dtype=np.uint32))

    def _make_ret(self, lo, hi):
        return (self._obs_view(lo, hi), self._agents(self._reward[lo:hi]), self._agents(self._done[lo:hi]), self._agents(self._trunc[lo:hi]), self._info(lo, hi))

    def _ret(self, j=None):
        if self.obs_type == "gobigger":                           # (five tensors per call: built per step, sub_batches == 1 only)
            return self._make_ret(0, None)
        return self._ret_full if j is None else self._ret_part[j]

    def _reset_part(self, j):
        p = self._parts[j]
        p.order_after_current()
        self._chk(self._L.agarcl_vec_reset(p.engine.h, ctypes.byref(self._spec), ctypes.byref(self._bufs[j])))

    def _step_part(self, j, move_ptr, kind_ptr):
        """one host call: the step, the episode bookkeeping with the same-step auto-reset, and the observation of sub-batch j"""
        sp, bf = self._byref[j]
        rc = self._vec_step(self._handles[j], sp, bf, move_ptr, kind_ptr, self._flags_ref)
        if rc != 0:
            self._chk(rc)
        return self._flags.value

    # ---- the vector-env surface -------------------------------------------------------------------------------------------------
    def reset(self, seed=None, options=None):
        """all arenas start a new episode.  seed: None (the arenas' random streams continue), an int s (arena a gets seed s + a, as
        num_envs separate `env.seed(s + a)` calls would give), or a sequence of num_envs seeds."""
        self.async_reset(seed)
        obs = self._ret()[0]                                       # (GoBigger: enqueues its observation kernel -- before the current stream is ordered after the engine's)
        for p in self._parts:
            p.order_current_after()
        return obs, {}

    def step(self, actions):
        """actions = (move, kind): move f32 [N, 2] (or [N, num_agents, 2]) in [-1, 1]^2, kind int [N] (or [N, num_agents]) in {0 none, 1 feed,
        2 split}; CUDA tensors of the right type are used in place, anything else is converted / uploaded.  Everything is enqueued -- on the
        current CUDA stream, or on the sub-batches' own streams ordered against it -- and nothing waits for it.  (Values are not range-checked
        here -- that would need them on the host; the engine clamps nothing either, as the reference's take_action: BaseEnvironment.hpp:162-176.)"""
        assert self._started, "reset() must be called before the first step()"
        torch = self.torch
        N, n = self.num_envs, self.num_agents
        move = self._as_device(actions[0], torch.float32, (N, n, 2)); kind = self._as_device(actions[1], torch.int32, (N, n))
        self._act_keep[0] = (move, kind)                           # (read when the step kernel executes)
        mp, kp = move.data_ptr(), kind.data_ptr()
        if self.pipe is None:
            p = self._parts[0]
            cur = p._current_raw_stream()
            if cur != p.stream_handle:                             # (stepping under another torch stream than the one the engine was bound to)
                p.engine.stream_wait(cur)
            flags = self._step_part(0, mp, kp)
            ret = self._ret()                                      # (GoBigger: its observation kernel is enqueued here, BEFORE the caller's stream is ordered after the engine's)
            if cur != p.stream_handle:
                p.engine.stream_signal(cur)
        else:
            cur = self._parts[0]._current_raw_stream()
            self.pipe.pipe.fork(cur)                               # the actions were produced on the current stream: one event, every sub-batch waits for it
            flags = 0
            for j, (lo, cnt) in enumerate(self._ranges):           # every range is stepped and joined even when one reports a flag: they stay in lock-step
                flags |= self._step_part(j, mp + lo * n * 8, kp + lo * n * 4)
            self.pipe.pipe.join(cur)
            ret = self._ret()
        if flags:
            self._flags.value = flags
            self._check_flags()                                    # raises once, after everything launched has been ordered (strict_flags)
        return ret

    # ---- the two halves, per sub-batch (double-buffered sampling) ---------------------------------------------------------------------
    def async_reset(self, seed=None):
        """reset() without the wait of the current stream: follow with recv(j) per sub-batch"""
        self._seed(seed)
        for j in range(self.sub_batches):
            self._reset_part(j)
        self._started = True

    def send(self, actions, j=0):
        """enqueue the step of sub-batch j: actions = (move [n_j, 2], kind [n_j]) for ITS arenas (ranges[j])"""
        assert self._started, "reset() / async_reset() must be called before the first send()"
        lo, cnt = self._ranges[j]
        n = self.num_agents
        move = self._as_device(actions[0], self.torch.float32, (cnt, n, 2)); kind = self._as_device(actions[1], self.torch.int32, (cnt, n))
        self._act_keep[j] = (move, kind)
        self._parts[j].order_after_current()                      # the actions were produced on the current stream (no-op when that is the engine's)
        if self._step_part(j, move.data_ptr(), kind.data_ptr()):
            self._check_flags()

    def recv(self, j=0):
        """the current stream waits (on the device) for sub-batch j's last reset / step; returns its rows (views of the full tensors):
        obs, reward, terminated, truncated, info"""
        self._parts[j].order_current_after()
        return self._ret(j)

    def stream(self, j=0):
        """the torch stream sub-batch j launches on: `with torch.cuda.stream(venv.stream(j)):` around recv(j) / policy / send(.., j) keeps range
        j's whole loop on ONE stream -- no cross-stream ordering at all, which is what makes sub-batches pay"""
        return self._parts[j].torch_stream()

    @property
    def ranges(self):
        """[(first arena, count)] per sub-batch"""
        return list(self._ranges)

    def close(self):
        (self.pipe if self.pipe is not None else self.env).close()
