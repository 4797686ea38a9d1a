"""Batched environment: N independent arenas stepped in lock-step by one HIP launch.

Host-side mirror of the reference's environment surface
(/root/reference/environment/bindings.cpp:99-135, environment/envs/BaseEnvironment.hpp:33-428) for
many arenas at once: same constructor argument names and meaning, same `seed / reset / take_actions /
step / dones` verbs, but tensors (one row per arena) instead of Python lists of one arena.
PyTorch is used only for device memory / streams; all simulation happens in libagarcl_hip.so.
"""
import numpy as np

from . import _capi


class _DevArray:
    """Zero-copy view of engine-owned HBM for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr,
                                         "version": 2, "strides": None}


class VecEnvironment:
    def __init__(self, num_arenas, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True,
                 num_pellets=1000, num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode_number=0,
                 device=0, dt=1.0 / 30, use_torch_stream=True, strict_flags=True, engine=None, **caps):
        """engine: an existing _capi.BatchedEngine to wrap instead of creating one (a sub-batch of a PipelinedVecEnvironment: it keeps its
        own, verified-concurrent stream, so use_torch_stream must be False for it)"""
        import torch
        if engine is not None and use_torch_stream:
            # agarcl_set_stream on an adopted sub-batch would replace the stream agarcl_pipe_create verified to be concurrent (and silently serialise the pipe)
            raise ValueError("VecEnvironment(engine=...) wraps a sub-batch that keeps its own stream: pass use_torch_stream=False")
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.num_arenas, self.num_agents, self.ticks_per_step = num_arenas, num_agents, ticks_per_step
        self.engine = engine if engine is not None else _capi.BatchedEngine(num_arenas, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets,
                                                                            num_viruses, num_bots, reward_type, c_death, mode_number, dt, device, **caps)
        self._torch_stream = bool(use_torch_stream)
        if use_torch_stream:
            # launch on torch's current stream so torch events / collectives order against the engine
            with torch.cuda.device(self.device):
                self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
        self.stream_handle = self.engine.stream()   # the hipStream_t (int) the engine launches on: torch's at construction, or its own
        p = self.engine.device_ptrs()
        A, n = num_arenas, num_agents
        self.rewards = torch.as_tensor(_DevArray(p["rewards"], (A, n), "<f8"), device=self.device)
        self.dones_u8 = torch.as_tensor(_DevArray(p["dones"], (A, n), "|u1"), device=self.device)
        self.masses = torch.as_tensor(_DevArray(p["masses"], (A, n), "<i4"), device=self.device)
        # (reward, done) f32 pairs in a ring of PACKED_SLOTS buffers: packed[engine.last_slot()] belongs to the last step;
        # packed_ring[B * h : B * h + B] is one contiguous block of B consecutive steps (what bench.py gathers per collective, B = 32)
        self.packed_ring = torch.as_tensor(_DevArray(p["packed"], (_capi.PACKED_SLOTS, A * n, 2), "<f4"), device=self.device)
        self.packed = [self.packed_ring[k] for k in range(_capi.PACKED_SLOTS)]
        self._act_keep = None
        # a capacity flag means that arena has left the reference's (unbounded-vector) semantics: by default that is an error
        # the caller hears about (within ~64 steps, without any synchronisation), not a silently diverged arena
        self.strict_flags = strict_flags

    def seed(self, seeds=None, base_seed=0):
        self.engine.seed(seeds, base_seed)

    def reset(self, mask=None, reset_ids=False):
        """mask: None (all arenas), a host uint8/bool array, or a CUDA uint8 tensor [A] such as self.dones_u8[:, 0] made
        contiguous -- the device form is a pure stream-ordered launch (no copy, no synchronisation) when the engine runs on torch's
        stream (use_torch_stream=True, the default).  With another stream current, the engine's stream first waits -- on the device -- for it
        (agarcl_stream_wait): nothing else orders the cast below against the reset kernel.
        Every reset also restarts the capacity-flag watch (strict_flags): after reset(mask) only arenas that are still flagged count."""
        torch = self.torch
        if isinstance(mask, torch.Tensor) and mask.is_cuda:
            m = mask.to(torch.uint8).contiguous().reshape(-1)
            if m.numel() != self.num_arenas:
                raise ValueError("mask must have one entry per arena")
            self._mask_keep = m
            self.order_after_current()        # (the cast above ran on torch's current stream; a no-op when that is the engine's stream)
            self.engine.reset_device(m.data_ptr(), reset_ids)
        else:
            self.engine.reset(None if mask is None else np.asarray(mask), reset_ids)

    def take_actions(self, dxdy, act):
        """dxdy: float32 [A, n_agents, 2]; act: int32 [A, n_agents]; torch CUDA tensors (zero copy) or
        host arrays.  Action enum as in the reference: 0 none, 1 feed, 2 split (core/types.hpp:59-61)."""
        torch = self.torch
        if isinstance(dxdy, torch.Tensor):
            dxdy, act = self._device_actions(dxdy, act)
            if self._current_raw_stream() != self.stream_handle:
                # a conversion above (int64 -> int32, a non-contiguous slice) ran on torch's CURRENT stream: the engine's stream waits for it
                self.order_after_current()
            self._act_keep = (dxdy, act)
            self.engine.set_actions_device(dxdy.data_ptr(), act.data_ptr())
        else:
            dxdy = np.asarray(dxdy, dtype=np.float32)
            act = np.asarray(act, dtype=np.int32)
            if dxdy.size != self.num_arenas * self.num_agents * 2 or act.size != self.num_arenas * self.num_agents:
                raise RuntimeError("Number of actions does not match number of agents")
            self.engine.set_actions(dxdy, act)

    def _device_actions(self, dxdy, act):
        """CUDA action tensors in the layout the kernels read (f32 / i32, contiguous).  Any conversion is a torch kernel on the CURRENT stream:
        callers order the engine's stream after it (order_after_current / pipe.fork) AFTER this returns, never before."""
        torch = self.torch
        if dxdy.device != self.device or act.device != self.device:
            raise ValueError("action tensors must live on %s" % self.device)
        dxdy = dxdy.to(torch.float32).contiguous()
        act = act.to(torch.int32).contiguous()
        if dxdy.numel() != self.num_arenas * self.num_agents * 2 or act.numel() != self.num_arenas * self.num_agents:
            raise RuntimeError("Number of actions does not match number of agents")  # BaseEnvironment.hpp:142-144
        return dxdy, act

    def step(self, ticks=0):
        """Enqueue one env step (ticks_per_step engine ticks); results land in self.rewards /
        self.dones_u8 / self.masses (HBM, valid once the stream reaches this point)."""
        self.engine.step(ticks)
        if self.strict_flags:
            fl = self.engine.poll_flags()
            if fl:
                raise _capi.AgarclError(-6, "capacity flags 0x%x were raised in at least one arena (it has diverged from the reference's "
                                            "unbounded containers): read engine.flags() and reset those arenas" % fl)
        return self.rewards

    def ram_obs(self, out=None, k_cells=16, k_pellets=16, k_viruses=8, k_others=16):
        """float32 CUDA tensor [A, n_agents, D] of the "ram" observation (include/agarcl_batch.h agarcl_ram_obs), written by one launch on
        the env's stream into the env's own tensor of this configuration (or into `out`)."""
        torch = self.torch
        D = 4 + 3 * k_cells + 2 * k_pellets + 3 * k_viruses + 3 * k_others
        if out is None:
            out = self._obs_tensor(("ram", k_cells, k_pellets, k_viruses, k_others), (self.num_arenas, self.num_agents, D), torch.float32)
        self.engine.ram_obs(k_cells, k_pellets, k_viruses, k_others, out_ptr=out.data_ptr())
        return out

    # ---- observations as persistent CUDA tensors: one tensor per (kind, configuration), created on first use and rewritten in place by
    # every later call on the env's stream -- no allocation, no host copy, no synchronisation inside a rollout loop --------------------------
    def _obs_tensor(self, key, shape, dtype):
        cache = self.__dict__.setdefault("_obs_cache", {})
        t = cache.get(key)
        if t is None:
            t = cache[key] = self.torch.zeros(shape, dtype=dtype, device=self.device)
        return t

    def grid_obs(self, grid_size=128, cells=True, others=True, viruses=True, pellets=True, out=None):
        """int32 CUDA tensor [A, n_agents, C, G, G], C = 1 + cells + 2 others + 2 viruses + 2 pellets (GridEnvironment.hpp:188-196), of the
        state after the last step.  Without `out` the env's own tensor of this configuration is rewritten through the incremental path
        (agarcl_grid_obs on_device = 2: only the words the previous call scattered are cleared); it belongs to the env -- read it, copy it,
        do not write into it.  With `out` (a contiguous CUDA int32 tensor of that shape) the stateless full-clear path is used."""
        torch = self.torch
        C = 1 + int(bool(cells)) + 2 * (int(bool(others)) + int(bool(viruses)) + int(bool(pellets)))
        shape = (self.num_arenas, self.num_agents, C, int(grid_size), int(grid_size))
        if out is None:
            t = self._obs_tensor(("grid", int(grid_size), bool(cells), bool(others), bool(viruses), bool(pellets)), shape, torch.int32)
            self.engine.grid_obs(grid_size, cells, others, viruses, pellets, out_ptr=t.data_ptr(), persistent=True)
            return t
        if out.dtype != torch.int32 or tuple(out.shape) != shape or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous int32 tensor of shape %s on %s" % (shape, self.device))
        self.engine.grid_obs(grid_size, cells, others, viruses, pellets, out_ptr=out.data_ptr(), persistent=False)
        return out

    def screen_obs(self, width=84, height=84, agent_view=False, out=None):
        """uint8 CUDA tensor [A, n_agents, height, width, 3] (4 channels with agent_view), rows bottom-up like glReadPixels
        (include/agarcl_batch.h agarcl_screen_obs); the env's own tensor of this configuration unless `out` is given."""
        torch = self.torch
        shape = (self.num_arenas, self.num_agents, int(height), int(width), 4 if agent_view else 3)
        if out is None:
            out = self._obs_tensor(("screen", int(width), int(height), bool(agent_view)), shape, torch.uint8)
        elif out.dtype != torch.uint8 or tuple(out.shape) != shape or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous uint8 tensor of shape %s on %s" % (shape, self.device))
        self.engine.screen_obs(width, height, out_ptr=out.data_ptr(), agent_view=agent_view)
        return out

    def gobigger_obs(self, grid_size=128, cap_food=256, cap_virus=64, cap_spore=64, cap_clone=32):
        """the GoBigger observation as padded CUDA tensors (include/agarcl_batch.h agarcl_gobigger_obs): {"hdr": i32 [A, P, 8],
        "food" / "virus" / "spore": f32 [A, P, cap, 4], "clone": f32 [A, P, cap_clone, 7]}, P = players per arena; the env's own tensors."""
        torch = self.torch
        A, P = self.num_arenas, self.engine.players
        key = ("gobigger", int(grid_size), int(cap_food), int(cap_virus), int(cap_spore), int(cap_clone))
        t = {"hdr": self._obs_tensor(key + ("hdr",), (A, P, 8), torch.int32),
             "food": self._obs_tensor(key + ("food",), (A, P, cap_food, 4), torch.float32),
             "virus": self._obs_tensor(key + ("virus",), (A, P, cap_virus, 4), torch.float32),
             "spore": self._obs_tensor(key + ("spore",), (A, P, cap_spore, 4), torch.float32),
             "clone": self._obs_tensor(key + ("clone",), (A, P, cap_clone, 7), torch.float32)}
        self.engine.gobigger_obs(grid_size, cap_food, cap_virus, cap_spore, cap_clone,
                                 out_ptrs=[t[k].data_ptr() for k in ("hdr", "food", "virus", "spore", "clone")])
        return t

    def dones(self):
        return self.dones_u8.bool()

    def torch_stream(self):
        """the engine's stream as a torch stream object (`with torch.cuda.stream(env.torch_stream()):` runs torch ops in order with the engine)"""
        if self.stream_handle == 0:
            return self.torch.cuda.default_stream(self.device)
        return self.torch.cuda.ExternalStream(self.stream_handle, device=self.device)

    def _current_raw_stream(self):
        try:       # (the raw handle without building a Stream object: this sits in the per-step path)
            return self.torch._C._cuda_getCurrentRawStream(self.device.index)
        except AttributeError:
            return self.torch.cuda.current_stream(self.device).cuda_stream

    def order_after_current(self):
        """the engine's next launches come after everything enqueued so far on torch's CURRENT stream (no-op when that is the engine's stream)"""
        cur = self._current_raw_stream()
        if cur != self.stream_handle:
            self.engine.stream_wait(cur)

    def order_current_after(self):
        """torch's CURRENT stream waits, on the device, for everything the engine has enqueued so far"""
        cur = self._current_raw_stream()
        if cur != self.stream_handle:
            self.engine.stream_signal(cur)

    def sync(self):
        self.engine.sync()

    def close(self):
        self.engine.close()


def default_sub_batches(num_arenas, num_agents=1, num_bots=0, mode_number=0, **_):
    """How many independent sub-batches a batch of this configuration is stepped as when the caller does not say.

    4 where the general engine (k_step) handles most arena-steps -- several players per arena (bots or several agents), or an agent that
    starts at mass 1000 (modes 5 and 6): a launch lasts as long as its slowest arena, and four ranges on streams of their own stop 3/4 of the
    arenas from waiting for it (measured at 4096 arenas, DESIGN.md section 5: mode 6 351 -> 303 us per step, C1 90 -> 67 us).
    1 for the quiet configurations (one mass-25 agent per arena: the front kernel finishes nearly every arena-step in ~10 us): there the
    fork / join of a pipelined step costs more than the step.  Small batches stay whole (a range below 256 arenas cannot fill the chip)."""
    heavy = (int(num_agents) + int(num_bots) > 1) or int(mode_number) in (5, 6)
    return 4 if heavy and int(num_arenas) >= 1024 else 1


class PipelinedVecEnvironment:
    """`num_arenas` arenas as `sub_batches` independent sub-batches (include/agarcl_batch.h agarcl_pipe_*): contiguous arena ranges, each a
    VecEnvironment (`parts[j]`, arenas `ranges[j] = (first, count)`) on a HIP stream of its own that was verified to run concurrently with
    the others.  The reference's vectorised runner lets every engine run ahead on its own pool thread and waits once at the end
    (/root/reference/agario/bots/benchmark.cpp:149-167); here a sub-batch's step does not wait for the slowest arena of another, and one
    sub-batch's observation kernel runs under another's step.

        send(j, dxdy_j, act_j)   enqueue take_actions + step of sub-batch j (after whatever produced the tensors on the current stream)
        recv(j)                  the current stream waits (on the device) for sub-batch j; returns parts[j] (rewards / dones_u8 / masses / *_obs)
        step(dxdy, act)          all sub-batches with full-batch tensors [A, n, ..]: send each, then recv each

    Arena first_j + a computes exactly what arena first_j + a of one VecEnvironment over all arenas computes (seeds by global index)."""

    def __init__(self, num_arenas, sub_batches="auto", num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                 num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode_number=0, device=0, dt=1.0 / 30, strict_flags=True, **caps):
        import torch
        if sub_batches == "auto":      # by workload (default_sub_batches); a quiet configuration becomes ONE range: the lock-step env behind this surface
            sub_batches = default_sub_batches(num_arenas, num_agents, num_bots, mode_number)
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.num_arenas, self.num_agents, self.sub_batches = num_arenas, num_agents, sub_batches
        self.pipe = _capi.PipelinedEngine(num_arenas, sub_batches, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses,
                                          num_bots, reward_type, c_death, mode_number, dt, device, **caps)
        self.ranges = list(self.pipe.ranges)
        self.concurrent = self.pipe.concurrent     # sub-batch streams verified to overlap (== sub_batches unless hardware queues ran out)
        self.parts = [VecEnvironment(n, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets, num_viruses, num_bots, reward_type, c_death,
                                     mode_number, device, dt, use_torch_stream=False, strict_flags=strict_flags, engine=self.pipe.parts[j])
                      for j, (lo, n) in enumerate(self.ranges)]

    def seed(self, seeds=None, base_seed=0):
        self.pipe.seed(seeds, base_seed)

    def reset(self, reset_ids=False):
        for p in self.parts:
            p.reset(None, reset_ids)

    def send(self, j, dxdy, act, ticks=0):
        p = self.parts[j]
        if isinstance(dxdy, self.torch.Tensor):
            dxdy, act = p._device_actions(dxdy, act)      # convert FIRST (torch kernels on the current stream) ...
            p.order_after_current()                       # ... then order the sub-batch's stream after the current one: the event covers them
            p._act_keep = (dxdy, act)
            p.engine.set_actions_device(dxdy.data_ptr(), act.data_ptr())
        else:
            p.order_after_current()
            p.take_actions(dxdy, act)
        p.step(ticks)

    def recv(self, j):
        p = self.parts[j]
        p.order_current_after()
        return p

    def step(self, dxdy, act, ticks=0):
        torch = self.torch
        cur = self.parts[0]._current_raw_stream()
        if isinstance(dxdy, torch.Tensor):
            # convert the FULL-batch tensors first (torch kernels on the current stream), then fork: the event covers the conversion
            A, n = self.num_arenas, self.num_agents
            if dxdy.device != self.device or act.device != self.device:
                raise ValueError("action tensors must live on %s" % self.device)
            if dxdy.numel() != A * n * 2 or act.numel() != A * n:
                raise RuntimeError("Number of actions does not match number of agents")
            dxdy = dxdy.to(torch.float32).reshape(A, n, 2).contiguous(); act = act.to(torch.int32).reshape(A, n).contiguous()
            self._act_keep = (dxdy, act)
            self.pipe.fork(cur)                               # one event on the current stream, every sub-batch waits for it (one host call)
            for j, (lo, cnt) in enumerate(self.ranges):
                p = self.parts[j]
                p.engine.set_actions_device(dxdy.data_ptr() + lo * n * 8, act.data_ptr() + lo * n * 4)
        else:
            self.pipe.fork(cur)
            for j, (lo, cnt) in enumerate(self.ranges):
                self.parts[j].take_actions(dxdy[lo:lo + cnt], act[lo:lo + cnt])
        # every sub-batch is stepped and the caller's stream joined even when one of them reports a capacity flag: the ranges stay in
        # lock-step and ordered; the flag is raised once, after the join
        err = None
        for p in self.parts:
            try:
                p.step(ticks)
            except _capi.AgarclError as ex:
                if ex.code != -6:
                    raise
                err = err or ex
        self.pipe.join(cur)
        if err is not None:
            raise err
        return list(self.parts)

    def sync(self):
        self.pipe.sync()

    def close(self):
        self.pipe.close()
