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
                 device=0, dt=1.0 / 30, use_torch_stream=True, strict_flags=True, **caps):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.num_arenas, self.num_agents, self.ticks_per_step = num_arenas, num_agents, ticks_per_step
        self.engine = _capi.BatchedEngine(num_arenas, num_agents, ticks_per_step, arena_size, pellet_regen, num_pellets,
                                          num_viruses, num_bots, reward_type, c_death, mode_number, dt, device, **caps)
        self._torch_stream = bool(use_torch_stream)
        if use_torch_stream:
            # launch on torch's current stream so torch events / collectives order against the engine
            with torch.cuda.device(self.device):
                self.engine.set_stream(torch.cuda.current_stream().cuda_stream)
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
        stream (use_torch_stream=True, the default).  With an engine-owned stream the mask's producer (torch's current stream) is
        waited for first: nothing else orders the cast below against the reset kernel.
        Every reset also restarts the capacity-flag watch (strict_flags): after reset(mask) only arenas that are still flagged count."""
        torch = self.torch
        if isinstance(mask, torch.Tensor) and mask.is_cuda:
            m = mask.to(torch.uint8).contiguous().reshape(-1)
            if m.numel() != self.num_arenas:
                raise ValueError("mask must have one entry per arena")
            self._mask_keep = m
            if not self._torch_stream:
                torch.cuda.current_stream(self.device).synchronize()
            self.engine.reset_device(m.data_ptr(), reset_ids)
        else:
            self.engine.reset(None if mask is None else np.asarray(mask), reset_ids)

    def take_actions(self, dxdy, act):
        """dxdy: float32 [A, n_agents, 2]; act: int32 [A, n_agents]; torch CUDA tensors (zero copy) or
        host arrays.  Action enum as in the reference: 0 none, 1 feed, 2 split (core/types.hpp:59-61)."""
        torch = self.torch
        if isinstance(dxdy, torch.Tensor):
            if dxdy.device != self.device or act.device != self.device:
                raise ValueError("action tensors must live on %s" % self.device)
            dxdy = dxdy.to(torch.float32).contiguous()
            act = act.to(torch.int32).contiguous()
            if dxdy.numel() != self.num_arenas * self.num_agents * 2 or act.numel() != self.num_arenas * self.num_agents:
                raise RuntimeError("Number of actions does not match number of agents")  # BaseEnvironment.hpp:142-144
            self._act_keep = (dxdy, act)
            self.engine.set_actions_device(dxdy.data_ptr(), act.data_ptr())
        else:
            dxdy = np.asarray(dxdy, dtype=np.float32)
            act = np.asarray(act, dtype=np.int32)
            if dxdy.size != self.num_arenas * self.num_agents * 2 or act.size != self.num_arenas * self.num_agents:
                raise RuntimeError("Number of actions does not match number of agents")
            self.engine.set_actions(dxdy, act)

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
        the env's stream; pass the previous tensor as `out` to reuse it."""
        torch = self.torch
        D = 4 + 3 * k_cells + 2 * k_pellets + 3 * k_viruses + 3 * k_others
        if out is None:
            out = torch.empty((self.num_arenas, self.num_agents, D), dtype=torch.float32, device=self.device)
        self.engine.ram_obs(k_cells, k_pellets, k_viruses, k_others, out_ptr=out.data_ptr())
        return out

    def dones(self):
        return self.dones_u8.bool()

    def sync(self):
        self.engine.sync()

    def close(self):
        self.engine.close()
