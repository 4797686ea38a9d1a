"""ctypes binding of the C ABI declared in include/agarcl_batch.h (libagarcl_hip.so).

The HIP engine is the only implementation this package ships: if the shared library is missing or
no HIP device is present, loading/creating fails loudly -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (AGARCL_HIP_SO: A/B timing of differently compiled builds of the same source, scripts/ only)
HIP_SO = os.environ.get("AGARCL_HIP_SO") or os.path.join(_HERE, "libagarcl_hip.so")

E_UNSUPPORTED = -3
ARENA_WORDS = 48     # include/agarcl_batch.h AGARCL_ARENA_WORDS
PACKED_SLOTS = 64   # include/agarcl_batch.h AGARCL_PACKED_SLOTS


class AgarclError(RuntimeError):
    """Raised for every non-zero return of the C ABI (the reference raises RuntimeError through
    pybind11 for EngineException / EnvironmentException)."""

    def __init__(self, code, msg):
        super().__init__("agarcl error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("num_agents", C.c_int32), ("ticks_per_step", C.c_int32), ("arena_size", C.c_int32),
        ("pellet_regen", C.c_int32), ("num_pellets", C.c_int32), ("num_viruses", C.c_int32),
        ("num_bots", C.c_int32), ("reward_type", C.c_int32), ("c_death", C.c_int32),
        ("mode_number", C.c_int32), ("dt", C.c_double), ("cap_cells", C.c_int32),
        ("cap_viruses", C.c_int32), ("cap_foods", C.c_int32), ("screen_respawn", C.c_int32), ("example_bots", C.c_int32), ("reserved", C.c_int32 * 3),
    ]


# every symbol include/agarcl_batch.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("agarcl_last_error", C.c_char_p, []),
    ("agarcl_device_count", C.c_int, []),
    ("agarcl_create", C.c_int, [C.POINTER(Config), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("agarcl_destroy", C.c_int, [C.c_void_p]),
    ("agarcl_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_sync", C.c_int, [C.c_void_p]),
    ("agarcl_seed", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    ("agarcl_reset", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ("agarcl_reset_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ("agarcl_set_actions", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    ("agarcl_step", C.c_int, [C.c_void_p, C.c_int32]),
    ("agarcl_step_actions", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    ("agarcl_timer_mark", C.c_int, [C.c_void_p, C.c_int32]),
    ("agarcl_timer_elapsed_ms", C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    ("agarcl_tick", C.c_int, [C.c_void_p, C.c_int32]),
    ("agarcl_set_targets", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("agarcl_respawn_dead", C.c_int, [C.c_void_p]),
    ("agarcl_rewards_dev", C.c_void_p, [C.c_void_p]),
    ("agarcl_dones_dev", C.c_void_p, [C.c_void_p]),
    ("agarcl_masses_dev", C.c_void_p, [C.c_void_p]),
    ("agarcl_packed_dev", C.c_void_p, [C.c_void_p, C.c_int32]),
    ("agarcl_last_slot", C.c_int, [C.c_void_p]),
    ("agarcl_get_rewards", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_get_dones", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_get_masses", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_get_flags", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_poll_flags", C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    ("agarcl_get_counts", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_get_events", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    ("agarcl_grid_obs", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ("agarcl_screen_obs", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32]),
    ("agarcl_ram_obs", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ("agarcl_gobigger_obs", C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    ("agarcl_dump_arena", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    ("agarcl_load_arena", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]),
    ("agarcl_adopt_arena", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    ("agarcl_seed_arena", C.c_int, [C.c_void_p, C.c_int32, C.c_uint32]),
    ("agarcl_get_seeds", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_get_arena_words", C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    ("agarcl_player_words", C.c_int, []),
    ("agarcl_num_arenas", C.c_int, [C.c_void_p]),
    ("agarcl_players_per_arena", C.c_int, [C.c_void_p]),
    ("agarcl_state_bytes", C.c_int64, [C.c_void_p]),
    ("agarcl_device_bytes", C.c_int64, [C.c_void_p]),
    ("agarcl_get_stream", C.c_void_p, [C.c_void_p]),
    ("agarcl_stream_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_stream_signal", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_pipe_create", C.c_int, [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ("agarcl_pipe_destroy", C.c_int, [C.c_void_p]),
    ("agarcl_pipe_sub_batches", C.c_int, [C.c_void_p]),
    ("agarcl_pipe_spin_timeouts", C.c_int, [C.c_void_p]),
    ("agarcl_pipe_env", C.c_void_p, [C.c_void_p, C.c_int32]),
    ("agarcl_pipe_range", C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    ("agarcl_pipe_seed", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32]),
    ("agarcl_pipe_concurrent", C.c_int, [C.c_void_p]),
    ("agarcl_pipe_sync", C.c_int, [C.c_void_p]),
    ("agarcl_pipe_fork", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_pipe_join", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_debug_fused", C.c_int, [C.c_void_p]),
    ("agarcl_debug_qinfo", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_debug_prof", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("agarcl_debug_prof_raw", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_debug_work", C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    ("agarcl_debug_qstat", C.c_int, [C.c_void_p, C.c_void_p]),
    ("agarcl_debug_sqrt_check", C.c_int, [C.c_void_p, C.c_void_p]),
]


# ---- include/agarcl_vec.h: one host call per step of the batched RL surface -------------------------------------------------------
OBS_NONE, OBS_GRID, OBS_SCREEN, OBS_RAM = 0, 1, 2, 3


class VecSpec(C.Structure):
    _fields_ = [("number_steps", C.c_int32), ("episodic", C.c_int32), ("reset_ids", C.c_int32), ("obs_kind", C.c_int32), ("obs_arg", C.c_int32 * 6),
                ("ticks", C.c_int32), ("reset_flagged", C.c_int32), ("reserved", C.c_int32 * 4)]


class VecBuffers(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("steps", "reward", "done", "truncated", "ended", "ep_return", "final_return", "final_length", "obs")]


VEC_SYMBOLS = [
    ("agarcl_vec_reset", C.c_int, [C.c_void_p, C.POINTER(VecSpec), C.POINTER(VecBuffers)]),
    ("agarcl_vec_step", C.c_int, [C.c_void_p, C.POINTER(VecSpec), C.POINTER(VecBuffers), C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
]


def bind(cdll):
    """Attach prototypes for every declared symbol (both headers); raises AttributeError if one is missing."""
    for name, res, args in SYMBOLS + VEC_SYMBOLS:
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll


_hip = None


def hip_lib():
    """The product library.  Fails loudly when it has not been built (python -c 'import
    __graft_entry__ as g; g.build()' or agarcl_amd/build.py)."""
    global _hip
    if _hip is None:
        try:  # if PyTorch is going to be used in this process, its HIP runtime must be the one that is loaded first
            import torch  # noqa: F401
        except Exception:
            pass
        if not os.path.exists(HIP_SO):
            raise AgarclError(-4, "HIP extension %s is missing -- build it with agarcl_amd/build.py "
                                  "(there is no CPU fallback)" % HIP_SO)
        _hip = bind(C.CDLL(HIP_SO))
    return _hip


def _ptr(x):
    """host numpy array or raw device pointer (int) -> void*"""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.ctypes.data)


class BatchedEngine:
    """Thin object wrapper over one agarcl_env (N arenas stepped in lock-step)."""

    def __init__(self, num_arenas, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True,
                 num_pellets=1000, num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode=0, dt=1.0 / 30,
                 device=0, cap_cells=0, cap_viruses=0, cap_foods=0, screen_respawn=False, example_bots=0, lib=None, _adopt=None):
        self.L = lib if lib is not None else hip_lib()
        self.cfg = Config(num_agents, ticks_per_step, arena_size, int(bool(pellet_regen)), num_pellets, num_viruses,
                          num_bots, int(reward_type), c_death, mode, dt, cap_cells, cap_viruses, cap_foods, int(bool(screen_respawn)), int(example_bots))
        self.h = C.c_void_p()
        self.num_arenas = num_arenas
        self.num_agents = num_agents
        self.dt = dt
        self._owner = _adopt          # a PipelinedEngine: the handle is one of its sub-batches and dies with it
        if _adopt is not None:
            self.h = C.c_void_p(_adopt[1])
        else:
            self._chk(self.L.agarcl_create(C.byref(self.cfg), num_arenas, device, C.byref(self.h)))
        self.players = self.L.agarcl_players_per_arena(self.h)
        self._blob = np.zeros(1 << 16, dtype=np.uint32)

    def _chk(self, rc):
        if rc != 0:
            raise AgarclError(rc, (self.L.agarcl_last_error() or b"").decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            if getattr(self, "_owner", None) is None:
                self.L.agarcl_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- control ------------------------------------------------------------------------------
    def set_stream(self, stream_ptr):
        self._chk(self.L.agarcl_set_stream(self.h, C.c_void_p(int(stream_ptr))))

    def sync(self):
        self._chk(self.L.agarcl_sync(self.h))

    def stream(self):
        """the hipStream_t (int) the engine launches on"""
        return int(self.L.agarcl_get_stream(self.h) or 0)

    def stream_wait(self, producer_stream):
        """the engine's stream waits, on the device, for everything enqueued so far on `producer_stream` (a hipStream_t as int; 0 = the legacy default stream)"""
        self._chk(self.L.agarcl_stream_wait(self.h, C.c_void_p(int(producer_stream))))

    def stream_signal(self, consumer_stream):
        """`consumer_stream` waits, on the device, for everything the engine has enqueued so far"""
        self._chk(self.L.agarcl_stream_signal(self.h, C.c_void_p(int(consumer_stream))))

    def seed(self, seeds=None, base_seed=0):
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
            assert seeds.shape == (self.num_arenas,)
        self._chk(self.L.agarcl_seed(self.h, _ptr(seeds), base_seed))

    def reset(self, mask=None, reset_ids=False):
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint8)
            assert mask.shape == (self.num_arenas,)
        self._chk(self.L.agarcl_reset(self.h, _ptr(mask), int(reset_ids)))

    def reset_device(self, mask_ptr, reset_ids=False):
        """mask_ptr: raw HBM pointer (int) to u8[num_arenas]; stream-ordered, no copy, no synchronisation"""
        self._chk(self.L.agarcl_reset_device(self.h, C.c_void_p(int(mask_ptr)), int(reset_ids)))

    def set_actions(self, dxdy, act):
        """host arrays: dxdy [A, n_agents, 2] f32, act [A, n_agents] i32"""
        dxdy = np.ascontiguousarray(dxdy, dtype=np.float32).reshape(self.num_arenas, self.num_agents, 2)
        act = np.ascontiguousarray(act, dtype=np.int32).reshape(self.num_arenas, self.num_agents)
        self._chk(self.L.agarcl_set_actions(self.h, _ptr(dxdy), _ptr(act), 0))

    def set_actions_device(self, dxdy_ptr, act_ptr):
        """raw HBM pointers (ints); buffers must outlive the next step()"""
        self._chk(self.L.agarcl_set_actions(self.h, C.c_void_p(dxdy_ptr), C.c_void_p(act_ptr), 1))

    def step(self, ticks=0):
        self._chk(self.L.agarcl_step(self.h, ticks))

    def timer_mark(self, which):
        """HIP event on the env's stream: 0 = start, 1 = stop"""
        self._chk(self.L.agarcl_timer_mark(self.h, which))

    def timer_elapsed_ms(self):
        ms = C.c_float(0.0)
        self._chk(self.L.agarcl_timer_elapsed_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def step_actions(self, dxdy_ptr, act_ptr, ticks=0):
        """set_actions_device + step in one call (raw HBM pointers as ints)"""
        rc = self.L.agarcl_step_actions(self.h, dxdy_ptr, act_ptr, ticks)
        if rc != 0:
            self._chk(rc)

    def tick(self, ticks=1):
        self._chk(self.L.agarcl_tick(self.h, ticks))

    def set_targets(self, txy, act):
        txy = np.ascontiguousarray(txy, dtype=np.float32).reshape(self.num_arenas, self.players, 2)
        act = np.ascontiguousarray(act, dtype=np.int32).reshape(self.num_arenas, self.players)
        self._chk(self.L.agarcl_set_targets(self.h, _ptr(txy), _ptr(act)))

    def respawn_dead(self):
        self._chk(self.L.agarcl_respawn_dead(self.h))

    # -- results ------------------------------------------------------------------------------
    def rewards(self):
        out = np.zeros((self.num_arenas, self.num_agents), dtype=np.float64)
        self._chk(self.L.agarcl_get_rewards(self.h, _ptr(out)))
        return out

    def dones(self):
        out = np.zeros((self.num_arenas, self.num_agents), dtype=np.uint8)
        self._chk(self.L.agarcl_get_dones(self.h, _ptr(out)))
        return out.astype(bool)

    def masses(self):
        out = np.zeros((self.num_arenas, self.num_agents), dtype=np.int32)
        self._chk(self.L.agarcl_get_masses(self.h, _ptr(out)))
        return out

    def flags(self):
        out = np.zeros(self.num_arenas, dtype=np.uint32)
        self._chk(self.L.agarcl_get_flags(self.h, _ptr(out)))
        return out

    def poll_flags(self):
        """OR of the capacity flags the engine's asynchronous watch has seen so far (never blocks; may lag ~64 steps)"""
        v = C.c_uint32(0)
        self._chk(self.L.agarcl_poll_flags(self.h, C.byref(v)))
        return int(v.value)

    def sqrt_check(self):
        """self-test: (patterns, lowest pattern) for which the relaxation's short square root differs from sqrtf, over all 2^32 floats"""
        out = np.zeros(2, dtype=np.uint64)
        self._chk(self.L.agarcl_debug_sqrt_check(self.h, _ptr(out)))
        return int(out[0]), int(out[1])

    def work(self, reset=False):
        """diagnostics: [arena-steps finished by the front part, arena-steps through the general engine, pellet-array
        transfers] since the counters were last reset"""
        out = np.zeros(4, dtype=np.int64)
        self._chk(self.L.agarcl_debug_work(self.h, _ptr(out), int(reset)))
        return out

    def counts(self):
        out = np.zeros((self.num_arenas, 4), dtype=np.int32)
        self._chk(self.L.agarcl_get_counts(self.h, _ptr(out)))
        return out

    def events(self, cap=256, cap_v=16):
        n = np.zeros((self.num_arenas, 2), dtype=np.int32)
        pe = np.full((self.num_arenas, cap), -1, dtype=np.int32)
        ve = np.full((self.num_arenas, cap_v), -1, dtype=np.int32)
        self._chk(self.L.agarcl_get_events(self.h, _ptr(n), _ptr(pe), cap, _ptr(ve), cap_v))
        return n, pe, ve

    def grid_obs(self, grid_size=128, cells=True, others=True, viruses=True, pellets=True, out_ptr=None, persistent=False):
        """int32 [A, n_agents, C, G, G] (host copy), or written to the HBM pointer `out_ptr` (returns C).  persistent=True: that
        HBM buffer is rewritten every step and nobody else modifies it -- only what the previous call scattered into it is cleared."""
        ch = C.c_int32(0)
        if out_ptr is not None:
            self._chk(self.L.agarcl_grid_obs(self.h, grid_size, int(cells), int(others), int(viruses), int(pellets), C.c_void_p(out_ptr), 2 if persistent else 1, C.byref(ch)))
            return ch.value
        self._chk(self.L.agarcl_grid_obs(self.h, grid_size, int(cells), int(others), int(viruses), int(pellets), None, 0, C.byref(ch)))
        out = np.zeros((self.num_arenas, self.num_agents, ch.value, grid_size, grid_size), dtype=np.int32)
        self._chk(self.L.agarcl_grid_obs(self.h, grid_size, int(cells), int(others), int(viruses), int(pellets), _ptr(out), 0, C.byref(ch)))
        return out

    def screen_obs(self, width=84, height=84, out_ptr=None, agent_view=False):
        """uint8 [A, n_agents, height, width, 3 or 4] (rows bottom-up, like glReadPixels), host copy or written to HBM `out_ptr`."""
        if out_ptr is not None:
            self._chk(self.L.agarcl_screen_obs(self.h, width, height, int(bool(agent_view)), C.c_void_p(out_ptr), 1))
            return None
        out = np.zeros((self.num_arenas, self.num_agents, height, width, 4 if agent_view else 3), dtype=np.uint8)
        self._chk(self.L.agarcl_screen_obs(self.h, width, height, int(bool(agent_view)), _ptr(out), 0))
        return out

    def ram_obs(self, k_cells=16, k_pellets=16, k_viruses=8, k_others=16, out_ptr=None):
        """float32 [A, n_agents, D] (host copy), or written to the HBM pointer `out_ptr` (returns D): include/agarcl_batch.h agarcl_ram_obs"""
        d = C.c_int32(0)
        if out_ptr is not None:
            self._chk(self.L.agarcl_ram_obs(self.h, k_cells, k_pellets, k_viruses, k_others, C.c_void_p(out_ptr), 1, C.byref(d)))
            return d.value
        self._chk(self.L.agarcl_ram_obs(self.h, k_cells, k_pellets, k_viruses, k_others, None, 0, C.byref(d)))
        out = np.zeros((self.num_arenas, self.num_agents, d.value), dtype=np.float32)
        self._chk(self.L.agarcl_ram_obs(self.h, k_cells, k_pellets, k_viruses, k_others, _ptr(out), 0, C.byref(d)))
        return out

    def gobigger_obs(self, grid_size=128, cap_food=256, cap_virus=64, cap_spore=64, cap_clone=32, out_ptrs=None):
        """GoBigger observation as padded tensors (include/agarcl_batch.h agarcl_gobigger_obs): dict of host arrays hdr i32
        [A, P, 8], food / virus / spore f32 [A, P, cap, 4], clone f32 [A, P, cap_clone, 7]; or, with out_ptrs = (hdr, food,
        virus, spore, clone) raw HBM pointers, written there."""
        if out_ptrs is not None:
            self._chk(self.L.agarcl_gobigger_obs(self.h, grid_size, cap_food, cap_virus, cap_spore, cap_clone, *[C.c_void_p(int(p)) for p in out_ptrs], 1))
            return None
        A, P = self.num_arenas, self.players
        out = {"hdr": np.zeros((A, P, 8), np.int32), "food": np.zeros((A, P, cap_food, 4), np.float32), "virus": np.zeros((A, P, cap_virus, 4), np.float32),
               "spore": np.zeros((A, P, cap_spore, 4), np.float32), "clone": np.zeros((A, P, cap_clone, 7), np.float32)}
        self._chk(self.L.agarcl_gobigger_obs(self.h, grid_size, cap_food, cap_virus, cap_spore, cap_clone, _ptr(out["hdr"]), _ptr(out["food"]),
                                             _ptr(out["virus"]), _ptr(out["spore"]), _ptr(out["clone"]), 0))
        return out

    def device_ptrs(self):
        return {"rewards": self.L.agarcl_rewards_dev(self.h), "dones": self.L.agarcl_dones_dev(self.h),
                "masses": self.L.agarcl_masses_dev(self.h),
                "packed": self.L.agarcl_packed_dev(self.h, 0)}   # base of the ring of PACKED_SLOTS buffers

    def last_slot(self):
        return int(self.L.agarcl_last_slot(self.h))

    # -- snapshots (agarcl_amd/snapshot.py builds / parses the reference's JSON) -----------------
    def adopt(self, arena, blob, kinds, hm_buckets, hm_next_resize):
        blob = np.ascontiguousarray(blob, dtype=np.uint32); kinds = np.ascontiguousarray(kinds, dtype=np.int32)
        self._chk(self.L.agarcl_adopt_arena(self.h, arena, _ptr(blob), len(blob), _ptr(kinds), hm_buckets, hm_next_resize))

    def seed_arena(self, arena, seed):
        self._chk(self.L.agarcl_seed_arena(self.h, arena, int(seed) & 0xFFFFFFFF))

    def seeds(self):
        out = np.zeros(self.num_arenas, dtype=np.uint32)
        self._chk(self.L.agarcl_get_seeds(self.h, _ptr(out)))
        return out

    def arena_words(self, arena):
        """(ar i32[ARENA_WORDS], pl i32[players][agarcl_player_words()]) raw words of one arena (agar_types.h AR_* / PL_*)."""
        P = int(self.L.agarcl_players_per_arena(self.h))
        ar = np.zeros(ARENA_WORDS, dtype=np.int32); pl = np.zeros((max(P, 1), int(self.L.agarcl_player_words())), dtype=np.int32)[:P]
        self._chk(self.L.agarcl_get_arena_words(self.h, arena, _ptr(ar), _ptr(pl)))
        return ar, pl

    def state_bytes(self):
        return int(self.L.agarcl_state_bytes(self.h))

    def device_bytes(self):
        """HBM the engine has allocated (incl. the event spill of dense / crowded configurations)"""
        return int(self.L.agarcl_device_bytes(self.h))

    # -- state exchange -------------------------------------------------------------------------
    def dump(self, arena=0):
        n = self.L.agarcl_dump_arena(self.h, arena, _ptr(self._blob), len(self._blob))
        if n < -1000:
            self._blob = np.zeros(-n, dtype=np.uint32)
            n = self.L.agarcl_dump_arena(self.h, arena, _ptr(self._blob), len(self._blob))
        if n < 0:
            self._chk(n)
        return self._blob[:n].copy()

    def load(self, blob, arena=0):
        blob = np.ascontiguousarray(blob, dtype=np.uint32)
        self._chk(self.L.agarcl_load_arena(self.h, arena, _ptr(blob), len(blob)))


class PipelinedEngine:
    """k sub-batches of one job (include/agarcl_batch.h agarcl_pipe_*): contiguous arena ranges, each a BatchedEngine on a HIP stream of its
    own that was verified to execute concurrently with the others.  parts[j] is used like any BatchedEngine; ranges[j] = (first arena, count)."""

    def __init__(self, num_arenas, sub_batches=2, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000, num_viruses=0,
                 num_bots=0, reward_type=1, c_death=0, mode=0, dt=1.0 / 30, device=0, cap_cells=0, cap_viruses=0, cap_foods=0, screen_respawn=False,
                 example_bots=0, lib=None):
        self.L = lib if lib is not None else hip_lib()
        self.cfg = Config(num_agents, ticks_per_step, arena_size, int(bool(pellet_regen)), num_pellets, num_viruses,
                          num_bots, int(reward_type), c_death, mode, dt, cap_cells, cap_viruses, cap_foods, int(bool(screen_respawn)), int(example_bots))
        self.p = C.c_void_p()
        self.num_arenas, self.num_agents, self.sub_batches = num_arenas, num_agents, sub_batches
        rc = self.L.agarcl_pipe_create(C.byref(self.cfg), num_arenas, sub_batches, device, C.byref(self.p))
        if rc != 0:
            raise AgarclError(rc, (self.L.agarcl_last_error() or b"").decode())
        kw = dict(num_agents=num_agents, ticks_per_step=ticks_per_step, arena_size=arena_size, pellet_regen=pellet_regen, num_pellets=num_pellets,
                  num_viruses=num_viruses, num_bots=num_bots, reward_type=reward_type, c_death=c_death, mode=mode, dt=dt, device=device,
                  cap_cells=cap_cells, cap_viruses=cap_viruses, cap_foods=cap_foods, screen_respawn=screen_respawn, example_bots=example_bots, lib=self.L)
        self.parts, self.ranges = [], []
        for j in range(sub_batches):
            lo, n = C.c_int32(0), C.c_int32(0)
            self.L.agarcl_pipe_range(self.p, j, C.byref(lo), C.byref(n))
            self.ranges.append((int(lo.value), int(n.value)))
            self.parts.append(BatchedEngine(int(n.value), _adopt=(self, int(self.L.agarcl_pipe_env(self.p, j))), **kw))
        self.concurrent = int(self.L.agarcl_pipe_concurrent(self.p))

    def _chk(self, rc):
        if rc != 0:
            raise AgarclError(rc, (self.L.agarcl_last_error() or b"").decode())

    def seed(self, seeds=None, base_seed=0):
        """seeds by GLOBAL arena index: uint32 [num_arenas], or arena i gets base_seed + i"""
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
            assert seeds.shape == (self.num_arenas,)
        self._chk(self.L.agarcl_pipe_seed(self.p, _ptr(seeds), base_seed))

    def reset(self, reset_ids=False):
        for e in self.parts:
            e.reset(None, reset_ids)

    def sync(self):
        self._chk(self.L.agarcl_pipe_sync(self.p))

    def fork(self, producer_stream):
        """every sub-batch's stream waits, on the device, for everything enqueued so far on `producer_stream` (hipStream_t as int): one call"""
        self._chk(self.L.agarcl_pipe_fork(self.p, C.c_void_p(int(producer_stream))))

    def join(self, consumer_stream):
        """`consumer_stream` waits, on the device, for everything every sub-batch has enqueued so far: one call"""
        self._chk(self.L.agarcl_pipe_join(self.p, C.c_void_p(int(consumer_stream))))

    def spin_timeouts(self):
        """diagnostics: 1 if a spin of the flag-word fork / join ever ran into its time bound (synchronises the pipe)"""
        return int(self.L.agarcl_pipe_spin_timeouts(self.p))

    def close(self):
        if getattr(self, "p", None) and self.p.value:
            for e in self.parts:
                e.close()
            self.L.agarcl_pipe_destroy(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
