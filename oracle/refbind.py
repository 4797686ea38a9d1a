"""ctypes binding of oracle/_ref/libagar_ref.so (the real reference engine).  TEST INFRASTRUCTURE.

Only tests/, the fixture generators under tests/golden/ and bench.py's cpu_baseline leg may import
this.  The .so is built by oracle/Makefile from /root/reference (this container only); a prebuilt
copy travels to the GPU box."""
import ctypes as C
import math
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libagar_ref.so")
BLOB_CAP = 1 << 20


def available():
    return os.path.exists(REF_SO)


def recomb_ticks_for(dt):
    return int(math.ceil(10.0 / dt - 1e-9))


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        L.ref_env_create.restype = C.c_void_p
        L.ref_env_create.argtypes = [C.c_int] * 11
        L.ref_env_create_ex.restype = C.c_void_p
        L.ref_env_create_ex.argtypes = [C.c_int] * 12
        L.ref_env_destroy.argtypes = [C.c_void_p]
        L.ref_env_seed.argtypes = [C.c_void_p, C.c_uint]
        L.ref_env_reset.argtypes = [C.c_void_p, C.c_int]
        L.ref_env_take_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ref_env_step.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_env_dones.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_env_pids.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_tick.argtypes = [C.c_void_p, C.c_double]
        L.ref_set_player.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.ref_take_action.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.ref_respawn_dead.argtypes = [C.c_void_p]
        L.ref_ticks.restype = C.c_longlong
        L.ref_ticks.argtypes = [C.c_void_p]
        L.ref_player_masses.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ref_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ref_set_global_id.argtypes = [C.c_int]
        L.ref_env_save_json.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_env_load_json.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.ref_run_random.restype = C.c_longlong
        L.ref_run_random.argtypes = [C.c_void_p, C.c_longlong, C.c_double, C.c_uint, C.c_int]
        _lib = L
    return _lib


class RefEnv:
    """One reference arena (BaseEnvironment<false> + Engine<false>)."""

    def __init__(self, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                 num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode=0, dt=1.0 / 30, example_bots=0):
        self.L = lib()
        self.num_agents = num_agents
        self.dt = dt
        self.h = self.L.ref_env_create_ex(num_agents, ticks_per_step, arena_size, int(pellet_regen), num_pellets,
                                       num_viruses, num_bots, int(reward_type), c_death, mode, recomb_ticks_for(dt), int(example_bots))
        if not self.h:
            raise RuntimeError("reference env construction failed")
        self._buf = np.zeros(BLOB_CAP, dtype=np.uint32)

    def close(self):
        if self.h:
            self.L.ref_env_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_screen_hook(self, on=True):
        """ScreenEnvironment's respawn hook around the unmodified BaseEnvironment::step (oracle/ref_harness.cpp RefEnv)"""
        self.L.ref_env_set_screen_hook.argtypes = [C.c_void_p, C.c_int]
        self.L.ref_env_set_screen_hook(self.h, 1 if on else 0)

    def seed(self, s):
        self.L.ref_env_seed(self.h, s)

    def reset(self, reset_ids=True):
        self.L.ref_env_reset(self.h, int(reset_ids))

    def take_actions(self, dxdy, act):
        dxdy = np.ascontiguousarray(dxdy, dtype=np.float32).reshape(-1, 2)
        act = np.ascontiguousarray(act, dtype=np.int32).reshape(-1)
        r = self.L.ref_env_take_actions(self.h, dxdy.ctypes.data, act.ctypes.data, len(act))
        if r != 0:
            raise RuntimeError("take_actions failed")

    def step(self):
        out = np.zeros(max(self.num_agents, 1), dtype=np.float64)
        n = self.L.ref_env_step(self.h, out.ctypes.data)
        return out[:n].copy()

    def dones(self):
        out = np.zeros(max(self.num_agents, 1), dtype=np.uint8)
        self.L.ref_env_dones(self.h, out.ctypes.data)
        return out[:self.num_agents].astype(bool)

    def pids(self):
        out = np.zeros(64, dtype=np.int32)
        n = self.L.ref_env_pids(self.h, out.ctypes.data)
        return out[:n].tolist()

    def tick(self, dt=None):
        self.L.ref_tick(self.h, self.dt if dt is None else dt)

    def set_player(self, pid, tx, ty, action):
        if self.L.ref_set_player(self.h, pid, tx, ty, action) != 0:
            raise RuntimeError("unknown pid")

    def take_action(self, pid, dx, dy, action):
        if self.L.ref_take_action(self.h, pid, dx, dy, action) != 0:
            raise RuntimeError("unknown pid")

    def respawn_dead(self):
        self.L.ref_respawn_dead(self.h)

    def ticks(self):
        return int(self.L.ref_ticks(self.h))

    def dump(self):
        n = self.L.ref_dump(self.h, self._buf.ctypes.data, len(self._buf))
        if n < 0:
            self._buf = np.zeros(-n + 1024, dtype=np.uint32)
            n = self.L.ref_dump(self.h, self._buf.ctypes.data, len(self._buf))
        return self._buf[:n].copy()

    def save_json(self, path):
        """BaseEnvironment::save_env_state (BaseEnvironment.hpp:213-310)."""
        if self.L.ref_env_save_json(self.h, str(path).encode()) != 0:
            raise RuntimeError("save_env_state failed")

    def load_json(self, path, reset_ids=True):
        """BaseEnvironment::load_env_state (BaseEnvironment.hpp:312-343); reset() is a no-op afterwards."""
        if self.L.ref_env_load_json(self.h, str(path).encode(), 1 if reset_ids else 0) != 0:
            raise RuntimeError("load_env_state failed")

    def load(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint32)
        r = self.L.ref_load(self.h, blob.ctypes.data, len(blob))
        if r != 0:
            raise RuntimeError("ref_load failed: %d" % r)

    def run_random(self, ticks, policy_seed=1, allow_actions=True):
        return int(self.L.ref_run_random(self.h, ticks, self.dt, policy_seed, int(allow_actions)))
