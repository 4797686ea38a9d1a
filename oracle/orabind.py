"""ctypes binding of oracle/_build/libagar_oracle.so (the plain-C restatement, oracle/agar_oracle.c).  TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product
package (agarcl_amd/) never does.  Built by oracle/Makefile (plain gcc; also on the GPU box)."""
import ctypes as C
import math
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORA_SO = os.path.join(_HERE, "_build", "libagar_oracle.so")
BLOB_CAP = 1 << 20


def available():
    return os.path.exists(ORA_SO)


def build():
    """(re)build the oracle .so with make (gcc only)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-f", os.path.join(_HERE, "Makefile"), "oracle"], cwd=_HERE)


def recomb_ticks_for(dt):
    return int(math.ceil(10.0 / dt - 1e-9))


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORA_SO)
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.c_int] * 11
        L.ora_create_ex.restype = C.c_void_p
        L.ora_create_ex.argtypes = [C.c_int] * 12
        L.ora_destroy.argtypes = [C.c_void_p]
        L.ora_seed.argtypes = [C.c_void_p, C.c_uint]
        L.ora_reset.argtypes = [C.c_void_p, C.c_int]
        L.ora_take_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ora_step.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_dones.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_pids.argtypes = [C.c_void_p, C.c_void_p]
        L.ora_tick.argtypes = [C.c_void_p, C.c_double]
        L.ora_set_player.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.ora_take_action.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]
        L.ora_respawn_dead.argtypes = [C.c_void_p]
        L.ora_ticks.restype = C.c_longlong
        L.ora_ticks.argtypes = [C.c_void_p]
        L.ora_player_masses.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ora_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ora_run_random.restype = C.c_longlong
        L.ora_run_random.argtypes = [C.c_void_p, C.c_longlong, C.c_double, C.c_uint, C.c_int]
        L.ora_last_events.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.ora_grid_obs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ora_mt_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.ora_mt_next.restype = C.c_uint64
        L.ora_mt_next.argtypes = [C.c_void_p]
        L.ora_uniform_float.restype = C.c_float
        L.ora_uniform_float.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.ora_rand_seed.argtypes = [C.c_void_p, C.c_uint]
        L.ora_rand_next.argtypes = [C.c_void_p]
        L.ora_hash_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ora_std_sort_by_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


class OraEnv:
    """One arena of the C restatement."""

    def __init__(self, num_agents=1, ticks_per_step=4, arena_size=1000, pellet_regen=True, num_pellets=1000,
                 num_viruses=0, num_bots=0, reward_type=1, c_death=0, mode=0, dt=1.0 / 30, example_bots=0):
        self.L = lib()
        self.num_agents = num_agents
        self.dt = dt
        self.h = self.L.ora_create_ex(num_agents, ticks_per_step, arena_size, int(pellet_regen), num_pellets,
                                       num_viruses, num_bots, int(reward_type), c_death, mode, recomb_ticks_for(dt), int(example_bots))
        if not self.h:
            raise RuntimeError("oracle env construction failed")
        self._buf = np.zeros(BLOB_CAP, dtype=np.uint32)

    def close(self):
        if self.h:
            self.L.ora_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_screen_hook(self, on=True):
        self.L.ora_set_screen_hook.argtypes = [C.c_void_p, C.c_int]
        self.L.ora_set_screen_hook(self.h, 1 if on else 0)

    def seed(self, s):
        self.L.ora_seed(self.h, s)

    def reset(self, reset_ids=True):
        self.L.ora_reset(self.h, int(reset_ids))

    def take_actions(self, dxdy, act):
        dxdy = np.ascontiguousarray(dxdy, dtype=np.float32).reshape(-1, 2)
        act = np.ascontiguousarray(act, dtype=np.int32).reshape(-1)
        r = self.L.ora_take_actions(self.h, dxdy.ctypes.data, act.ctypes.data, len(act))
        if r != 0:
            raise RuntimeError("take_actions failed")

    def step(self):
        out = np.zeros(max(self.num_agents, 1), dtype=np.float64)
        n = self.L.ora_step(self.h, out.ctypes.data)
        return out[:n].copy()

    def dones(self):
        out = np.zeros(max(self.num_agents, 1), dtype=np.uint8)
        self.L.ora_dones(self.h, out.ctypes.data)
        return out[:self.num_agents].astype(bool)

    def pids(self):
        out = np.zeros(64, dtype=np.int32)
        n = self.L.ora_pids(self.h, out.ctypes.data)
        return out[:n].tolist()

    def tick(self, dt=None):
        self.L.ora_tick(self.h, self.dt if dt is None else dt)

    def set_player(self, pid, tx, ty, action):
        if self.L.ora_set_player(self.h, pid, tx, ty, action) != 0:
            raise RuntimeError("unknown pid")

    def take_action(self, pid, dx, dy, action):
        if self.L.ora_take_action(self.h, pid, dx, dy, action) != 0:
            raise RuntimeError("unknown pid")

    def respawn_dead(self):
        self.L.ora_respawn_dead(self.h)

    def ticks(self):
        return int(self.L.ora_ticks(self.h))

    def dump(self):
        n = self.L.ora_dump(self.h, self._buf.ctypes.data, len(self._buf))
        if n < 0:
            self._buf = np.zeros(-n + 1024, dtype=np.uint32)
            n = self.L.ora_dump(self.h, self._buf.ctypes.data, len(self._buf))
        return self._buf[:n].copy()

    def load(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint32)
        r = self.L.ora_load(self.h, blob.ctypes.data, len(blob))
        if r != 0:
            raise RuntimeError("ora_load failed: %d" % r)

    def run_random(self, ticks, policy_seed=1, allow_actions=True):
        return int(self.L.ora_run_random(self.h, ticks, self.dt, policy_seed, int(allow_actions)))

    def last_events(self):
        pe = np.zeros(4096, dtype=np.int32); ve = np.zeros(256, dtype=np.int32); nv = C.c_int(0)
        n = self.L.ora_last_events(self.h, pe.ctypes.data, len(pe), ve.ctypes.data, len(ve), C.byref(nv))
        return pe[:n].copy(), ve[:nv.value].copy()

    def grid_obs(self, agent_index=0, grid_size=128, cells=True, others=True, viruses=True, pellets=True):
        c = 1 + int(cells) + 2 * int(others) + 2 * int(viruses) + 2 * int(pellets)
        out = np.zeros((c, grid_size, grid_size), dtype=np.int32)
        self.L.ora_grid_obs(self.h, agent_index, grid_size, int(cells), int(others), int(viruses), int(pellets), out.ctypes.data)
        return out
