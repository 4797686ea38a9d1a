"""The oracle's restatements of libstdc++ / glibc behaviour, checked against the real libraries."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def probe():
    out = os.path.join(ROOT, "tests", "_build", "libstl_probe.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", out, os.path.join(ROOT, "tests", "helpers", "stl_probe.cpp")])
    L = C.CDLL(out)
    L.probe_map_round.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.probe_intmap_order.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.probe_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.probe_uniform.argtypes = [C.c_uint, C.c_float, C.c_float, C.c_int, C.c_void_p]
    L.probe_rand.argtypes = [C.c_uint, C.c_int, C.c_void_p]
    return L


def test_unordered_map_order_across_resets(probe, oracle_lib):
    """players map: pids keep growing across reset() while clear() keeps the bucket array (GameState.hpp:61-67)."""
    L = oracle_lib.lib()
    for P in (1, 2, 5, 12, 13, 14, 20, 31):
        probe.probe_map_new()
        bc, nr = C.c_int(1), C.c_int(0)
        pid = 0
        for _round in range(6):
            keys = np.arange(pid, pid + P, dtype=np.int32) & 0xFFFF
            pid += P
            want = np.zeros(P, np.int32); got = np.zeros(P, np.int32)
            assert probe.probe_map_round(keys.ctypes.data, P, want.ctypes.data) == P
            assert L.ora_hash_order(keys.ctypes.data, P, got.ctypes.data, C.byref(bc), C.byref(nr)) == P
            assert np.array_equal(want, got), (P, _round, want, got)


def test_int_keyed_map_order(probe, oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.RandomState(0)
    for n in (1, 3, 11, 12, 13, 14, 30, 60, 150):
        keys = np.sort(rng.choice(400, size=n, replace=False)).astype(np.int32)
        want = np.zeros(n, np.int32); got = np.zeros(n, np.int32)
        probe.probe_intmap_order(keys.ctypes.data, n, want.ctypes.data)
        bc, nr = C.c_int(1), C.c_int(0)
        L.ora_hash_order(keys.ctypes.data, n, got.ctypes.data, C.byref(bc), C.byref(nr))
        assert np.array_equal(want, got), n


def test_std_sort_tie_order(probe, oracle_lib):
    L = oracle_lib.lib()
    rng = np.random.RandomState(1)
    for n in (0, 1, 2, 15, 16, 17, 33, 100, 1000):
        for trial in range(5):
            k = rng.randint(0, max(2, n // 3), size=n).astype(np.float32)   # many ties
            if trial == 4 and n:
                k = np.sort(k)[::-1].copy()
            p = np.arange(n, dtype=np.int32)
            k1, p1, k2, p2 = k.copy(), p.copy(), k.copy(), p.copy()
            probe.probe_sort(k1.ctypes.data, p1.ctypes.data, n)
            L.ora_std_sort_by_float(k2.ctypes.data, p2.ctypes.data, n)
            assert np.array_equal(p1, p2) and np.array_equal(k1, k2), (n, trial)


def test_mt19937_64_uniform_float(probe, oracle_lib):
    L = oracle_lib.lib()
    for seed in (0, 1, 42, 10000, 2**32 - 1):
        n = 2000
        want = np.zeros(n, np.float32)
        probe.probe_uniform(seed, 0.0, 998.8716, n, want.ctypes.data)
        mt = np.zeros(313, np.uint64)
        L.ora_mt_seed(mt.ctypes.data, seed)
        got = np.array([L.ora_uniform_float(mt.ctypes.data, 0.0, 998.8716) for _ in range(n)], np.float32)
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), seed


def test_glibc_rand(probe, oracle_lib):
    L = oracle_lib.lib()
    for seed in (0, 1, 42, 123456789):
        want = np.zeros(500, np.int32)
        probe.probe_rand(seed, 500, want.ctypes.data)
        st = np.zeros(35, np.int32)
        L.ora_rand_seed(st.ctypes.data, seed)
        got = np.array([L.ora_rand_next(st.ctypes.data) for _ in range(500)], np.int32)
        assert np.array_equal(want, got), seed
