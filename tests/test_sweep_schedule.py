"""The overlapped schedule of the self-collision sweeps (agarcl_amd/csrc/agar_core.inl self_collisions): visit (s, a, b) of
sweep s and pair a < b runs at level s * D + a + b with D = min(n, 2n - 3).  It must keep the reference's sequential order
(Engine.hpp:763-794: sweeps in order, pairs in lexicographic order) between any two visits that share a cell."""


def test_overlapped_sweep_schedule_keeps_every_dependency():
    for n in range(2, 33):
        D = min(n, 2 * n - 3)
        visits = [(s, a, b, s * D + a + b) for s in range(6) for a in range(n) for b in range(a + 1, n)]   # in sequential order
        last, width = {}, {}
        for s, a, b, t in visits:
            for cell in (a, b):
                assert last.get(cell, -1) < t, (n, s, a, b)      # strictly after the previous visit of either cell
                last[cell] = t
            width[t] = width.get(t, 0) + 1
        assert max(t for _, _, _, t in visits) == 5 * D + 2 * n - 3
        assert max(width.values()) <= 16                          # at most n / 2 pairs share a level


def test_level_enumeration_matches_the_schedule():
    """the kernel's incremental (sN, LN) / (sO, LO) bookkeeping enumerates exactly the visits of each level"""
    for n in range(2, 33):
        LL = 2 * n - 3; D = min(n, LL); total = 5 * D + LL
        want = {}
        for s in range(6):
            for a in range(n):
                for b in range(a + 1, n):
                    want.setdefault(s * D + a + b, set()).add((s, a, b))
        sN, LN = 0, 0
        for lev in range(1, total + 1):
            LN += 1
            if sN < 5 and LN > D:
                sN += 1; LN -= D
            sO, LO = sN - 1, LN + D
            got = set()
            if sO >= 0 and LO <= LL:
                a0 = max(LO - (n - 1), 0)
                got |= {(sO, a, LO - a) for a in range(a0, (LO - 1) // 2 + 1)}
            if LN <= LL:
                a0 = max(LN - (n - 1), 0)
                got |= {(sN, a, LN - a) for a in range(a0, (LN - 1) // 2 + 1)}
            assert got == want.get(lev, set()), (n, lev)
