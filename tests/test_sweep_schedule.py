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


def test_branch_free_pair_bookkeeping_up_to_16_cells():
    """agar_core.inl self_collisions, the straight-line forms used up to 16 cells: (1) the wave-uniform row walk gives lane k (and k + 64)
    the pair the row-major decode gives; (2) the packed level descriptor (first a | pair count << 8, one v_readlane per look-up) describes
    exactly the pairs (a, L - a) of local level L; (3) the quad -> pair mapping by selects (older sweep's pairs first) enumerates the
    level-time's pairs; (4) a lane's level bit 1 << (a + b) stays below 32."""
    for n in range(2, 17):
        NP = n * (n - 1) // 2; LL = 2 * n - 3
        pairs = [(a, b) for a in range(n) for b in range(a + 1, n)]            # row-major: the reference's loop order
        for k in range(min(NP, 128)):
            a = st_k = 0; st = 0; row = n - 1
            for r in range(n - 1):                                             # the uniform walk: lane k keeps the last row whose start it has reached
                if k >= st: a, st_k = r, st
                st += row; row -= 1
            b = a + 1 + (k - st_k)
            assert (a, b) == pairs[k], (n, k)
            assert a + b < 32
        for L in range(1, LL + 1):
            a0 = L - (n - 1) if L > n - 1 else 0
            desc = a0 | ((((L - 1) >> 1) - a0 + 1) << 8)
            a0_, w = desc & 255, desc >> 8
            assert {(a, L - a) for a in range(a0_, a0_ + w)} == {(a, b) for (a, b) in pairs if a + b == L}, (n, L)
        D = min(n, LL)
        for LN in range(1, LL + 1):                                            # a level-time of a phase: newer sweep's LN, older sweep's LN + D
            LO = LN + D; validO = LO <= LL
            d = lambda L: ((L - (n - 1) if L > n - 1 else 0), ((L - 1) >> 1) - (L - (n - 1) if L > n - 1 else 0) + 1)
            (a0O, wO), (a0N, wN) = (d(LO) if validO else (0, 0)), d(LN)
            got = []
            for j in range(16):
                has, older = j < wO + wN, j < wO
                offs, Lsel = (a0O if older else a0N - wO), (LO if older else LN)
                a = j + offs if has else 0; b = Lsel - a if has else 0
                if has: got.append((older, a, b))
            want = [(True, a, LO - a) for a in range(a0O, a0O + wO)] + [(False, a, LN - a) for a in range(a0N, a0N + wN)]
            assert got == want and wO + wN <= 16, (n, LN)
