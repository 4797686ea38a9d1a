#!/usr/bin/env python
"""Generates tests/golden/*.npz from the REAL reference engine (oracle/_ref/libagar_ref.so, built
from /root/reference by oracle/Makefile).  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture is pure data: environment arguments, the initial state blob (oracle/BLOB_FORMAT.md),
the per-step actions, and the reference's outputs (state blobs at checkpoints, a CRC of the state
after every step, rewards and dones).  Toolchain that produced them: g++ 11.4.0, glibc 2.35.
"""
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import blob, refbind  # noqa: E402
from lockstep import policy  # noqa: E402


def crc(b):
    # NaN payloads are canonicalised so that any-NaN == any-NaN
    f = b.view(np.float32)
    c = b.copy()
    m = blob._is_float_word_mask(b) & np.isnan(f)
    c[m] = 0x7FC00000
    return zlib.crc32(c.tobytes()) & 0xFFFFFFFF


ONLY = [a for a in sys.argv[1:] if not a.startswith("-")]   # name prefixes: regenerate only those fixtures (default: all)


def record(name, cfg, steps, seed=None, init=None, actions=None, policy_seed=1, sticky=4, allow_actions=True,
           checkpoints=None, note=""):
    """Env-level trace: take_actions + step (cfg['ticks_per_step'] ticks per step)."""
    if ONLY and not any(name.startswith(p) for p in ONLY):
        return
    env = refbind.RefEnv(**cfg)
    if seed is not None:
        env.seed(seed)
    env.reset(True)
    if init is not None:
        b0 = init(blob.parse(env.dump()))
        env.load(blob.build(b0))
    blob0 = env.dump()
    n = cfg.get("num_agents", 1)
    acts = np.zeros((steps, n, 3), dtype=np.float32)
    rewards = np.zeros((steps, n), dtype=np.float64)
    dones = np.zeros((steps, n), dtype=np.uint8)
    crcs = np.zeros(steps, dtype=np.uint32)
    checkpoints = sorted(set(checkpoints or [steps - 1]) | {steps - 1})
    blobs = {}
    for t in range(steps):
        if actions is not None:
            a = np.asarray(actions(t), dtype=np.float32).reshape(n, 3)
            dxdy, act = a[:, :2].copy(), a[:, 2].astype(np.int32)
        else:
            dxdy, act = policy(policy_seed, t, n, allow_actions, sticky)
        acts[t, :, :2] = dxdy
        acts[t, :, 2] = act
        env.take_actions(dxdy, act)
        rewards[t] = env.step()
        dones[t] = env.dones()
        b = env.dump()
        crcs[t] = crc(b)
        if t in checkpoints:
            blobs["blob_%d" % t] = b
    out = dict(cfg=json.dumps(cfg), seed=-1 if seed is None else seed, blob0=blob0, actions=acts, rewards=rewards,
               dones=dones, crcs=crcs, checkpoints=np.asarray(checkpoints, dtype=np.int32), note=note, **blobs)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    d = blob.parse(env.dump())
    print("%-28s steps=%4d pellets=%4d viruses=%3d foods=%3d cells=%s" % (
        name, steps, len(d["pellet_x"]), len(d["virus_x"]), len(d["food_x"]), [p["n_cells"] for p in d["players"]]))


# ---- helpers to hand-build states ---------------------------------------------------------------
def set_cells(pl, cells, recomb=0):
    """cells: list of (x, y, mass); ids continue from the player's first cell id."""
    base = int(pl["cell_id"][0]) if len(pl["cell_id"]) else 5000
    n = len(cells)
    pl["cell_f"] = np.array([[c[0], c[1], 0, 0, 0, 0] for c in cells], dtype=np.float32).reshape(n, 6)
    pl["cell_mass"] = np.array([c[2] for c in cells], dtype=np.int64)
    pl["cell_id"] = np.array([base + 10 * i for i in range(n)], dtype=np.int64)
    pl["cell_recomb"] = np.array([recomb if np.isscalar(recomb) else recomb[i] for i in range(n)], dtype=np.int64)
    pl["n_cells"] = n
    pl["highest_mass"] = max(int(pl["highest_mass"]), int(sum(c[2] for c in cells)))


def set_pellets(d, pts):
    base = int(d["pellet_id"][0]) if len(d["pellet_id"]) else 2
    d["pellet_x"] = np.array([p[0] for p in pts], dtype=np.float32)
    d["pellet_y"] = np.array([p[1] for p in pts], dtype=np.float32)
    d["pellet_id"] = np.arange(base, base + len(pts), dtype=np.int64)


def set_viruses(d, vs):
    """vs: list of (x, y, mass, hits)"""
    base = 900000
    d["virus_x"] = np.array([v[0] for v in vs], dtype=np.float32)
    d["virus_y"] = np.array([v[1] for v in vs], dtype=np.float32)
    d["virus_vx"] = np.zeros(len(vs), dtype=np.float32)
    d["virus_vy"] = np.zeros(len(vs), dtype=np.float32)
    d["virus_mass"] = np.array([v[2] for v in vs], dtype=np.int64)
    d["virus_hits"] = np.array([v[3] for v in vs], dtype=np.int64)
    d["virus_id"] = np.arange(base, base + len(vs), dtype=np.int64)


def main():
    if not refbind.available():
        raise SystemExit("oracle/_ref/libagar_ref.so missing: run `make -C oracle ref` in the build container")
    T1 = dict(ticks_per_step=1)
    C2 = dict(num_agents=1, ticks_per_step=4, arena_size=1000, num_pellets=1000, num_viruses=0, mode=0)
    C3 = dict(num_agents=1, ticks_per_step=4, arena_size=1000, num_pellets=1000, num_viruses=25, mode=0)
    C3m6 = dict(C3, mode=6)
    # ---- F1 / F10: seeded reset + random rollouts ------------------------------------------------
    record("roll_c2_s10000", C2, 300, seed=10000, allow_actions=False, checkpoints=[0, 99, 299])
    record("roll_c2_s10001", C2, 300, seed=10001, allow_actions=False, checkpoints=[0, 299])
    record("roll_c3_mode0_s42", C3, 300, seed=42, checkpoints=[0, 299])
    record("roll_c3_mode6_s42", C3m6, 400, seed=42, sticky=16, checkpoints=[0, 99, 199, 399])
    record("roll_c3_mode6_s7", C3m6, 400, seed=7, sticky=8, checkpoints=[0, 399])
    record("roll_small_mode6_s1", dict(num_agents=1, ticks_per_step=4, arena_size=250, num_pellets=500, num_viruses=10, mode=6),
           400, seed=1, sticky=8, checkpoints=[0, 9, 399])
    record("roll_mode1_s3", dict(num_agents=1, ticks_per_step=4, arena_size=300, num_pellets=300, num_viruses=5, mode=1), 200, seed=3, sticky=8)
    record("roll_mode5_s3", dict(num_agents=1, ticks_per_step=4, arena_size=300, num_pellets=300, num_viruses=5, mode=5), 300, seed=3, sticky=8)
    record("roll_mode3_s9", dict(num_agents=1, ticks_per_step=4, arena_size=1200, num_pellets=600, num_viruses=12, mode=3), 200, seed=9, sticky=8)
    record("roll_mode0_rewardmass_s4", dict(C3, reward_type=0), 100, seed=4)
    # CPU-only rows (bots / several agents: cell-eats-cell, unordered_map order)
    record("roll_c1_bots_s42", dict(num_agents=1, ticks_per_step=4, arena_size=250, num_pellets=500, num_viruses=10, num_bots=4, mode=0),
           500, seed=42, checkpoints=[0, 499])
    record("roll_multi3_mode6_s5", dict(num_agents=3, ticks_per_step=4, arena_size=250, num_pellets=500, num_viruses=10, mode=6),
           400, seed=5, sticky=8, checkpoints=[0, 399])
    # the paper's tasks (values of /root/reference/bench/tasks_configs/mode_{1,4,7,10}.json via tests/golden/paper_tasks.json: 350 x 350 arena,
    # 500 pellets, no viruses, one bot in modes 7-10), 300 steps each
    tasks = json.load(open(os.path.join(HERE, "paper_tasks.json")))["tasks"]
    for m, sd in ((1, 101), (4, 104), (7, 107), (10, 110)):
        t = tasks[str(m)]
        record("task_mode%d_s%d" % (m, sd), dict(num_agents=1, ticks_per_step=t["ticks_per_step"], arena_size=t["arena_size"], num_pellets=t["num_pellets"],
                                                 num_viruses=t["num_viruses"], num_bots=t["num_bots"], mode=m, reward_type=t["reward_type"], c_death=t["c_death"],
                                                 pellet_regen=bool(t["pellet_regen"])),
               300, seed=sd, sticky=8, checkpoints=[0, 149, 299], note="bench/tasks_configs/mode_%d.json" % m)
    record("roll_mode9_bot_s2", dict(num_agents=1, ticks_per_step=4, arena_size=200, num_pellets=500, num_viruses=5, num_bots=1, mode=9), 300, seed=2, sticky=8)

    # ---- bench/main.cpp's Tick population (bench/main.cpp:14-38): N ExampleBots (agario/bots/ExampleBot.hpp:45-51) on the default 250 x 250
    # engine, no Player at all; and the same bots as prey of a random-policy agent (they get eaten and respawn: mode 0)
    tick = dict(ticks_per_step=4, arena_size=250, num_pellets=500, num_viruses=10, mode=0)
    record("fb_tick30_examplebots", dict(tick, num_agents=0, example_bots=30), 300, seed=42, checkpoints=[0, 149, 299],
           note="Tick/30: 30 ExampleBots, nobody else; 1200 engine ticks")
    record("fb_tick5_examplebots", dict(tick, num_agents=0, example_bots=5), 150, seed=43, checkpoints=[0, 149], note="Tick/5")
    record("fb_agent_among_20_examplebots", dict(tick, num_agents=1, example_bots=20), 400, seed=44, sticky=8, checkpoints=[0, 199, 399],
           note="one random-policy agent among 20 ExampleBots: cell-eats-cell, respawns, the 13 -> 29 rehash of the players map")
    record("fb_mode6_agent_among_30_examplebots", dict(tick, num_agents=1, example_bots=30, mode=6, arena_size=300), 300, seed=45, sticky=8, checkpoints=[0, 299],
           note="a mass-1000 agent among 30 ExampleBots (31 players: the 29 -> 59 rehash)")

    # ---- F1: reset / spawn as standalone vectors: seeds {0, 1, 42} x the four (W, N_p, N_v) shapes -> every position and id
    for W, n_p, n_v in ((250, 500, 10), (1000, 1000, 0), (1000, 1000, 25), (350, 500, 0)):
        for sd in (0, 1, 42):
            record("f1_spawn_w%d_p%d_v%d_s%d" % (W, n_p, n_v, sd), dict(num_agents=1, ticks_per_step=4, arena_size=W, num_pellets=n_p, num_viruses=n_v, mode=0),
                   1, seed=sd, allow_actions=False, note="blob0 = the state right after seed + reset (Engine.hpp:98-148, BaseEnvironment.hpp:179-204)")
    # ---- F2: kinematics of one cell: scripted directions (incl. none, and into the walls of a small arena), 200 ticks, state CRC every
    # tick and the full state every 20th
    def f2_actions(t):
        k = t // 10
        dirs = [(1.0, 0.0), (0.0, 1.0), (-1.0, -1.0), (0.0, 0.0), (0.3, -0.7), (-1.0, 0.0), (-1.0, 0.0), (-1.0, 0.0), (0.0, -1.0), (0.0, -1.0),
                (1.0, 1.0), (1.0, 1.0), (1.0, 1.0), (1.0, 1.0), (1e-3, -1e-3), (0.0, 0.0), (-0.5, 0.9), (1.0, -1.0), (1.0, -1.0), (0.2, 0.2)]
        d = dirs[k % len(dirs)]
        return [[d[0], d[1], 0]]
    record("f2_kinematics_200", dict(num_agents=1, ticks_per_step=1, arena_size=60, num_pellets=0, num_viruses=0, mode=3, pellet_regen=False),
           200, seed=12, actions=f2_actions, checkpoints=list(range(0, 200, 20)), note="one cell, no pellets: Engine.hpp:609-630, core/types.hpp:176-223")
    # ---- F10 at the survey's length: single-arena rollouts of 4000 ticks (1000 steps), C2 x 8 seeds and C3 in modes 0 / 4 / 6
    for sd in range(8):
        record("roll4k_c2_s%d" % (20000 + sd), C2, 1000, seed=20000 + sd, allow_actions=False, checkpoints=[0, 499, 999])
    record("roll4k_c3_mode0_s11", C3, 1000, seed=11, checkpoints=[0, 499, 999])
    record("roll4k_c3_mode4_s12", dict(C3, mode=4), 1000, seed=12, checkpoints=[0, 499, 999])
    record("roll4k_c3_mode6_s13", C3m6, 1000, seed=13, sticky=8, checkpoints=[0, 499, 999])

    small = dict(num_agents=1, arena_size=200, num_pellets=40, num_viruses=0, mode=3, **T1)  # mode 3: no decay

    # ---- F3: pellet eating: double-eat (two cells over one pellet) and stale swap-pop indices ----
    def f3(d):
        pl = d["players"][0]
        set_cells(pl, [(100, 100, 400), (101, 100, 400)], recomb=300)
        pts = [(100.5, 100.0), (20, 20), (100.2, 100.3), (30, 30), (99, 99.5), (40, 40), (180, 180), (102, 101)]
        pts += [(10 + 3 * i, 150) for i in range(30)]
        pts += [(100.1, 99.9), (100.9, 100.1)]  # the two LAST pellets are eaten too (stale back index)
        set_pellets(d, pts)
        pl["target"] = np.array([100.5, 100.0], dtype=np.float32)
        return d
    record("f3_pellet_double_eat_stale", small, 6, seed=1, init=f3, actions=lambda t: [[0, 0, 0]], checkpoints=[0, 1, 5],
           note="two cells share pellets; last indices eaten; Engine.hpp:976-1009")

    # growth-dependent eat: a pellet just outside the radius is eaten only because an earlier one grew the cell
    def f3b(d):
        pl = d["players"][0]
        set_cells(pl, [(100, 100, 25)])
        r25, r26 = np.sqrt(25 / np.pi), np.sqrt(26 / np.pi)
        mid = np.float32((r25 + r26) / 2)
        set_pellets(d, [(100 + 1.0, 100), (100 + float(mid), 100), (100 - float(mid), 100), (100, 100 + float(mid))] + [(10 + 3 * i, 150) for i in range(10)])
        pl["target"] = np.array([100, 100], dtype=np.float32)
        return d
    record("f3b_pellet_growth_order", small, 3, seed=1, init=f3b, actions=lambda t: [[0, 0, 0]], checkpoints=[0, 2],
           note="radius grows during the scan (Engine.hpp:991-994)")

    # ---- F4: virus pop (mass 111 / 400) and virus eaten with 14 cells -------------------------------
    vcfg = dict(num_agents=1, arena_size=300, num_pellets=30, num_viruses=0, mode=3, **T1)

    def f4(mass, ncells):
        def init(d):
            pl = d["players"][0]
            cells = [(150, 150, mass)] + [(30 + 12 * i, 40, 30) for i in range(ncells - 1)]
            set_cells(pl, cells, recomb=300)
            set_viruses(d, [(150.5, 150.5, 100, 0), (200, 200, 100, 0), (151, 149, 100, 0)])
            pl["target"] = np.array([160, 155], dtype=np.float32)
            return d
        return init
    record("f4_virus_pop_m111", vcfg, 12, seed=2, init=f4(111, 1), actions=lambda t: [[0.3, 0.2, 0]], checkpoints=[0, 1, 11])
    record("f4_virus_pop_m400", vcfg, 40, seed=2, init=f4(400, 1), actions=lambda t: [[0.3, 0.2, 0]], checkpoints=[0, 1, 39])
    record("f4_virus_pop_m400_9cells", vcfg, 12, seed=2, init=f4(400, 9), actions=lambda t: [[0.3, 0.2, 0]], checkpoints=[0, 11])
    record("f4_virus_eat_14cells", vcfg, 12, seed=2, init=f4(400, 14), actions=lambda t: [[0.3, 0.2, 0]], checkpoints=[0, 11])

    # ---- F5: split / eject / food feeds virus until it spawns a new one (8th hit) -----------------------
    def f5(d):
        pl = d["players"][0]
        set_cells(pl, [(100, 150, 2000)], recomb=0)
        set_viruses(d, [(135, 150, 100, 0)])
        pl["target"] = np.array([200, 150], dtype=np.float32)
        return d
    record("f5_feed_virus_spawn", vcfg, 130, seed=3, init=f5, actions=lambda t: [[1.0, 0.0, 1]], checkpoints=[0, 60, 129],
           note="eject every 10 ticks toward a virus; Engine.hpp:632-687,1027-1054")
    record("f5_split_then_feed", vcfg, 90, seed=3, init=f5, actions=lambda t: [[0.7, 0.7 if t < 40 else -0.7, 2 if t % 35 == 0 else (1 if t % 7 == 0 else 0)]],
           checkpoints=[0, 1, 89])

    # ---- F6: self-collision relaxation + recombine (timers expired / not) ------------------------------
    def f6(rec):
        def init(d):
            pl = d["players"][0]
            set_cells(pl, [(100, 100, 200), (104, 101, 180), (97, 103, 60), (101, 96, 205), (140, 100, 90)], recomb=rec)
            pl["target"] = np.array([100, 100], dtype=np.float32)
            return d
        return init
    record("f6_selfcollide_norecombine", small, 40, seed=4, init=f6(300), actions=lambda t: [[0.1, -0.2, 0]], checkpoints=[0, 1, 39])
    record("f6_recombine_expired", small, 40, seed=4, init=f6(0), actions=lambda t: [[0.1, -0.2, 0]], checkpoints=[0, 1, 39])
    record("f6_recombine_mixed", small, 40, seed=4, init=f6([0, 5, 0, 300, 2]), actions=lambda t: [[0.1, -0.2, 0]], checkpoints=[0, 6, 39])

    # cells pinned in a corner (border ratios in avoid_static_overlap, Engine.hpp:723-738)
    def f6c(d):
        pl = d["players"][0]
        set_cells(pl, [(3, 3, 100), (4, 4, 100), (8, 3, 95)], recomb=300)
        pl["target"] = np.array([0, 0], dtype=np.float32)
        return d
    record("f6_corner_static_overlap", small, 25, seed=4, init=f6c, actions=lambda t: [[-1.0, -1.0, 0]], checkpoints=[0, 24])

    # ---- F7: decay + anti-team (3 virus ticks inside the window) -------------------------------------------
    dcfg = dict(num_agents=1, arena_size=300, num_pellets=30, num_viruses=0, mode=4, **T1)

    def f7(d):
        pl = d["players"][0]
        set_cells(pl, [(150, 150, 5000), (60, 60, 300), (240, 240, 26)], recomb=300)
        pl["elapsed"] = 3655
        pl["last_decay"] = 3600
        pl["virus_ticks"] = np.array([10, 100, 3000, 3600], dtype=np.int64)
        pl["target"] = np.array([150, 150], dtype=np.float32)
        return d
    record("f7_decay_anti_team", dcfg, 130, seed=5, init=f7, actions=lambda t: [[0, 0, 0]], checkpoints=[0, 4, 5, 64, 65, 129])

    # ---- F9: regen at ticks % 120 == 0 after eats ---------------------------------------------------------------
    rcfg = dict(num_agents=1, arena_size=200, num_pellets=60, num_viruses=3, mode=0, **T1)

    def f9(d):
        d["ticks"] = 118
        d["pellet_x"] = d["pellet_x"][:41]; d["pellet_y"] = d["pellet_y"][:41]; d["pellet_id"] = d["pellet_id"][:41]
        set_viruses(d, [(20, 20, 100, 0)])
        return d
    record("f9_regen_topup", rcfg, 8, seed=6, init=f9, actions=lambda t: [[0.5, 0.5, 0]], checkpoints=[0, 1, 2, 7])

    # ---- auto split at 22500 (n < 14 and n == 14) ------------------------------------------------------------------
    def fa(n):
        def init(d):
            pl = d["players"][0]
            set_cells(pl, [(100, 100, 22499)] + [(20 + 11 * i, 180, 30) for i in range(n - 1)], recomb=300)
            set_pellets(d, [(100.5, 100.5), (99.5, 100), (100, 99.5)] + [(10 + 3 * i, 30) for i in range(10)])
            pl["target"] = np.array([120, 120], dtype=np.float32)
            return d
        return init
    record("fa_autosplit_n3", small, 5, seed=7, init=fa(3), actions=lambda t: [[0.5, 0.5, 0]], checkpoints=[0, 4])
    record("fa_autosplit_n14", small, 5, seed=7, init=fa(14), actions=lambda t: [[0.5, 0.5, 0]], checkpoints=[0, 4])

    # ---- F8 (CPU-only row): cell eats cell, 2 and 3 players, strip / break quirks -----------------------------------
    def f8(d):
        big, small_, third = d["players"][0], d["players"][1], d["players"][2] if len(d["players"]) > 2 else None
        set_cells(big, [(100, 100, 900), (108, 100, 300)], recomb=300)
        set_cells(small_, [(101, 101, 100), (109, 99, 40), (150, 150, 30)], recomb=300)
        if third is not None:
            set_cells(third, [(100.5, 99.5, 120), (60, 60, 2000)], recomb=300)
        return d
    record("f8_cell_eats_cell_2p", dict(num_agents=2, arena_size=200, num_pellets=40, num_viruses=0, mode=3, **T1), 10, seed=8,
           init=f8, actions=lambda t: [[0, 0, 0], [0, 0, 0]], checkpoints=[0, 9])
    record("f8_cell_eats_cell_3p", dict(num_agents=3, arena_size=200, num_pellets=40, num_viruses=0, mode=3, **T1), 10, seed=8,
           init=f8, actions=lambda t: [[0, 0, 0], [0.2, 0.1, 0], [0, 0, 2]], checkpoints=[0, 9])


if __name__ == "__main__":
    main()
