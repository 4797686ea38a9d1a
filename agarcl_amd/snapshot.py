"""Snapshots in the reference's JSON wire format.

Writer: BaseEnvironment::save_env_state (/root/reference/environment/envs/BaseEnvironment.hpp:213-310).
Reader: Engine::load_env_state (/root/reference/agario/engine/Engine.hpp:247-348) followed by
BaseEnvironment::load_env_state (BaseEnvironment.hpp:312-343).  A file written here loads in the reference and the
other way round; tests/test_snapshot.py pins both directions against the real reference build.

The JSON text lives on the host; state moves in and out of HBM as the word blob of the C ABI
(agarcl_dump_arena / agarcl_adopt_arena, layout in oracle/BLOB_FORMAT.md, restated here for the product side).

What the reference's loader does, and this one therefore does too:
  * the player set is rebuilt: players get fresh pids 0..P-1 in file order (state.next_pid restarts at 0,
    Engine.hpp:103-109,266-283); bots are recognised by NAME; every add_player also respawns a throw-away cell,
    which costs one entity id (Engine.hpp:81);
  * cells keep their saved ids but still advance the id counter; pellets, viruses and foods get fresh ids;
    splitting velocities, recombine timers, virus food hits and the players' pending actions are not in the file
    (fresh cells may recombine at once: Entities.hpp:125-128);
  * ticks restart at 0 and the arena is re-seeded with the file's "seed" (Engine.hpp:345-347);
  * agents are the non-bot players in the NEW map iteration order (BaseEnvironment.hpp:324-335), and reset()
    is a no-op from then on (is_loading_env_state, :180-181) -- mirrored by agarcl_amd/agarcl.py.
"""
import json

import numpy as np

MAGIC = 0x31524741
KIND_BY_NAME = {"HungryBot": 1, "HungryShyBot": 2, "AggressiveBot": 3, "AggressiveShyBot": 4}   # Engine.hpp:272-281
NAME_BY_KIND = {v: k for k, v in KIND_BY_NAME.items()}
BOT_COLOR = {1: 4, 2: 5, 3: 0, 4: 1}  # blue, purple, red, orange: the bot classes' default colours (agario/bots/*.hpp, core/color.hpp:4)
CELL_MIN_SIZE = 25


def _f(u):
    return float(np.array([u], dtype=np.uint32).view(np.float32)[0])


def _u(x):
    return int(np.array([x], dtype=np.float32).view(np.uint32)[0])


# ---- libstdc++ unordered_map<pid, ...> iteration order for a cleared map that keeps its bucket array -------------
_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97, 103, 109, 113]
_FAST = [2, 2, 2, 3, 5, 5, 7, 7, 11, 11, 11, 11, 13, 13]


def _next_bkt(n):  # _Prime_rehash_policy::_M_next_bkt (hashtable_c++0x.cc); returns (buckets, next_resize)
    b = _FAST[n] if n < 14 else next(p for p in _PRIMES if p >= n)
    return b, b


def map_order_after_inserts(keys, bucket_count, next_resize):
    """Iteration order of a libstdc++ unordered_map<int, T> (identity hash, max load factor 1) after inserting
    `keys` in order into an EMPTY table that has `bucket_count` buckets (clear() keeps the bucket array).
    Returns (order as indices into keys, bucket_count, next_resize).  _M_insert_unique_node / _M_rehash_aux."""
    bc = max(int(bucket_count), 1); nr = int(next_resize)
    nxt, head, before = [], -1, {}      # before[b] = node preceding bucket b's first node (-1 = list head)

    def rehash(nb):
        nonlocal head, before, bc
        p, before, bc, head, bbegin = head, {}, nb, -1, 0
        while p != -1:
            n2 = nxt[p]; b = keys[p] % nb
            if b not in before:
                nxt[p] = head; head = p; before[b] = -1
                if nxt[p] != -1:
                    before[bbegin] = p
                bbegin = b
            else:
                bf = before[b]
                if bf == -1:
                    nxt[p] = head; head = p
                else:
                    nxt[p] = nxt[bf]; nxt[bf] = p
            p = n2

    for node, key in enumerate(keys):
        if node + 1 > nr:  # _M_need_rehash(n_bkt, n_elt, 1)
            min_bkts = max(node + 1, 0 if nr else 11)
            if min_bkts >= bc:
                nb, nr = _next_bkt(max(min_bkts + 1, bc * 2))
                rehash(nb)
            else:
                nr = bc
        nxt.append(-1)
        b = key % bc
        if b in before:  # _M_insert_bucket_begin
            bf = before[b]
            if bf == -1:
                nxt[node] = head; head = node
            else:
                nxt[node] = nxt[bf]; nxt[bf] = node
        else:
            nxt[node] = head; head = node
            if nxt[node] != -1:
                before[keys[nxt[node]] % bc] = node
            before[b] = -1
    order, p = [], head
    while p != -1:
        order.append(p); p = nxt[p]
    return order, bc, nr


# ---- blob <-> python structures (product-side restatement of oracle/BLOB_FORMAT.md) ------------------------------
def parse_blob(words):
    w = np.asarray(words, dtype=np.uint32)
    if int(w[0]) != MAGIC:
        raise ValueError("not an AGR1 blob")
    ticks, idc, next_pid, npel, nv, nf, npl = (int(x) for x in w[1:8])
    p = 8
    fl = lambda a: a.view(np.float32)
    pel = dict(x=fl(w[p:p + npel]), y=fl(w[p + npel:p + 2 * npel]), id=w[p + 2 * npel:p + 3 * npel].astype(np.int64)); p += 3 * npel
    vir = dict(x=fl(w[p:p + nv]), y=fl(w[p + nv:p + 2 * nv]), vx=fl(w[p + 2 * nv:p + 3 * nv]), vy=fl(w[p + 3 * nv:p + 4 * nv]),
               mass=w[p + 4 * nv:p + 5 * nv].astype(np.int64)); p += 7 * nv
    food = dict(x=fl(w[p:p + nf]), y=fl(w[p + nf:p + 2 * nf]), vx=fl(w[p + 2 * nf:p + 3 * nf]), vy=fl(w[p + 3 * nf:p + 4 * nf])); p += 5 * nf
    players = []
    for _ in range(npl):
        h = w[p:p + 17]; nt = int(h[16]); p += 17
        vt = [int(np.int32(x)) for x in w[p:p + nt]]; p += nt
        nc = int(h[2]); cells = w[p:p + 9 * nc].reshape(nc, 9); p += 9 * nc
        players.append(dict(pid=int(np.int32(h[0])), is_bot=int(h[1]), target=(_f(h[4]), _f(h[5])), split_cd=int(h[6]), feed_cd=int(h[7]),
                            elapsed=int(h[8]), last_decay=int(h[9]), anti_team=_f(h[10]), food_eaten=int(h[11]), highest_mass=int(h[12]),
                            cells_eaten=int(h[13]), viruses_eaten=int(h[14]), virus_ticks=vt, cells=cells))
    return dict(ticks=ticks, id_counter=idc, next_pid=next_pid, pellets=pel, viruses=vir, foods=food, players=players)


def blob_to_json(words, cfg, seed, names, colors):
    """The dict BaseEnvironment::save_env_state writes (BaseEnvironment.hpp:213-310) for one arena.
    cfg: the env's constructor arguments; names / colors: per player in the blob's (= map iteration) order.
    cfg["mode_number"] is what goes into the file's "mode_number": the reference writes Engine::mode_number
    (BaseEnvironment.hpp:227), a member the constructor never sets (its parameter shadows it, Engine.hpp:45-48,354) --
    0 unless a snapshot was loaded before, then that file's value (Engine.hpp:263).  Callers that mirror the reference
    (agarcl_amd/agarcl.py) pass exactly that; the true mode is a constructor argument and is not in the file."""
    d = parse_blob(words)
    fx = lambda v: float(np.float32(v))
    out = dict(num_agents=int(cfg["num_agents"]), ticks_per_step=int(cfg["ticks_per_step"]), arena_size=int(cfg["arena_size"]),
               num_bots=int(cfg["num_bots"]), reward_type=bool(cfg["reward_type"]), seed=int(np.int32(np.uint32(seed))), c_death=int(cfg["c_death"]),
               mode_number=int(cfg["mode_number"]), pellet_regen=bool(cfg["pellet_regen"]), pellet_count=len(d["pellets"]["x"]))
    out["players"] = []
    for pl, name, color in zip(d["players"], names, colors):
        cells = [dict(id=int(np.int32(c[7])), x=_f(c[0]), y=_f(c[1]), mass=int(c[6]), velocity_x=_f(c[2]), velocity_y=_f(c[3]), color=int(color))
                 for c in pl["cells"]]
        out["players"].append(dict(pid=pl["pid"], name=name, target_x=fx(pl["target"][0]), target_y=fx(pl["target"][1]), is_bot=bool(pl["is_bot"]),
                                   dead=len(cells) == 0, split_cooldown=pl["split_cd"], feed_cooldown=pl["feed_cd"], virus_eaten_ticks=list(pl["virus_ticks"]),
                                   cells=cells, anti_team_decay=fx(pl["anti_team"]), elapsed_ticks=pl["elapsed"], last_decay_tick=pl["last_decay"],
                                   food_eaten=pl["food_eaten"], highest_mass=pl["highest_mass"], cells_eaten=pl["cells_eaten"],
                                   viruses_eaten=pl["viruses_eaten"], top_position=0))
    out["pellets"] = [dict(x=fx(x), y=fx(y)) for x, y in zip(d["pellets"]["x"], d["pellets"]["y"])]
    out["viruses"] = [dict(x=fx(x), y=fx(y), velocity_x=fx(vx), velocity_y=fx(vy), mass=float(m))
                      for x, y, vx, vy, m in zip(d["viruses"]["x"], d["viruses"]["y"], d["viruses"]["vx"], d["viruses"]["vy"], d["viruses"]["mass"])]
    out["foods"] = [dict(x=fx(x), y=fx(y), velocity_x=fx(vx), velocity_y=fx(vy))
                    for x, y, vx, vy in zip(d["foods"]["x"], d["foods"]["y"], d["foods"]["vx"], d["foods"]["vy"])]
    return out


def dumps(snapshot):
    """nlohmann's `out << std::setw(4) << json` (BaseEnvironment.hpp:309): keys sorted, 4-space indent."""
    return json.dumps(snapshot, indent=4, sort_keys=True) + "\n"


def json_to_adopt(snapshot, num_players, hm_buckets, hm_next_resize, id_counter=1, virus_initial_mass=100):
    """Engine::load_env_state restated: (blob words with the players in the NEW map iteration order, kinds in the
    same order, names in the same order, hm_buckets, hm_next_resize, seed)."""
    players = snapshot["players"]
    if len(players) != num_players:
        raise RuntimeError("snapshot has %d players, the environment was built for %d" % (len(players), num_players))
    idc = int(id_counter)
    recs = []
    for pid, pd in enumerate(players):   # fresh pids in file order; one id for the respawned throw-away cell
        idc += 1
        cells = []
        for cd in pd["cells"]:
            idc += 1
            mass = int(np.float32(cd["mass"]))     # .get<float>() -> agario::mass (unsigned), then set_mass's floor (Entities.hpp:171-177)
            cells.append((_u(cd["x"]), _u(cd["y"]), _u(cd["velocity_x"]), _u(cd["velocity_y"]), 0, 0, max(mass, CELL_MIN_SIZE), int(cd["id"]) & 0xFFFFFFFF, 0))
        recs.append((pid, pd, cells))
    order, hb, hr = map_order_after_inserts(list(range(len(players))), hm_buckets, hm_next_resize)
    w = [MAGIC, 0, 0, len(players), len(snapshot["pellets"]), len(snapshot["viruses"]), len(snapshot["foods"]), len(players)]
    pel = snapshot["pellets"]; ids = list(range(idc + 1, idc + 1 + len(pel))); idc += len(pel)
    w += [_u(p["x"]) for p in pel] + [_u(p["y"]) for p in pel] + ids
    vir = snapshot["viruses"]; ids = list(range(idc + 1, idc + 1 + len(vir))); idc += len(vir)
    w += [_u(v["x"]) for v in vir] + [_u(v["y"]) for v in vir] + [_u(v["velocity_x"]) for v in vir] + [_u(v["velocity_y"]) for v in vir]
    w += [int(np.float32(v["mass"])) for v in vir] + [0] * len(vir) + ids
    foods = snapshot["foods"]; ids = list(range(idc + 1, idc + 1 + len(foods))); idc += len(foods)
    w += [_u(f["x"]) for f in foods] + [_u(f["y"]) for f in foods] + [_u(f["velocity_x"]) for f in foods] + [_u(f["velocity_y"]) for f in foods] + ids
    kinds, names = [], []
    for k in order:
        pid, pd, cells = recs[k]
        kind = KIND_BY_NAME.get(pd["name"], 0)
        kinds.append(kind); names.append(pd["name"])
        vt = [int(t) & 0xFFFFFFFF for t in pd["virus_eaten_ticks"]]
        w += [pid, 1 if pd["is_bot"] else 0, len(cells), 0, _u(pd["target_x"]), _u(pd["target_y"]), int(pd["split_cooldown"]), int(pd["feed_cooldown"]),
              int(pd["elapsed_ticks"]), int(pd["last_decay_tick"]), _u(pd["anti_team_decay"]), int(pd["food_eaten"]), int(pd["highest_mass"]),
              int(pd["cells_eaten"]), int(pd["viruses_eaten"]), CELL_MIN_SIZE, len(vt)] + vt
        for c in cells:
            w += list(c)
    w[2] = idc
    return np.array([x & 0xFFFFFFFF for x in w], dtype=np.uint32), np.array(kinds, dtype=np.int32), names, hb, hr, int(snapshot["seed"])


# ---- engine-level helpers ----------------------------------------------------------------------------------------
AR_IDC, AR_ORDER0, AR_HM_BUCKETS, AR_HM_RESIZE = 2, 13, 45, 46   # agarcl_amd/csrc/agar_types.h (32 order slots)
PL_KIND = 16


def default_names(engine, arena):
    """Names as BaseEnvironment::reset gives them (:187, "agent<i>"; bots carry their class name), in the engine's
    iteration order; colours: bots have class colours, agents' random colours are not tracked (0)."""
    ar, pl = engine.arena_words(arena)
    names, colors = [], []
    for k in range(pl.shape[0]):
        slot = int(ar[AR_ORDER0 + k]); kind = int(pl[slot, PL_KIND])
        names.append("agent%d" % slot if kind == 0 else NAME_BY_KIND[kind]); colors.append(BOT_COLOR.get(kind, 0))
    return names, colors


def save_arena(engine, arena, cfg, names=None, colors=None):
    """One arena of a BatchedEngine as the reference's snapshot dict."""
    dn, dc = default_names(engine, arena)
    return blob_to_json(engine.dump(arena), cfg, engine.seeds()[arena], names or dn, colors or dc)


def load_arena(engine, arena, snapshot, reset_ids=True):
    """BaseEnvironment::load_env_state for one arena of a BatchedEngine; returns the players' names in the new
    iteration order (keep them if the arena is to be saved again)."""
    ar, _ = engine.arena_words(arena)
    blob, kinds, names, hb, hr, seed = json_to_adopt(snapshot, engine.players, int(ar[AR_HM_BUCKETS]), int(ar[AR_HM_RESIZE]),
                                                     1 if reset_ids else int(ar[AR_IDC]))
    engine.adopt(arena, blob, kinds, hb, hr)
    engine.seed_arena(arena, seed)
    return names
