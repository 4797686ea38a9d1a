"""Shared body of the snapshot tests (CPU: wave-emulation build; GPU: the HIP engine)."""
import glob
import json
import os

import numpy as np

from oracle import blob

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def snapshot_cases():
    return sorted(p[:-4] for p in glob.glob(os.path.join(GOLDEN, "snapshot_*.npz")))


def strip(s):
    """what cannot be compared after a load: the reference draws a fresh random colour for re-created agents"""
    s = json.loads(json.dumps(s))
    for p in s["players"]:
        for c in p["cells"]:
            c.pop("color")
    return s


def replay_snapshot_case(engine_cls, base, arena=1, num_arenas=3):
    """Load the reference-written snapshot into one arena of a batched engine that already has some history, and
    follow the reference's recorded continuation.  Returns (ok, message)."""
    from agarcl_amd import snapshot
    z = np.load(base + ".npz")
    cfg = json.loads(str(z["cfg"]))
    na = cfg["num_agents"]
    snap = json.load(open(base + ".json"))
    eng = engine_cls(num_arenas, **cfg)
    eng.seed(np.full(num_arenas, int(z["loader_seed"]), dtype=np.uint32)); eng.reset(reset_ids=True)
    for _ in range(int(z["history_steps"])):
        eng.set_actions(np.zeros((num_arenas, na, 2), np.float32), np.zeros((num_arenas, na), np.int32)); eng.step()
    other_before = eng.dump(0)
    names = snapshot.load_arena(eng, arena, snap, reset_ids=True)
    d = blob.diff(z["post_blob"], eng.dump(arena))
    if d:
        return False, "right after the load: %s" % d
    if not np.array_equal(other_before, eng.dump(0)):
        return False, "loading arena %d disturbed arena 0" % arena
    for t in range(len(z["dxdy"])):
        dxdy = np.zeros((num_arenas, na, 2), np.float32); act = np.zeros((num_arenas, na), np.int32)
        dxdy[arena] = z["dxdy"][t]; act[arena] = z["act"][t]
        eng.set_actions(dxdy, act); eng.step()
        if not np.array_equal(eng.rewards()[arena], z["rewards"][t]):
            return False, "step %d: rewards %r vs reference %r" % (t, eng.rewards()[arena], z["rewards"][t])
        if not np.array_equal(eng.dones()[arena], z["dones"][t]):
            return False, "step %d: dones differ" % t
    d = blob.diff(z["final_blob"], eng.dump(arena))
    if d:
        return False, "after the continuation: %s" % d
    # writer: the same state must serialise to what the reference wrote (it had loaded the same file: mode_number field
    # = the file's, seed = the env-level seed, names from the file)
    wcfg = dict(num_agents=na, ticks_per_step=4, arena_size=cfg["arena_size"], num_bots=cfg["num_bots"], reward_type=True, c_death=0,
                mode_number=int(snap["mode_number"]), pellet_regen=True)
    ours = snapshot.save_arena(eng, arena, wcfg, names=names)
    ref = json.load(open(base + "_resaved.json"))
    a, b = strip(ref), strip(json.loads(snapshot.dumps(ours)))
    if a != b:
        bad = [k for k in a if a[k] != b.get(k)]
        return False, "re-saved JSON differs from the reference's in %s" % bad
    eng.close()
    return True, ""
