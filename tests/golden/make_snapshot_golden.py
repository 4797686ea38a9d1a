"""Golden vectors for the JSON snapshot path, recorded from the REAL reference build
(oracle/_ref/libagar_ref.so = /root/reference compiled as is).  Run in the build container only:

    python tests/golden/make_snapshot_golden.py

For each configuration: (1) a reference env plays `pre` steps and writes snapshot_<name>.json with its own
save_env_state; (2) a second reference env (different seed, a few steps of history) loads that file with
load_env_state; its state blob right after the load, the rewards/dones of `post` further steps under recorded
actions, the final blob, and the file it writes when asked to save again (snapshot_<name>_resaved.json) are stored
in snapshot_<name>.npz.  The .json files are the reference's serializer output (data), not source."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refbind  # noqa: E402

CASES = {
    "single_m6": (dict(num_agents=1, arena_size=200, num_pellets=60, num_viruses=4, num_bots=0, mode=6), 150, 200),
    "bots_m0": (dict(num_agents=2, arena_size=200, num_pellets=80, num_viruses=3, num_bots=3, mode=0), 120, 200),
    "agents14_m4": (dict(num_agents=14, arena_size=300, num_pellets=150, num_viruses=5, num_bots=0, mode=4), 40, 60),  # 14 inserts: map rehash
    "bot_m8": (dict(num_agents=1, arena_size=200, num_pellets=100, num_viruses=5, num_bots=1, mode=8), 100, 150),
}


def main():
    for name, (cfg, pre, post) in CASES.items():
        na = cfg["num_agents"]
        rng = np.random.RandomState(abs(hash(name)) % 1000 + 7)
        e = refbind.RefEnv(**cfg); e.seed(5); e.reset(True)
        for t in range(pre):
            e.take_actions(rng.uniform(-1, 1, (na, 2)).astype(np.float32), rng.randint(0, 3, na).astype(np.int32)); e.step()
            if any(e.dones()):
                e.reset(False)
        path = os.path.join(HERE, "snapshot_%s.json" % name)
        e.save_json(path)
        f = refbind.RefEnv(**cfg); f.seed(99); f.reset(True)
        hist = 7
        for t in range(hist):
            f.take_actions(np.zeros((na, 2), np.float32), np.zeros(na, np.int32)); f.step()
        f.load_json(path, reset_ids=True)
        post_blob = f.dump()
        acts_d, acts_a, rewards, dones = [], [], [], []
        for t in range(post):
            dxdy = rng.uniform(-1, 1, (na, 2)).astype(np.float32); act = rng.randint(0, 3, na).astype(np.int32)
            f.take_actions(dxdy, act); r = f.step()
            acts_d.append(dxdy); acts_a.append(act); rewards.append(r); dones.append(f.dones())
        f.save_json(os.path.join(HERE, "snapshot_%s_resaved.json" % name))
        np.savez_compressed(os.path.join(HERE, "snapshot_%s.npz" % name), cfg=json.dumps(cfg), history_steps=hist, loader_seed=99,
                            post_blob=post_blob, dxdy=np.array(acts_d), act=np.array(acts_a), rewards=np.array(rewards, dtype=np.float64),
                            dones=np.array(dones, dtype=np.uint8), final_blob=f.dump())
        print(name, "players", len(json.load(open(path))["players"]), "post-load words", len(post_blob))


if __name__ == "__main__":
    main()
