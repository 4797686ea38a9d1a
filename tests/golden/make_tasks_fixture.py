#!/usr/bin/env python
"""Writes tests/golden/paper_tasks.json: the VALUES of the reference's ten RL task configurations
(/root/reference/bench/tasks_configs/mode_{1..10}.json -- the workloads the paper trains on), as data.  Run in the build container only
(the reference tree does not exist on the GPU box):

    python tests/golden/make_tasks_fixture.py

Only the keys that configure the environment are kept (the engine / env constructor arguments of gym_agario/AgarioEnv.py:313-349 and the
observation settings); file names and video paths are not.  The sha256 of every source file is recorded so that a change of the reference's
configurations is noticed when the script is run again."""
import hashlib
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/bench/tasks_configs"
KEEP = ("ticks_per_step", "num_frames", "arena_size", "num_pellets", "num_viruses", "num_bots", "pellet_regen", "grid_size", "screen_len",
        "observe_cells", "observe_others", "observe_viruses", "observe_pellets", "obs_type", "reward_type", "c_death", "agent_view", "add_noise",
        "mode", "number_steps", "env_type", "load_env_snapshot")


def main():
    tasks, shas = {}, {}
    for m in range(1, 11):
        path = os.path.join(SRC, "mode_%d.json" % m)
        raw = open(path, "rb").read()
        cfg = json.loads(raw)
        assert cfg["mode"] == m, (path, cfg["mode"])
        tasks[str(m)] = {k: cfg[k] for k in KEEP if k in cfg}
        shas["mode_%d.json" % m] = hashlib.sha256(raw).hexdigest()[:16]
    out = {"source": "machado-research/AgarCL bench/tasks_configs/mode_{1..10}.json (values only)", "source_sha256_16": shas, "tasks": tasks}
    json.dump(out, open(os.path.join(HERE, "paper_tasks.json"), "w"), indent=1, sort_keys=True)
    for m in range(1, 11):
        t = tasks[str(m)]
        print("mode %2d: arena %d, pellets %d, viruses %d, bots %d, %s %dx%d agent_view=%s, number_steps %d, env_type %d"
              % (m, t["arena_size"], t["num_pellets"], t["num_viruses"], t["num_bots"], t["obs_type"], t["screen_len"], t["screen_len"], t["agent_view"], t["number_steps"], t["env_type"]))


if __name__ == "__main__":
    main()
