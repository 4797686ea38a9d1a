"""A rollout loop through the batched user surface (agarcl_amd/vector_env.py: AgarioVectorEnv): N arenas, a random policy on the device,
T steps collected into a rollout buffer -- the shape of a PPO / IMPALA actor -- with nothing inside the loop that waits for the GPU.

    python examples/vector_rollout.py [--envs 4096] [--steps 256] [--obs grid|ram|screen|none] [--difficulty normal] [--mode 6] [--sub-batches auto|k [--halves]]

--sub-batches k: the arenas as k independent ranges on HIP streams of their own (a range never waits for the slowest arena of another, one
range's observation kernel runs under another's step).  The default, "auto", is what AgarioVectorEnv picks: ONE range for the full-batch
step() (which orders every range against the caller's stream every step: measured slower with ranges, scripts/gpu_vec_pipe_ab.py), and with
--halves agarcl_amd/vec_env.py default_sub_batches: 4 where the general engine handles most arena-steps -- bots, several agents, modes 5 / 6
(try --mode 6 --halves) --, 1 for quiet batches such as the default mode 0.
--halves: double-buffered sampling through recv(j) / send(actions_j, j) -- the policy works on range j's observations while the other ranges step.

Prints env-steps per second of the whole loop (engine step + observation + Python), which is what a learner sees; `bench.py` times the
engine alone."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # (run from a checkout: the package is in-tree)
import torch
from agarcl_amd.vector_env import AgarioVectorEnv

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--steps", type=int, default=256)
ap.add_argument("--obs", default="ram"); ap.add_argument("--difficulty", default="normal"); ap.add_argument("--number-steps", type=int, default=500)
ap.add_argument("--bare", action="store_true", help="time venv.step alone: one fixed action batch, no rollout buffer")
ap.add_argument("--sub-batches", default="auto", type=lambda v: v if v == "auto" else int(v)); ap.add_argument("--halves", action="store_true", help="recv / send per sub-batch instead of full-batch step()")
ap.add_argument("--mode", type=int, default=0)
a = ap.parse_args()
venv = AgarioVectorEnv(a.envs, obs_type=a.obs, difficulty=a.difficulty, number_steps=a.number_steps, env_type=0, sub_batches=a.sub_batches, halves=a.halves, mode=a.mode,
                       **({"num_viruses": 25} if a.mode else {}))
dev = venv.device
obs, _ = venv.reset(seed=1)
g = torch.Generator(device=dev); g.manual_seed(0)
buf_obs = None if obs is None else torch.empty((a.steps,) + tuple(obs.shape), dtype=obs.dtype, device=dev)
buf_rew = torch.empty((a.steps, a.envs), dtype=torch.float32, device=dev)
buf_done = torch.empty((a.steps, a.envs), dtype=torch.bool, device=dev)
def policy():   # uniform moves, action kind ~ U{none, feed, split}
    return torch.rand((a.envs, 2), generator=g, device=dev) * 2 - 1, torch.randint(0, 3, (a.envs,), generator=g, device=dev, dtype=torch.int32)
for _ in range(8): venv.step(policy())                      # warm-up
if a.bare:
    act = policy(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(a.steps): venv.step(act)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d envs x %d steps, obs=%s, step() alone: %.3g env-steps/s (%.1f us per vector step)" % (a.envs, a.steps, a.obs, a.envs * a.steps / dt, dt / a.steps * 1e6))
    venv.close(); sys.exit(0)
if a.halves:   # every range on its own: range j's policy output is computed from range j's observation while the other ranges step
    k = venv.sub_batches
    def policy_j(n):
        return torch.rand((n, 2), generator=g, device=dev) * 2 - 1, torch.randint(0, 3, (n,), generator=g, device=dev, dtype=torch.int32)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ended = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(k)]
    for t in range(a.steps):
        for j in range(k):
            with torch.cuda.stream(venv.stream(j)):                   # range j's copies and policy on range j's own stream: no cross-stream ordering
                obs_j, rew_j, term_j, trunc_j, info_j = venv.recv(j)
                lo, cnt = venv.ranges[j]
                if buf_obs is not None: buf_obs[t, lo:lo + cnt].copy_(obs_j)
                buf_rew[t, lo:lo + cnt].copy_(rew_j); buf_done[t, lo:lo + cnt].copy_(term_j)
                ended[j] += info_j["ended"].sum()
                venv.send(policy_j(cnt), j)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%d envs x %d steps as %d ranges (recv / send halves), obs=%s: %.3g env-steps/s (%.1f us per vector step); %d episodes ended"
          % (a.envs, a.steps, k, a.obs, a.envs * a.steps / dt, dt / a.steps * 1e6, sum(int(e.item()) for e in ended)))
    venv.close(); sys.exit(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
episodes = torch.zeros((), dtype=torch.int64, device=dev); ret_sum = torch.zeros((), dtype=torch.float32, device=dev)
for t in range(a.steps):
    obs, rew, term, trunc, info = venv.step(policy())
    if buf_obs is not None: buf_obs[t].copy_(obs)           # (the env rewrites its tensors in place: a rollout buffer copies them)
    buf_rew[t].copy_(rew); buf_done[t].copy_(term)
    episodes += info["ended"].sum(); ret_sum += (info["final_return"] * info["ended"]).sum()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
n = int(episodes.item())
print("%d envs x %d steps, obs=%s: %.3g env-steps/s through AgarioVectorEnv.step (%.1f us per vector step); %d episodes ended, mean return %.1f"
      % (a.envs, a.steps, a.obs, a.envs * a.steps / dt, dt / a.steps * 1e6, n, float(ret_sum.item()) / max(n, 1)))
venv.close()
