"""us per 4-tick step of configurations a user is likely to run (presets, bots, several agents, larger arenas) at 4096 arenas: looks for instantiations
that are pathologically slow (round 6 found two: scratch of the several-player kernels).  python scripts/gpu_config_sweep.py   (AGARCL_HIP_SO selects a build)"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from agarcl_amd.vec_env import VecEnvironment
A=4096
CASES=[("C1-like (agent + 4 bots, 250x250, 500 pellets, 10 viruses)", dict(arena_size=250, num_pellets=500, num_viruses=10, num_bots=4)),
       ("task 10-like (1 bot, 350x350, 500 pellets, mode 10)", dict(arena_size=350, num_pellets=500, num_bots=1, mode_number=10)),
       ("30 ExampleBots, no agent (Tick/30)", dict(num_agents=0, example_bots=30, arena_size=250, num_pellets=500, num_viruses=10)),
       ("normal + 25 bots (1000x1000, 1000 pellets)", dict(num_bots=25)),
       ("normal + 4 bots", dict(num_bots=4)),
       ("trivial preset (50x50, 200 pellets)", dict(arena_size=50, num_pellets=200)),
       ("trivial + mode 6", dict(arena_size=50, num_pellets=200, mode_number=6)),
       ("3 agents, 1000x1000", dict(num_agents=3)),
       ("3 agents mode 6 + 25 viruses", dict(num_agents=3, mode_number=6, num_viruses=25)),
       ("1100x1100 1300 pellets 2 agents + 3 bots", dict(arena_size=1100, num_pellets=1300, num_agents=2, num_bots=3)),
       ("1100x1100 1000 pellets 1 agent + 3 bots", dict(arena_size=1100, num_pellets=1000, num_bots=3)),
       ("mode 6, 350x350, 500 pellets, 25 viruses", dict(arena_size=350, num_pellets=500, num_viruses=25, mode_number=6)),
       ]
for name,kw in CASES:
    na=kw.get("num_agents",1)
    if na == 0:
        from agarcl_amd import _capi
        eng=_capi.BatchedEngine(A, **{("mode" if k=="mode_number" else k):v for k,v in kw.items()}); eng.seed(None,10000); eng.reset(reset_ids=True)
        for t in range(20): eng.tick(4)
        eng.sync(); t0=time.perf_counter()
        for t in range(40): eng.tick(4)
        eng.sync(); dt=(time.perf_counter()-t0)/40
        print("%-62s %8.1f us/step  %7.1f M env-steps/s" % (name, dt*1e6, A*4/dt/1e6), flush=True); eng.close(); continue
    env=VecEnvironment(A, strict_flags=False, **kw); env.seed(base_seed=10000); env.reset()
    g=torch.Generator(device="cuda"); g.manual_seed(0)
    dx=torch.rand((60,A,na,2),generator=g,device="cuda")*2-1; ac=torch.randint(0,3,(60,A,na),generator=g,device="cuda",dtype=torch.int32)
    for t in range(20): env.take_actions(dx[t],ac[t]); env.step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for t in range(20,60): env.take_actions(dx[t],ac[t]); env.step()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/40
    fl=int((env.engine.flags()!=0).sum())
    print("%-62s %8.1f us/step  %7.1f M env-steps/s  flagged %d  device MB %.0f" % (name, dt*1e6, A*4/dt/1e6, fl, env.engine.device_bytes()/1e6), flush=True)
    env.close()
