"""What do the sub-batches' kernels do on the device's clock when AgarioVectorEnv.step() is pipelined?  Run under
rocprofv3 --kernel-trace --output-format csv (python3 directly behind `--`); argv: sub_batches [steps]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from agarcl_amd.vector_env import AgarioVectorEnv
k = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
A = 4096
venv = AgarioVectorEnv(A, obs_type="none", mode=6, num_viruses=25, sub_batches=k, strict_flags=False, number_steps=100000)
venv.reset(seed=10000)
g = torch.Generator(device=venv.device); g.manual_seed(0)
move = torch.rand((A, 2), generator=g, device=venv.device) * 2 - 1; kind = torch.randint(0, 3, (A,), generator=g, device=venv.device, dtype=torch.int32)
for _ in range(steps):
    venv.step((move, kind))
torch.cuda.synchronize()
venv.close()
