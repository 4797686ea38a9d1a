"""bench.py's N > 1 path end to end on a 1-GPU box.  This file sorts FIRST on purpose: every test here only starts child processes, and
they are started before the pytest process has made any GPU call (a process that has initialised the GPU must not exec, and that state is
inherited by a forked child)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--gather", "step"], ["--gather", "step", "--gather-obs", "screen", "--arenas", "1024"]],
                         ids=["block", "step", "step+screen"])
def test_bench_two_ranks_end_to_end_gpu(extra):
    """bench.py --gpus 2 end to end on a 1-GPU box: the self-spawning launcher (children are started before anything touches the GPU and
    nothing is ever re-exec'd), 2 ranks over gloo sharing GPU 0 (AGAR_BENCH_BACKEND: the driver's multi-GPU runs use nccl == RCCL, a branch
    a 1-GPU box cannot execute), both result-gather modes and the observation gather.  The JSON line must say what ran."""
    import json
    env = dict(os.environ); env["AGAR_BENCH_BACKEND"] = "gloo"; env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10"] + extra,
                       env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE JSON line: %r" % p.stdout[-500:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["world_size"] == 2 and b["backend"] == "gloo" and len(b["rank_devices"]) == 2
    assert b["steps"] == 40 and b["warmup"] == 10 and b["scaling"] == "weak" and b["value"] > 0
    arenas = 1024 if "--arenas" in extra else 4096
    assert b["config"]["arenas_total"] == 2 * arenas
    assert abs(b["value"] - 2 * arenas * 4 * 40 / (b["ms_per_step"] * 1e-3 * 40)) / b["value"] < 1e-6   # whole-job aggregate
    assert ("screen frames" in b["config"]["parallelism"]) == ("--gather-obs" in extra)


@pytest.mark.gpu
def test_bench_rank_count_mismatch_is_an_error_gpu():
    env = dict(os.environ); env["WORLD_SIZE"] = "3"; env["AGAR_BENCH_BACKEND"] = "gloo"
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120, cwd=root)
    assert p.returncode != 0 and "must equal --gpus" in (p.stderr + p.stdout)

