"""bench.py's N > 1 path end to end on a 1-GPU box.  This file sorts FIRST on purpose: every test here only starts child processes, and
they are started before the pytest process has made any GPU call (a process that has initialised the GPU must not exec, and that state is
inherited by a forked child)."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--gather", "step"], ["--gather", "step", "--gather-obs", "screen", "--arenas", "1024"],
                                   ["--sub-batches", "2"], ["--sub-batches", "2", "--gather", "step", "--gather-obs", "screen", "--arenas", "1024"]],
                         ids=["block", "step", "step+screen", "pipe2", "pipe2+step+screen"])
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
    assert len(lines[0]) < 6500 and all(len(l) < 300 for l in p.stderr.splitlines()[-20:])   # the driver keeps an 8 KB tail of stdout + stderr
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["backend"] == "gloo" and len(line["rank_devices"]) == 2
    assert len(line["ranks"]) == 2 and all(len(r) == len(line["ranks_columns"]) and r[0] > 0 and r[1] > 0 for r in line["ranks"])
    b = json.load(open(os.path.join(root, line["full_record"])))            # the full record of the same run, beside bench.py
    for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus", "config"):
        assert b[k] == line[k], k
    assert b["steps"] == 40 and b["warmup"] == 10 and b["scaling"] == "weak" and b["value"] > 0
    arenas = 1024 if "--arenas" in extra else 4096
    assert b["config"]["arenas_total"] == 2 * arenas
    assert abs(b["value"] - 2 * arenas * 4 * 40 / (b["ms_per_step"] * 1e-3 * 40)) / b["value"] < 1e-6   # whole-job aggregate
    assert ("screen frames" in b["config"]["parallelism"]) == ("--gather-obs" in extra)
    # what a poor scaling curve would be read from: one record per rank, and whether the observation gather ran
    assert b["obs_gather_ran"] == ("--gather-obs" in extra) and b["result_gather"] == ("step" if "--gather" in extra else "block")
    assert [r["rank"] for r in b["ranks"]] == [0, 1]
    for r in b["ranks"]:
        assert r["ms_per_step"] > 0 and r["kernel_ms_per_step"] > 0 and r["gather_wait_ms_total"] >= 0
        sub = 2 if "--sub-batches" in extra else 1       # every sub-batch gathers its own result ring: `sub` collectives where one engine makes one
        assert ("sub-batches" in b["config"]["parallelism"]) == (sub > 1)
        if "--gather" in extra:   # one collective per step, 8 bytes per arena
            assert r["result_collectives"] == 40 * sub and r["result_bytes_sent"] == 40 * arenas * 8
        else:                     # whole 32-step blocks of the result ring: steps 10 .. 49 complete block 0 and leave block 1 partial (flushed as a block)
            assert r["result_collectives"] == 2 * sub and r["result_bytes_sent"] == 2 * 32 * arenas * 8
        assert r["obs_collectives"] == (40 if "--gather-obs" in extra else 0) and r["obs_bytes_sent"] == (40 * arenas * 84 * 84 * 3 if "--gather-obs" in extra else 0)


@pytest.mark.gpu
def test_bench_rank_count_mismatch_is_an_error_gpu():
    env = dict(os.environ); env["WORLD_SIZE"] = "3"; env["AGAR_BENCH_BACKEND"] = "gloo"
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=120, cwd=root)
    assert p.returncode != 0 and "must equal --gpus" in (p.stderr + p.stdout)



@pytest.mark.gpu
def test_bench_line_as_the_driver_runs_it_gpu():
    """`python bench.py --gpus 1 --steps 20 --warmup 5`: ONE JSON line with the contract's fields, the roofline and cpu_baseline objects, and the
    blocks VERDICT r2 asked for (the full rule set and the mid-game in the same run, the C3 / mode 6 CPU baseline, >= 50 k arenas)."""
    import json
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stderr[-1500:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # round 5's line was 52 KB and the driver, which keeps an 8 KB tail of stdout + stderr, could not parse it: the size IS part of the contract
    assert len(lines[0]) < 6500, len(lines[0])
    assert len(p.stderr) < 1500, p.stderr[-1500:]
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert all(len(v) <= 120 for v in list(line["config"].values()) + list(line["roofline"].values()) + list(line["cpu_baseline"].values()) if isinstance(v, str))
    lr, lc = line["roofline"], line["cpu_baseline"]
    assert lr["bound"] == "hbm" and lr["unit"] == "GB/s" and 0 < lr["frac"] <= 1 and abs(lr["frac"] - lr["achieved"] / lr["peak"]) < 1e-3 and "traffic" in lr and lr["kernel_ms"] > 0
    assert lc["kind"] in ("reference", "port") and lc["cores"] >= 1 and lc["value"] > 0 and lc["c3m6_value"] > 0 and lc["unit"] == "env-steps/s" and lc["sample"]
    lcols, lby = lr["by_workload_columns"], lr["by_workload"]
    assert all(isinstance(row, list) and len(row) == len(lcols) and row[0] > 0 and row[1] > 0 for row in lby.values()), lby
    for k in ("ms_C3m6", "ms_C3m6_pipe4", "ms_mid", "ms_C1", "ms_C5", "ms_C5s", "ms_task5", "ms_task6", "ms_tick30", "ms_C2_65536", "ms_C3m6_32768"):
        assert lr[k] > 0, k
    assert set(lr["gym_vector_us"]) == set(lr["gym_vector_auto_us"]) == {"no_obs", "ram_obs", "screen_obs_84"}
    b = json.load(open(os.path.join(root, line["full_record"])))            # the full record of the same run, beside bench.py
    assert set(lby) == set(b["roofline"]["by_workload"])
    for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus", "config"):
        assert b[k] == line[k] or k == "config", k
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in b, k
    assert (b["n_gpus"], b["steps"], b["warmup"], b["dtype"], b["data"], b["vs_baseline"], b["higher_is_better"]) == (1, 20, 5, "f32", "synthetic", None, True)
    assert "workload" in b["config"] and "model" not in b["config"]
    assert abs(b["value"] - 4096 * 4 * 20 / (b["ms_per_step"] * 1e-3 * 20)) / b["value"] < 1e-6
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "requested_bytes_per_step", "frac_of_requested", "kernel_ms"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if r["traffic"] is not None:      # PMC traffic is quoted only while profiles/ was recorded on this kernel source
        assert r["traffic"] >= r["requested_bytes_per_step"] * 0.99 and 0 < r["frac_of_requested"] <= 1.01
    c = b["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["c3m6_value"] > 0 and c["unit"] == "env-steps/s" and c["sample"]
    assert b["roofline_large"]["arenas"] == 65536 and b["roofline_xlarge"]["arenas"] == 262144
    assert [e["arenas"] for e in b["roofline_sweep"]] == [16384, 131072] and all("error" not in e and e["ms_per_step"] > 0 and 0 < e["frac"] <= 1 for e in b["roofline_sweep"])
    full = b["roofline_full"]
    base = {"C3m6@4096", "mid@4096", "C1@4096", "C5@4096", "C5s@4096"}                    # every BASELINE config on the driver's clock ...
    assert {k for k in full if "/" not in k and not k.startswith("task")} == base | {"C3m6@32768"}
    assert {k for k in full if "/pipe" in k} == {k + "/pipe4" for k in base} | {"task%d@4096/pipe4" % m for m in range(5, 11)}   # ... and as four independent sub-batches
    assert {k for k in full if k.startswith("task") and "/" not in k} == {"task%d@4096" % m for m in range(1, 11)}
    for k, v in full.items():
        assert "error" not in v, (k, v)
        assert v["kernel_ms"] > 0 and v["work_per_step"]["general_engine_arena_steps"] > (4000 if not k.startswith("task") else -1)
        if "/pipe" in k:
            assert v["sub_batches"] == int(k[-1]) and v["sub_batches_concurrent"] == int(k[-1]), (k, v.get("sub_batches_concurrent"))
    assert full["C1@4096"]["cpu_reference_ticks_per_s_1core"] > 0                        # bench/main.cpp's population on the reference engine
    assert full["C5@4096"]["streaming_model_bytes_per_step"] > 4096 * 8 * 128 * 128 * 4 and "k_grid_obs" in full["C5@4096"]["kernel"]
    assert "k_screen_obs" in full["C5s@4096"]["kernel"]
    # both fractions under unambiguous names in every block, and the regime the size is in
    blocks = [r, b["roofline_large"], b["roofline_xlarge"]] + list(full.values())
    for v in blocks:
        assert v["frac_streaming_model"] > 0 and v["regime"].split(":")[0] in ("latency", "bandwidth")
        assert (v["frac_hbm_traffic"] is None) == (v["traffic"] is None)
        if v["traffic"] is not None:
            assert abs(v["frac_hbm_traffic"] - v["frac"]) < 1e-12
    assert b["ranks"][0]["rank"] == 0 and b["obs_gather_ran"] is False
    # bench/main.cpp's own population (Tick/N: N ExampleBots, no Player) batched, the reference engine's rate beside it
    tick = b["bench_main_cpp_tick"]
    assert set(tick) == {"Tick/%d@4096" % n for n in (0, 5, 10, 20, 30)}
    for v in tick.values():
        assert "error" not in v and v["gpu_env_steps_per_s"] > 0 and v["cpu_ticks_per_s_1core"] > 0 and v["cpu_kind"] in ("reference", "port")
    assert full["mid@4096"]["mean_counts_pellets_viruses_foods_cells"][3] > 1.5      # the agents have grown and split
    assert b["capacity_flags_raised"] == 0
    # everything the driver's parse keeps lives inside `roofline`: one compact row per workload measured in this run, the issue-rate
    # figures that bind the general engine (None until the round's SQ pass is committed for this kernel source), the tasks table, the
    # vector surface
    cols = r["by_workload_columns"]
    assert cols[:2] == ["ms_per_step", "env_steps_per_s"] and "frac_valu_issue" in cols and "mean_wave_residency" in cols and "cpu_reference_env_steps_per_s" in cols
    by = r["by_workload"]
    want = ({"C2@%d" % a for a in (4096, 16384, 65536, 131072, 262144)} | set(full) | set(tick))
    assert set(by) == want, sorted(set(by) ^ want)
    for k, row in by.items():
        assert isinstance(row, list) and len(row) == len(cols) and row[0] > 0 and row[1] > 0, (k, row)
    assert by["C2@4096"][cols.index("cpu_reference_env_steps_per_s")] == c["value"] and by["C3m6@4096"][cols.index("cpu_reference_env_steps_per_s")] == c["c3m6_value"]
    for k in ("frac_valu_issue", "mean_wave_residency", "clock_ghz_measured", "guide_copy_GBs", "frac_of_guide_copy", "frac_of_measured_copy"):
        assert k in r, k
    assert r["guide_copy_GBs"] == 6300.0
    assert set(r["tasks"]["rows"]) == {"task%d" % m for m in range(1, 11)} and all(row[1] > 0 and row[6] > 0 for row in r["tasks"]["rows"].values())
    gv = r["gym_vector"]
    assert set(gv) == {"no_obs", "ram_obs", "screen_obs_84"} and all(v["host_us_per_step"] > 0 and v["gym_vector_steps_per_s"] > 0 for v in gv.values())
    assert b["gym_vector_steps_per_s"] == gv["no_obs"]["gym_vector_steps_per_s"]
    # the surface's own choice of sub-batching never costs more than the un-pipelined form: on these light workloads it IS that form
    ga = r["gym_vector_auto"]
    assert set(ga) == set(gv) and all(ga[k]["us_per_step"] < 1.5 * gv[k]["us_per_step"] + 10 for k in gv), (ga, gv)
