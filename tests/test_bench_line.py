"""bench.py's ONE line, checked without a GPU: round 5's line grew to 52 KB and the driver, which keeps an 8 KB tail of stdout + stderr, could not parse
it.  compact_line() is a pure function of the full record; the committed record of the round's driver-style run (profiles/r06_bench_driver20_full.json,
written by bench.py itself on the GPU box) goes through it here."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(line):
    assert len(line) < bench.LINE_LIMIT + 500 and "\n" not in line
    b = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in b, k
    assert "workload" in b["config"] and "model" not in b["config"]
    r = b["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"):
        assert k in r, k
    assert all(len(v) <= 120 for v in list(b["config"].values()) + list(r.values()) if isinstance(v, str))
    return b


def test_compact_line_of_the_committed_driver_run():
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_driver20_full.json")))
    b = _check(bench.compact_line(full))
    assert b["value"] == full["value"] and b["ms_per_step"] == full["ms_per_step"] and b["steps"] == 20 and b["warmup"] == 5
    c = b["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and len(c["sample"]) <= 120
    rows = b["roofline"]["by_workload"]
    assert set(rows) == set(full["roofline"]["by_workload"]) and all(len(v) == len(b["roofline"]["by_workload_columns"]) for v in rows.values())


def test_compact_line_sheds_rows_before_it_outgrows_the_limit():
    """a record with ten times the rows still yields a line inside the limit: side rows go first, the contract's fields never"""
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_driver20_full.json")))
    by = full["roofline"]["by_workload"]
    for i in range(400):
        by["task%d@4096/pipe%d" % (i, i)] = list(by["task3@4096"])
    b = _check(bench.compact_line(full))
    assert len(bench.compact_line(full)) <= bench.LINE_LIMIT and b["value"] == full["value"]
