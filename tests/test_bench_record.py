"""bench.py's stdout record: compact, parseable, contract keys present (the 24.6 KB line of round 5 reached the driver as
`parsed: null`).  CPU test: the record is built from canned `measure()` results -- the committed full report of round 5 and a minimal one."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline")


def _check(rec, n1=True):
    line = json.dumps(rec, separators=(",", ":"))
    assert len(line) < bench.MAX_LINE_BYTES, len(line)
    back = json.loads(line)
    for k in CONTRACT:
        if k == "cpu_baseline" and not n1:
            continue
        assert k in back, k
    assert set(back["config"]) <= {"workload", "global_batch", "parallelism"} and "workload" in back["config"]
    r = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma")
    assert r["traffic"] is None or isinstance(r["traffic"], (int, float))
    if n1:
        for k in ("value", "unit", "cores", "kind"):
            assert k in back["cpu_baseline"], k
    return back


def test_record_from_the_round5_full_report():
    fn = os.path.join(ROOT, "profiles", "r05_t8_bench.json")
    if not os.path.exists(fn):
        pytest.skip("no committed full report")
    full = json.load(open(fn))
    assert len(json.dumps(full)) > 20000          # what the driver could not parse
    back = _check(bench.compact_record(full, detail="bench_detail.json"))
    assert back["value"] == full["value"] and back["ms_per_step"] == full["ms_per_step"]
    assert abs(back["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-9
    assert back["roofline"]["frac_in_step_bracket"] < back["roofline"]["frac"]
    assert back["roofline"]["traffic"] == full["roofline"]["traffic"]["bytes_per_launch"]
    for wl in ("acdc", "pancreas"):
        e = back["extra_workloads"][wl]
        assert e["value"] == full["extra_workloads"][wl]["value"]
        assert e["roofline_frac"] == full["extra_workloads"][wl]["roofline"]["frac"]
        assert e["cpu_baseline_value"] == full["extra_workloads"][wl]["cpu_baseline"]["value"]


def test_record_minimal_and_multi_rank_and_failed_extra():
    out = {"metric": "m", "value": 1.0, "unit": "volumes/s", "n_gpus": 2, "steps": 3, "warmup": 1, "ms_per_step": 2.0, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": bench.DTYPE, "data": "synthetic",
           "config": {"workload": "w", "global_batch": 8, "parallelism": "dp2", "host_path": "x" * 5000, "arithmetic": "y" * 5000},
           "roofline": {"bound": "mfma", "achieved": None, "peak": 157.3, "unit": "TFLOP/s", "frac": None, "traffic": None},
           "kernels": [{"op": "o" * 100}] * 200, "ranks_seen": 2, "exposed_allreduce_ms_per_step": 0.1,
           "extra_workloads": {"acdc": {"error": "RuntimeError: " + "z" * 1000}}}
    back = _check(bench.compact_record(out), n1=False)
    assert back["ranks_seen"] == 2 and "kernels" not in back and "host_path" not in back["config"]
    assert len(back["extra_workloads"]["acdc"]["error"]) <= 120
