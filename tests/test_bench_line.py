"""The driver keeps only an ~8 KB tail of bench.py's stdout (round 3's 24 KB line could not be parsed): the ONE JSON line is
built by bench.compact_line from the full record and must stay under 4 KB whatever the full record carries."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = os.path.join(ROOT, "profiles", "r03_bench_driver_short.json")     # a real 24 KB record of round 3


def _full_record():
    with open(CANNED) as f:
        return json.loads(f.read())


def test_compact_line_is_short_and_carries_the_contract():
    full = _full_record()
    assert len(json.dumps(full)) > 20000                   # the canned record is the oversized one
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < 4096
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == full["value"] and d["steps"] == full["steps"] and d["ms_per_step"] == full["ms_per_step"]
    assert d["config"]["workload"]
    r = d["roofline"]
    assert r["frac"] == full["roofline"]["frac"] and r["bound"] in ("hbm", "mfma") and r["achieved"] and r["peak"]
    assert "kernels" not in r and "compute_counters" not in r
    c = d["cpu_baseline"]
    assert c["value"] == full["cpu_baseline"]["value"] and c["cores"] == 1 and c["kind"] == "port" and len(c["sample"]) <= 120
    assert c["all_cores"]["value"] == full["cpu_baseline"]["all_cores"]["value"]
    assert d["upload_inclusive"]["frames_per_s"] == full["upload_inclusive"]["frames_per_s"]
    assert [m["streams_per_gpu"] for m in d["multi_stream"]] == [m["streams_per_gpu"] for m in full["multi_stream"]]
    assert d["distributed"]["world_size"] == 1


def test_compact_line_survives_a_bloated_record_and_missing_parts():
    full = _full_record()
    full["roofline"]["kernels"] = {("k%d" % i): {"x" * 40: list(range(50))} for i in range(200)}
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["config"]["workload"] = "w" * 5000
    full["distributed"]["ranks"] = [dict(full["distributed"]["ranks"][0], rank=i) for i in range(8)]
    line = bench.compact_line(full)
    assert len(line) < 4096 and json.loads(line)["roofline"]["frac"]
    for k in ("roofline", "cpu_baseline", "multi_stream", "upload_inclusive", "latency_mode", "orb_only"):
        full[k] = None                                      # --no-cpu-baseline, N > 1 ranks ...
    d = json.loads(bench.compact_line(full))
    assert d["roofline"] is None and d["cpu_baseline"] is None and d["value"] == full["value"]
