"""The line bench.py prints last on stdout (bench_common.headline_line): small, strict JSON, self-sufficient.

Round 5's line had grown to 21.6 KB (18 extras with prose notes) and the driver could not parse it: BENCH_r05.json holds
`parsed: null` and the round's headline went unmeasured.  The canned input here is that very record
(profiles/r05_bench_n1.json), plus a multi-GPU one and a degenerate one with NaNs and numpy scalars."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, ROOT)
import bench_common as BC                                   # noqa: E402


def strict(s):
    def refuse(c):
        raise ValueError(f"non-standard JSON constant {c}")
    return json.loads(s, parse_constant=refuse)


def canned():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))


def test_round5_record_becomes_a_small_line(tmp_path):
    full = canned()
    assert len(json.dumps(full)) > 20000                     # the record that broke the driver
    f = tmp_path / "x" / "extras.json"
    s = BC.headline_line(full, extras_file=str(f))
    assert "\n" not in s and len(s.encode()) <= BC.LINE_LIMIT < 6000
    d = strict(s)
    for k in BC.REQUIRED_KEYS:
        assert k in d, k
    assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"] and d["n_gpus"] == 1
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] == full["roofline"]["traffic"] and r["kernel"] == "spmm_plan_kernel"
    assert all(not isinstance(v, (dict, list)) for v in r.values())              # scalars only
    c = d["cpu_baseline"]
    assert c["value"] == full["cpu_baseline"]["value"] and c["cores"] == 16 and c["kind"] == "port" and c["sample"]
    assert all(not isinstance(v, (dict, list)) for v in c.values())
    assert d["one_shot_ms_per_step"] == full["one_shot_ms_per_step"]
    assert "workload" in d["config"] and "BASELINE configs[1]" in d["config"]["workload"]
    assert "model" not in d["config"]
    es = d["extras_summary"]
    assert 1 <= len(es) <= 12 and es["cfg5_shard"] == [full["extras"]["cfg5_shard"]["ms_per_step"],
                                                        full["extras"]["cfg5_shard"]["roofline"]["frac"]]
    assert es["spmv_cfg3"][0] == full["extras"]["spmv_cfg3"]["ms"]
    # the full record went to the side file, nothing lost
    side = json.load(open(f))
    assert side["extras"].keys() == full["extras"].keys() and side["value"] == full["value"]


def test_line_survives_nan_numpy_and_a_full_house(tmp_path):
    full = canned()
    full["parity_max_err_over_max_abs_vs_oracle"] = float("nan")
    full["value"] = np.float64(full["value"])
    full["n_gpus"] = np.int64(8)
    full["roofline"]["traffic"] = None
    full["roofline"]["traffic_refused"] = "x" * 900                              # prose is cut, never the numbers
    full["config"]["workload"] = full["config"]["workload"] + "; " + "y" * 2000
    full["distributed"] = {"backend": "nccl", "world_size": 8, "ranks_in_an_rccl_all_reduce_of_ones": 8, "scaling": "strong",
                           "row_blocks": [125000] * 8, "rows_total": 1000000}
    full["allgather"] = {"bytes_received_per_gpu": 896000000, "approx_ms": 1.2, "approx_GBps_in_per_gpu": 746.0}
    full["compute_only_gflops"] = 30000.0
    full["extras"]["errors"] = {"cfg5_strong": "RuntimeError('" + "z" * 400 + "')"}
    for i in range(40):                                                            # many more extras than the summary takes
        full["extras"][f"more_{i}"] = {"ms": float(i), "roofline": {"frac": 0.5}, "note": "n" * 500}
    s = BC.headline_line(full, extras_file=str(tmp_path / "e.json"))
    assert len(s.encode()) <= BC.LINE_LIMIT
    d = strict(s)
    assert d["parity_max_err_over_max_abs_vs_oracle"] is None and d["n_gpus"] == 8 and isinstance(d["value"], float)
    assert len(d.get("extras_summary", {})) <= 12
    assert d["distributed"]["world_size"] == 8 and d["compute_only_gflops"] == 30000.0
    for k in BC.REQUIRED_KEYS:
        assert k in d, k


def test_line_without_extras_or_side_file():
    full = canned()
    del full["extras"]
    s = BC.headline_line(full, extras_file=None)
    d = strict(s)
    assert "extras_summary" not in d and "extras_file" not in d and d["roofline"]["frac"] > 0


def test_bench_prints_through_headline_line():
    """bench.py has exactly one place that prints the line, and it goes through headline_line (no bare json.dumps(out))"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "print(json.dumps(" not in src and src.count("print(headline_line(") == 3
