"""BASELINE configs[4] at its STATED size on one GPU, through the export-level boundary (tools/cfg5_full.py): the 8M x 200k
CSR with nnz 512M (6.1 GB), f32 dense 200k x 256, 8.2 GB column-major f32 result in plain malloc'ed memory — sharded over the
device listed eight times (mx_set_devices), unsharded, and as one device-level launch.  Size-independent checks (column
checksum in f64, linearity) plus oracle row blocks around the 2^31- / 2^32-byte marks of the result
(reference: src/matmul.cpp:53-57,138,316-343,361-375)."""
import json
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_cfg5_whole_matrix_through_the_boundary(gpu):
    import cfg5_full
    r = cfg5_full.run()
    print(json.dumps(r))
    assert r["dims"]["rows"] == 8_000_000 and r["dims"]["nnz"] == 512_000_000 and r["dims"]["result_elements"] == 2_048_000_000
    assert r["algorithmic_bytes"] == 14_572_800_004                       # 14,572.8 MB
    assert r["partition_rows_cuts_equal_bench_blocks"]
    assert r["device_level"]["kernel"] == "spmm_plan_kernel"              # AUTO's choice at this size
    assert r["linearity_rel_err"] <= 2e-5
    assert r["sharded_equals_unsharded_bitwise"]                          # row cuts at multiples of 1024 rows: the same plan generations
    # 8 shards on ONE GPU (the second call: its workers, their streams and plan buffers exist): round 3 measured 2.0-2.5x the
    # unsharded cold call (369-467 ms vs 189); round 4 — pieces of the result registered as they are touched, downloads queued
    # behind each block's product, persistent workers, B once per device, the shards of one device taking turns on its link —
    # 1.4x (267 ms vs 192).  What is left is the duplex rate of the one link all eight share and the last shard's download
    # tail; the bound below guards against falling back.
    assert r["export_sharded_ms"][1] <= 1.7 * r["export_unsharded_ms"]["cold"], (r["export_sharded_ms"], r["export_unsharded_ms"])
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "cfg5_full_test.json"), "w") as f:
            json.dump(r, f, indent=1)
