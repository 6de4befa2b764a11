"""The N > 1 code path of bench.py on ONE GPU: a one-rank RCCL process group with the collectives forced on
(MXGPU_BENCH_FORCE_DIST=1, MXGPU_DIST_ALWAYS_COLLECTIVE=1).  What a multi-GPU node would run — process-group set-up on
the "nccl" backend, RowShardedSpMM / PipelinedRowShardedSpMM with the in-place asynchronous all_gather_into_tensor, the
per-step events, the parity check over BOTH gathered buffers against the oracle, the JSON line — runs here with one
rank; only the link traffic is missing.  (Two ranks on one GPU are refused by RCCL; the world-2 / world-3 logic is covered
on CPU by tests/test_distributed.py.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra_env, *argv, fixed=("--no-extras", "--no-cpu-baseline")):
    import socket
    with socket.socket() as sk:                       # a free port for the one-rank rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MXGPU_BENCH_FORCE_DIST="1", MXGPU_DIST_ALWAYS_COLLECTIVE="1", MASTER_PORT=str(port), **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        *fixed, *argv], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 6000, f"the bench line grew to {len(last)} bytes (VERDICT r5: 21.6 KB was not parsed)"

    def refuse(c):
        raise ValueError(f"non-standard JSON constant {c}")
    d = json.loads(last, parse_constant=refuse)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline"):
        assert k in d, k
    assert d["config"]["layout"] in ("colmajor", "rowmajor")
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_bench_distributed_path_one_rank(gpu, overlap):
    d = run_bench({"MXGPU_BENCH_OVERLAP": overlap}, "--scaling", "weak", "--rows", "131072", "--cols", "30000", "--nnz-row", "24", "--n", "128",
                  "--dtype", "f32")
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["scaling"] == "weak"
    assert "allgather" in d and d["allgather"]["bytes_received_per_gpu"] == 0      # (world - 1) blocks arrive
    assert d["parity_max_err_over_max_abs_vs_oracle"] <= 2e-5


@pytest.mark.gpu
def test_bench_strong_scaling_path_one_rank(gpu):
    """--scaling strong: --rows is the WHOLE matrix (64 device-drawn row chunks, the same matrix whatever N is), cut into
    nnz-balanced row blocks; the line says so and carries what the driver needs to check "N ranks" (world size, the
    number of ranks seen by an RCCL all-reduce of ones, the row blocks)."""
    d = run_bench({"MXGPU_BENCH_OVERLAP": "1"}, "--scaling", "strong", "--rows", "131072", "--cols", "30000", "--nnz-row", "24", "--n", "128",
                  "--dtype", "f32")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["value"] > 0
    dd = d["distributed"]
    assert dd["world_size"] == 1 and dd["ranks_in_an_rccl_all_reduce_of_ones"] == 1 and dd["backend"] == "nccl"
    assert dd["rows_total"] == 131072 and sum(dd["row_blocks"]) == 131072
    assert d["parity_max_err_over_max_abs_vs_oracle"] <= 2e-5


@pytest.mark.gpu
def test_bench_default_multi_gpu_line_is_the_metrics_workload(gpu):
    """What `python bench.py --gpus N` runs on every rank, with one rank here: NO shape arguments -> BASELINE configs[1]
    (1M x 100k, 32 / row, n = 128, f64), the N = 1 matrix itself cut into nnz-balanced row blocks (strong scaling), the
    CPU baseline and rank 0's roofline on the line, `value` with the all-gather and `compute_only_gflops` without it, and
    configs[4] (f32, strong) beside it as extras.cfg5_strong (fewer rows here: one GPU holds the whole of it)."""
    d = run_bench({"MXGPU_BENCH_OVERLAP": "1"}, "--cfg5-strong-rows", "1048576", "--cpu-seconds", "3", fixed=())
    assert d["dtype"] == "f64" and d["scaling"] == "strong" and d["n_gpus"] == 1
    w = d["config"]["workload"]
    assert "1000000x100000" in w and "nnz/row=32" in w and "100000x128 f64" in w and "BASELINE configs[1]" in w
    assert "fp64, 1M x 100k, 32 nnz/row, k=128" in d["metric"]
    dd = d["distributed"]
    assert dd["world_size"] == 1 and dd["ranks_in_an_rccl_all_reduce_of_ones"] == 1 and dd["rows_total"] == 1_000_000
    assert d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["kind"] == "port"
    assert d["roofline"]["frac"] > 0 and "rank 0" in d["roofline"]["scope"]
    assert d["compute_only_gflops"] >= d["value"] > 0
    assert d["parity_max_err_over_max_abs_vs_oracle"] <= 1e-10
    assert d["config"]["layout"] == "rowmajor"                      # row-major blocks are what the all-gather moves
    assert d["extras_summary"]["cfg5_strong"][0] > 0
    full = json.load(open(os.path.join(ROOT, d["extras_file"])))   # the full record beside the line
    assert full["value"] == d["value"]
    c5 = full["extras"]["cfg5_strong"]
    assert c5["dtype"] == "f32" and c5["value"] > 0 and c5["distributed"]["rows_total"] == 1048576
    assert c5["parity_max_err_over_max_abs_vs_oracle"] <= 2e-5
