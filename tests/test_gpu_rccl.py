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


def run_bench(extra_env, *argv):
    import socket
    with socket.socket() as sk:                       # a free port for the one-rank rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MXGPU_BENCH_FORCE_DIST="1", MXGPU_DIST_ALWAYS_COLLECTIVE="1", MASTER_PORT=str(port), **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1",
                        "--no-extras", "--no-cpu-baseline", *argv], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["1", "0"])
def test_bench_distributed_path_one_rank(gpu, overlap):
    d = run_bench({"MXGPU_BENCH_OVERLAP": overlap}, "--rows", "131072", "--cols", "30000", "--nnz-row", "24", "--n", "128",
                  "--dtype", "f32")
    assert d["n_gpus"] == 1 and d["value"] > 0
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
