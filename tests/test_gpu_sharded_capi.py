"""mx_spmm_sharded_* (csrc/sharded.hip): row blocks on their devices, products on all of them at once, ONE in-place
ncclAllGather of equal slots so that every device holds the full C — behind the C-ABI, one process, RCCL dlopen()ed.

A one-GPU box runs (a) the one-rank communicator — ncclCommInitAll over {0}, the all-gather call, teardown — against the
unsharded device-level product bit for bit, and (b) three shards on the one device (shared gathered buffer, nothing to
exchange): the cut / slot arithmetic, ragged blocks, empty blocks, both dtypes, the host copy in both layouts.  The
reference product: tcrossprod_csr_dense (src/matmul.cpp:316-343)."""
import ctypes as C

import numpy as np
import pytest
import torch

from matrixextra_amd import _lib, device as D, sharded as S, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _oracle_rows(p, j, x, B, rows):
    out = []
    for r in rows:
        s, e = p[r], p[r + 1]
        out.append(x[s:e] @ B[j[s:e]].astype(np.float64))
    return np.array(out)


def test_one_rank_communicator_matches_the_unsharded_product_bitwise(gpu):
    m, K, n = 200_000, 40_000, 128
    p, j, x = synth.csr_fixed(m, K, 24, seed=3)
    B = synth.dense_normal(K, n)
    sh = S.ShardedSpMM([0], p, j, x, K)
    try:
        assert sh.uses_rccl and sh.nshards == 1 and sh.cuts == [0, m] and sh.rccl_version > 0
        got = sh.run(B)                                             # host copy from device 0, row-major
        A = D.DeviceCSR.from_host(p, j, x, K)
        want = D.spmm(A, torch.from_numpy(B).cuda()).cpu().numpy()  # the same AUTO with the plan kept, unsharded
        assert sh.kernel(0) == _lib.load().mxd_spmm_last_kernel().decode()
        assert np.array_equal(got, want)
        assert np.array_equal(sh.gathered(0), want)                 # what the device holds after the all-gather
        rows = np.array([0, 1, 77_777, m - 1])
        np.testing.assert_allclose(got[rows], _oracle_rows(p, j, x, B, rows), rtol=1e-12, atol=1e-12)
        # device-resident B, asynchronous products: the gather of product k under product k + 1, two buffers alternating
        Bd = torch.from_numpy(B).cuda()
        B2 = torch.from_numpy(synth.dense_normal(K, n, seed=9)).cuda()
        for _ in range(3):
            sh.run_dev([Bd.data_ptr()], n, np.float64, asynchronous=True)
            sh.run_dev([B2.data_ptr()], n, np.float64, asynchronous=True)
        sh.sync()
        assert np.array_equal(sh.gathered(0), D.spmm(A, B2).cpu().numpy())
        cm = sh.run(B, colmajor=True)                               # column-major for R: transposed on the device
        assert cm.flags.f_contiguous and np.array_equal(cm, want)
    finally:
        sh.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("equal_rows", [False, True])
def test_three_shards_on_one_device_fill_their_slots(gpu, dtype, equal_rows):
    rng = np.random.default_rng(5)
    m, K, n = 50_001, 9_000, 96
    lens = rng.integers(0, 30, size=m)
    lens[:4000] = 300                                               # heavy head: the balanced cuts are ragged
    lens[20_000:20_050] = 0
    p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    j = rng.integers(0, K, size=int(p[-1])).astype(np.int32)        # unsorted, duplicates: the reference accepts both
    x = rng.uniform(-1, 1, size=j.size)
    B = rng.normal(size=(K, n)).astype(dtype)
    sh = S.ShardedSpMM([0, 0, 0], p, j, x, K, equal_rows=equal_rows)
    try:
        assert not sh.uses_rccl and sh.nshards == 3 and sh.cuts[0] == 0 and sh.cuts[-1] == m
        if not equal_rows:
            assert sh.cuts[1] < m // 3                              # fewer rows where the entries are
        for colmajor in (False, True):
            got = sh.run(B, colmajor=colmajor)
            want = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, True)
            tol = 1e-12 if dtype == np.float64 else 2e-5
            np.testing.assert_allclose(got, want, rtol=tol, atol=tol * np.abs(want).max())
        assert np.array_equal(sh.gathered(2), np.ascontiguousarray(got))
    finally:
        sh.close()


def test_more_devices_than_rows_and_an_empty_matrix(gpu):
    p = np.array([0, 2, 2, 3], dtype=np.int32)
    j = np.array([1, 0, 2], dtype=np.int32)
    x = np.array([1.0, 2.0, 3.0])
    B = np.arange(12, dtype=np.float64).reshape(3, 4)
    sh = S.ShardedSpMM([0] * 5, p, j, x, 3)
    try:
        got = sh.run(B)
        assert np.array_equal(got, np.array([B[1] + 2 * B[0], np.zeros(4), 3 * B[2]]))
    finally:
        sh.close()
    sh = S.ShardedSpMM([0], np.zeros(5, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0), 3)
    try:
        assert np.array_equal(sh.run(B), np.zeros((4, 4)))
    finally:
        sh.close()


def test_errors(gpu):
    p = np.array([0, 1], dtype=np.int32)
    with pytest.raises(_lib.MxError, match="device 99"):
        S.ShardedSpMM([99], p, np.zeros(1, dtype=np.int32), np.ones(1), 1)
    sh = S.ShardedSpMM([0], p, np.zeros(1, dtype=np.int32), np.ones(1), 1)
    try:
        with pytest.raises(_lib.MxError, match="no product has run"):
            sh.result_ptr(0)
    finally:
        sh.close()


def test_bench_single_process_line(gpu):
    """`python bench.py --gpus 1 --single-process`: the metric's line from the C-ABI's sharded product (one process, one-rank
    communicator here), small custom shape, parity check against the oracle inside."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--single-process", "--steps", "4", "--warmup", "1",
                        "--rows", "131072", "--cols", "30000", "--nnz-row", "24", "--n", "128", "--no-cpu-baseline"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    assert len(last.encode()) < 6000
    d = json.loads(last)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["parallelism"] == "rowshard1-single-process"
    assert d["single_process"]["uses_rccl"] is True and d["single_process"]["rccl_version"] > 0
    assert d["parity_max_err_over_max_abs_vs_oracle"] <= 1e-10 and d["config"]["layout"] == "rowmajor"
