"""world_size-2 (and 3, ragged) gloo process groups on CPU: the row partition + all-gather assembly of the
sharded SpMM.  The local compute step is the CPU oracle here (test infrastructure); on the GPU box the same
RowShardedSpMM drives matrixextra_amd.device.spmm over RCCL (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, m, K, n, balanced, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from matrixextra_amd import distributed as MD, synth
        from oracle import oracle as O
        if balanced:
            p, j, x = synth.csr_fixed(m, K, 8, seed=1)
        else:
            p, j, x = synth.csr_skewed(m, K, 6, seed=2)
        B = synth.dense_normal(K, n)
        blocks = MD.nnz_balanced_row_blocks(p, world) if not balanced else \
            [(r * (m // world), (r + 1) * (m // world)) for r in range(world)]
        r0, r1 = blocks[rank]
        lp, lj, lx = MD.shard_csr(p, j, x, r0, r1)

        def spmm_local(local, Bt, out):
            pp, jj, xx = local
            rows = pp.size - 1
            if rows == 0:
                return
            res = np.zeros(rows * n)
            O.gemm_csr_drm_as_drm(rows, n, pp, np.ascontiguousarray(jj), np.ascontiguousarray(xx),
                                  Bt.numpy().reshape(-1), n, res, n, 1, False)
            out.copy_(torch.from_numpy(res.reshape(rows, n)))

        op = MD.RowShardedSpMM((lp, lj, lx), blocks, spmm_local)
        C = op(torch.from_numpy(B)).numpy()
        full = np.zeros(m * n)
        O.gemm_csr_drm_as_drm(m, n, p, j, x, B.reshape(-1), n, full, n, 1, False)
        ok = np.array_equal(C, full.reshape(m, n))
        # a stream of products with the all-gather of one running under the next: every product must come out right —
        # equal blocks (the gathered buffer IS C) and ragged ones (one slot of max(rows) rows per rank, assembled on request)
        pipe = MD.PipelinedRowShardedSpMM(op, n, torch.float64, "cpu")
        assert pipe.bufs[0].shape[0] == world * max(b - a for a, b in blocks)
        for k in range(5):
            Bk = torch.from_numpy(B * 2.0 ** k)                  # powers of two: the products scale exactly
            i = pipe.step(Bk)
            if k >= 1:                                           # the other buffer holds product k - 1 once waited for
                pipe._wait(1 - i)
                ok = ok and np.array_equal(pipe.assemble(pipe.bufs[1 - i]).numpy(), full.reshape(m, n) * 2.0 ** (k - 1))
        last = pipe.finish()
        ok = ok and np.array_equal(pipe.assemble(last).numpy(), full.reshape(m, n) * 2.0 ** 4)
        for rk, (a, b) in enumerate(blocks):                     # every rank's rows sit at the start of its slot
            ok = ok and np.array_equal(pipe.block(last, rk).numpy(), full.reshape(m, n)[a:b] * 2.0 ** 4)
        q.put((rank, bool(ok), blocks))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,balanced,m", [(2, True, 400), (2, False, 301), (3, False, 250)])
def test_row_sharded_spmm_gloo(world, balanced, m):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, 120, 16, balanced, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    results = [q.get(timeout=10) for _ in range(world)]
    assert all(ok for _, ok, _ in results)
    blocks = results[0][2]
    assert blocks[0][0] == 0 and blocks[-1][1] == m and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))


def test_nnz_balanced_blocks_properties():
    from matrixextra_amd.distributed import nnz_balanced_row_blocks, shard_csr
    from matrixextra_amd import synth
    p, j, x = synth.csr_skewed(2000, 500, 10, seed=7)
    for world in (1, 2, 3, 8):
        blocks = nnz_balanced_row_blocks(p, world)
        assert len(blocks) == world and blocks[0][0] == 0 and blocks[-1][1] == 2000
        nnz = [int(p[b] - p[a]) for a, b in blocks]
        assert sum(nnz) == int(p[-1])
        assert max(nnz) - min(nnz) <= 2 * int(np.diff(p).max())          # balanced up to a couple of rows
        parts = [shard_csr(p, j, x, a, b) for a, b in blocks]
        assert np.array_equal(np.concatenate([q[1] for q in parts]), j)
        assert all(q[0][0] == 0 and q[0][-1] == q[1].size for q in parts)
    empty = nnz_balanced_row_blocks(np.zeros(5, dtype=np.int32), 4)
    assert empty[0][0] == 0 and empty[-1][1] == 4
