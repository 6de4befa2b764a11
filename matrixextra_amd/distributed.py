"""Row-sharded SpMM across the GPUs of one node (new design — the reference is single-process).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  A is cut into contiguous row
blocks balanced by nonzeros, B is replicated, every rank computes its block of C with the HIP SpMM kernel and
the blocks are exchanged with ONE all-gather so that every rank holds the full C.  Row blocks of a
row-major C are contiguous, so the gathered buffer *is* C (row-major, gemm_csr_drm_as_drm layout); the
column-major matrix R needs is produced by the strided D2H copy at the boundary.

xGMI is point-to-point (7 links per GPU): an all-gather of s bytes per rank moves (N-1)*s bytes INTO every
GPU; it is the dominant cost of the sharded product (C is ~10x larger than A+B for the headline shapes).

The compute step is injected (`spmm_local`) so that the partition / exchange logic can be exercised with
world_size-2 gloo process groups on CPU (tests/test_distributed.py) — the product path passes
matrixextra_amd.device.spmm, there is no CPU implementation in this package.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Sequence, Tuple

import os

import numpy as np
import torch
import torch.distributed as dist

# MXGPU_DIST_ALWAYS_COLLECTIVE=1: issue the all-gather even in a one-rank group (a one-GPU box then exercises the RCCL
# calls, in place and asynchronous, exactly as N ranks would: tests/test_gpu_rccl.py)
ALWAYS_COLLECTIVE = os.environ.get("MXGPU_DIST_ALWAYS_COLLECTIVE") == "1"


def nnz_balanced_row_blocks(indptr: np.ndarray, world: int) -> List[Tuple[int, int]]:
    """Contiguous row ranges [r0, r1) per rank with (almost) equal numbers of nonzeros.
    Every rank gets a (possibly empty) range; ranges tile [0, m) in rank order."""
    indptr = np.asarray(indptr)
    m = indptr.size - 1
    nnz = int(indptr[m])
    cuts = [0]
    for r in range(1, world):
        target = nnz * r // world
        c = int(np.searchsorted(indptr, target, side="left"))
        cuts.append(min(max(c, cuts[-1]), m))
    cuts.append(m)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_csr(indptr, indices, values, r0: int, r1: int):
    """Rows [r0, r1) as a self-contained CSR (indptr rebased to 0). Views where possible."""
    indptr = np.asarray(indptr)
    s, e = int(indptr[r0]), int(indptr[r1])
    p = (indptr[r0:r1 + 1] - indptr[r0]).astype(np.int32)
    return p, np.asarray(indices)[s:e], None if values is None else np.asarray(values)[s:e]


@dataclass
class RowShardedSpMM:
    """Per-rank handle: my row block of A (any object `spmm_local` understands) + the global row layout."""
    local_A: object
    row_blocks: Sequence[Tuple[int, int]]
    spmm_local: Callable          # (local_A, B, out) -> None, writes the (rows_local x n) row-major block into `out`
    group: object = None

    def __post_init__(self):
        self.world = dist.get_world_size(self.group)
        self.rank = dist.get_rank(self.group)
        assert len(self.row_blocks) == self.world
        self.rows = [b - a for a, b in self.row_blocks]
        self.max_rows = max(self.rows) if self.rows else 0
        self.equal = all(r == self.max_rows for r in self.rows)

    def __call__(self, B: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """Returns the full C (sum(rows) x n, row-major) on every rank."""
        n = int(B.shape[1])
        m_total = sum(self.rows)
        if out is None:
            out = torch.empty((m_total, n), dtype=B.dtype, device=B.device)
        r0, r1 = self.row_blocks[self.rank]
        if self.equal:
            # equal blocks: compute straight into my slot of the gathered buffer, one all-gather in place
            mine = out[r0:r1]
            self.spmm_local(self.local_A, B, mine)
            if self.world > 1 or ALWAYS_COLLECTIVE:
                dist.all_gather_into_tensor(out, mine, group=self.group) if _has_into_tensor(B) else \
                    _all_gather_list(out, mine, self.rows, self.group)
            return out
        # ragged blocks: gather padded blocks, then compact
        pad = torch.zeros((self.max_rows, n), dtype=B.dtype, device=B.device)
        self.spmm_local(self.local_A, B, pad[: r1 - r0])
        bufs = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(bufs, pad, group=self.group)
        for rk, (a, b) in enumerate(self.row_blocks):
            out[a:b] = bufs[rk][: b - a]
        return out


class PipelinedRowShardedSpMM:
    """A stream of products with the same A: the all-gather of product k runs while product k + 1 is computed.

    Two gathered buffers alternate; `step(B)` computes this rank's block straight into its slot of the next buffer and
    starts the all-gather asynchronously (RCCL on its own stream; on xGMI the exchange is ~10x longer than the product,
    so hiding the product behind it is all there is to gain).  A buffer is handed out again only after its previous
    all-gather has completed.  `finish()` waits for everything and returns the most recent gathered buffer.

    Row blocks may be RAGGED (nnz_balanced_row_blocks of any matrix, any world size): the gathered buffer then has one
    SLOT of max(rows) rows per rank — rank r's block is the first rows[r] rows of slot r — so that the exchange stays ONE
    in-place all-gather of equal pieces (the padding rows, at most the imbalance of the cut, travel with it).  With equal
    blocks the slots are the blocks and the buffer IS C; otherwise `block(buf, r)` is rank r's rows and `assemble(buf)`
    the contiguous C (a copy)."""

    def __init__(self, sharded: RowShardedSpMM, n: int, dtype, device):
        self.s = sharded
        self.slot = sharded.max_rows
        self.bufs = [torch.empty((sharded.world * self.slot, n), dtype=dtype, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.k = 0

    def block(self, buf: torch.Tensor, r: int) -> torch.Tensor:
        return buf[r * self.slot: r * self.slot + self.s.rows[r]]

    def assemble(self, buf: torch.Tensor) -> torch.Tensor:
        return buf if self.s.equal else torch.cat([self.block(buf, r) for r in range(self.s.world)], dim=0)

    def _wait(self, i):
        w = self.pending[i]
        if w is not None:
            w.wait()                     # NCCL: the current stream waits, the host does not; gloo: blocks the host
            if not self.bufs[i].is_cuda:
                _copy_back(self.bufs[i], self._chunks[i], [self.slot] * self.s.world)
            self.pending[i] = None

    def step(self, B: torch.Tensor) -> int:
        i = self.k % 2
        self._wait(i)
        out = self.bufs[i]
        rk = self.s.rank
        mine = out[rk * self.slot:(rk + 1) * self.slot]              # my whole slot travels
        self.s.spmm_local(self.s.local_A, B, mine[: self.s.rows[rk]])
        if self.s.world > 1 or ALWAYS_COLLECTIVE:
            if out.is_cuda:
                self.pending[i] = dist.all_gather_into_tensor(out, mine, group=self.s.group, async_op=True)
            else:
                if not hasattr(self, "_chunks"):
                    self._chunks = [None, None]
                self._chunks[i] = list(torch.split(out, self.slot, dim=0))
                self.pending[i] = dist.all_gather(self._chunks[i], mine.clone(), group=self.s.group, async_op=True)
        self.k += 1
        return i

    def finish(self) -> torch.Tensor:
        for i in (0, 1):
            self._wait(i)
        return self.bufs[(self.k - 1) % 2]


def _copy_back(out, chunks, rows):
    off = 0
    for c, r in zip(chunks, rows):
        if c.data_ptr() != out[off:off + r].data_ptr():
            out[off:off + r] = c
        off += r


def _has_into_tensor(t: torch.Tensor) -> bool:
    return t.is_cuda          # gloo has no all_gather_into_tensor on CPU tensors in every build: use the list form there


def _all_gather_list(out, mine, rows, group):
    chunks = list(torch.split(out, rows, dim=0))
    dist.all_gather(chunks, mine.contiguous(), group=group)
    # torch.split gives views of `out`, but gloo may write into temporaries: copy back if needed
    off = 0
    for c, r in zip(chunks, rows):
        if c.data_ptr() != out[off:off + r].data_ptr():
            out[off:off + r] = c
        off += r


def gpu_spmm_local(algo: int = 0):
    """The product compute step: HIP SpMM on a DeviceCSR, row-major block written in place."""
    from . import device as D

    def run(local_A, B, out):
        if local_A.m:
            D.spmm(local_A, B, out=out, colmajor=False, algo=algo)
    return run
