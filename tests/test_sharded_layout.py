"""Slot / cut arithmetic of the single-process sharded SpMM (csrc/sharded.hip mx_spmm_sharded_layout) — host only, no GPU.

Block r = rows [cuts[r], cuts[r+1]) of C lives at the head of slot r of the gathered buffer; every slot holds slot_rows rows
(equal slots are what one in-place ncclAllGather moves).  Nnz-balanced cuts are mx_partition_rows' (the sharded exports',
tests/test_gpu_cfg5_full.py); equal_rows makes the slots the blocks, so the gathered buffer is the contiguous matrix."""
import numpy as np
import pytest

from matrixextra_amd import sharded as S


def _indptr(lens):
    p = np.zeros(len(lens) + 1, dtype=np.int32)
    p[1:] = np.cumsum(lens)
    return p


@pytest.mark.parametrize("m,ndev", [(0, 1), (1, 1), (1, 4), (5, 8), (64, 2), (1000, 3), (100_000, 8), (100_001, 7)])
@pytest.mark.parametrize("equal_rows", [False, True])
def test_cuts_cover_every_row_once_and_fit_their_slots(m, ndev, equal_rows):
    rng = np.random.default_rng(m + ndev)
    lens = rng.integers(0, 40, size=m)
    if m > 10:
        lens[rng.integers(0, m, size=3)] = 5000                      # a few giant rows move the nnz-balanced cuts
    cuts, slot = S.layout(_indptr(lens), ndev, equal_rows=equal_rows)
    assert len(cuts) == ndev + 1 and cuts[0] == 0 and cuts[-1] == m
    assert all(a <= b for a, b in zip(cuts, cuts[1:]))
    assert max([b - a for a, b in zip(cuts, cuts[1:])], default=0) <= slot
    if equal_rows:
        per = -(-m // ndev) if m else 0
        assert slot == per and cuts == [min(m, per * r) for r in range(ndev + 1)]
        # the gathered buffer is the matrix: global row i sits at buffer row i
        for r in range(ndev):
            assert cuts[r] == min(m, r * slot)
    else:
        assert slot % 64 == 0 or m == 0


def test_nnz_balanced_cuts_follow_the_entries_not_the_rows():
    m = 80_000
    lens = np.full(m, 4, dtype=np.int64)
    lens[:10_000] = 400                                               # the first eighth of the rows holds most entries
    p = _indptr(lens)
    cuts, slot = S.layout(p, 4)
    nnz = [int(p[b] - p[a]) for a, b in zip(cuts, cuts[1:])]
    assert max(nnz) < 1.5 * (p[-1] / 4)
    assert cuts[1] < m // 4                                           # far fewer rows in the heavy block
    eq, _ = S.layout(p, 4, equal_rows=True)
    nnz_eq = [int(p[b] - p[a]) for a, b in zip(eq, eq[1:])]
    assert max(nnz_eq) > 2.5 * (p[-1] / 4)                            # what equal rows would have given the first device
