"""Deterministic synthetic workloads of BASELINE.json / SURVEY.md §8(d).

numpy Generator(PCG64(seed)); fixed nnz per row, distinct column ids per row,
sorted ascending, approximately uniform over [0, K); values ~ U(-1, 1) f64;
dense operands ~ N(0, 1).  Seeds: A=1, B=2, rows=3, second merge operand=4.
Used by bench.py and the tests; generates inputs only (no arithmetic of the path).
"""
from __future__ import annotations

import numpy as np

SEED_A, SEED_B, SEED_ROWS, SEED_A2 = 1, 2, 3, 4


def _distinct_sorted_columns(rng, m, K, k, chunk=1 << 18):
    """(m, k) int32, each row strictly increasing in [0, K): sorted draws from [0, K-k] plus 0..k-1."""
    if k > K:
        raise ValueError("nnz per row exceeds the number of columns")
    out = np.empty((m, k), dtype=np.int32)
    off = np.arange(k, dtype=np.int32)
    for s in range(0, m, chunk):
        e = min(m, s + chunk)
        c = rng.integers(0, K - k + 1, size=(e - s, k), dtype=np.int32)
        c.sort(axis=1)
        out[s:e] = c + off
    return out


def csr_fixed(m, K, nnz_row, seed=SEED_A):
    """CSR with exactly nnz_row entries per row. Returns (indptr int32[m+1], indices int32, values f64)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    cols = _distinct_sorted_columns(rng, m, K, nnz_row)
    indptr = (np.arange(m + 1, dtype=np.int64) * nnz_row).astype(np.int32)
    values = rng.uniform(-1.0, 1.0, size=m * nnz_row)
    return indptr, cols.reshape(-1), values


def csr_skewed(m, K, mean_nnz, seed=SEED_A, sigma=1.0, max_nnz=None):
    """Log-normal row lengths with the given mean (the 'skewed' variant of §8d); some rows empty."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    mu = np.log(mean_nnz) - 0.5 * sigma * sigma
    lens = np.floor(rng.lognormal(mu, sigma, size=m)).astype(np.int64)
    lens = np.minimum(lens, K if max_nnz is None else min(K, max_nnz))
    indptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr[1:])
    indices = np.empty(indptr[-1], dtype=np.int32)
    for r in range(m):
        n = lens[r]
        if n:
            indices[indptr[r]:indptr[r + 1]] = np.sort(rng.choice(K, size=n, replace=False))
    values = rng.uniform(-1.0, 1.0, size=indptr[-1])
    return indptr.astype(np.int32), indices, values


def csr_skewed_fast(m, K, mean_nnz, seed=SEED_A, sigma=1.0):
    """The skewed variant of §8d at full size, vectorised (csr_skewed draws row by row: minutes for 1M rows): log-normal row
    lengths with the given mean (floor; clipped to K), column ids uniform over [0, K), sorted per row, duplicates inside a
    row dropped (so a few rows are one or two entries shorter than drawn), values ~ U(-1, 1)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    mu = np.log(mean_nnz) - 0.5 * sigma * sigma
    lens = np.minimum(np.floor(rng.lognormal(mu, sigma, size=m)).astype(np.int64), K)
    total = int(lens.sum())
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = row * np.int64(K) + rng.integers(0, K, size=total, dtype=np.int64)
    key.sort()                                                        # by row, then by column
    keep = np.ones(total, dtype=bool)
    keep[1:] = key[1:] != key[:-1]
    key = key[keep]
    row = key // K
    indices = (key - row * K).astype(np.int32)
    indptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=m), out=indptr[1:])
    values = rng.uniform(-1.0, 1.0, size=indices.size)
    return indptr.astype(np.int32), indices, values


def zipf_column_cdf(K, alpha, seed):
    """Column popularity of real sparse data (bag-of-words / one-hot features; the vignette's own application is LibSVM
    real-sim, vignettes/Introducing_MatrixExtra.Rmd:442-502): P(column of popularity rank r) ~ 1 / (r + 1)^alpha, the ranks
    dealt to column ids by a seeded permutation (hot columns sit anywhere in [0, K)).  Returns (cdf over ranks float64[K],
    rank -> column id int32[K])."""
    w = 1.0 / np.power(np.arange(1, K + 1, dtype=np.float64), alpha)
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    perm = np.random.default_rng(np.random.PCG64(seed + 7919)).permutation(K).astype(np.int32)
    return cdf, perm


def csr_zipf(m, K, mean_nnz, alpha=1.0, sigma=1.0, seed=SEED_A):
    """Realistic dgRMatrix contents: power-law COLUMNS (zipf_column_cdf: exponent alpha) and log-normal ROW lengths (mean
    mean_nnz, shape sigma; sigma = 0: every row draws mean_nnz ids).  Column ids drawn with replacement, sorted per row,
    duplicates inside a row dropped — hot columns collide, so rows come out a little shorter than drawn — values ~ U(-1, 1).
    real-sim's shape: csr_zipf(72_309, 20_958, 51)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    if sigma > 0:
        mu = np.log(mean_nnz) - 0.5 * sigma * sigma
        lens = np.minimum(np.floor(rng.lognormal(mu, sigma, size=m)).astype(np.int64), K)
    else:
        lens = np.full(m, min(int(mean_nnz), K), dtype=np.int64)
    total = int(lens.sum())
    cdf, perm = zipf_column_cdf(K, alpha, seed)
    cols = perm[np.minimum(np.searchsorted(cdf, rng.random(total), side="left"), K - 1)].astype(np.int64)
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = row * np.int64(K) + cols
    key.sort()                                                        # by row, then by column
    keep = np.ones(total, dtype=bool)
    keep[1:] = key[1:] != key[:-1]
    key = key[keep]
    row = key // K
    indices = (key - row * K).astype(np.int32)
    indptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(row, minlength=m), out=indptr[1:])
    values = rng.uniform(-1.0, 1.0, size=indices.size)
    return indptr.astype(np.int32), indices, values


def dense_normal(rows, cols, seed=SEED_B, dtype=np.float64, order="C"):
    rng = np.random.default_rng(np.random.PCG64(seed))
    a = rng.standard_normal(size=(rows, cols))
    return np.asarray(a, dtype=dtype, order=order)


def rows_with_replacement(r, m, seed=SEED_ROWS):
    """cfg3: r draws with replacement from [0, m), unsorted (0-based, as copy_csr_rows receives them)."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    return rng.integers(0, m, size=r, dtype=np.int32)


def csr_overlapping(indptr, indices, K, nnz_row, share=0.5, seed=SEED_A2):
    """cfg4 second operand: per row keep ~share of A's columns and draw the rest fresh, so that both
    the coincident and the one-sided branches of the merge run.  A must have nnz_row entries per row."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    m = indptr.size - 1
    a_cols = indices.reshape(m, nnz_row)
    keep = int(round(nnz_row * share))
    fresh = _distinct_sorted_columns(rng, m, K, nnz_row)
    # first `keep` columns from a random subset of A's columns, the rest fresh; then sort + dedupe per row
    pick = np.argsort(rng.random((m, nnz_row), dtype=np.float32), axis=1)[:, :keep]
    shared = np.take_along_axis(a_cols, pick.astype(np.int64), axis=1)
    cols = np.concatenate([shared, fresh[:, : nnz_row - keep]], axis=1)
    cols.sort(axis=1)
    dup = np.zeros_like(cols, dtype=bool)
    dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
    lens = (~dup).sum(axis=1)
    indptr2 = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=indptr2[1:])
    indices2 = cols[~dup]
    values2 = rng.uniform(-1.0, 1.0, size=indices2.size)
    return indptr2.astype(np.int32), indices2.astype(np.int32), values2


# ---- the same generators on the device (torch RNG, seeded): bench.py's cfg4 operands are 2 x 1e8 entries, which
# numpy draws and sorts in ~45 s on one host core and torch on the GPU in well under a second.  Inputs only.
def device_csr_fixed(m, K, nnz_row, seed=SEED_A, device="cuda"):
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    c = torch.randint(0, K - nnz_row + 1, (m, nnz_row), dtype=torch.int32, device=device, generator=g)
    c = torch.sort(c, dim=1).values
    c += torch.arange(nnz_row, dtype=torch.int32, device=device)
    indptr = (torch.arange(m + 1, dtype=torch.int64, device=device) * nnz_row).to(torch.int32)
    values = torch.rand(m * nnz_row, dtype=torch.float64, device=device, generator=g) * 2.0 - 1.0
    return indptr, c.reshape(-1).contiguous(), values


def device_csr_zipf(m, K, mean_nnz, alpha=1.0, sigma=1.0, seed=SEED_A, device="cuda"):
    """Device twin of csr_zipf (torch RNG: the same distribution, not the same draws): power-law columns, log-normal row
    lengths, sorted rows, duplicates dropped.  Returns (indptr int32, indices int32, values f64) on the device."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if sigma > 0:
        mu = float(np.log(mean_nnz) - 0.5 * sigma * sigma)
        lens = torch.empty(m, dtype=torch.float64, device=device).log_normal_(mu, sigma, generator=g).floor_().clamp_(max=K).to(torch.int64)
    else:
        lens = torch.full((m,), min(int(mean_nnz), K), dtype=torch.int64, device=device)
    total = int(lens.sum().item())
    cdf_h, perm_h = zipf_column_cdf(K, alpha, seed)
    cdf, perm = torch.from_numpy(cdf_h).to(device), torch.from_numpy(perm_h).to(device)
    u = torch.rand(total, dtype=torch.float64, device=device, generator=g)
    cols = perm[torch.searchsorted(cdf, u).clamp_(max=K - 1)].to(torch.int64)
    del u
    row = torch.repeat_interleave(torch.arange(m, dtype=torch.int64, device=device), lens)
    key = torch.sort(row * K + cols).values
    del row, cols
    keep = torch.ones(total, dtype=torch.bool, device=device)
    keep[1:] = key[1:] != key[:-1]
    key = key[keep]
    row = key // K
    indices = (key - row * K).to(torch.int32).contiguous()
    indptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.bincount(row, minlength=m), 0, out=indptr[1:])
    values = torch.rand(indices.numel(), dtype=torch.float64, device=device, generator=g) * 2.0 - 1.0
    return indptr.to(torch.int32), indices, values


def device_csr_overlapping(indices, m, K, nnz_row, share=0.5, seed=SEED_A2, device="cuda"):
    """Device twin of csr_overlapping: ~share of A's columns per row plus fresh draws, sorted, duplicates dropped."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    keep = int(round(nnz_row * share))
    a_cols = indices.reshape(m, nnz_row)
    pick = torch.rand((m, nnz_row), dtype=torch.float32, device=device, generator=g).argsort(dim=1)[:, :keep]
    shared = torch.gather(a_cols, 1, pick)
    fresh = torch.randint(0, K - nnz_row + 1, (m, nnz_row), dtype=torch.int32, device=device, generator=g)
    fresh = torch.sort(fresh, dim=1).values + torch.arange(nnz_row, dtype=torch.int32, device=device)
    cols = torch.sort(torch.cat([shared, fresh[:, : nnz_row - keep]], dim=1), dim=1).values
    dup = torch.zeros_like(cols, dtype=torch.bool)
    dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
    lens = (~dup).sum(dim=1)
    indptr = torch.zeros(m + 1, dtype=torch.int64, device=device)
    torch.cumsum(lens, 0, out=indptr[1:])
    ind2 = cols[~dup].contiguous()
    values = torch.rand(ind2.numel(), dtype=torch.float64, device=device, generator=g) * 2.0 - 1.0
    return indptr.to(torch.int32), ind2, values


def spmm_algorithmic_bytes(m, K, n, nnz, s_dense):
    """4(m+1) + 4 nnz + 8 nnz + s*K*n + s*m*n."""
    return 4 * (m + 1) + 12 * nnz + s_dense * K * n + s_dense * m * n
