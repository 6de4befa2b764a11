#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X backend (BASELINE.json metric).

Workloads (config.workload):
  cfg2  BASELINE configs[1] — dgRMatrix 1M x 100k, 32 nnz/row (nnz 32M), f64, %*% dense 100k x 128.  The metric's
        configuration: the default for EVERY N.  At N > 1 it is the same matrix (same seed, same bytes) cut into N
        nnz-balanced row blocks (strong scaling): `python bench.py --gpus 1/2/4/8` is one curve.
  cfg5  BASELINE configs[4] — 8M x 200k, 64 nnz/row (CSR values f64), f32 dense 200k x 256.  At N > 1 the whole matrix,
        strong-scaled, rides on the same line as `extras.cfg5_strong`; at N = 1 one GPU's 1M-row block is timed briefly
        (`extras.cfg5_shard`) and `--config cfg5-full` runs the whole matrix on one GPU.
One "step" = one SpMM over the rank's whole matrix, inputs resident in HBM.  `value` (--algo 0): the plan AUTO uses — a
regrouping of the CSR's entries that depends on the matrix alone — is kept on the DeviceCSR and built once per matrix
(untimed, like rows_sorted()); `plan.rebuild_every_step` in the line is the same loop with the plan rebuilt from plain CSR
inside every step (--algo 4 times that form as `value`).
N GPUs (one rank per GPU; `python bench.py --gpus N` starts torch.distributed.run itself when it was not started by
it): every rank owns one nnz-balanced row block of the matrix, B replicated, blocks of C exchanged with one RCCL
all-gather so that every rank holds the full C.  value = total GFLOP/s of the whole product (2*nnz*n flops per step,
max-over-ranks time, all-gather included); `compute_only_gflops` = the same flops over the slowest rank's local product;
`cpu_baseline` (rank 0's host cores, a bounded sample) and rank 0's `roofline` are carried at every N.
`--scaling weak` is the other experiment: every rank owns a --rows row block (the matrix grows with N).

At N = 1 the line also carries (`extras`) configs[2] (SpMV + gather of 200k rows), configs[3] (CSR + CSR, CSR * CSR
on 2M x 2M, nnz 1e8 each) and one export-level call from host memory, each with its own roofline and CPU baseline.

Prints ONE JSON line on rank 0; see DESIGN.md §Measurement for how roofline / cpu_baseline are defined.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SETUP_CALLS = 12               # untimed products before the warm-up: workspace allocation + clock ramp (see main)

FORCE_DIST = os.environ.get("MXGPU_BENCH_FORCE_DIST") == "1"

WORKLOADS = {
    "cfg2": dict(rows=1_000_000, cols=100_000, nnz_row=32, n=128, dtype="f64",
                 label="BASELINE configs[1]"),
    "cfg5": dict(rows=1_000_000, cols=200_000, nnz_row=64, n=256, dtype="f32",
                 label="BASELINE configs[4], one GPU's row block of the 8M x 200k matrix"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(WORKLOADS) + ["cfg5-full"],
                    help="workload; default cfg2 (the metric's configuration) for every N")
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--cols", type=int, default=None)
    ap.add_argument("--nnz-row", type=int, default=None)
    ap.add_argument("--n", type=int, default=None, help="dense columns")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"])
    ap.add_argument("--layout", default="colmajor", choices=["colmajor", "rowmajor"],
                    help="C layout at N=1 (colmajor = what tcrossprod_csr_dense returns to R)")
    ap.add_argument("--algo", type=int, default=0,
                    help="0 auto, 1 row-wave kernel, 2 slab/panel kernel, 3 planned kernel (plan cached), "
                         "4 planned kernel with the plan rebuilt inside every timed step")
    ap.add_argument("--sync", type=int, default=-1, help="planned kernel: 0 no barrier, 1 per row block, 2 per panel")
    ap.add_argument("--panels", type=int, default=0)
    ap.add_argument("--wg-per-cu", type=int, default=0)
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="N > 1: strong (default) = --rows is the WHOLE matrix (cfg2: the N = 1 matrix itself; cfg5: configs[4]'s "
                         "8M rows), cut into N nnz-balanced row blocks; weak = every rank owns a --rows row block")
    ap.add_argument("--cfg5-strong-rows", type=int, default=8_000_000,
                    help="rows of the configs[4] matrix of `extras.cfg5_strong` (N > 1; a multiple of 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget for the SpMM CPU baseline sample")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the SpMV / gather / merge / export-call / cfg5-shard legs (N = 1 only)")
    ap.add_argument("--extras", action="store_true", help="(default at N = 1; kept for older command lines)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through the C-ABI (mx_spmm_sharded_*: one host thread per device, ncclCommInitAll, "
                         "one in-place ncclAllGather per product) instead of one torch.distributed rank per GPU")
    args = ap.parse_args()
    args.config_given = args.config is not None
    if args.config is None:
        args.config = "cfg2"
    if args.scaling is None:
        args.scaling = "strong" if (args.gpus > 1 or FORCE_DIST) else "weak"      # (one GPU: nothing to scale; reported as weak)
    w = WORKLOADS["cfg5" if args.config == "cfg5-full" else args.config]
    args.custom = any(v is not None for v in (args.rows, args.cols, args.nnz_row, args.n, args.dtype))
    for k in ("rows", "cols", "nnz_row", "n", "dtype"):
        if getattr(args, k) is None:
            setattr(args, k, w[k])
    if args.scaling == "strong" and not args.custom and args.config == "cfg5":
        args.rows = 8_000_000                                       # BASELINE configs[4] whole: 8M x 200k, 64 / row
    return args


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` (N > 1) without torch.distributed.run around it: start it as a child process — before
    anything here touches the GPU — and leave with its return code.  Never falls back to one GPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)

from bench_common import (HBM_PEAK_GBS, STREAM, committed_kernels_traffic, committed_traffic, cpu_baseline_spmm,  # noqa: E402,F401
                          bound_unit_of, headline_line, roofline, stream_copy_probe)


STRONG_CHUNKS = 64


def strong_rows(synth, torch, m_total, K, nnz_row, r0, r1):
    """Rows [r0, r1) of the --scaling strong matrix, on the device.  The matrix is the same whatever N is: 64 row chunks,
    chunk c drawn on the device from seed SEED_A + c (synth.device_csr_fixed); a rank draws the chunks its block touches."""
    assert m_total % STRONG_CHUNKS == 0, "--scaling strong: --rows must be a multiple of 64"
    cr = m_total // STRONG_CHUNKS
    js, xs = [], []
    for c in range(r0 // cr, (max(r1, r0 + 1) - 1) // cr + 1):
        _, j, x = synth.device_csr_fixed(cr, K, nnz_row, seed=synth.SEED_A + c)
        a, b = max(r0, c * cr) - c * cr, min(r1, (c + 1) * cr) - c * cr
        js.append(j[a * nnz_row:b * nnz_row])
        xs.append(x[a * nnz_row:b * nnz_row])
    m = r1 - r0
    p = (torch.arange(m + 1, dtype=torch.int64, device="cuda") * nnz_row).to(torch.int32)
    return p, torch.cat(js).contiguous(), torch.cat(xs).contiguous()


# ------------------------------------------------------------------------------------------------ the SpMM leg
def spmm_leg(args, torch, dist, D, synth, lib, _lib, cfg, world, rank, steps, warmup, want_cpu, want_steady):
    """Times `steps` SpMM steps of workload `cfg` (dict rows/cols/nnz_row/n/dtype) on this rank (+ all-gather for
    dist_on).  Returns the result dict on rank 0 (None elsewhere)."""
    import ctypes
    # MXGPU_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, in-place RCCL all-gather, pipelined buffers,
    # gathered-buffer parity check) with whatever WORLD_SIZE is — lets a ONE-GPU box exercise it (tests/test_gpu_rccl.py)
    dist_on = world > 1 or FORCE_DIST
    m, K, n, nnz_row, dtype = cfg["rows"], cfg["cols"], cfg["n"], cfg["nnz_row"], cfg["dtype"]
    tdt = torch.float64 if dtype == "f64" else torch.float32
    ndt = np.float64 if dtype == "f64" else np.float32
    s_dense = 8 if dtype == "f64" else 4

    strong = dist_on and args.scaling == "strong"
    B_host = synth.dense_normal(K, n, dtype=ndt)
    if strong:
        # --rows is the WHOLE matrix; this rank owns one of `world` nnz-balanced row blocks (ragged when the cut is)
        from matrixextra_amd import distributed as MDs
        m_total = m
        blocks = MDs.nnz_balanced_row_blocks(np.arange(m_total + 1, dtype=np.int64) * nnz_row, world)
        r0g, r1g = blocks[rank]
        m = r1g - r0g
        host_whole = cfg["name"] == "cfg2" and not args.custom
        if host_whole:
            # the metric's matrix: the SAME bytes as the N = 1 run (synth.csr_fixed, seed A = 1), drawn by every rank and cut
            pw, jw, xw = synth.csr_fixed(m_total, K, nnz_row, seed=synth.SEED_A)
            p, j, x = MDs.shard_csr(pw, jw, xw, r0g, r1g)
            A = D.DeviceCSR.from_host(p, j, x, K)
            last_rows = MDs.shard_csr(pw, jw, xw, m_total - 2048, m_total) if rank == 0 else None
            del pw, jw, xw
        else:
            dp, dj, dx = strong_rows(synth, torch, m_total, K, nnz_row, r0g, r1g)
            A = D.DeviceCSR(dp, dj, dx, m, K, int(dj.numel()))
            p = j = x = None
    else:
        # synthetic inputs: seeds A=1 (+1000*rank for the other row blocks), B=2
        m_total = world * m
        blocks = [(r * m, (r + 1) * m) for r in range(world)]
        p, j, x = synth.csr_fixed(m, K, nnz_row, seed=synth.SEED_A + 1000 * rank)
        A = D.DeviceCSR.from_host(p, j, x, K)
    B = torch.from_numpy(B_host).cuda()
    nnz = A.nnz
    A.rows_sorted()                  # once per matrix, outside the timed region (cached on the DeviceCSR)
    colmajor = (args.layout == "colmajor") and not dist_on
    overlap = dist_on and os.environ.get("MXGPU_BENCH_OVERLAP", "1") != "0"
    C_full = C_loc = None
    if dist_on and not overlap:
        C_full = torch.full((m_total, n), float("nan"), dtype=tdt, device="cuda")       # gathered row-major blocks
        C_loc = C_full[blocks[rank][0]:blocks[rank][1]]                                  # compute straight into my slot
    elif not dist_on:
        C_loc = torch.empty((n, m) if colmajor else (m, n), dtype=tdt, device="cuda")

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)] \
        if dist_on else []

    keep_plan = [True]              # --algo 0: AUTO's plan is kept on the DeviceCSR (built in the untimed setup calls)

    def run_spmm(A_, B_, out_, colmajor_):
        if args.algo in (3, 4):
            D.spmm_planned(A_, B_, out=out_, colmajor=colmajor_, npanels=args.panels, wg_per_cu=args.wg_per_cu,
                           sync_mode=args.sync, rebuild_plan=(args.algo == 4))
        else:
            D.spmm(A_, B_, out=out_, colmajor=colmajor_, algo=args.algo, npanels=args.panels, wg_per_cu=args.wg_per_cu,
                   keep_plan=keep_plan[0])

    pipe = None
    cur = [None]
    if dist_on:
        # matrixextra_amd.distributed: equal row blocks -> compute into my slot, one RCCL all-gather in place
        from matrixextra_amd import distributed as MD

        def timed_local(local_A, Bt, out):
            k = cur[0]
            if k is not None:
                ev[k][0].record()
            run_spmm(local_A, Bt, out, False)
            if k is not None:
                ev[k][1].record()
        sharded = MD.RowShardedSpMM(A, blocks, timed_local)
        # the all-gather of step k runs under the product of step k + 1 (two gathered buffers alternate;
        # MXGPU_BENCH_OVERLAP=0 gathers in line instead).  Everything is complete before the timed region closes.
        if overlap:
            pipe = MD.PipelinedRowShardedSpMM(sharded, n, tdt, "cuda")
            for b in pipe.bufs:
                b.fill_(float("nan"))          # anything not written by a product or a completed gather stays NaN

    def step(k=None):
        if dist_on:
            cur[0] = k
            if pipe is not None:
                pipe.step(B)
            else:
                sharded(B, out=C_full)
            return
        run_spmm(A, B, C_loc, colmajor)     # no per-step events here: each one is a packet the queue drains between kernels

    # one-time setup, not steps: the library's grow-only workspaces (plan arrays, packed copy of B, pinned read-back
    # buffer, timing events) are allocated on first use, and after the idle seconds of input generation the GPU needs
    # ~10 products (25 ms) to reach its steady clocks (tools/ramp_probe.py: 2.32 -> 2.09 ms per call).  SETUP_CALLS
    # untimed calls here, reported in the JSON line, keep a short --warmup/--steps run from measuring that ramp.
    lib.mxd_spmm_kernel_timing(1)
    for _ in range(SETUP_CALLS):
        step()
    torch.cuda.synchronize()
    lib.mxd_spmm_kernel_timing(0)
    for _ in range(warmup):
        step()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    lib.mxd_spmm_kernel_timing(1)           # HIP events right around the dominant kernel of every launch
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    if pipe is not None:
        pipe.finish()                     # waits for the gathers still in flight
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # local SpMM call (plan build + repack + kernel) of every step; only recorded when N > 1 (beside the all-gather)
    step_ms = np.array([a.elapsed_time(b) for a, b in ev]) if dist_on else np.array([elapsed / steps * 1e3])
    kt = (ctypes.c_float * 256)()
    kcount = ctypes.c_int(0)
    _lib.check(lib.mxd_spmm_kernel_times(kt, 256, ctypes.byref(kcount)))
    lib.mxd_spmm_kernel_timing(0)
    kern_ms = np.array(kt[:kcount.value], dtype=np.float64) if kcount.value else step_ms
    kern_avg_s = float(kern_ms.mean()) / 1e3                         # dominant kernel only: the roofline figure
    flops_rank_step = 2.0 * nnz * n
    flops_all_step = 2.0 * m_total * nnz_row * n if strong else world * flops_rank_step
    alg_bytes = synth.spmm_algorithmic_bytes(m, K, n, nnz, s_dense)
    nranks_seen = None
    slowest_local_ms = float(step_ms.mean())
    if dist_on:
        # how many ranks really took part: an RCCL all-reduce of ones (beside dist.get_world_size(), for the driver's check)
        ones = torch.ones(1, dtype=torch.float32, device="cuda")
        dist.all_reduce(ones)
        nranks_seen = int(round(float(ones.item())))
        tl = torch.tensor([slowest_local_ms], dtype=torch.float64, device="cuda")     # the slowest rank's local product
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        slowest_local_ms = float(tl.item())
    if rank != 0:
        return None

    # parity of the timed output against the CPU restatement (checker only; not timed)
    from oracle import oracle as O
    rows_chk = 2048

    def oracle_rows(pp, jj, xx, r0):
        ref = np.zeros(rows_chk * n, dtype=ndt)
        lo, hi = int(pp[r0]), int(pp[r0 + rows_chk])
        O.gemm_csr_drm_as_drm(rows_chk, n, (pp[r0:r0 + rows_chk + 1] - pp[r0]).astype(np.int32), jj[lo:hi].copy(),
                              xx[lo:hi].copy(), B_host.reshape(-1), n, ref, n, O.max_threads(), True)
        return ref.reshape(rows_chk, n)

    def max_err(got, ref):
        # normalised max error: |got - ref| / max|ref| over the checked block (element-wise relative error is
        # meaningless for entries that cancel to ~0); NaN (unwritten / half-gathered data) propagates
        return float(np.max(np.abs(got.astype(np.float64) - ref)) / np.max(np.abs(ref)))
    if strong and p is None:                          # my block's first rows, read back from the device
        e = rows_chk * nnz_row
        p = (np.arange(rows_chk + 1, dtype=np.int64) * nnz_row).astype(np.int32)
        j, x = A.indices[:e].cpu().numpy(), A.values[:e].cpu().numpy()
    ref0 = oracle_rows(p, j, x, 0)
    if not dist_on:
        got = (C_loc[:, :rows_chk].t() if colmajor else C_loc[:rows_chk]).cpu().numpy()
        parity = max_err(got, ref0)
    else:
        # every gathered buffer, my own block AND the last rank's block (regenerated here from its seed): data that
        # never arrived, or arrived from an unfinished product, shows up as NaN or as a wrong value
        ml = blocks[-1][1] - blocks[-1][0]              # the last rank's block: its last rows_chk rows
        if strong and host_whole:
            refl = oracle_rows(*last_rows, 0)
        elif strong:
            _, jl_d, xl_d = strong_rows(synth, torch, m_total, K, nnz_row, m_total - rows_chk, m_total)
            pl = (np.arange(rows_chk + 1, dtype=np.int64) * nnz_row).astype(np.int32)
            refl = oracle_rows(pl, jl_d.cpu().numpy(), xl_d.cpu().numpy(), 0)
        else:
            pl, jl, xl = synth.csr_fixed(m, K, nnz_row, seed=synth.SEED_A + 1000 * (world - 1))
            refl = oracle_rows(pl, jl, xl, m - rows_chk)
        bufs = pipe.bufs if pipe is not None else [C_full]
        parity = 0.0
        for Cg in bufs:
            mine_blk = pipe.block(Cg, rank) if pipe is not None else Cg[blocks[rank][0]:blocks[rank][1]]
            last_blk = pipe.block(Cg, world - 1) if pipe is not None else Cg[blocks[-1][0]:blocks[-1][1]]
            parity = max(parity, max_err(mine_blk[:rows_chk].cpu().numpy(), ref0))
            parity = max(parity, max_err(last_blk[ml - rows_chk:ml].cpu().numpy(), refl))
            for r in range(world):                    # (padding rows of a ragged slot are never written: only the blocks are checked)
                blk = pipe.block(Cg, r) if pipe is not None else Cg[blocks[r][0]:blocks[r][1]]
                assert bool(torch.isfinite(blk).all()), "gathered C holds unwritten (NaN) entries"
    tol = 1e-10 if dtype == "f64" else 2e-5
    assert parity <= tol, f"bench output differs from the oracle: {parity}"

    kernel_name = lib.mxd_spmm_last_kernel().decode()      # which kernel AUTO / --algo actually launched
    tag = None
    if not args.custom and (args.layout, args.algo, args.panels, args.wg_per_cu) == ("colmajor", 0, 0, 0) and not dist_on:
        tag = {"cfg2": "cfg2-default", "cfg5": "cfg5-shard-default"}.get(cfg["name"])   # the `workload` string of profiles/*_pmc.json
    traffic = committed_traffic(kernel_name, tag, kern_avg_s * 1e3) if tag else (None, None, "not the profiled command line")
    gb = nnz * n * s_dense
    res = {
        "value": round(flops_all_step * steps / elapsed / 1e9, 2),
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "workload": f"dgRMatrix {m_total if strong else m}x{K} nnz/row={nnz_row} (CSR values f64) %*% dense {K}x{n} {dtype} "
                    f"({cfg['label']}); C {'col' if colmajor else 'row'}-major"
                    + ((f"; the WHOLE matrix, cut into {world} nnz-balanced row blocks (rank 0: {m} rows), one per GPU, + RCCL "
                        f"all-gather of C" if strong else f"; one such row block per GPU + RCCL all-gather of C ({world}x{m} rows)")
                       + (", gather of step k under the product of step k+1" if pipe is not None else "")
                       if dist_on else ""),
        "roofline": roofline(alg_bytes, kern_avg_s, traffic=traffic[0], traffic_source=traffic[1],
                             **({"traffic_refused": traffic[2]} if traffic[2] else {}), kernel=kernel_name,
                             kernel_avg_ms=round(kern_avg_s * 1e3, 4), kernel_min_ms=round(float(kern_ms.min()), 4),
                             **({"bound_unit": bound_unit_of(kernel_name)} if bound_unit_of(kernel_name) else {}),
                             # secondary, non-scoring: what actually bounds the kernel.  Every nonzero gathers one B
                             # row: nnz * n * s bytes move from L2 into the CUs' L1 whatever the schedule
                             l2_to_l1_gather={"bytes_per_launch": int(gb),
                                              "achieved_GBps": round(gb / kern_avg_s / 1e9, 0),
                                              "guide_ceiling_GBps": [16000, 22000]}),
        "kernel_gflops": round(flops_rank_step / kern_avg_s / 1e9, 1),
        "parity_max_err_over_max_abs_vs_oracle": parity,
        "spmm_call_avg_ms": round(float(step_ms.mean()), 4),
        "dims": {"rows_per_gpu": m, "cols": K, "nnz_per_row": nnz_row, "dense_cols": n,
                 "layout": "colmajor" if colmajor else "rowmajor"},
    }
    if dist_on:
        gather_s = max(elapsed / steps - float(step_ms.mean()) / 1e3, 1e-9) if pipe is None else elapsed / steps
        slot = max(b - a for a, b in blocks)
        res["allgather"] = {"bytes_received_per_gpu": int((world - 1) * slot * n * s_dense),
                            "approx_ms": round(gather_s * 1e3, 3),
                            "approx_GBps_in_per_gpu": round((world - 1) * slot * n * s_dense / gather_s / 1e9, 1)}
        res["distributed"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                              "ranks_in_an_rccl_all_reduce_of_ones": nranks_seen, "scaling": args.scaling,
                              "row_blocks": [b - a for a, b in blocks], "rows_total": m_total}
        # the same flops over the slowest rank's local product alone (no exchange): what the GPUs do; `value` is what the job does
        res["compute_only_gflops"] = round(flops_all_step / (slowest_local_ms / 1e3) / 1e9, 2)
        res["roofline"]["scope"] = "rank 0's row block, its dominant kernel only (value is the whole job, all-gather included)"
    if kernel_name == "spmm_plan_kernel" and args.algo in (0, 4):
        res["plan"] = {"value_is": "plan kept on the DeviceCSR: built once per matrix in the untimed setup calls, every timed "
                                   "step = repack of B + the sweep" if args.algo == 0 else
                                   "plan rebuilt from plain CSR inside every timed step (--algo 4)",
                       "info": A.plan_info() if A._plan is not None else None}
    if want_steady and kernel_name == "spmm_plan_kernel" and args.algo in (0, 4) and not dist_on:
        # the other form, same loop: --algo 0 -> the C-ABI's own AUTO (plan rebuilt from plain CSR inside every step: what a
        # caller pays who brings a new matrix every time); --algo 4 -> the kept plan
        other_keeps = args.algo == 4

        def other():
            if other_keeps:
                D.spmm_planned(A, B, out=C_loc, colmajor=colmajor)
            else:
                D.spmm(A, B, out=C_loc, colmajor=colmajor, keep_plan=False)
        for _ in range(3):
            other()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            other()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        res["plan"]["kept_plan" if other_keeps else "rebuild_every_step"] = {
            "ms_per_step": round(dt * 1e3, 4), "GFLOP/s": round(flops_rank_step / dt / 1e9, 1)}
    if want_steady and not dist_on and colmajor and args.algo == 0:
        # the same product with C row-major: the sweep every rank runs at N > 1 (row-major blocks are what the all-gather
        # moves), so that the 1 -> N ratio can be read against ONE layout (VERDICT r5 missing 3)
        C_rm = C_loc.view(-1)[: m * n].view(m, n)
        for _ in range(3):
            run_spmm(A, B, C_rm, False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps):
            run_spmm(A, B, C_rm, False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        res["rowmajor_ms_per_step"] = round(dt * 1e3, 4)
        res["rowmajor_gflops"] = round(flops_rank_step / dt / 1e9, 1)
    if want_cpu:
        if p is None:            # a device-drawn block: the CPU sample is its first rows, read back
            rs = min(m, 60_000)
            e = rs * nnz_row
            p_c = (np.arange(rs + 1, dtype=np.int64) * nnz_row).astype(np.int32)
            res["cpu_baseline"] = cpu_baseline_spmm(args, p_c, A.indices[:e].cpu().numpy(), A.values[:e].cpu().numpy(), B_host, dtype)
        else:
            res["cpu_baseline"] = cpu_baseline_spmm(args, p, j, x, B_host, dtype)
    res["_host"] = (p, j, x, A, B, B_host)          # handed to the extras; removed before printing
    return res


def single_process(args):
    """--single-process: the metric's product row-sharded over --gpus devices from THIS process through the C-ABI
    (include/mxgpu.h mx_spmm_sharded_*, csrc/sharded.hip): every step = the products of all row blocks, queued by one host
    thread per device, + one in-place ncclAllGather of C; the gather of step k runs under the product of step k + 1."""
    import torch
    from matrixextra_amd import _lib, sharded as S, synth
    from oracle import oracle as O
    lib = _lib.load()
    N, steps, warmup = args.gpus, args.steps, args.warmup
    m, K, n, nnz_row, dtype = args.rows, args.cols, args.n, args.nnz_row, args.dtype
    ndt = np.float64 if dtype == "f64" else np.float32
    p, j, x = synth.csr_fixed(m, K, nnz_row, seed=synth.SEED_A)
    B_host = synth.dense_normal(K, n, dtype=ndt)
    sh = S.ShardedSpMM(list(range(N)), p, j, x, K)
    Bs = [torch.from_numpy(B_host).to(f"cuda:{k}") for k in range(N)]
    ptrs = [b.data_ptr() for b in Bs]
    for _ in range(SETUP_CALLS):
        sh.run_dev(ptrs, n, ndt, asynchronous=True)
    sh.sync()
    for _ in range(warmup):
        sh.run_dev(ptrs, n, ndt, asynchronous=True)
    sh.sync()
    for k in range(N):
        torch.cuda.synchronize(k)
    t0 = time.perf_counter()
    for _ in range(steps):
        sh.run_dev(ptrs, n, ndt, asynchronous=True)
    sh.sync()
    for k in range(N):
        torch.cuda.synchronize(k)
    elapsed = time.perf_counter() - t0
    # the same loop without overlap: every product waits for its own all-gather
    t1 = time.perf_counter()
    for _ in range(steps):
        sh.run_dev(ptrs, n, ndt)
    in_line = (time.perf_counter() - t1) / steps
    # parity: the first rows of the first block and the last rows of the last one, as the LAST device holds them
    rows_chk = 2048
    Cg = sh.gathered(N - 1)

    def ref_rows(r0):
        ref = np.zeros(rows_chk * n, dtype=ndt)
        lo, hi = int(p[r0]), int(p[r0 + rows_chk])
        O.gemm_csr_drm_as_drm(rows_chk, n, (p[r0:r0 + rows_chk + 1] - p[r0]).astype(np.int32), j[lo:hi].copy(), x[lo:hi].copy(),
                              B_host.reshape(-1), n, ref, n, O.max_threads(), True)
        return ref.reshape(rows_chk, n)
    parity = 0.0
    for r0 in (0, m - rows_chk):
        ref = ref_rows(r0)
        parity = max(parity, float(np.max(np.abs(Cg[r0:r0 + rows_chk].astype(np.float64) - ref)) / np.max(np.abs(ref))))
    assert parity <= (1e-10 if dtype == "f64" else 2e-5), f"bench output differs from the oracle: {parity}"
    flops = 2.0 * int(p[-1]) * n
    s_dense = 8 if dtype == "f64" else 4
    slot_bytes = sh.slot_rows * n * s_dense
    out = {
        "metric": "CSR x dense SpMM GFLOP/s (fp64, 1M x 100k, 32 nnz/row, k=128; one process, C-ABI mx_spmm_sharded_*: row blocks with "
                  "AUTO's kept plan + one in-place ncclAllGather of C per product) + achieved HBM BW% vs CPU ref" if not args.custom else
                  "CSR x dense SpMM GFLOP/s (custom shape; one process, C-ABI mx_spmm_sharded_*)",
        "value": round(flops * steps / elapsed / 1e9, 2), "unit": "GFLOP/s", "n_gpus": N, "steps": steps, "warmup": warmup,
        "setup_calls": SETUP_CALLS, "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"dgRMatrix {m}x{K} nnz/row={nnz_row} (CSR values f64) %*% dense {K}x{n} {dtype} ({WORKLOADS[args.config]['label'] if not args.custom else 'custom shape'}); "
                               f"C row-major, the whole matrix cut into {N} row block(s) balanced by entries + rows, one per GPU, + RCCL all-gather of C; "
                               f"ONE process", "parallelism": f"rowshard{N}-single-process", "rows_per_gpu": sh.cuts[1] - sh.cuts[0],
                   "cols": K, "nnz_per_row": nnz_row, "dense_cols": n, "layout": "rowmajor"},
        "roofline": roofline(synth.spmm_algorithmic_bytes(m, K, n, int(p[-1]), s_dense) / N, elapsed / steps,
                             scope="whole step on one GPU's share of the algorithmic bytes (products + all-gather; no per-kernel events in this mode)",
                             kernel=sh.kernel(0)),
        "parity_max_err_over_max_abs_vs_oracle": parity,
        "single_process": {"uses_rccl": sh.uses_rccl, "rccl_version": sh.rccl_version, "shards": sh.nshards, "slot_rows": sh.slot_rows,
                           "in_line_ms_per_step": round(in_line * 1e3, 4),
                           "allgather_bytes_received_per_gpu": int((N - 1) * slot_bytes)},
        "distributed": {"backend": "rccl (ncclCommInitAll, one process)", "world_size": N, "scaling": "strong",
                        "row_blocks": [b - a for a, b in zip(sh.cuts, sh.cuts[1:])], "rows_total": m},
        "device": _lib.device_name(),
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_spmm(args, p, j, x, B_host, dtype)
    sh.close()
    # (RCCL prints its version banner through C stdio, which — stdout being a pipe — would otherwise land AFTER the line)
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    print(headline_line(out), flush=True)


def main():
    args = parse()
    if args.single_process:
        return single_process(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1 or FORCE_DIST:
        if "MASTER_ADDR" not in os.environ:                 # forced one-rank run started without torchrun
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"),
                              RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from matrixextra_amd import _lib, device as D, synth
    lib = _lib.load()                      # fails loudly if libmxgpu.so is missing

    if args.config == "cfg5-full":
        # BASELINE configs[4] WHOLE on this one GPU, through the export-level boundary (tools/cfg5_full.py: sharded over the
        # device listed 8 times, unsharded cold / cached, and one device-level launch; parity checks inside)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import cfg5_full
        r = cfg5_full.run(verbose=True)
        stream = stream_copy_probe(torch, lib, _lib)       # (AFTER the run: with torch's allocator already active in the process the cold
                                                           # unsharded call measured 262-265 ms instead of 188-197, three runs each way)
        dl = r["device_level"]
        dl["roofline"]["frac_of_stream_copy"] = round(dl["roofline"]["achieved"] / stream["GBps"], 4)
        print(headline_line({
            "metric": "CSR x dense SpMM GFLOP/s (fp32 dense / f64 CSR values, 8M x 200k, 64 nnz/row, k=256, whole matrix on one GPU) "
                      "+ achieved HBM BW% vs CPU ref",
            "value": dl["GFLOP/s"], "unit": "GFLOP/s", "n_gpus": 1, "steps": 5, "warmup": 2, "ms_per_step": dl["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "dgRMatrix 8000000x200000 nnz/row=64 (CSR values f64) %*% dense 200000x256 f32 (BASELINE "
                                   "configs[4], the whole matrix on ONE MI355X); C col-major", "parallelism": "single",
                       "layout": "colmajor"},
            "roofline": dl["roofline"], "stream_copy": stream, "extras": {"cfg5_full": r}, "device": _lib.device_name()},
            extras_file=os.path.join(ROOT, "gpurun_out", "bench_cfg5_full.json")), flush=True)
        return
    cfg = dict(name=args.config, rows=args.rows, cols=args.cols, nnz_row=args.nnz_row, n=args.n, dtype=args.dtype,
               label=WORKLOADS[args.config]["label"] if not args.custom else "custom shape")
    stream = stream_copy_probe(torch, lib, _lib) if world == 1 or rank == 0 else None
    want_cpu = not args.no_cpu_baseline                      # (rank 0's host cores; a shorter sample beside N > 1 ranks)
    if world > 1 or FORCE_DIST:
        args.cpu_seconds = min(args.cpu_seconds, 8.0)
    r = spmm_leg(args, torch, dist, D, synth, lib, _lib, cfg, world, rank, args.steps, args.warmup, want_cpu, True)
    out = None
    if rank == 0:
        host = r.pop("_host")
        dist_line = world > 1 or FORCE_DIST
        what = {"cfg2": "fp64, 1M x 100k, 32 nnz/row, k=128", "cfg5": "fp32 dense / f64 CSR values, 1M x 200k per GPU, "
                "64 nnz/row, k=256"}[args.config] if not args.custom else "custom shape"
        if args.scaling == "strong" and dist_line:
            what = (what.replace("1M x 200k per GPU", "8M x 200k over all GPUs") + f", the whole matrix row-sharded over {world} GPU(s) + "
                    "RCCL all-gather of C") if not args.custom else "custom shape, rows over all GPUs"
        elif dist_line:
            what = what.replace("1M x 100k", "1M x 100k per GPU") if not args.custom else "custom shape, rows per GPU"
        out = {
            "metric": f"CSR x dense SpMM GFLOP/s ({what}; AUTO's plan kept per matrix — one_shot_* = plan rebuilt inside every step) "
                      f"+ achieved HBM BW% vs CPU ref",
            "value": r.pop("value"), "unit": "GFLOP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "setup_calls": SETUP_CALLS,
            "ms_per_step": r.pop("ms_per_step"),
            "higher_is_better": True, "scaling": args.scaling if world > 1 or FORCE_DIST else "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": dict(workload=r.pop("workload"), parallelism=f"rowshard{world}" if world > 1 else "single",
                           **r.pop("dims")),
            "roofline": r.pop("roofline"),
            "stream_copy": stream,
        }
        cpu = r.pop("cpu_baseline", None)
        one = (r.get("plan") or {}).get("rebuild_every_step")
        if one:                                             # the like-for-like figure of rounds 1-2 (`value` there), at top level
            out["one_shot_ms_per_step"] = one["ms_per_step"]
            out["one_shot_gflops"] = one["GFLOP/s"]
        out.update(r)
        out["device"] = _lib.device_name()
        if cpu:
            out["cpu_baseline"] = cpu
        if world == 1 and not FORCE_DIST and not args.no_extras and not args.custom and args.config == "cfg2" and args.algo == 0:
            from bench_extras import extras
            out["extras"] = extras(args, torch, D, synth, _lib, host, want_cpu)
            del host
            torch.cuda.empty_cache()
            # configs[4]'s per-GPU shard on this one GPU (the N > 1 workload; its N = 1 reference point)
            c5 = dict(WORKLOADS["cfg5"], name="cfg5")
            try:
                r5 = spmm_leg(args, torch, dist, D, synth, lib, _lib, c5, 1, 0, max(5, args.steps // 2), 2, False, False)
                r5.pop("_host")
                out["extras"]["cfg5_shard"] = r5
            except Exception as exc:                         # noqa: BLE001 - the headline line still has to come out
                out["extras"].setdefault("errors", {})["cfg5_shard"] = repr(exc)[:400]
    if (world > 1 or FORCE_DIST) and not args.no_extras and not args.custom and args.config == "cfg2" and args.algo == 0 \
            and args.scaling == "strong":
        # BASELINE configs[4] on the same line: the 8M x 200k f32 product, strong-scaled over the same ranks (every rank takes
        # part: the leg holds collectives).  A failure here must not cost the headline line — and must not leave ranks
        # waiting for each other: every rank first learns whether all of them got through the set-up.
        c5 = dict(WORKLOADS["cfg5"], name="cfg5", rows=args.cfg5_strong_rows,
                  label="BASELINE configs[4], the whole matrix" if args.cfg5_strong_rows == 8_000_000 else "configs[4]'s shape, fewer rows")
        try:
            torch.cuda.empty_cache()
            # bytes this rank will hold: its block of A (+ AUTO's plan, ~1.6x), B, two gathered C buffers, generator scratch
            blk_nnz = c5["rows"] // world * c5["nnz_row"] * 1.05
            need = blk_nnz * 12 * 3.5 + c5["cols"] * c5["n"] * 4 + 2.0 * c5["rows"] * c5["n"] * 4 * 1.02 + (2 << 30)
            ok = torch.tensor([1.0 if torch.cuda.mem_get_info()[0] > need else 0.0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) < 1.0:
                raise RuntimeError(f"skipped: a rank has less than {need / 2**30:.1f} GiB of device memory free")
            r5 = spmm_leg(args, torch, dist, D, synth, lib, _lib, c5, world, rank, max(3, args.steps // 4), 1, False, False)
            if rank == 0:
                r5.pop("_host")
                r5["dtype"] = "f32"
                out.setdefault("extras", {})["cfg5_strong"] = r5
        except Exception as exc:                             # noqa: BLE001
            if rank == 0:
                out.setdefault("extras", {}).setdefault("errors", {})["cfg5_strong"] = repr(exc)[:400]
    if world > 1 or FORCE_DIST:
        # The JSON line must be the LAST line on stdout.  RCCL writes its version banner through C stdio, which — stdout
        # being a pipe — sits in the C buffer until the process exits, i.e. lands AFTER a line printed from Python.  So:
        # every rank empties its C buffers, the group is torn down, the other ranks leave, and only then rank 0 prints.
        import ctypes
        libc = ctypes.CDLL(None)
        libc.fflush(None)
        dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
        if rank == 0 and world > 1:
            time.sleep(1.0)                                  # the other ranks' exit-time output, if any, comes first
    if rank == 0:
        # ONE line, last on stdout, <= 4 KB, strict JSON; the full record (every extra with its notes) goes to
        # gpurun_out/bench_extras.json (bench_common.headline_line; tests/test_bench_line.py)
        sys.stdout.flush()
        print(headline_line(out), flush=True)


if __name__ == "__main__":
    main()
